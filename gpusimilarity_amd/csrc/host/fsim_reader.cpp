// fsim_reader.cpp -- see fsim_reader.h.
#include "fsim_reader.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <atomic>
#include <cstdio>
#include <stdexcept>
#include <thread>

#include "qds.h"

namespace gpusim
{

namespace
{
// A qCompress block where it lies in the mapped file: nothing of the file is copied before it is inflated.
struct BlockView {
    const unsigned char* p = nullptr;
    size_t n = 0;
};

// The whole file mapped read-only (the reference reads it through a QFile + QDataStream, gpusim.cpp:176-183; a 100 M-row
// database is tens of GB compressed: a second copy of it on the heap is what this avoids).
class MappedFile
{
  public:
    explicit MappedFile(const std::string& fname)
    {
        m_fd = ::open(fname.c_str(), O_RDONLY | O_CLOEXEC);
        if (m_fd < 0) throw std::runtime_error("cannot open database file " + fname);
        struct stat st;
        if (::fstat(m_fd, &st) != 0 || !S_ISREG(st.st_mode)) {
            ::close(m_fd);
            throw std::runtime_error("cannot open database file " + fname);
        }
        m_size = static_cast<size_t>(st.st_size);
        if (m_size) {
            void* m = ::mmap(nullptr, m_size, PROT_READ, MAP_PRIVATE, m_fd, 0);
            if (m == MAP_FAILED) {
                ::close(m_fd);
                throw std::runtime_error("cannot map database file " + fname);
            }
            m_data = static_cast<const unsigned char*>(m);
            ::madvise(const_cast<unsigned char*>(m_data), m_size, MADV_SEQUENTIAL);
        }
    }
    ~MappedFile()
    {
        if (m_data) ::munmap(const_cast<unsigned char*>(m_data), m_size);
        ::close(m_fd);
    }
    MappedFile(const MappedFile&) = delete;
    MappedFile& operator=(const MappedFile&) = delete;
    const unsigned char* data() const { return m_data; }
    size_t size() const { return m_size; }

  private:
    int m_fd = -1;
    const unsigned char* m_data = nullptr;
    size_t m_size = 0;
};

std::vector<BlockView> read_block_list(QdsReader& r, const unsigned char* base)
{
    const int n = r.i32();
    if (n < 0) throw std::runtime_error("database file: negative block count");
    // (every block costs at least its 4-byte length: a count the rest of the file cannot hold is a corrupt file, not a
    // reason to reserve n entries)
    if (static_cast<size_t>(n) > r.remaining() / 4) throw std::runtime_error("QDataStream: truncated input");
    std::vector<BlockView> blocks(static_cast<size_t>(n));
    for (auto& b : blocks) {
        const uint32_t len = r.u32();
        if (len == 0xFFFFFFFFu) continue; // a null QByteArray
        b.p = base + r.consumed();
        b.n = len;
        r.skip(len);
    }
    return blocks;
}

// qUncompress straight into `out` (resized to the size the block announces).
template <class Vec> void inflate_into(const BlockView& b, Vec& out)
{
    out.clear();
    if (b.n < 4) return;
    const uint32_t expect = (uint32_t(b.p[0]) << 24) | (uint32_t(b.p[1]) << 16) | (uint32_t(b.p[2]) << 8) | uint32_t(b.p[3]);
    out.resize(expect ? expect : 1);
    uLongf len = static_cast<uLongf>(out.size());
    const int rc = uncompress(reinterpret_cast<Bytef*>(out.data()), &len, b.p + 4, static_cast<uLong>(b.n - 4));
    if (rc != Z_OK || len != expect) throw std::runtime_error("qUncompress: corrupt block in database file");
    out.resize(expect);
}

void strings_from_block(const BlockView& compressed, std::vector<char*>& out)
{
    std::vector<unsigned char> raw;
    inflate_into(compressed, raw);
    QdsReader r(raw);
    while (!r.atEnd()) out.push_back(r.cstr_new());
}
} // namespace

std::vector<unsigned char> q_uncompress(const std::vector<unsigned char>& blob)
{
    std::vector<unsigned char> out;
    inflate_into(BlockView{blob.data(), blob.size()}, out);
    return out;
}

unsigned extract_threads(size_t jobs)
{
    unsigned hw = std::thread::hardware_concurrency();
    if (hw == 0) hw = 4;
    const unsigned cap = hw < kMaxExtractThreads ? hw : kMaxExtractThreads;
    return static_cast<unsigned>(jobs < cap ? jobs : cap);
}

void extractData(const std::string& database_fname, int& fp_bitcount, int& fp_count, std::string& dbkey,
                 std::vector<std::vector<char>>& fingerprint_data, std::vector<char*>& smiles_vector,
                 std::vector<char*>& ids_vector)
{
    const MappedFile file(database_fname);
    QdsReader r(file.data(), file.size());
    const int version = r.i32();
    if (version != DATABASE_VERSION) {
        throw std::runtime_error("Database version incompatible with this GPUSim version");
    }
    dbkey = r.cstr();
    fp_bitcount = r.i32();
    fp_count = r.i32();

    const auto fp_blocks = read_block_list(r, file.data());
    const auto smi_blocks = read_block_list(r, file.data());
    const auto id_blocks = read_block_list(r, file.data());

    fingerprint_data.assign(fp_blocks.size(), {});
    std::vector<std::vector<char*>> smiles_data(smi_blocks.size()), ids_data(id_blocks.size());
    // One job per block (the reference queues one QRunnable per block on the global QThreadPool, gpusim.cpp:193-236,
    // which runs as many at a time as the host has cores): the workers take the jobs in file order from one counter.
    // A block's inflated bytes are held once -- a fingerprint block is inflated into the vector the caller keeps.
    const size_t nfp = fp_blocks.size(), nsmi = smi_blocks.size(), njobs = nfp + nsmi + id_blocks.size();
    std::vector<std::string> errors(njobs);
    std::atomic<size_t> next{0};
    auto worker = [&] {
        for (size_t job = next.fetch_add(1); job < njobs; job = next.fetch_add(1)) {
            try {
                if (job < nfp) {
                    std::fprintf(stderr, "  loading FP %zu of %zu\n", job + 1, nfp);
                    inflate_into(fp_blocks[job], fingerprint_data[job]);
                } else if (job < nfp + nsmi) {
                    std::fprintf(stderr, "  loading SMI %zu of %zu\n", job - nfp + 1, nsmi);
                    strings_from_block(smi_blocks[job - nfp], smiles_data[job - nfp]);
                } else {
                    std::fprintf(stderr, "  loading ID %zu of %zu\n", job - nfp - nsmi + 1, id_blocks.size());
                    strings_from_block(id_blocks[job - nfp - nsmi], ids_data[job - nfp - nsmi]);
                }
            } catch (const std::exception& e) {
                errors[job] = e.what();
            }
        }
    };
    const unsigned nthreads = extract_threads(njobs);
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < nthreads; t++) pool.emplace_back(worker);
    worker(); // (the caller's thread is one of them)
    for (auto& t : pool) t.join();
    for (const auto& e : errors)
        if (!e.empty()) throw std::runtime_error(e);
    for (auto& v : smiles_data) smiles_vector.insert(smiles_vector.end(), v.begin(), v.end());
    for (auto& v : ids_data) ids_vector.insert(ids_vector.end(), v.begin(), v.end());
}

} // namespace gpusim
