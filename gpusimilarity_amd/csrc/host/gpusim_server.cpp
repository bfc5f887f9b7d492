// gpusim_server.cpp -- see gpusim_server.h.  POSIX sockets + poll(); the wire
// format is in qds.h.
#include "gpusim_server.h"

#include <poll.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/un.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unordered_map>
#include <sstream>
#include <stdexcept>
#include <string_view>

#include "fsim_reader.h"
#include "qds.h"

namespace gpusim
{

namespace
{
// QFileInfo::baseName(): file name up to (not including) the FIRST '.'
std::string base_name(const std::string& path)
{
    const size_t slash = path.find_last_of('/');
    std::string name = slash == std::string::npos ? path : path.substr(slash + 1);
    const size_t dot = name.find('.');
    return dot == std::string::npos ? name : name.substr(0, dot);
}
} // namespace

// Where the tables go and whether they have to be folded (gpusim.cpp:121-163 decides the fold factor
// from the SUM of free memory but then places every storage whole on one GPU; here the decision is
// made against the placement that will actually happen).
//
//   * requested == 1 (one GPU per table, round-robin like get_next_gpu): simulate that placement with
//     every device's free memory.  If a table does not fit any single device but the devices together
//     hold everything, the tables are sharded over all devices instead of failing or folding.
//   * requested == 0 / n: every table is sharded over all / the first n devices: capacity = their sum.
//   * Reserve per table: the four-kernel fallback pipeline's scratch (24 B per row, allocated on the
//     first query that needs it) + 64 MB of fixed buffers.
//   * What still does not fit is folded by ceil(bytes / capacity), and --gpu_bitcount may ask for more
//     (never for less: the reference's std::invalid_argument, :146-149).
struct TableFacts {
    size_t bytes;
    size_t rows;
    int fp_bits;
};

struct PlacementPlan {
    unsigned int fold_factor = 1;
    int ndevices = 1; // argument of FingerprintDB::copyToGPU
    bool full_on_device = true; // folded tables: the full fingerprints may be kept in HBM as well (device re-score)
};

static PlacementPlan plan_placement(const std::vector<TableFacts>& tables, int gpu_bitcount, int requested,
                                    const std::vector<size_t>& free_bytes)
{
    PlacementPlan plan;
    const int ndev = static_cast<int>(free_bytes.size());
    plan.ndevices = requested < 0 ? 1 : (requested > ndev ? ndev : requested);
    auto reserve = [](const TableFacts& t) { return t.rows * 24 + (size_t(64) << 20); };
    size_t total = 0, total_reserve = 0;
    int max_bits = 0;
    for (const auto& t : tables) {
        total += t.bytes;
        total_reserve += reserve(t);
        max_bits = std::max(max_bits, t.fp_bits);
    }
    auto pool = [&](int n) { // free memory of the first n devices (0 = all), minus the reserves
        size_t sum = 0;
        for (int d = 0; d < (n == 0 ? ndev : n) && d < ndev; d++) sum += free_bytes[d];
        return sum > total_reserve ? sum - total_reserve : 0;
    };
    auto whole_tables_fit = [&]() { // round-robin, first device with room, as gsim_next_device does
        std::vector<size_t> left = free_bytes;
        int next = 0;
        for (const auto& t : tables) {
            const size_t need = t.bytes + reserve(t);
            bool placed = false;
            for (int i = 0; i < ndev && !placed; i++) {
                const int d = next++ % ndev;
                if (left[d] > need) {
                    left[d] -= need;
                    placed = true;
                }
            }
            if (!placed) return false;
        }
        return true;
    };
    size_t capacity = pool(plan.ndevices);
    if (plan.ndevices == 1) {
        if (whole_tables_fit()) {
            capacity = total; // fits as it is
        } else if (ndev > 1 && pool(0) >= total) {
            plan.ndevices = 0; // together the devices hold it: shard every table over all of them
            capacity = pool(0);
        } else {
            capacity = pool(0); // the reference's sum; folded storages are placed one by one
        }
    }
    std::fprintf(stderr, "Database:   %zu MB GPU Memory:  %zu MB\n", total / 1024 / 1024, capacity / 1024 / 1024);
    if (total > capacity)
        plan.fold_factor = static_cast<unsigned int>(std::ceil(static_cast<float>(total) / static_cast<float>(capacity ? capacity : 1)));
    if (gpu_bitcount > 0) {
        const unsigned int arg_fold_factor = static_cast<unsigned int>(max_bits / gpu_bitcount);
        if (arg_fold_factor < plan.fold_factor) throw std::invalid_argument("GPU bitset not sufficiently small to fit on GPU");
        plan.fold_factor = arg_fold_factor;
    }
    // Folding to fit (total > capacity): the full fingerprints cannot be resident as well, and copies made database by
    // database would take the memory the plan counted on for the later databases' folded rows.  Folding on request
    // (--gpu_bitcount) with room for everything: each database may keep them (decided per table at finalize, all
    // storages or none).
    plan.full_on_device = plan.fold_factor > 1 && total + total / plan.fold_factor <= capacity;
    return plan;
}

GPUSimServer::GPUSimServer(const std::vector<std::string>& database_fnames, int gpu_bitcount, bool open_socket,
                           bool use_gpu, int ndevices, bool rccl_merge)
    : m_use_gpu(use_gpu)
{
    std::fprintf(stderr, "--------------------------\nStarting up GPUSim Server\n--------------------------\n");
    std::fprintf(stderr, "Utilizing %u GPUs for calculation.\n", get_gpu_count());
    if (open_socket && !setupSocket()) return;

    std::vector<TableFacts> facts;
    for (const auto& database_fname : database_fnames) {
        if (database_fname.rfind("synthetic:", 0) == 0) {
            // benchmark table, no file: "synthetic:<rows>[:<sparse|dense|morgan>[:<bits>]]" -> database "synthetic", key "pass"
            unsigned long long rows = 0;
            char kind_s[16] = "sparse";
            int bits = 1024;
            if (std::sscanf(database_fname.c_str(), "synthetic:%llu:%15[a-z]:%d", &rows, kind_s, &bits) < 1 || rows == 0 || bits <= 0 ||
                bits % 32 != 0)
                throw std::invalid_argument("bad synthetic database spec: " + database_fname);
            const std::string ks = kind_s;
            const int kind = ks == "dense" ? 1 : ks == "morgan" ? 2 : 0;
            m_databases["synthetic"] = std::make_shared<FingerprintDB>(bits, rows, "pass", 0x5EED0001ull, kind);
            continue;
        }
        int fp_bitcount = 0, fp_count = 0;
        std::string dbkey;
        std::vector<std::vector<char>> fingerprint_data;
        std::vector<char*> smiles_vector, ids_vector;
        std::fprintf(stderr, "Extracting data: %s\n", database_fname.c_str());
        extractData(database_fname, fp_bitcount, fp_count, dbkey, fingerprint_data, smiles_vector, ids_vector);
        std::fprintf(stderr, "Finished extracting data\n");
        auto fps = std::make_shared<FingerprintDB>(fp_bitcount, fp_count, dbkey, fingerprint_data, smiles_vector,
                                                   ids_vector);
        m_databases[base_name(database_fname)] = fps;
    }
    if (!usingGPU()) {
        std::fprintf(stderr, "Ready for searches.\n");
        return;
    }
    for (auto& kv : m_databases)
        facts.push_back({kv.second->getFingerprintDataSize(), kv.second->count(), kv.second->getFingerprintBitcount()});
    std::vector<size_t> free_bytes(get_gpu_count());
    for (size_t d = 0; d < free_bytes.size(); d++) free_bytes[d] = get_gpu_free_memory(static_cast<unsigned int>(d));
    const PlacementPlan plan = plan_placement(facts, gpu_bitcount, ndevices, free_bytes);
    std::fprintf(stderr, "Putting graphics card data up.\n");
    if (plan.fold_factor > 1) std::fprintf(stderr, "Folding databases by at least %u to fit in gpu memory\n", plan.fold_factor);
    if (ndevices == 1 && plan.ndevices == 0) std::fprintf(stderr, "Sharding every database over all GPUs (no single GPU holds the largest)\n");
    if (rccl_merge && plan.fold_factor <= 1) std::fprintf(stderr, "Per-GPU results merged through an RCCL all-gather (--merge rccl)\n");
    for (auto& kv : m_databases) kv.second->copyToGPU(plan.fold_factor, plan.ndevices, plan.full_on_device, rccl_merge);
    std::fprintf(stderr, "Finished putting graphics card data up.\n");
    std::fprintf(stderr, "Ready for searches.\n");
}

GPUSimServer::~GPUSimServer()
{
    if (m_listen_fd >= 0) {
        close(m_listen_fd);
        unlink(socketPath().c_str());
    }
}

bool GPUSimServer::usingGPU()
{
    return m_use_gpu && (get_gpu_count() != 0);
}

bool GPUSimServer::setupSocket()
{
    const std::string path = socketPath();
    auto try_listen = [&]() -> int {
        int fd = socket(AF_UNIX, SOCK_STREAM, 0);
        if (fd < 0) return -1;
        sockaddr_un addr;
        std::memset(&addr, 0, sizeof(addr));
        addr.sun_family = AF_UNIX;
        std::strncpy(addr.sun_path, path.c_str(), sizeof(addr.sun_path) - 1);
        if (bind(fd, reinterpret_cast<sockaddr*>(&addr), sizeof(addr)) != 0 || listen(fd, 50) != 0) {
            close(fd);
            return -1;
        }
        return fd;
    };
    m_listen_fd = try_listen();
    if (m_listen_fd < 0) {
        unlink(path.c_str()); // stale socket file: remove and retry once (gpusim.cpp:259-262)
        m_listen_fd = try_listen();
        if (m_listen_fd < 0) {
            std::fprintf(stderr, "Server start failed on %s\n", path.c_str());
            return false;
        }
    }
    return true;
}

void GPUSimServer::similaritySearch(const Fingerprint& reference, const std::string& dbname,
                                    const std::string& dbkey, unsigned int max_return_count,
                                    float similarity_cutoff, CalcType calc_type, std::vector<char*>& results_smiles,
                                    std::vector<char*>& results_ids, std::vector<float>& results_scores,
                                    unsigned long& approximate_result_count)
{
    auto it = m_databases.find(dbname);
    if (it == m_databases.end()) throw std::invalid_argument("unknown database " + dbname);
    if (calc_type == CalcType::GPU) {
        it->second->search(reference, dbkey, max_return_count, similarity_cutoff, results_smiles, results_ids,
                           results_scores, approximate_result_count);
    } else {
        it->second->search_cpu(reference, dbkey, max_return_count, similarity_cutoff, results_smiles, results_ids,
                               results_scores, approximate_result_count);
    }
}

void GPUSimServer::searchDatabases(const Fingerprint& query, int results_requested, float similarity_cutoff,
                                   std::map<std::string, std::string>& dbname_to_key,
                                   std::vector<char*>& results_smiles, std::vector<char*>& results_ids,
                                   std::vector<float>& results_scores, unsigned long& approximate_result_count)
{
    // Every database returns its hits in canonical order (score desc, row asc), so the merged order
    // -- score desc; equal scores: database (map) order, then row order -- comes out of a k-way merge
    // on a small heap, and only as far as it is needed.  (The reference concatenates, std::sorts
    // (score, (smiles*, id*)) pairs and reverses, gpusim.cpp:330-336, which leaves equal scores in
    // heap-address order; this order is deterministic.)
    struct List {
        std::vector<char*> smiles, ids;
        std::vector<float> scores;
        size_t pos = 0;
    };
    std::vector<List> lists;
    for (const auto& name_key : dbname_to_key) {
        const std::string& local_dbname = name_key.first;
        if (m_databases.find(local_dbname) == m_databases.end()) {
            std::fprintf(stderr, "Unknown database  %s  requested.\n", local_dbname.c_str());
            continue;
        }
        lists.emplace_back();
        List& l = lists.back();
        unsigned long local_approx = 0; // the reference leaves it uninitialised on the CPU path
        similaritySearch(query, local_dbname, name_key.second, static_cast<unsigned int>(results_requested),
                         similarity_cutoff, usingGPU() ? CalcType::GPU : CalcType::CPU, l.smiles, l.ids, l.scores,
                         local_approx);
        approximate_result_count += local_approx;
    }
    if (results_requested <= 0) return;
    auto later = [&](size_t a, size_t b) { // "a comes after b": heap keeps the earliest entry on top
        const float sa = lists[a].scores[lists[a].pos], sb = lists[b].scores[lists[b].pos];
        return sa < sb || (sa == sb && a > b);
    };
    std::vector<size_t> heap;
    for (size_t i = 0; i < lists.size(); i++)
        if (!lists[i].scores.empty()) heap.push_back(i);
    std::make_heap(heap.begin(), heap.end(), later);

    // Hits with equal SMILES fold into one result whose id is the ids joined by ";:;" (:342-357); the
    // merge stops with the entry that completes the results_requested-th distinct SMILES (:355).
    // (no allocation per hit: the map's keys are views of the databases' own SMILES strings, an id is the database's own
    // string until a second hit with the same SMILES makes it a joined copy -- at k = 1000 the four std::string
    // constructions per hit were a third of a millisecond of the server-side latency)
    struct Result {
        float score;
        char* smiles;
        char* id;           // the first hit's id (the database's string)
        std::string joined; // ... or, once a second hit has the same SMILES, the ids joined by ";:;"
    };
    // (sized by what the lists can yield, not by the count a client wrote on the socket: INT_MAX would ask for ~150 GB)
    size_t available = 0;
    for (const List& l : lists) available += l.scores.size();
    const size_t cap = std::min(static_cast<size_t>(results_requested), available);
    std::vector<Result> results;
    results.reserve(cap);
    std::unordered_map<std::string_view, size_t> by_smiles;
    by_smiles.reserve(cap * 2);
    while (!heap.empty()) {
        std::pop_heap(heap.begin(), heap.end(), later);
        List& l = lists[heap.back()];
        const size_t p = l.pos++;
        const std::string_view key(l.smiles[p]);
        auto found = by_smiles.find(key);
        if (found != by_smiles.end()) {
            Result& r = results[found->second];
            if (r.joined.empty()) r.joined = r.id;
            r.joined += ";:;";
            r.joined += l.ids[p];
        } else {
            by_smiles.emplace(key, results.size());
            results.push_back({l.scores[p], l.smiles[p], l.ids[p], std::string()});
        }
        if (results.size() >= static_cast<size_t>(results_requested)) break;
        if (l.pos < l.scores.size()) std::push_heap(heap.begin(), heap.end(), later);
        else heap.pop_back();
    }
    results_scores.reserve(results.size());
    results_smiles.reserve(results.size());
    results_ids.reserve(results.size());
    for (const auto& r : results) {
        results_scores.push_back(r.score);
        results_smiles.push_back(r.smiles);
        results_ids.push_back(strdup(r.joined.empty() ? r.id : r.joined.c_str())); // the receiver frees them (:369-370)
    }
}

Fingerprint GPUSimServer::getFingerprint(const int index, const std::string& dbname)
{
    auto it = m_databases.find(dbname);
    if (it == m_databases.end()) throw std::invalid_argument("unknown database " + dbname);
    return it->second->getFingerprint(static_cast<unsigned int>(index));
}

std::vector<unsigned char> GPUSimServer::handleRequest(const std::vector<unsigned char>& request)
{
    QdsReader qds(request);
    const int database_search_count = qds.i32();
    std::map<std::string, std::string> dbname_to_key;
    for (int i = 0; i < database_search_count; i++) {
        const std::string dbname = qds.cstr();
        const std::string dbkey = qds.cstr();
        dbname_to_key[dbname] = dbkey;
    }
    const int request_num = qds.i32();
    const int results_requested = qds.i32();
    const float similarity_cutoff = static_cast<float>(qds.f64()); // `qds >> float` reads a double
    const std::vector<unsigned char> fp_data = qds.bytearray();
    const size_t fp_int_size = fp_data.size() / sizeof(int);
    Fingerprint query(fp_int_size);
    if (fp_int_size) std::memcpy(query.data(), fp_data.data(), fp_int_size * sizeof(int));

    std::vector<char*> results_smiles, results_ids;
    std::vector<float> results_scores;
    const auto t0 = std::chrono::steady_clock::now();
    unsigned long approximate_result_count = 0;
    // (a count <= 0 from the socket -- the reference would cast it to a huge unsigned -- gets an empty reply; counts
    // above the table sizes are clamped inside FingerprintDB::search)
    if (results_requested > 0)
        searchDatabases(query, results_requested, similarity_cutoff, dbname_to_key, results_smiles, results_ids,
                    results_scores, approximate_result_count);
    const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::fprintf(stderr, "Search completed, time elapsed: %g\n", el);

    // gpusim.cpp:432-453: ints, then all smiles, all ids, all scores
    QdsWriter out;
    out.i32(request_num);
    out.i32(static_cast<int>(results_smiles.size()));
    out.u64(static_cast<uint64_t>(approximate_result_count));
    for (char* s : results_smiles) out.cstr(s);
    for (char* s : results_ids) out.cstr(s);
    for (float f : results_scores) out.f64(static_cast<double>(f));
    for (char* s : results_ids) free(s); // strdup'd by searchDatabases (the reference leaks them)
    return out.bytes();
}

namespace
{
// Length of the first complete request in buf (i32 ndb, ndb x (cstr, cstr), i32, i32, f64, QByteArray),
// 0 when more bytes are needed, -1 for garbage.
long request_length(const std::vector<unsigned char>& buf)
{
    try {
        QdsReader r(buf);
        const int ndb = r.i32();
        if (ndb < 0 || ndb > 4096) return -1;
        for (int i = 0; i < ndb; i++) {
            r.cstr();
            r.cstr();
        }
        r.i32();
        r.i32();
        r.f64();
        r.bytearray();
        return static_cast<long>(r.consumed());
    } catch (const std::exception&) {
        return 0;
    }
}

constexpr size_t kMaxRequestBytes = size_t(4) << 20; // names + keys + one fingerprint: kilobytes in practice

bool write_all(int fd, const unsigned char* p, size_t n)
{
    while (n) {
        const ssize_t w = send(fd, p, n, MSG_NOSIGNAL);
        if (w < 0) {
            if (errno == EINTR) continue;
            return false;
        }
        p += w;
        n -= static_cast<size_t>(w);
    }
    return true;
}
} // namespace

int GPUSimServer::exec()
{
    if (m_listen_fd < 0) return 1;
    auto len_error = [](int& fd, const char* what) {
        std::fprintf(stderr, "request failed: %s\n", what);
        close(fd);
        fd = -1;
    };
    struct Client {
        int fd;
        std::vector<unsigned char> buf;
    };
    std::vector<Client> clients;
    while (!m_stop) {
        std::vector<pollfd> fds;
        fds.push_back({m_listen_fd, POLLIN, 0});
        for (auto& c : clients) fds.push_back({c.fd, POLLIN, 0});
        const int rc = poll(fds.data(), fds.size(), 200);
        if (rc < 0) {
            if (errno == EINTR) continue;
            return 1;
        }
        if (rc == 0) continue;
        if (fds[0].revents & POLLIN) {
            const int cfd = accept(m_listen_fd, nullptr, nullptr);
            if (cfd >= 0) clients.push_back({cfd, {}});
        }
        for (size_t i = 1; i < fds.size(); i++) {
            if (!(fds[i].revents & (POLLIN | POLLHUP | POLLERR))) continue;
            Client& c = clients[i - 1];
            unsigned char tmp[65536];
            const ssize_t n = recv(c.fd, tmp, sizeof(tmp), 0);
            if (n <= 0) {
                close(c.fd);
                c.fd = -1;
                continue;
            }
            c.buf.insert(c.buf.end(), tmp, tmp + n);
            // the reference assumes a whole request per readyRead (gpusim.cpp:381); here partial frames are
            // buffered until they parse and several requests in one read are answered one after the other
            for (;;) {
                const long len = request_length(c.buf);
                if (len == 0) {
                    if (c.buf.size() > kMaxRequestBytes) len_error(c.fd, "request larger than 4 MB");
                    break;
                }
                if (len < 0) {
                    len_error(c.fd, "malformed request");
                    break;
                }
                try {
                    const std::vector<unsigned char> frame(c.buf.begin(), c.buf.begin() + len);
                    const std::vector<unsigned char> reply = handleRequest(frame);
                    c.buf.erase(c.buf.begin(), c.buf.begin() + len);
                    if (!write_all(c.fd, reply.data(), reply.size())) {
                        close(c.fd);
                        c.fd = -1;
                        break;
                    }
                } catch (const std::exception& e) {
                    len_error(c.fd, e.what());
                    break;
                }
            }
        }
        clients.erase(std::remove_if(clients.begin(), clients.end(), [](const Client& c) { return c.fd < 0; }),
                      clients.end());
    }
    for (auto& c : clients) close(c.fd);
    return 0;
}

} // namespace gpusim
