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
#include <set>
#include <sstream>
#include <stdexcept>

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

GPUSimServer::GPUSimServer(const std::vector<std::string>& database_fnames, int gpu_bitcount, bool open_socket,
                           bool use_gpu, int ndevices)
    : m_use_gpu(use_gpu)
{
    std::fprintf(stderr, "--------------------------\nStarting up GPUSim Server\n--------------------------\n");
    std::fprintf(stderr, "Utilizing %u GPUs for calculation.\n", get_gpu_count());
    if (open_socket && !setupSocket()) return;

    for (const auto& database_fname : database_fnames) {
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

    // gpusim.cpp:121-163: does everything fit?  If not (or if --gpu_bitcount asks for it) the
    // tables are folded exactly as the reference does.
    size_t total_db_memory = 0;
    unsigned int max_compounds_in_db = 0;
    int max_fp_bitcount = 0;
    for (auto& kv : m_databases) {
        total_db_memory += kv.second->getFingerprintDataSize();
        max_compounds_in_db = std::max(max_compounds_in_db, kv.second->count());
        max_fp_bitcount = std::max(max_fp_bitcount, kv.second->getFingerprintBitcount());
    }
    if (usingGPU()) {
        size_t gpu_memory = get_available_gpu_memory();
        // search scratch: ~24 bytes per row worst case (candidate + finalist slots)
        const size_t scratch = static_cast<size_t>(max_compounds_in_db) * 24;
        gpu_memory = gpu_memory > scratch ? gpu_memory - scratch : 0;
        std::fprintf(stderr, "Database:   %zu MB GPU Memory:  %zu MB\n", total_db_memory / 1024 / 1024,
                     gpu_memory / 1024 / 1024);
        unsigned int fold_factor = 1;
        if (total_db_memory > gpu_memory) {
            fold_factor = static_cast<unsigned int>(
                std::ceil(static_cast<float>(total_db_memory) / static_cast<float>(gpu_memory ? gpu_memory : 1)));
        }
        if (gpu_bitcount > 0) {
            const unsigned int arg_fold_factor = static_cast<unsigned int>(max_fp_bitcount / gpu_bitcount);
            if (arg_fold_factor < fold_factor) {
                throw std::invalid_argument("GPU bitset not sufficiently small to fit on GPU"); // :146-149
            }
            fold_factor = arg_fold_factor;
        }
        std::fprintf(stderr, "Putting graphics card data up.\n");
        if (fold_factor > 1) {
            std::fprintf(stderr, "Folding databases by at least %u to fit in gpu memory\n", fold_factor);
        }
        for (auto& kv : m_databases) kv.second->copyToGPU(fold_factor, ndevices);
        std::fprintf(stderr, "Finished putting graphics card data up.\n");
    }
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
    // (score, arrival order): the reference sorts (score, (smiles*, id*)) pairs and reverses,
    // which leaves equal scores in heap-address order; here ties keep database order
    // (map order) then row order -- deterministic.
    struct Entry {
        float score;
        size_t seq;
        char* smiles;
        char* id;
    };
    std::vector<Entry> sortable;
    for (const auto& name_key : dbname_to_key) {
        const std::string& local_dbname = name_key.first;
        if (m_databases.find(local_dbname) == m_databases.end()) {
            std::fprintf(stderr, "Unknown database  %s  requested.\n", local_dbname.c_str());
            continue;
        }
        std::vector<char*> l_smiles, l_ids;
        std::vector<float> l_scores;
        unsigned long local_approx = 0; // the reference leaves it uninitialised on the CPU path
        similaritySearch(query, local_dbname, name_key.second, static_cast<unsigned int>(results_requested),
                         similarity_cutoff, usingGPU() ? CalcType::GPU : CalcType::CPU, l_smiles, l_ids, l_scores,
                         local_approx);
        approximate_result_count += local_approx;
        for (size_t i = 0; i < l_smiles.size(); i++) sortable.push_back({l_scores[i], sortable.size(), l_smiles[i], l_ids[i]});
    }
    std::stable_sort(sortable.begin(), sortable.end(), [](const Entry& a, const Entry& b) { return a.score > b.score; });

    std::map<std::string, std::string> smiles_to_ids;
    for (const auto& r : sortable) {
        const std::string smiles(r.smiles);
        auto it = smiles_to_ids.find(smiles);
        if (it != smiles_to_ids.end()) {
            it->second += ";:;";
            it->second += r.id;
        } else {
            smiles_to_ids[smiles] = r.id;
        }
        if (smiles_to_ids.size() >= static_cast<size_t>(results_requested)) break;
    }
    int written = 0;
    std::set<std::string> smiles_written;
    for (const auto& r : sortable) {
        if (written >= results_requested) break;
        const std::string smiles(r.smiles);
        if (smiles_written.count(smiles) > 0) continue;
        smiles_written.insert(smiles);
        results_scores.push_back(r.score);
        results_smiles.push_back(r.smiles);
        results_ids.push_back(strdup(smiles_to_ids[smiles].c_str()));
        ++written;
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
// A request is complete when it parses: i32 ndb, ndb x (cstr, cstr), i32, i32, f64, QByteArray.
bool request_complete(const std::vector<unsigned char>& buf)
{
    try {
        QdsReader r(buf);
        const int ndb = r.i32();
        if (ndb < 0 || ndb > 4096) return true; // garbage: let the handler fail and drop the client
        for (int i = 0; i < ndb; i++) {
            r.cstr();
            r.cstr();
        }
        r.i32();
        r.i32();
        r.f64();
        r.bytearray();
        return true;
    } catch (const std::exception&) {
        return false;
    }
}

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
            // the reference assumes a whole request per readyRead (gpusim.cpp:381); here
            // partial frames are buffered until they parse
            if (!request_complete(c.buf)) continue;
            try {
                const std::vector<unsigned char> reply = handleRequest(c.buf);
                if (!write_all(c.fd, reply.data(), reply.size())) {
                    close(c.fd);
                    c.fd = -1;
                }
            } catch (const std::exception& e) {
                std::fprintf(stderr, "request failed: %s\n", e.what());
                close(c.fd);
                c.fd = -1;
            }
            c.buf.clear();
        }
        clients.erase(std::remove_if(clients.begin(), clients.end(), [](const Client& c) { return c.fd < 0; }),
                      clients.end());
    }
    for (auto& c : clients) close(c.fd);
    return 0;
}

} // namespace gpusim
