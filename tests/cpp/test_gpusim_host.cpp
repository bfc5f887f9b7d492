// test_gpusim_host.cpp -- the reference's Boost tests (test/test_gpusim.cpp) restated
// for the Qt-free host twin (no Boost in the image: a 30-line harness instead).
//
//   test_gpusim_host cpu <small.fsim> <small_copy.fsim>   CPU-only cases (no GPU needed)
//   test_gpusim_host gpu <small.fsim> <small_copy.fsim>   the cases that need a GPU
//
// Exit code 0 = all checks passed.  Driven by tests/test_host_cpp.py.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "fingerprintdb.h"
#include "fsim_reader.h"
#include "gpusim_server.h"
#include "qds.h"

using namespace gpusim;
using std::vector;

static int g_fail = 0;
#define CHECK(cond)                                                              \
    do {                                                                         \
        if (!(cond)) {                                                           \
            std::fprintf(stderr, "CHECK FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            g_fail++;                                                            \
        }                                                                        \
    } while (0)
#define CHECK_EQ(a, b) CHECK((a) == (b))

// test_gpusim.cpp:134-146
static void CPUSort()
{
    vector<int> indices = {0, 1, 2, 3, 4, 5};
    vector<float> scores = {1, 3, 2, 4, 0, 7};
    top_results_bubble_sort(indices, scores, 3);
    CHECK_EQ(indices[0], 5);
    CHECK_EQ(scores[0], 7);
    CHECK_EQ(indices[2], 1);
    CHECK_EQ(scores[2], 3);
}

static void LoadFsim(const std::string& small)
{
    int bits = 0, count = 0;
    std::string key;
    vector<vector<char>> fp;
    vector<char*> smiles, ids;
    extractData(small, bits, count, key, fp, smiles, ids);
    CHECK_EQ(bits, 1024);
    CHECK_EQ(count, 100);
    CHECK_EQ(key, std::string("pass"));
    CHECK_EQ(fp.size(), 1u);
    CHECK_EQ(fp[0].size(), 12800u);
    CHECK_EQ(smiles.size(), 100u);
    CHECK_EQ(ids.size(), 100u);
    CHECK_EQ(std::string(ids[0]), std::string("ZINC00000007"));
    CHECK_EQ(std::string(ids[3]), std::string("ZINC00000022"));
    bool threw = false;
    try {
        vector<vector<char>> bad = fp;
        vector<char*> s2, i2;
        FingerprintDB db(1024, 99, "pass", bad, s2, i2); // fingerprintdb_cuda.cu:153-156
    } catch (const std::runtime_error&) {
        threw = true;
    }
    CHECK(threw);
}

// SURVEY.md Appendix B example request (51 bytes) decodes; reply framing round-trips
static void Codec()
{
    const unsigned char req[] = {0, 0, 0, 1, 0, 0, 0, 6, 's', 'm', 'a', 'l', 'l', 0, 0, 0, 0, 5, 'p', 'a', 's', 's', 0,
                                 0x11, 0x22, 0x33, 0x44, 0, 0, 0, 0x0a, 0x3f, 0xd3, 0x33, 0x33, 0x40, 0, 0, 0,
                                 0, 0, 0, 8, 0, 1, 2, 3, 4, 5, 6, 7};
    CHECK_EQ(sizeof(req), 51u);
    QdsReader r(req, sizeof(req));
    CHECK_EQ(r.i32(), 1);
    CHECK_EQ(r.cstr(), std::string("small"));
    CHECK_EQ(r.cstr(), std::string("pass"));
    CHECK_EQ(r.i32(), 0x11223344);
    CHECK_EQ(r.i32(), 10);
    CHECK_EQ(static_cast<float>(r.f64()), 0.3f);
    CHECK_EQ(r.bytearray().size(), 8u);
    CHECK(r.atEnd());
    QdsWriter w;
    w.i32(1);
    w.cstr("small");
    w.cstr("pass");
    w.i32(0x11223344);
    w.i32(10);
    w.f64(static_cast<double>(0.3f));
    const unsigned char fp[8] = {0, 1, 2, 3, 4, 5, 6, 7};
    w.bytearray(fp, 8);
    CHECK_EQ(w.bytes().size(), 51u);
    CHECK(std::memcmp(w.bytes().data(), req, 51) == 0);
    QdsWriter e;
    e.cstr("");
    const unsigned char empty[] = {0, 0, 0, 1, 0};
    CHECK(e.bytes().size() == 5 && std::memcmp(e.bytes().data(), empty, 5) == 0);
}

// test_gpusim.cpp:71-99 (on the CPU route when use_gpu == false)
static void TestSearchMultiple(const std::string& small, const std::string& copy, bool use_gpu)
{
    GPUSimServer server({small, copy}, 0, false, use_gpu);
    const Fingerprint fp = server.getFingerprint(3, "small"); // rand() % 20 == 3 in the reference's process
    int return_count = 10;
    vector<char*> smiles, ids;
    vector<float> scores;
    std::map<std::string, std::string> dbname_to_key;
    dbname_to_key["small"] = "pass";
    dbname_to_key["small_copy"] = "pass";
    unsigned long approximate_result_count = 0;
    server.searchDatabases(fp, return_count, 0, dbname_to_key, smiles, ids, scores, approximate_result_count);
    CHECK_EQ(static_cast<int>(smiles.size()), return_count);
    CHECK_EQ(std::string(ids[0]), std::string("ZINC00000022;:;ZINC00000022"));
    if (use_gpu) CHECK_EQ(approximate_result_count, 200ul);
    for (size_t i = 0; i + 1 < scores.size(); i++) CHECK(scores[i] >= scores[i + 1]);
    // unknown database is skipped, wrong key yields nothing (gpusim.cpp:322-325, fingerprintdb_cuda.cu:349-352)
    vector<char*> s2, i2;
    vector<float> sc2;
    std::map<std::string, std::string> bad;
    bad["nope"] = "pass";
    bad["small"] = "wrong";
    unsigned long ap2 = 0;
    server.searchDatabases(fp, 10, 0, bad, s2, i2, sc2, ap2);
    CHECK(s2.empty());
    // one request frame through the protocol handler
    QdsWriter w;
    w.i32(2);
    w.cstr("small");
    w.cstr("pass");
    w.cstr("small_copy");
    w.cstr("pass");
    w.i32(777);
    w.i32(5);
    w.f64(0.0);
    w.bytearray(reinterpret_cast<const unsigned char*>(fp.data()), fp.size() * sizeof(int));
    const auto reply = server.handleRequest(w.bytes());
    QdsReader r(reply);
    CHECK_EQ(r.i32(), 777);
    const int n = r.i32();
    CHECK_EQ(n, 5);
    r.u64();
    for (int i = 0; i < n; i++) r.cstr();
    CHECK_EQ(r.cstr(), std::string("ZINC00000022;:;ZINC00000022"));
    for (int i = 1; i < n; i++) r.cstr();
    CHECK_EQ(static_cast<float>(r.f64()), 1.0f);
    for (int i = 1; i < n; i++) r.f64();
    CHECK(r.atEnd());
}

// test_gpusim.cpp:29-69
static void CompareGPUtoCPU(const std::string& small)
{
    GPUSimServer server({small}, 0, false, true);
    const Fingerprint fp = server.getFingerprint(3, "small");
    for (int return_count : {10, 15}) {
        unsigned long approx = 0;
        vector<char*> gs, gi, cs, ci;
        vector<float> gsc, csc;
        server.similaritySearch(fp, "small", "pass", return_count, 0, CalcType::GPU, gs, gi, gsc, approx);
        server.similaritySearch(fp, "small", "pass", return_count, 0, CalcType::CPU, cs, ci, csc, approx);
        CHECK_EQ(static_cast<int>(gs.size()), return_count);
        CHECK_EQ(gs.size(), cs.size());
        for (size_t i = 0; i < gs.size() && i < cs.size(); i++) {
            CHECK_EQ(gs[i], cs[i]); // same pointers, as the reference compares
            CHECK_EQ(gsc[i], csc[i]);
        }
    }
}

// test_gpusim.cpp:101-128
static void TestSimilarityCutoff(const std::string& small)
{
    GPUSimServer server({small}, 0, false, true);
    const Fingerprint fp = server.getFingerprint(0, "small");
    const vector<float> cutoffs = {0, 0.1f, 0.3f, 0.4f};
    const vector<int> result_counts = {10, 10, 3, 1};
    const vector<unsigned long> approximate_counts = {100, 86, 3, 1};
    for (size_t i = 0; i < cutoffs.size(); i++) {
        vector<char*> smiles, ids;
        vector<float> scores;
        unsigned long approx = 0;
        server.similaritySearch(fp, "small", "pass", 10, cutoffs[i], CalcType::GPU, smiles, ids, scores, approx);
        CHECK_EQ(static_cast<int>(smiles.size()), result_counts[i]);
        CHECK_EQ(approx, approximate_counts[i]);
    }
}

// test_gpusim.cpp:168-181
static void getNextGPU()
{
    const unsigned int gpucount = get_gpu_count();
    CHECK(gpucount >= 1);
    vector<unsigned int> first, second;
    for (unsigned int i = 0; i < gpucount; i++) first.push_back(get_next_gpu(1));
    for (unsigned int i = 0; i < gpucount; i++) second.push_back(get_next_gpu(1));
    CHECK(first == second);
    vector<bool> seen(gpucount, false);
    for (auto d : first) seen[d] = true;
    for (bool s : seen) CHECK(s);
}

int main(int argc, char** argv)
{
    if (argc < 4) {
        std::fprintf(stderr, "usage: %s cpu|gpu small.fsim small_copy.fsim\n", argv[0]);
        return 2;
    }
    const std::string mode = argv[1], small = argv[2], copy = argv[3];
    try {
        CPUSort();
        LoadFsim(small);
        Codec();
        if (mode == "cpu") {
            TestSearchMultiple(small, copy, false);
        } else {
            if (get_gpu_count() == 0) {
                std::fprintf(stderr, "no GPU\n");
                return 3;
            }
            getNextGPU();
            CompareGPUtoCPU(small);
            TestSearchMultiple(small, copy, true);
            TestSimilarityCutoff(small);
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "EXCEPTION: %s\n", e.what());
        return 1;
    }
    std::fprintf(stderr, "%s: %d failed checks\n", mode.c_str(), g_fail);
    return g_fail ? 1 : 0;
}
