// main.cpp -- `gpusimserver`: the backend process python/gpusim_server.py spawns.
// Flags as the reference's main.cpp:21-62: --cpu_only, --gpu_bitcount N, positional
// .fsim files; plus --gpus N (shard every table over N GPUs, 0 = all), --merge host|rccl (where the shards' results meet:
// on the host as in the reference, fingerprintdb_cuda.cu:363-380, or through the C ABI's RCCL all-gather) and, in place of a file,
// "synthetic:<rows>[:<kind>[:<bits>]]" (a benchmark table generated in HBM; scripts/server_latency.py).
#include <sys/stat.h>

#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "gpusim_server.h"

namespace
{
gpusim::GPUSimServer* g_server = nullptr;
void on_signal(int)
{
    if (g_server) g_server->stop();
}
} // namespace

int main(int argc, char* argv[])
{
    bool cpu_only = false, rccl_merge = false;
    int gpu_bitcount = 0, ndevices = 1;
    std::vector<std::string> db_fnames;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto int_value = [&](const char* name, int& out) -> bool {
            const std::string prefix = std::string(name) + "=";
            const char* v = nullptr;
            if (a.rfind(prefix, 0) == 0) v = a.c_str() + prefix.size();
            else if (i + 1 < argc) v = argv[++i];
            char* end = nullptr;
            const long x = v ? std::strtol(v, &end, 10) : 0;
            if (!v || !*v || *end) return false;
            out = static_cast<int>(x);
            return true;
        };
        if (a == "--cpu_only") {
            cpu_only = true;
        } else if (a == "--gpu_bitcount" || a.rfind("--gpu_bitcount=", 0) == 0) {
            if (!int_value("--gpu_bitcount", gpu_bitcount)) {
                std::fprintf(stderr, "GPU Bitcount must be an integer\n");
                return 1;
            }
        } else if (a == "--gpus" || a.rfind("--gpus=", 0) == 0) {
            if (!int_value("--gpus", ndevices)) {
                std::fprintf(stderr, "--gpus must be an integer\n");
                return 1;
            }
        } else if (a == "--merge" || a.rfind("--merge=", 0) == 0) {
            const std::string v = a.size() > 8 ? a.substr(8) : (i + 1 < argc ? std::string(argv[++i]) : std::string());
            if (v != "host" && v != "rccl") {
                std::fprintf(stderr, "--merge must be host or rccl\n");
                return 1;
            }
            rccl_merge = v == "rccl";
        } else if (a == "--help" || a == "-h") {
            std::fprintf(stderr, "Arg parsing is only done in a reasonable way in the python gpusim_server.py.  "
                                 "Handling here is very error prone and not intended for direct use.\n");
            return 1;
        } else if (!a.empty() && a[0] == '-') {
            std::fprintf(stderr, "Unknown option '%s'.\n", a.c_str());
            return 1;
        } else {
            db_fnames.push_back(a);
        }
    }
    if (cpu_only && gpu_bitcount != 0) {
        std::fprintf(stderr, "--cpu_only and --gpu_bitcount are incompatible options\n");
        return 1;
    }
    for (const auto& f : db_fnames) {
        if (f.rfind("synthetic:", 0) == 0) continue; // a generated benchmark table, not a file (gpusim_server.cpp)
        struct stat st;
        if (stat(f.c_str(), &st) != 0) {
            std::fprintf(stderr, "File: \" %s \" not found.\n", f.c_str());
            return 1;
        }
    }
    try {
        // the reference constructs (and uploads) first and applies --cpu_only after
        // (main.cpp:64-65); here --cpu_only also skips the upload
        gpusim::GPUSimServer server(db_fnames, gpu_bitcount, true, !cpu_only, ndevices, rccl_merge);
        if (!server.socketOk()) return 1;
        g_server = &server;
        std::signal(SIGINT, on_signal);
        std::signal(SIGTERM, on_signal);
        return server.exec();
    } catch (const std::exception& e) {
        std::fprintf(stderr, "gpusimserver: %s\n", e.what());
        return 1;
    }
}
