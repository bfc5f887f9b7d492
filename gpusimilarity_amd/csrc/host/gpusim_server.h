// gpusim_server.h -- Qt-free twin of the reference's GPUSimServer (gpusim.h/.cpp):
// loads .fsim databases, serves similarity searches on the local socket
// /tmp/gpusimilarity with the reference's wire protocol, so that the reference's
// unmodified python/gpusim_server.py and gpusim_search.py keep working.
#pragma once

#include <map>
#include <memory>
#include <string>
#include <vector>

#include "fingerprintdb.h"

namespace gpusim
{

enum class CalcType { GPU, CPU }; // gpusim.h

class GPUSimServer
{
  public:
    // gpusim.cpp:87-171.  open_socket = false builds the server without binding
    // /tmp/gpusimilarity (in-process use, like the reference's unit tests).
    // gpu_bitcount: the reference's folding request (:144-151): fold factor =
    // widest fingerprint / gpu_bitcount.
    // ndevices: GPUs a table is sharded over (1 = one GPU per table, round-robin;
    // 0 = every table over all GPUs).
    explicit GPUSimServer(const std::vector<std::string>& database_fnames, int gpu_bitcount = 0,
                          bool open_socket = true, bool use_gpu = true, int ndevices = 1, bool rccl_merge = false);
    ~GPUSimServer();

    // gpusim.cpp:276-293
    void similaritySearch(const Fingerprint& reference, const std::string& dbname, const std::string& dbkey,
                          unsigned int max_return_count, float similarity_cutoff, CalcType calc_type,
                          std::vector<char*>& results_smiles, std::vector<char*>& results_ids,
                          std::vector<float>& results_scores, unsigned long& approximate_result_count);

    // gpusim.cpp:306-374: search several databases, merge by score, fold hits with
    // equal SMILES into one result whose id is the ids joined by ";:;".
    // results_ids are strdup'd (the receiver frees them, :369-370).
    void searchDatabases(const Fingerprint& reference, int results_requested, float similarity_cutoff,
                         std::map<std::string, std::string>& dbname_to_key, std::vector<char*>& results_smiles,
                         std::vector<char*>& results_ids, std::vector<float>& results_scores,
                         unsigned long& approximate_result_count);

    Fingerprint getFingerprint(const int index, const std::string& dbname); // gpusim.cpp:456-459

    void setUseGPU(bool use_gpu) { m_use_gpu = use_gpu; }
    bool usingGPU(); // gpusim.cpp:168-171

    // gpusim.cpp:376-454 as a pure function: one request frame in, one reply frame out.
    std::vector<unsigned char> handleRequest(const std::vector<unsigned char>& request);

    // Event loop (replaces QCoreApplication::exec + QLocalServer signals): accept
    // clients on the socket and answer their requests until stop() or a signal.
    int exec();
    void stop() { m_stop = true; }
    bool socketOk() const { return m_listen_fd >= 0; }
    static std::string socketPath() { return "/tmp/gpusimilarity"; } // gpusim.cpp:257-261

  private:
    std::map<std::string, std::shared_ptr<FingerprintDB>> m_databases;
    bool m_use_gpu = true;
    int m_listen_fd = -1;
    volatile bool m_stop = false;

    bool setupSocket(); // gpusim.cpp:255-274
};

} // namespace gpusim
