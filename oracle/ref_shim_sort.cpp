// ref_shim_sort.cpp -- extern "C" driver for the REFERENCE's own
// gpusim::top_results_bubble_sort (fingerprintdb_cuda.cpp:92-103, compiled in
// place by oracle/build_ref.sh).  TEST INFRASTRUCTURE ONLY; contains no
// reference code.  Declared here rather than via fingerprintdb_cuda.h to keep
// this TU free of the Qt headers that header pulls in.
#include <vector>

namespace gpusim
{
void top_results_bubble_sort(std::vector<int>& indices, std::vector<float>& scores,
                             int number_required);
}

extern "C" void gsref_bubble_sort(int* indices, float* scores, int count,
                                  int number_required)
{
    std::vector<int> idx(indices, indices + count);
    std::vector<float> sc(scores, scores + count);
    gpusim::top_results_bubble_sort(idx, sc, number_required);
    for (int i = 0; i < count; i++) {
        indices[i] = idx[i];
        scores[i] = sc[i];
    }
}
