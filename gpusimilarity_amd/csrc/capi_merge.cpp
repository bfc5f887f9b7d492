// capi_merge.cpp -- FingerprintDB::search's merge of per-storage results (fingerprintdb_cuda.cu:363-380): on the host
// for the in-process multi-device path, on the device for gathered result blocks.
#include "capi_internal.h"

namespace gsim_host
{

bool hit_before(const gsim_hit& x, const gsim_hit& y)
{
    if (x.score > y.score) return true;
    if (x.score < y.score) return false;
    return x.row < y.row;
}

// FingerprintDB::search's merge (fingerprintdb_cuda.cu:363-380: std::sort of all storages' results, first k kept).  The
// shards' lists arrive in canonical order and their keys are unique, so the first k of the sorted union are the first
// k of a k-way merge: O(k log #lists) instead of sorting #lists x k hits (8 x 1000: ~20 us instead of ~0.4 ms per query).
// `lists` holds the concatenated lists, `ends[i]` the end of list i in it.  Returns the number of hits written.
uint32_t merge_canonical_lists(const std::vector<gsim_hit>& lists, const std::vector<size_t>& ends, uint32_t k, gsim_hit* out)
{
    struct Head {
        size_t pos, end;
    };
    std::vector<Head> heads;
    size_t begin = 0;
    for (size_t e : ends) {
        if (e > begin) heads.push_back({begin, e});
        begin = e;
    }
    auto later = [&](const Head& x, const Head& y) { return hit_before(lists[y.pos], lists[x.pos]); }; // (a max-heap on "comes first")
    std::make_heap(heads.begin(), heads.end(), later);
    uint32_t n = 0;
    while (n < k && !heads.empty()) {
        std::pop_heap(heads.begin(), heads.end(), later);
        Head& h = heads.back();
        out[n++] = lists[h.pos++];
        if (h.pos < h.end) std::push_heap(heads.begin(), heads.end(), later);
        else heads.pop_back();
    }
    return n;
}

} // namespace gsim_host

using namespace gsim_host;

extern "C" {

int gsim_merge_device(int device, void* hip_stream, const void* d_blocks, uint32_t nblocks, size_t block_bytes,
                      uint32_t k, void* d_result)
{
    if (!d_blocks || !d_result || nblocks == 0) return fail(GSIM_ERR_INVALID, "NULL / empty argument");
    if (block_bytes < gsim_result_block_bytes(k)) return fail(GSIM_ERR_INVALID, "block_bytes too small for k");
    GSIM_HIP(set_device(device));
    GSIM_HIP(gsim::launch_merge_batch(d_blocks, nblocks, 1, block_bytes, k, d_result,
                                      static_cast<hipStream_t>(hip_stream)));
    return GSIM_OK;
}

int gsim_merge_device_batch(int device, void* hip_stream, const void* d_blocks, uint32_t nranks, uint32_t nq,
                            size_t block_bytes, uint32_t k, void* d_results)
{
    if (!d_blocks || !d_results || nranks == 0) return fail(GSIM_ERR_INVALID, "NULL / empty argument");
    if (block_bytes < gsim_result_block_bytes(k)) return fail(GSIM_ERR_INVALID, "block_bytes too small for k");
    if (nq == 0) return GSIM_OK;
    GSIM_HIP(set_device(device));
    GSIM_HIP(gsim::launch_merge_batch(d_blocks, nranks, nq, block_bytes, k, d_results,
                                      static_cast<hipStream_t>(hip_stream)));
    return GSIM_OK;
}

int gsim_merge_host(const void* blocks, uint32_t nblocks, size_t block_bytes, uint32_t k, void* result)
{
    if (!blocks || !result || nblocks == 0) return fail(GSIM_ERR_INVALID, "NULL / empty argument");
    if (block_bytes < sizeof(gsim_result_header)) return fail(GSIM_ERR_INVALID, "block_bytes too small");
    std::vector<gsim_hit> all;
    std::vector<size_t> ends;
    bool canonical = true;
    uint64_t approx = 0;
    uint32_t flags = 0;
    for (uint32_t i = 0; i < nblocks; i++) {
        const unsigned char* b = static_cast<const unsigned char*>(blocks) + static_cast<size_t>(i) * block_bytes;
        gsim_result_header h;
        std::memcpy(&h, b, sizeof(h));
        if (sizeof(h) + static_cast<size_t>(h.count) * sizeof(gsim_hit) > block_bytes)
            return fail(GSIM_ERR_INVALID, "result block count exceeds block_bytes");
        const size_t old = all.size();
        all.resize(old + h.count);
        if (h.count) std::memcpy(all.data() + old, b + sizeof(h), static_cast<size_t>(h.count) * sizeof(gsim_hit));
        canonical = canonical && std::is_sorted(all.begin() + static_cast<std::ptrdiff_t>(old), all.end(), hit_before);
        ends.push_back(all.size());
        approx += h.approx;
        flags |= h.flags;
    }
    gsim_result_header out;
    if (canonical) { // blocks made by the search kernels are in canonical order: a k-way merge is the sorted union's head
        std::vector<gsim_hit> head(std::min<size_t>(all.size(), k));
        const uint32_t n = merge_canonical_lists(all, ends, static_cast<uint32_t>(head.size()), head.data());
        head.resize(n);
        all.swap(head);
    } else {
        std::sort(all.begin(), all.end(), hit_before);
    }
    out.count = static_cast<uint32_t>(std::min<size_t>(all.size(), k));
    out.flags = flags;
    out.approx = approx;
    std::memcpy(result, &out, sizeof(out));
    if (out.count)
        std::memcpy(static_cast<unsigned char*>(result) + sizeof(out), all.data(), sizeof(gsim_hit) * out.count);
    return GSIM_OK;
}

} // extern "C"
