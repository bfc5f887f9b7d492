// capi_folded.cpp -- folded tables (fingerprintdb_cuda.cu:168-195, 284-331; fold functor calculation_functors.cpp:22-41):
// the reference's capacity fallback, reproduced exactly -- OR-folded rows on the GPU, candidates re-scored with the full
// fingerprints (on the device when they are resident too).
#include "capi_internal.h"

namespace gsim_host
{

// FoldFingerprintFunctorCPU (calculation_functors.cpp:22-41): bit `pos` of the fingerprint is
// OR-ed into bit `pos % (32 * Wf)`; since 32 * Wf is a multiple of 32 that is word (w % Wf), same
// bit -- i.e. the F consecutive blocks of Wf words are OR-ed together.
void fold_row(const uint32_t* row, uint32_t W, uint32_t F, uint32_t* out)
{
    const uint32_t Wf = W / F;
    for (uint32_t j = 0; j < Wf; j++) out[j] = 0;
    for (uint32_t w = 0; w < W; w++) out[w % Wf] |= row[w];
}

// fold_data (fingerprintdb_cuda.cpp:56-69) over a row range, on all host threads
void fold_rows_mt(const uint32_t* rows, uint64_t nrows, uint32_t W, uint32_t F, uint32_t* out)
{
    unsigned nt = std::thread::hardware_concurrency();
    if (nt == 0) nt = 1;
    if (nt > nrows) nt = nrows ? static_cast<unsigned>(nrows) : 1;
    const uint32_t Wf = W / F;
    const uint64_t per = (nrows + nt - 1) / nt;
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < nt; t++) {
        const uint64_t lo = per * t, hi = std::min<uint64_t>(lo + per, nrows);
        if (lo >= hi) break;
        pool.emplace_back([=] {
            for (uint64_t r = lo; r < hi; r++) fold_row(rows + r * W, W, F, out + r * Wf);
        });
    }
    for (auto& th : pool) th.join();
}

// Search of a folded table, fingerprintdb_cuda.cu:228-339 with m_fold_factor > 1, per storage:
// folded query vs folded rows on the GPU for the k*F*(int)log2(2F) best FOLDED scores (:284-287),
// re-score those with the full fingerprints on the host (:307-314), stable partial bubble sort
// (:315), keep min(k, .) and stop at the first re-scored value below the cutoff (:317-331);
// then FingerprintDB::search's merge over the storages (:363-380).
int search_folded(gsim_db* db, const uint32_t* queries, uint32_t nq, uint32_t kout, float cutoff, gsim_hit* hits,
                  uint32_t* counts, uint64_t* approx)
{
    // kout is only the stride of the caller's hits array: every buffer and the selection are sized by the count clamped
    // to the table (a count from a socket must not size pinned and device blocks; the answer is the same)
    const uint32_t k = static_cast<uint32_t>(std::min<uint64_t>(kout, db->nrows));
    const uint32_t F = db->fold, W = db->W, Wf = W / F;
    int lg = 0;
    while ((1u << (lg + 1)) <= 2 * F) lg++;
    const uint64_t want = static_cast<uint64_t>(k) * F * static_cast<uint64_t>(lg);
    std::vector<uint32_t> fq(Wf);
    std::vector<gsim_hit> merged;
    std::vector<int> idx;
    std::vector<float> sc;
    // The re-score runs on the device when every storage also holds its full fingerprints in HBM (gsim_db_finalize puts
    // them there when they fit) and the candidate list fits the device sort; GSIM_FOLD_RESCORE=host forces the host path.
    const bool force_host = db->knobs.fold_rescore_host != 0;
    bool on_device = !force_host && want <= 65536 && k > 0;
    for (auto& s : db->shards) on_device = on_device && s.d_full != nullptr;
    // the host path for one storage: its folded candidates (in s.h_result) re-scored with the host copy of the full rows
    auto rescore_on_host = [&](Shard& s, const uint32_t* query) {
        const gsim_result_header* h = reinterpret_cast<const gsim_result_header*>(s.h_result);
        const gsim_hit* hh = reinterpret_cast<const gsim_hit*>(h + 1);
        const uint32_t n = h->count;
        idx.resize(n);
        sc.resize(n);
        std::vector<uint16_t> cm(n), pc(n);
        for (uint32_t j = 0; j < n; j++) { // tanimoto_similarity_cpu on the FULL fingerprints (:387-399)
            const uint32_t* d = db->host_rows.data() + (s.first_row + hh[j].row) * W;
            int total = 0, common = 0, pd = 0;
            for (uint32_t w = 0; w < W; w++) {
                const int p2 = __builtin_popcount(d[w]);
                pd += p2;
                total += __builtin_popcount(query[w]) + p2;
                common += __builtin_popcount(query[w] & d[w]);
            }
            idx[j] = static_cast<int>(j);
            sc[j] = static_cast<float>(common) / static_cast<float>(total - common);
            cm[j] = static_cast<uint16_t>(common);
            pc[j] = static_cast<uint16_t>(pd);
        }
        // top_results_bubble_sort(indices, scores, k) (fingerprintdb_cuda.cpp:92-103): k passes of a bubble sort with
        // a strict '>' -- stable, so its first k entries are the first k of a stable descending sort.  That sort is
        // what runs here (O(n log n) instead of O(k n): k = 1000, F = 8 means 32 k candidates x 1000 passes per
        // storage and query); the literal bubble sort only when a NaN score (0/0: two empty fingerprints) is
        // present, for which '>' is not an order and the two would differ.
        bool has_nan = false;
        for (uint32_t j = 0; j < n; j++) has_nan = has_nan || sc[j] != sc[j];
        if (!has_nan) {
            std::stable_sort(idx.begin(), idx.end(), [&](int x, int y) { return sc[x] > sc[y]; });
            std::vector<float> sorted(n);
            for (uint32_t j = 0; j < n; j++) sorted[j] = sc[idx[j]];
            sc.swap(sorted);
        } else {
            for (uint32_t a = 0; a < k && a < n; a++) {
                for (uint32_t b = n - 1; b > a; b--) {
                    if (sc[b] > sc[b - 1]) {
                        std::swap(idx[b], idx[b - 1]);
                        std::swap(sc[b], sc[b - 1]);
                    }
                }
            }
        }
        const uint32_t keep = std::min(k, n);
        for (uint32_t a = 0; a < keep; a++) {
            if (sc[a] < cutoff) break;
            gsim_hit o;
            o.row = db->row_base + static_cast<uint32_t>(s.first_row) + hh[idx[a]].row;
            o.score = sc[a];
            o.common = cm[idx[a]];
            o.popc_db = pc[idx[a]];
            merged.push_back(o);
        }
        return h->approx;
    };
    for (uint32_t q = 0; q < nq; q++) {
        const uint32_t* query = queries + static_cast<size_t>(q) * W;
        fold_row(query, W, F, fq.data());
        const uint32_t qa = popcount_words(query, W);
        std::vector<uint32_t> kshard(db->shards.size());
        for (size_t i = 0; i < db->shards.size(); i++) {
            Shard& s = db->shards[i];
            kshard[i] = static_cast<uint32_t>(std::min<uint64_t>(want, s.nrows));
            int rc = ensure_result_capacity(s, std::max(kshard[i], k));
            if (rc != GSIM_OK) return rc;
            if (!on_device) {
                rc = enqueue_query(db, s, fq.data(), kshard[i], cutoff, GSIM_METRIC_TANIMOTO, 0.f, 0.f, 0, s.h_result, true);
                if (rc != GSIM_OK) return rc;
                continue;
            }
            // device route, all enqueued on the storage's stream: folded search -> candidates' block in device memory ->
            // re-score with the full rows, sort, first k at or above the cutoff -> pinned host block
            GSIM_HIP(set_device(s.device));
            const uint32_t npad = next_pow2_u32(kshard[i] ? kshard[i] : 1);
            if (!s.d_fq) {
                GSIM_HIP(hipMalloc(reinterpret_cast<void**>(&s.d_fq), static_cast<size_t>(W) * 4 + 64));
                GSIM_HIP(hipMalloc(reinterpret_cast<void**>(&s.d_key2), static_cast<size_t>(65536) * 16)); // (keys + the sort's second buffer)
                GSIM_HIP(hipMalloc(reinterpret_cast<void**>(&s.d_cb2), static_cast<size_t>(65536) * 4));
                GSIM_HIP(hipHostMalloc(reinterpret_cast<void**>(&s.h_fq), static_cast<size_t>(W) * 4 + 64, kHostPinned));
            }
            GSIM_HIP(hipStreamSynchronize(s.stream)); // (the pinned staging of the previous query's fingerprint is free)
            std::memcpy(s.h_fq, query, static_cast<size_t>(W) * 4);
            s.h_fq[W] = 0;
            GSIM_HIP(hipMemcpyAsync(s.d_fq, s.h_fq, static_cast<size_t>(W) * 4 + 4, hipMemcpyHostToDevice, s.stream)); // (+ the NaN flag, cleared)
            rc = enqueue_query(db, s, fq.data(), kshard[i], cutoff, GSIM_METRIC_TANIMOTO, 0.f, 0.f, 0, s.d_result, false);
            if (rc != GSIM_OK) return rc;
            GSIM_HIP(gsim::launch_fold_rescore(s.d_result, s.d_full, s.d_fq, W, qa, s.d_key2, s.d_cb2, npad, s.d_fq + W, k, cutoff,
                                               db->row_base + static_cast<uint32_t>(s.first_row), s.h_result, s.stream));
            GSIM_HIP(hipMemcpyAsync(s.h_fq + W, s.d_fq + W, 4, hipMemcpyDeviceToHost, s.stream));
        }
        uint64_t ap = 0;
        merged.clear();
        for (size_t i = 0; i < db->shards.size(); i++) {
            Shard& s = db->shards[i];
            GSIM_HIP(set_device(s.device));
            if (on_device) {
                int rc = wait_stream(s.stream);
                if (rc != GSIM_OK) return rc;
                if (s.h_fq[W] == 0) { // (no NaN among the re-scored candidates: the block in s.h_result is the storage's answer)
                    const gsim_result_header* h = reinterpret_cast<const gsim_result_header*>(s.h_result);
                    const gsim_hit* hh = reinterpret_cast<const gsim_hit*>(h + 1);
                    ap += h->approx;
                    merged.insert(merged.end(), hh, hh + h->count);
                    continue;
                }
                rc = enqueue_query(db, s, fq.data(), kshard[i], cutoff, GSIM_METRIC_TANIMOTO, 0.f, 0.f, 0, s.h_result, true);
                if (rc != GSIM_OK) return rc;
            }
            int rc = finish_query_sync(db, s, fq.data(), kshard[i], cutoff, GSIM_METRIC_TANIMOTO, 0.f, 0.f, 0, s.h_result);
            if (rc != GSIM_OK) return rc;
            ap += rescore_on_host(s, query);
        }
        if (db->shards.size() > 1) std::stable_sort(merged.begin(), merged.end(), hit_before);
        const uint32_t n = static_cast<uint32_t>(std::min<size_t>(merged.size(), k));
        if (n) std::memcpy(hits + static_cast<size_t>(q) * kout, merged.data(), sizeof(gsim_hit) * n);
        counts[q] = n;
        if (approx) approx[q] = ap;
    }
    return GSIM_OK;
}

} // namespace gsim_host

using namespace gsim_host;

extern "C" {

int gsim_fold_fingerprint(const uint32_t* fingerprint, uint32_t words, uint32_t fold_factor, uint32_t* out)
{
    if (!fingerprint || !out) return fail(GSIM_ERR_INVALID, "NULL argument");
    if (fold_factor == 0 || words == 0 || words % fold_factor != 0)
        return fail(GSIM_ERR_INVALID, "fold factor must divide the word count");
    fold_row(fingerprint, words, fold_factor, out);
    return GSIM_OK;
}

} // extern "C"
