// capi_batch.cpp -- multi-query batches: nq >= 4 queries share table passes (the matrix-core contraction for
// 256..2048-bit rows, the VALU pass otherwise); every query gets exactly the result a call with nq = 1 would return.
#include "capi_internal.h"

namespace gsim_host
{

// Buffers of the multi-query path, sized for kBatchMaxQ queries and result blocks of k hits.
int ensure_batch_buffers(gsim_db* db, Shard& s, uint32_t k)
{
    GSIM_HIP(set_device(s.device));
    if (s.bq_cap == 0) {
        const int wpc = db->knobs.batch_waves_per_cu;
        s.bgeo = gsim::scan_geometry(s.nrows, s.W, s.num_cus, wpc, 8, db->knobs.scan_ragged != 0);
        const uint64_t nchunks = (s.nrows + 63) / 64;
        uint64_t nw = static_cast<uint64_t>(s.num_cus) * static_cast<uint64_t>(wpc);
        if (nw > nchunks) nw = nchunks ? nchunks : 1;
        nw = (nw + 3) / 4 * 4;
        s.bgeo.nwaves = static_cast<uint32_t>(nw);
        // candidate slots per wave: the worst case (every pair a candidate) when that is small, else 4 Ki entries to start
        // with (134 MB for a 256-CU grid; round 3 allocated the 64 Ki worst case up front: ~2.1 GB): a wave that needs more
        // reports how many (flags[15]) and the host grows the segments -- up to 64 Ki -- and runs the batch again
        // (run_batch); only beyond that do the queries fall back to the single-query path
        const uint64_t rows_per_wave = ((nchunks + nw - 1) / nw) * 64 + 256; // chunks are 64 x (1..4) rows
        uint64_t cap = rows_per_wave * gsim::kBQ;
        // a wave of the matrix-core pass meets 32 queries and every row of its workgroup
        const uint64_t mfma_waves = gsim::batch_mfma_waves(s.num_cus);
        if (gsim::batch_mfma_supported(s.W)) {
            cap = std::max<uint64_t>(cap, (s.nrows / static_cast<uint64_t>(s.num_cus) + 512) * 32);
            nw = std::max<uint64_t>(nw, mfma_waves);
        }
        const uint64_t lim = static_cast<uint64_t>(db->knobs.batch_seg_cap);
        if (cap > lim) cap = lim;
        if (cap < 256) cap = 256;
        s.bseg_max = static_cast<uint32_t>(cap);
        const uint64_t init = static_cast<uint64_t>(db->knobs.batch_seg_cap_init);
        if (cap > init) cap = init < 16 ? 16 : init;
        s.bseg_cap = static_cast<uint32_t>(cap);
        s.bseg_waves = static_cast<uint32_t>(nw);
        const size_t slots = static_cast<size_t>(nw) * cap;
        GSIM_HIP(hipMalloc(&s.d_bqueries, static_cast<size_t>(kBatchMaxQ) * s.W * 4));
        GSIM_HIP(hipMalloc(&s.d_bqpop, kBatchMaxQ * 4));
        GSIM_HIP(hipMalloc(&s.d_bstate, sizeof(gsim::BatchQueryState) * kBatchMaxQ));
        GSIM_HIP(hipMalloc(&s.d_bcand, slots * 8));
        GSIM_HIP(hipMalloc(&s.d_bcand_cb, slots * 4));
        GSIM_HIP(hipMalloc(&s.d_bcand_q, slots * 4));
        GSIM_HIP(hipMalloc(&s.d_bseg_count, nw * 4));
        GSIM_HIP(hipMalloc(&s.d_bfin_key, static_cast<size_t>(kBatchMaxQ) * gsim::kSelectCap * 8));
        GSIM_HIP(hipMalloc(&s.d_bfin_cb, static_cast<size_t>(kBatchMaxQ) * gsim::kSelectCap * 4));
        GSIM_HIP(hipMalloc(&s.d_bflags, 64));
        GSIM_HIP(hipMalloc(&s.d_brare, sizeof(gsim::BatchRare)));
        GSIM_HIP(hipHostMalloc(&s.h_brare, sizeof(gsim::BatchRare), kHostPinned));
        GSIM_HIP(hipHostMalloc(&s.h_bflags, 64, kHostPinned));
        GSIM_HIP(hipHostMalloc(&s.h_bqueries, static_cast<size_t>(kBatchMaxQ) * (s.W + 1) * 4, kHostPinned));
        s.bq_cap = kBatchMaxQ;
    }
    const size_t need = gsim_result_block_bytes(k) * kBatchMaxQ;
    if (need > s.h_bresult_bytes) {
        if (s.h_bresult) GSIM_HIP(hipHostFree(s.h_bresult));
        if (s.d_bresult) GSIM_HIP(hipFree(s.d_bresult));
        s.h_bresult = nullptr;
        s.d_bresult = nullptr;
        GSIM_HIP(hipHostMalloc(&s.h_bresult, need, kHostPinned));
        GSIM_HIP(hipMalloc(&s.d_bresult, need));
        s.h_bresult_bytes = need;
    }
    return GSIM_OK;
}

// Enqueue nq (<= kBatchMaxQ) queries on one shard: ceil(nq / kBQ) passes over the table (one on the
// matrix cores), the result blocks land in `results` (device memory; NULL = the shard's pinned host
// block array s.h_bresult, through s.d_bresult).  No host synchronisation.
int enqueue_batch(gsim_db* db, Shard& s, const uint32_t* queries, uint32_t nq, uint32_t k, float cutoff, int metric,
                  float alpha, float beta, uint32_t row_base, void* results, bool allow_mfma)
{
    int rc = ensure_batch_buffers(db, s, k);
    if (rc != GSIM_OK) return rc;
    // NULL: blocks go to device memory and then, in one copy, to the pinned array (256 blocks of
    // 12 KB written by the select kernel straight over PCIe cost ~5 ms per batch)
    const bool to_host = results == nullptr;
    if (to_host) results = s.d_bresult; // (allocated or grown just above)
    GSIM_HIP(set_device(s.device));
    const size_t qbytes = static_cast<size_t>(nq) * s.W * 4;
    std::memcpy(s.h_bqueries, queries, qbytes);
    uint32_t* hp = s.h_bqueries + static_cast<size_t>(nq) * s.W;
    for (uint32_t q = 0; q < nq; q++) hp[q] = popcount_words(queries + static_cast<size_t>(q) * s.W, s.W);
    GSIM_HIP(hipMemcpyAsync(s.d_bqueries, s.h_bqueries, qbytes, hipMemcpyHostToDevice, s.stream));
    GSIM_HIP(hipMemcpyAsync(s.d_bqpop, hp, static_cast<size_t>(nq) * 4, hipMemcpyHostToDevice, s.stream));
    GSIM_HIP(hipMemsetAsync(s.d_bstate, 0, sizeof(gsim::BatchQueryState) * nq, s.stream));
    GSIM_HIP(hipMemsetAsync(s.d_bflags, 0, 64, s.stream));
    gsim::BatchRare& rr = *s.h_brare;
    rr.qstate = s.d_bstate;
    rr.cand = s.d_bcand;
    rr.cand_cb = s.d_bcand_cb;
    rr.cand_q = s.d_bcand_q;
    rr.seg_count = s.d_bseg_count;
    rr.fin_key = s.d_bfin_key;
    rr.fin_cb = s.d_bfin_cb;
    rr.flags = s.d_bflags;
    rr.ticket = s.d_bflags + 1;
    rr.seg_cap = s.bseg_cap;
    rr.pad = 0;
    GSIM_HIP(hipMemcpyAsync(s.d_brare, s.h_brare, sizeof(gsim::BatchRare), hipMemcpyHostToDevice, s.stream));
    gsim::BatchArgs a{};
    a.rows = s.d_rows;
    a.nrows = s.nrows;
    a.W = s.W;
    a.queries = s.d_bqueries;
    a.qpop = s.d_bqpop;
    a.rare = s.d_brare;
    a.k = k;
    a.cutoff = cutoff;
    a.metric = metric;
    a.alpha = alpha;
    a.beta = beta;
    const uint32_t sample = static_cast<uint32_t>(db->knobs.batch_sample_chunks);
    a.opts = (db->knobs.batch_mfma_sample ? 1u : 0u) | (db->knobs.batch_mfma_dense ? 2u : 0u) | ((static_cast<uint32_t>(db->knobs.batch_rpl) & 255u) << 8);
    // One contraction pass on the matrix cores for all of them (gsim_batch_mfma.hip: with fewer
    // than 8 x 32 queries the waves of a workgroup share query tiles and split the rows).  With a
    // cutoff it needs the matrix-core sample pass (large tables), which also estimates how many
    // rows the cutoff keeps: a cutoff that keeps many sets bit 3 of the flags and the kernel leaves
    // the batch to the VALU pass (the callers re-enqueue with allow_mfma = false).
    hipEvent_t* bev = nullptr;
    if (db->timing && s.bev_used < kTimingRing) {
        if (s.bev.size() < static_cast<size_t>(2 * (s.bev_used + 1))) {
            for (int i = 0; i < 2; i++) {
                hipEvent_t e;
                GSIM_HIP(hipEventCreate(&e));
                s.bev.push_back(e);
            }
        }
        bev = &s.bev[2 * s.bev_used];
        s.bev_used++;
    }
    const int mfma_min_q = db->knobs.batch_mfma_min_q;
    if (allow_mfma && mfma_min_q > 0 && nq >= static_cast<uint32_t>(mfma_min_q) &&
        nq <= static_cast<uint32_t>(gsim::kMfmaQueries) && gsim::batch_mfma_supported(s.W) &&
        (!(cutoff > 0.0f) || gsim::batch_mfma_sample_applies(s.W, s.nrows, nq, k, s.num_cus, db->knobs.batch_mfma_sample != 0))) {
        a.q0 = 0;
        a.nq = nq;
        // the rows' popcounts: once per table; borrowed rows (gsim_db_attach_device_rows) may have changed since the
        // last call, so theirs are recounted every time (one more read of the table)
        if (!s.d_rowpop) GSIM_HIP(hipMalloc(reinterpret_cast<void**>(&s.d_rowpop), gsim::row_popcount_bytes(s.nrows)));
        if (!s.rowpop_valid || !s.owns_rows) {
            GSIM_HIP(gsim::launch_row_popcounts(s.d_rows, s.nrows, s.W, s.d_rowpop, s.stream));
            s.rowpop_valid = true;
        }
        a.rowpop = s.d_rowpop;
        GSIM_HIP(gsim::launch_batch_mfma_pass(a, s.bgeo, s.num_cus, sample, row_base, results,
                                              gsim_result_block_bytes(k), s.stream, bev ? bev[0] : nullptr,
                                              bev ? bev[1] : nullptr));
        if (to_host)
            GSIM_HIP(hipMemcpyAsync(s.h_bresult, s.d_bresult, gsim_result_block_bytes(k) * nq, hipMemcpyDeviceToHost, s.stream));
        GSIM_HIP(hipMemcpyAsync(s.h_bflags, s.d_bflags, 64, hipMemcpyDeviceToHost, s.stream));
        return GSIM_OK;
    }
    if (bev) GSIM_HIP(hipEventRecord(bev[0], s.stream));
    for (uint32_t q0 = 0; q0 < nq; q0 += gsim::kBQ) {
        a.q0 = q0;
        a.nq = std::min<uint32_t>(gsim::kBQ, nq - q0);
        GSIM_HIP(gsim::launch_batch_pass(a, rr, s.bgeo, sample, row_base, results, gsim_result_block_bytes(k),
                                         s.stream));
    }
    if (bev) GSIM_HIP(hipEventRecord(bev[1], s.stream));
    if (to_host)
        GSIM_HIP(hipMemcpyAsync(s.h_bresult, s.d_bresult, gsim_result_block_bytes(k) * nq, hipMemcpyDeviceToHost, s.stream));
    GSIM_HIP(hipMemcpyAsync(s.h_bflags, s.d_bflags, 64, hipMemcpyDeviceToHost, s.stream));
    return GSIM_OK;
}

// A candidate segment overflowed (flags bit 0; flags[15] = the most slots a wave asked for): larger segments, up to the
// worst case the first allocation was spared.  false: already there.
static int grow_batch_segments(Shard& s, uint32_t wanted, bool* grown)
{
    *grown = false;
    if (s.bseg_cap >= s.bseg_max) return GSIM_OK;
    uint64_t cap = s.bseg_cap;
    const uint64_t target = static_cast<uint64_t>(wanted) + wanted / 4 + 64; // (thresholds move a little from run to run)
    while (cap < target && cap < s.bseg_max) cap *= 2;
    if (cap <= s.bseg_cap) cap = static_cast<uint64_t>(s.bseg_cap) * 2;
    if (cap > s.bseg_max) cap = s.bseg_max;
    GSIM_HIP(set_device(s.device));
    GSIM_HIP(hipStreamSynchronize(s.stream));
    GSIM_HIP(hipFree(s.d_bcand));
    GSIM_HIP(hipFree(s.d_bcand_cb));
    GSIM_HIP(hipFree(s.d_bcand_q));
    s.d_bcand = nullptr, s.d_bcand_cb = nullptr, s.d_bcand_q = nullptr;
    const size_t slots = static_cast<size_t>(s.bseg_waves) * cap;
    GSIM_HIP(hipMalloc(&s.d_bcand, slots * 8));
    GSIM_HIP(hipMalloc(&s.d_bcand_cb, slots * 4));
    GSIM_HIP(hipMalloc(&s.d_bcand_q, slots * 4));
    s.bseg_cap = static_cast<uint32_t>(cap);
    *grown = true;
    return GSIM_OK;
}

// nb <= kBatchMaxQ queries on every shard at once (shard i's result blocks: out_of(i), device memory, or nullptr = the
// shard's pinned host array), then -- per shard -- what has to be repeated: a dense cutoff without a usable band goes to
// the VALU pass, a batch whose candidates overflowed a segment runs again with larger segments.  On return every shard's
// stream is idle and s.h_bflags[0] says what is left for the caller (bit 0: overflow at the largest segments, bit 2:
// too many ties for the multi-query select).
int run_batch(gsim_db* db, const uint32_t* qb, uint32_t nb, uint32_t k, float cutoff, int metric, float alpha, float beta,
              const std::vector<void*>& outs)
{
    const size_t n = db->shards.size();
    std::vector<char> allow(n, 1), todo(n, 1);
    for (int round = 0; round < 8; round++) {
        bool any = false;
        for (size_t i = 0; i < n; i++) {
            Shard& s = db->shards[i];
            if (!todo[i] || s.nrows == 0) continue;
            const int rc = enqueue_batch(db, s, qb, nb, k, cutoff, metric, alpha, beta, db->row_base + static_cast<uint32_t>(s.first_row),
                                         outs[i], allow[i] != 0);
            if (rc != GSIM_OK) return rc;
            any = true;
        }
        if (!any) break;
        for (size_t i = 0; i < n; i++) {
            Shard& s = db->shards[i];
            if (!todo[i] || s.nrows == 0) continue;
            todo[i] = 0;
            GSIM_HIP(set_device(s.device));
            int rc = wait_stream(s.stream);
            if (rc != GSIM_OK) return rc;
            const uint32_t fl = s.h_bflags[0];
            if (fl & 16u) db->dense_batches++; // (the matrix-core pass counted a dense cutoff itself)
            if ((fl & 24u) == 8u && allow[i]) { // the cutoff keeps too many rows for the exact path and has no band: VALU pass
                allow[i] = 0;
                todo[i] = 1;
            } else if (fl & 1u) {
                bool grown = false;
                rc = grow_batch_segments(s, s.h_bflags[15], &grown);
                if (rc != GSIM_OK) return rc;
                if (grown) {
                    db->batch_regrown++;
                    todo[i] = 1;
                }
            }
        }
    }
    return GSIM_OK;
}

// gsim_db_search with nq >= 4 on a batch-capable width: per 256 queries one enqueue per shard, one wait, host merge
// across shards; queries the shared pass could not finish (heavy ties, candidate overflow) go through search_one.
int search_batched(gsim_db* db, const uint32_t* queries, uint32_t nq, uint32_t k, uint32_t kout, float cutoff, int metric, float alpha,
                   float beta, gsim_hit* hits, uint32_t* counts, uint64_t* approx)
{
    int rc = GSIM_OK;
    std::vector<gsim_hit> merged;
    // Multi-query path: kBQ queries share each pass over the table (VALU-bound).
    const size_t blk = gsim_result_block_bytes(k);
    for (uint32_t base = 0; base < nq; base += kBatchMaxQ) {
        const uint32_t nb = std::min<uint32_t>(kBatchMaxQ, nq - base);
        const uint32_t* qb = queries + static_cast<size_t>(base) * db->W;
        std::vector<char> redo(nb, 0);
        rc = run_batch(db, qb, nb, k, cutoff, metric, alpha, beta, std::vector<void*>(db->shards.size(), nullptr));
        if (rc != GSIM_OK) return rc;
        bool overflow = false;
        for (auto& s : db->shards) {
            if (s.nrows == 0) continue;
            if (s.h_bflags[0] & 1u) overflow = true;
            if (db->knobs.debug_batch) { // counters of instrumented builds (GSIM_MF_TIMING)
                std::fprintf(stderr, "batch flags %u dbg", s.h_bflags[0]);
                for (int d = 2; d < 16; d++) std::fprintf(stderr, " %u", s.h_bflags[d]);
                std::fprintf(stderr, "\n");
            }
        }
        for (uint32_t q = 0; q < nb; q++) {
            uint64_t ap = 0;
            merged.clear();
            std::vector<size_t> ends;
            bool bad = overflow;
            for (auto& s : db->shards) {
                if (s.nrows == 0) continue;
                const gsim_result_header* h = reinterpret_cast<const gsim_result_header*>(s.h_bresult + q * blk);
                if (h->flags & 2u) bad = true;
                ap += h->approx;
                const gsim_hit* hh = reinterpret_cast<const gsim_hit*>(h + 1);
                merged.insert(merged.end(), hh, hh + h->count);
                ends.push_back(merged.size());
            }
            if (bad) {
                redo[q] = 1;
                continue;
            }
            const uint32_t n = merge_canonical_lists(merged, ends, k, hits + static_cast<size_t>(base + q) * kout);
            counts[base + q] = n;
            if (approx) approx[base + q] = ap;
        }
        // heavy ties / candidate overflow: those queries go through the single-query path
        for (uint32_t q = 0; q < nb; q++) {
            if (!redo[q]) continue;
            rc = search_one(db, qb + static_cast<size_t>(q) * db->W, k, cutoff, metric, alpha, beta,
                            hits + static_cast<size_t>(base + q) * kout, &counts[base + q],
                            approx ? &approx[base + q] : nullptr, merged);
            if (rc != GSIM_OK) return rc;
        }
    }
    return GSIM_OK;
}

// nb <= kBatchMaxQ queries on one shard through the shared passes, result blocks in device memory at `out`.
int batch_to_device(gsim_db* db, Shard& s, const uint32_t* qb, uint32_t nb, uint32_t k, float cutoff, int metric, float alpha,
                    float beta, uint32_t row_base, unsigned char* out)
{
    const size_t blk = gsim_result_block_bytes(k);
    // (a single-shard handle: row_base is the handle's) the one host synchronisation of a batch: did a query overflow its
    // candidate segment at the largest size, or collect too many ties for the multi-query select (bit 2)?
    int rc = run_batch(db, qb, nb, k, cutoff, metric, alpha, beta, std::vector<void*>(1, out));
    if (rc != GSIM_OK) return rc;
    if ((s.h_bflags[0] & 5u) != 0) { // those cases are rare: the whole chunk goes through the single-query pipeline
        for (uint32_t q = 0; q < nb; q++) {
            rc = enqueue_query(db, s, qb + static_cast<size_t>(q) * db->W, k, cutoff, metric, alpha, beta, row_base, out + q * blk, false);
            if (rc != GSIM_OK) return rc;
        }
    }
    return GSIM_OK;
}

} // namespace gsim_host

using namespace gsim_host;

extern "C" {

int gsim_db_search_batch_device(gsim_db* db, const uint32_t* queries, uint32_t nq, uint32_t k, float cutoff, int metric,
                                float alpha, float beta, void* d_results)
{
    int rc = check_search_args(db, queries, metric);
    if (rc != GSIM_OK) return rc;
    if (!d_results) return fail(GSIM_ERR_INVALID, "d_results is NULL");
    if (db->shards.size() != 1) return fail(GSIM_ERR_STATE, "search_batch_device needs a single-shard handle");
    if (db->fold > 1) return fail(GSIM_ERR_STATE, "search_batch_device does not support folded tables");
    std::lock_guard<std::mutex> guard(db->search_mutex);
    Shard& s = db->shards[0];
    const size_t blk = gsim_result_block_bytes(k);
    unsigned char* out = static_cast<unsigned char*>(d_results);
    const bool batched = nq >= 4 && k <= static_cast<uint32_t>(gsim::kSelectCap) && k > 0 && gsim::batch_supported(db->W) &&
                         s.nrows > 0 && db->knobs.batch != 0;
    for (uint32_t base = 0; base < nq; base += kBatchMaxQ) {
        const uint32_t nb = std::min<uint32_t>(kBatchMaxQ, nq - base);
        const uint32_t* qb = queries + static_cast<size_t>(base) * db->W;
        if (batched) {
            rc = batch_to_device(db, s, qb, nb, k, cutoff, metric, alpha, beta, db->row_base, out + base * blk);
            if (rc != GSIM_OK) return rc;
            continue;
        }
        for (uint32_t q = 0; q < nb; q++) {
            rc = enqueue_query(db, s, qb + static_cast<size_t>(q) * db->W, k, cutoff, metric, alpha, beta, db->row_base,
                               out + (base + q) * blk, false);
            if (rc != GSIM_OK) return rc;
        }
    }
    return GSIM_OK;
}

} // extern "C"
