// capi_query.cpp -- the single-query path of the C ABI: the per-query launch sequence on a shard (single launch, or
// sample -> scan -> compact -> select), completion, the shard fan-out + host merge of FingerprintDB::search
// (fingerprintdb_cuda.cu:341-381), gsim_db_search / _each / _device.
#include "capi_internal.h"

#include <chrono>

namespace gsim_host
{

// Scratch of the four-kernel pipeline (per-wave candidate segments + finalists: the worst case is
// every row a candidate, 24 B per row).  The single-launch path keeps its candidates in LDS and
// needs none of it, so it is only allocated when a query takes the classic pipeline: k above
// kFusedMaxK, widths without a specialised scan, or a query the single-launch path handed back
// (heavy ties, adversarial row orders).
int ensure_classic_scratch(Shard& s)
{
    if (s.classic_ready) return GSIM_OK;
    GSIM_HIP(set_device(s.device));
    const uint64_t slots = static_cast<uint64_t>(s.geo.nwaves) * s.geo.seg_cap;
    GSIM_HIP(hipMalloc(&s.d_cand, static_cast<size_t>(slots) * 8));
    GSIM_HIP(hipMalloc(&s.d_cand_cb, static_cast<size_t>(slots) * 4));
    GSIM_HIP(hipMalloc(&s.d_seg_count, static_cast<size_t>(s.geo.nwaves) * 4));
    if (s.d_final) { // (the publishing route's smaller list: nothing is in flight on it when a classic query is about to be enqueued
        GSIM_HIP(hipStreamSynchronize(s.stream)); //  behind it on the same stream -- but the free must not overtake the kernels)
        GSIM_HIP(hipFree(s.d_final));
        s.d_final = nullptr;
    }
    s.final_cap = next_pow2_u32(slots);
    GSIM_HIP(hipMalloc(&s.d_final, static_cast<size_t>(s.final_cap) * 8));
    GSIM_HIP(hipMalloc(&s.d_final_cb, static_cast<size_t>(s.final_cap) * 4));
    s.classic_ready = true;
    return GSIM_OK;
}

// The finalist list alone, sized by what the single launch can publish (its workgroups' regions): all a synchronous caller's
// large-k query needs behind the publishing launch -- 16 MB, not the four-kernel pipeline's 24 bytes per row.
int ensure_publish_scratch(Shard& s)
{
    if (s.d_final) return GSIM_OK;
    GSIM_HIP(set_device(s.device));
    s.final_cap = next_pow2_u32(static_cast<uint64_t>(s.fgeo.nwaves / 4) * gsim::kFusedRegion);
    GSIM_HIP(hipMalloc(&s.d_final, static_cast<size_t>(s.final_cap) * 8));
    return GSIM_OK;
}

int ensure_result_capacity(Shard& s, uint32_t k)
{
    const size_t need = gsim_result_block_bytes(k);
    if (need > s.result_bytes) {
        GSIM_HIP(set_device(s.device));
        if (s.d_result) GSIM_HIP(hipFree(s.d_result));
        s.d_result = nullptr;
        GSIM_HIP(hipMalloc(&s.d_result, need));
        s.result_bytes = need;
    }
    if (need > s.h_result_bytes) {
        if (s.h_result) GSIM_HIP(hipHostFree(s.h_result));
        s.h_result = nullptr;
        GSIM_HIP(hipHostMalloc(&s.h_result, need, kHostPolled));
        s.h_result_bytes = need;
    }
    return GSIM_OK;
}

bool fused_applies(const gsim_db* db, const Shard& s, uint32_t k)
{
    const int enabled = db->knobs.fused;
    const long long max_rows = db->knobs.fused_max_rows;
    if (!enabled || k == 0 || k > gsim::kFusedMaxK || s.nrows == 0 || !gsim::fused_supported(s.fgeo)) return false;
    // thresholds need >= k summary keys; without them every row is published (tiny tables only)
    if ((gsim::fused_summary_keys(s.fgeo.nwaves, k) == 0 || gsim::fused_final_keys(s.fgeo.nwaves / 4, k) == 0) && s.nrows > 8192) return false;
    return max_rows < 0 || s.nrows <= static_cast<uint64_t>(max_rows);
}

// k in (fused_select_max_k, kFusedPublishMaxK]: the single launch scans and publishes (kFusedPublishOnly), the large-k kernels
// rank what it published.  Rows of 512 bits and more only: the narrow widths want the sampled seed (enqueue_query_impl), and
// their scan is bound by the per-row arithmetic either way.  From k = 2049, not 8193: the selectors' own ranking of thousands of
// rows is issue-bound (one wave per SIMD; profiles/EXPERIMENTS.md) -- k = 4096 at 1 M rows 89 us a query inside the launch, 63
// published and ranked by score bin; k = 8192 103 -> 74 (Morgan-shaped rows 139 -> 76, and nothing is handed back for "too many
// finalists"); at 100 M rows k = 4096 0.856 -> 0.872 of the roofline.
bool fused_publish_applies(const gsim_db* db, const Shard& s, uint32_t k)
{
    if (!db->knobs.fused || !db->knobs.fused_publish || k <= static_cast<uint32_t>(db->knobs.fused_select_max_k) || k > static_cast<uint32_t>(db->knobs.fused_publish_max_k) || s.nrows == 0 ||
        !gsim::fused_supported(s.fgeo))
        return false;
    if (s.fgeo.ragged_words || s.fgeo.ragged_loads) return false;
    if (s.fgeo.lanes_per_row < 4 && !db->knobs.publish_narrow) return false; // (128 / 256-bit rows: seeded like the ranking launch, enqueue_query_impl)
    if (gsim::fused_summary_keys(s.fgeo.nwaves, k, gsim::fused_publish_max_m(k)) == 0) return false; // (no thresholds: every row would be published)
    if (s.nrows < static_cast<uint64_t>(db->knobs.publish_min_rows_per_k) * k) return false; // (a short table: the thresholds come late and most of it is published)
    if (k > 65536u && s.nrows < 32ull * k) return false; // (measured: 1 M rows, k = 100 000 -- a tenth of the table -- 286 us classic, 303 published)
    const long long max_rows = db->knobs.fused_max_rows;
    return max_rows < 0 || s.nrows <= static_cast<uint64_t>(max_rows);
}

namespace
{
// A synchronous query that does not end in the single launch's self-announcing block is waited for through an event of its
// own: the stream holds up to kPipe - 1 later queries of a pipelined call, and waiting for IT to drain idled the device between
// batches of eight (k = 8192 at 1 M rows: 74 us a query for 59 us of kernels).
// The tag of a launch's published lists (FusedArgs::pub_tag): never 0 (what the regions hold when they are allocated), and its
// low 26 bits -- the part a header carries -- never 0 either.
uint32_t next_pub_tag(Shard& s)
{
    do s.pub_tag++;
    while ((s.pub_tag & 0x3FFFFFFu) == 0);
    return s.pub_tag;
}

int record_slot_event(Shard& s, uint32_t pipe_slot)
{
    Shard::PipeSlot& sl = s.slot[pipe_slot];
    if (!sl.ev) GSIM_HIP(hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming));
    GSIM_HIP(hipEventRecord(sl.ev, s.stream));
    sl.ev_set = true;
    return GSIM_OK;
}

// Enqueue one query on one shard; the result block ends up at `out`, which is
// device memory or device-visible pinned host memory (zero-copy).
//
// Single-launch path (fused_applies): ONE kernel does scan + publish + select.  Synchronous
// callers (caller_syncs) get the query's epoch stored into s.h_done when the block is complete
// and check header flag 2 ("handed back": re-run with mode kClassic).  Enqueue-only callers
// (the RCCL path) get the four classic kernels enqueued behind it, gated on QueryState::redo:
// they return at once unless the single launch handed the query back.
//
// Classic path: sample -> scan -> compact -> select.  The query is read by the kernels straight
// from a pinned ring slot (no upload op) and the last kernel re-zeroes the per-query state (no
// memset op).  Nothing here synchronises with the host, whatever k.
int enqueue_query_impl(gsim_db* db, Shard& s, const uint32_t* query, uint32_t k, float cutoff, int metric, float alpha,
                       float beta, uint32_t row_base, void* out, bool caller_syncs, QueryMode mode, uint32_t pipe_slot)
{
    Shard::PipeSlot& sl = s.slot[pipe_slot]; // (only touched for synchronous callers)
    GSIM_HIP(set_device(s.device));
    if (s.state_dirty) { // a previous enqueue failed half way: the per-query state may not be zero
        GSIM_HIP(hipMemsetAsync(s.d_state, 0, offsetof(gsim::QueryState, redo_why), s.stream)); // (the per-query part)
        if (s.d_lk) GSIM_HIP(hipMemsetAsync(s.d_lk, 0, sizeof(gsim::LargeKState), s.stream));   // (a large-k enqueue may have failed between its radix passes)
        GSIM_HIP(hipMemsetAsync(s.d_summ, 0, kSummBytes, s.stream));
        if (s.d_bincur) GSIM_HIP(hipMemsetAsync(s.d_bincur, 0, static_cast<size_t>(gsim::kScanBins) * 4, s.stream));
        s.state_dirty = false;
    }
    // (k between the knob fused_select_max_k and kFusedMaxK: the single launch could rank it, the publishing route is preferred)
    bool publish = mode == kAuto && fused_publish_applies(db, s, k);
    uint8_t why = 0;
    if (publish && caller_syncs && s.publish_skip) { // (tables that make the publishing launch hand every query back: scanned twice)
        s.publish_skip--;
        publish = false;
        why |= kQSkipPublish;
    }
    bool fused = mode == kAuto && !publish && fused_applies(db, s, k);
    if (fused && caller_syncs && s.fused_skip) {
        s.fused_skip--;
        fused = false;
        why |= kQSkipFused;
    }
    if (why) db->backoff_skips++;
    const bool classic = (!fused && !(publish && caller_syncs)) || !caller_syncs;
    if (classic) {
        const int rc = ensure_classic_scratch(s);
        if (rc != GSIM_OK) return rc;
    } else if (publish) {
        const int rc = ensure_publish_scratch(s);
        if (rc != GSIM_OK) return rc;
    }
    const uint32_t slot = s.q_next++ % kQueryRing;
    uint32_t* hq = s.h_query + static_cast<size_t>(slot) * s.W;
    if (s.q_pending[slot]) { // only set by asynchronous searches
        GSIM_HIP(hipEventSynchronize(s.q_ev[slot]));
        s.q_pending[slot] = false;
    }
    std::memcpy(hq, query, static_cast<size_t>(s.W) * 4); // `query` is already folded for a folded table

    gsim::ScanArgs a{};
    a.rows = s.d_rows;
    a.nrows = s.nrows;
    a.W = s.W;
    a.query = hq; // hipHostMalloc memory: device-visible at the same address
    a.query_dev = s.d_query;
    a.qpop = popcount_words(query, s.W);
    a.k = k;
    a.cutoff = cutoff;
    a.metric = metric;
    a.alpha = alpha;
    a.beta = beta;
    a.cand = s.d_cand;
    a.cand_cb = s.d_cand_cb;
    a.seg_count = s.d_seg_count;
    a.state = s.d_state;
    a.gate = nullptr;
    if (s.geo.lanes_per_row == 0 || s.nrows == 0) {
        // generic-width scan reads the query per word: give it a device copy
        GSIM_HIP(hipMemcpyAsync(s.d_query, hq, static_cast<size_t>(s.W) * 4, hipMemcpyHostToDevice, s.stream));
        a.query = s.d_query;
    }

    hipEvent_t* ev = nullptr;
    if (db->timing && s.ev_used < kTimingRing) {
        if (s.ev.size() < static_cast<size_t>(3 * (s.ev_used + 1))) {
            for (int i = 0; i < 3; i++) {
                hipEvent_t e;
                GSIM_HIP(hipEventCreate(&e));
                s.ev.push_back(e);
            }
        }
        ev = &s.ev[3 * s.ev_used];
    }
    if (caller_syncs) {
        sl.fused = false;
        sl.publish = false;
        sl.binrank = false;
        sl.ev_set = false;
        sl.rerun = false; // (a slot is enqueued again only after it was finished: a flag still set here is stale -- ADVICE r05)
        sl.inflight = true;
        if (mode == kAuto) sl.why = why; // (a re-run, kClassic, keeps what the first run recorded)
    }
    if (fused) {
        gsim::FusedArgs f{};
        f.pub = s.d_pub;
        f.hdr = s.d_hdr;
        f.arrive = s.d_summ + 4096 + kTicketWords;
        f.summ = s.d_summ;
        f.summ_keys = gsim::fused_summary_keys(s.fgeo.nwaves, k);
        f.final_keys = gsim::fused_final_keys(s.fgeo.nwaves / 4, k);
        f.tickets = s.d_summ + 4096;
        f.result = out;
        f.row_base = row_base;
        // Synchronous callers poll the result block's own header: the closing workgroup stores {count, flags | epoch << 8,
        // approx} in ONE 16-byte write when the hits are out (a separate completion word meant waiting for the header's
        // acknowledgement over PCIe first: ~1.3 us per query); finish_query_sync clears the epoch bits again.
        f.done_flag = caller_syncs ? s.h_done + pipe_slot : nullptr; // (non-null = "the caller polls the header")
        f.epoch = ++s.epoch & 0xFFFFFFu;
        if (f.epoch == 0) f.epoch = ++s.epoch & 0xFFFFFFu; // 0: what a clean header holds
        f.pub_tag = next_pub_tag(s);
        if (caller_syncs) static_cast<gsim_result_header*>(out)->flags = 0;
        const int dbg_on = db->knobs.fused_debug;
        if (dbg_on && !s.d_dbg) GSIM_HIP(hipMalloc(reinterpret_cast<void**>(&s.d_dbg), (static_cast<size_t>(s.fgeo.nwaves / 4) * 24 + 8) * 8));
        f.dbg = s.d_dbg;
        // grid-wide waits give up after 2 ms + four scan times at 4 TB/s (only reached when the GPU is shared)
        f.wait_ticks = static_cast<uint32_t>(std::min<uint64_t>(200000ull + static_cast<uint64_t>(s.nrows) * s.W * 4 / 10000ull, 0xFFFFFFFFull));
        const int xflags = db->knobs.fused_flags;
        f.xflags = static_cast<uint32_t>(xflags);
        if (ev) GSIM_HIP(hipEventRecord(ev[0], s.stream));
        // Narrow rows (128 / 256 bits): a wave meets 512 / 256 rows per trip and its LDS store (2048 slots) is full after a
        // few trips -- before the first in-loop threshold has been elected (~12 us) -- so that sparse tables were handed back
        // (1/64 of the queries at 256 bits, nearly all at 128: scanned twice).  A strided sample first (K0: the
        // four-kernel pipeline's own) leaves a valid starting threshold in QueryState::gtau as a coarse BIN; the single
        // launch turns it into a score key (xflags bit 2).
        const int seed_narrow = db->knobs.fused_seed_narrow;
        if (seed_narrow && ((s.geo.lanes_per_row != 0 && s.geo.lanes_per_row <= 2) || s.fgeo.ragged_words) && s.nrows > 1500ull * s.fgeo.nwaves &&
            s.sample_chunks > 0 && k > 0) {
            bool seeded = false;
            GSIM_HIP(gsim::launch_sample(a, s.geo, 1u, s.stream, &seeded, db->knobs.sample_shift)); // (sample_rows_kernel: 64 Ki ... 1 Mi rows, by k and the table)
            if (seeded) f.xflags |= 4u; // (the launcher has its own size rule: the flag says a threshold WAS left in gtau)
        }
        GSIM_HIP(gsim::launch_fused(a, s.fgeo, f, s.stream));
        if (ev) GSIM_HIP(hipEventRecord(ev[1], s.stream));
        if (caller_syncs) {
            sl.fused = true;
            sl.epoch = f.epoch;
            if (ev) {
                GSIM_HIP(hipEventRecord(ev[2], s.stream));
                s.ev_used++;
            }
            return GSIM_OK;
        }
        a.gate = &s.d_state->redo; // the classic kernels behind it run only if it handed the query back
    }
    if (publish) {
        // large k, first half: the single launch's scan with its in-loop thresholds (one read of the table at 0.88 of the HBM
        // roofline instead of the classic scan's 0.82 + compaction), told to stop after publishing; the hand-off kernel makes
        // its lists the finalists.  The classic kernels follow, gated as above: they run only for a query handed back.
        gsim::FusedArgs f{};
        f.pub = s.d_pub;
        f.hdr = s.d_hdr;
        f.arrive = s.d_summ + 4096 + kTicketWords;
        f.summ = s.d_summ;
        f.summ_keys = gsim::fused_summary_keys(s.fgeo.nwaves, k, gsim::fused_publish_max_m(k));
        f.final_keys = 0; // (no end-of-scan reports: nobody elects a final threshold)
        f.tickets = s.d_summ + 4096;
        f.wait_ticks = static_cast<uint32_t>(std::min<uint64_t>(200000ull + static_cast<uint64_t>(s.nrows) * s.W * 4 / 10000ull, 0xFFFFFFFFull));
        f.xflags = (static_cast<uint32_t>(db->knobs.fused_flags) & ~4u) | gsim::kFusedPublishOnly;
        f.pub_tag = next_pub_tag(s);
        if (!s.d_lk) {
            GSIM_HIP(hipMalloc(reinterpret_cast<void**>(&s.d_lk), sizeof(gsim::LargeKState)));
            GSIM_HIP(hipMemsetAsync(s.d_lk, 0, sizeof(gsim::LargeKState), s.stream));
        }
        if (!s.d_bincur) {
            GSIM_HIP(hipMalloc(reinterpret_cast<void**>(&s.d_bincur), static_cast<size_t>(gsim::kScanBins) * 8)); // (cursors + the layout)
            GSIM_HIP(hipMemsetAsync(s.d_bincur, 0, static_cast<size_t>(gsim::kScanBins) * 8, s.stream));
        }
        if (ev) GSIM_HIP(hipEventRecord(ev[0], s.stream));
        // narrow rows: the sampled seed, as for the ranking launch above (a wave's store fills before the first election otherwise)
        if (db->knobs.fused_seed_narrow && s.geo.lanes_per_row != 0 && s.geo.lanes_per_row <= 2 && s.nrows > 1500ull * s.fgeo.nwaves && s.sample_chunks > 0) {
            bool seeded = false;
            GSIM_HIP(gsim::launch_sample(a, s.geo, 1u, s.stream, &seeded, db->knobs.sample_shift));
            if (seeded) f.xflags |= 4u;
        }
        GSIM_HIP(gsim::launch_fused(a, s.fgeo, f, s.stream));
        db->large_k_published++;
        if (caller_syncs) {
            sl.publish = true;
            // The caller reads the block: a hand-back costs a second run, not a wrong answer -- so the finalists are placed by coarse
            // bin and ranked inside their bins (two launches; the radix select + gather + sort are four and a gap).  Tables whose
            // top bins hold more than kBinRankCap rows (ties) hand that back: the next large-k queries take the radix tail.
            // (128 / 256-bit rows score coarsely: from k ~ 10 000 on a top bin holds more than the emission takes -- 7 % of the k = 32 768
            // queries of an 8 M x 128-bit soak were handed back for it; their large k goes straight to the radix tail)
            const uint32_t binrank_max = (s.fgeo.lanes_per_row != 0 && s.fgeo.lanes_per_row <= 2) ? std::min<uint32_t>(8192u, static_cast<uint32_t>(db->knobs.largek_binrank_max_k))
                                                                                                 : static_cast<uint32_t>(db->knobs.largek_binrank_max_k);
            if (db->knobs.largek_binrank && s.binrank_skip == 0 && k <= binrank_max) {
                GSIM_HIP(gsim::launch_fused_binsort(a, f, s.fgeo.nwaves / 4, s.d_final, s.final_cap, s.d_bincur, s.stream));
                if (ev) GSIM_HIP(hipEventRecord(ev[1], s.stream));
                GSIM_HIP(gsim::launch_binrank_emit(a, s.d_final, s.final_cap, s.d_bincur, s.d_lk, row_base, s.nrows, 1u, out, s.stream));
                sl.binrank = true;
                if (ev) {
                    GSIM_HIP(hipEventRecord(ev[2], s.stream));
                    s.ev_used++;
                }
                return record_slot_event(s, pipe_slot);
            }
            if (s.binrank_skip && k <= binrank_max) {
                s.binrank_skip--;
                sl.why |= kQSkipPublish;
                db->backoff_skips++;
            }
        }
        GSIM_HIP(gsim::launch_fused_handoff(a, f, s.fgeo.nwaves / 4, s.d_final, s.final_cap, s.stream));
        a.gate = &s.d_state->redo;
    }
    // (a synchronous caller of the publishing launch learns of a hand-back from the block's header and runs the query again --
    // finish_query_sync -- instead of paying for three gated launches, ~4.5 us each, behind every query)
    const bool scan_classic = !(publish && caller_syncs);
    if (scan_classic && s.nrows > 0 && s.sample_chunks > 0)
        GSIM_HIP(gsim::launch_sample(a, s.geo, static_cast<uint32_t>(s.sample_chunks), s.stream, nullptr, db->knobs.sample_shift));
    if (ev && !fused && !publish) GSIM_HIP(hipEventRecord(ev[0], s.stream));
    if (scan_classic && s.nrows > 0) GSIM_HIP(gsim::launch_scan(a, s.geo, s.stream));
    if (!caller_syncs) { // the ring slot is free once the scan has run
        GSIM_HIP(hipEventRecord(s.q_ev[slot], s.stream));
        s.q_pending[slot] = true;
    }
    if (ev && !fused) GSIM_HIP(hipEventRecord(ev[1], s.stream));
    if (scan_classic && s.nrows > 0) GSIM_HIP(gsim::launch_compact(a, s.geo, s.d_final, s.d_final_cb, s.final_cap, s.stream));
    if (k <= static_cast<uint32_t>(gsim::kSelectCap) && !publish) {
        GSIM_HIP(gsim::launch_select(a, s.d_final, s.d_final_cb, s.final_cap, row_base, out, s.stream));
    } else {
        // large k: the k-th largest finalist key by a radix select on the device (the finalist count never reaches the
        // host: nothing here waits), the keys at or above it gathered and sorted in global memory (sized by k)
        const uint32_t np2 = next_pow2_u32(k);
        if (np2 > s.large_cap) {
            if (s.d_large) GSIM_HIP(hipFree(s.d_large));
            s.d_large = nullptr;
            s.large_cap = 0;
            GSIM_HIP(hipMalloc(reinterpret_cast<void**>(&s.d_large), static_cast<size_t>(np2) * 8));
            s.large_cap = np2;
        }
        if (!s.d_lk) {
            GSIM_HIP(hipMalloc(reinterpret_cast<void**>(&s.d_lk), sizeof(gsim::LargeKState)));
            GSIM_HIP(hipMemsetAsync(s.d_lk, 0, sizeof(gsim::LargeKState), s.stream));
        }
        // (the route by the finalist count of the previous large-k query, left in pinned memory by its kernels: one workgroup in
        // one launch -- two reads of the finalists, the radix passes in LDS over the keys of the boundary bin: select + gather
        // 56 us instead of 87 at k = 10 000 -- up to 32 Ki finalists, the grid's eight passes + gather beyond (52 k finalists, a
        // boundary bin of more than 16 Ki keys: 121 us either way).  A wrong guess is slower, never wrong)
        const int one_block_max = db->knobs.largek_one_block_max;
        static_assert(kPipe <= 15, "the large-k hint word shares h_done's 64-byte block with the pipeline's completion words");
        uint32_t* hint = s.h_done + 15;
        const bool one_block = *static_cast<volatile uint32_t*>(hint) <= static_cast<uint32_t>(one_block_max);
        GSIM_HIP(gsim::launch_largek_select(a, s.d_final, s.final_cap, s.d_lk, s.d_large, np2, hint, one_block, s.stream));
        // (tiles sorted, then positions by counting + the hits + the header + the state's reset in one launch)
        GSIM_HIP(gsim::launch_largek_sort_emit(a, s.d_large, np2, s.d_lk, row_base, s.nrows, 1u | (publish && caller_syncs ? 0x80000000u : 0u), out, s.stream));
    }
    if (ev) {
        GSIM_HIP(hipEventRecord(ev[2], s.stream));
        s.ev_used++;
    }
    return caller_syncs ? record_slot_event(s, pipe_slot) : GSIM_OK;
}
} // namespace

int enqueue_query(gsim_db* db, Shard& s, const uint32_t* query, uint32_t k, float cutoff, int metric, float alpha,
                  float beta, uint32_t row_base, void* out, bool caller_syncs, QueryMode mode, uint32_t pipe_slot)
{
    const int rc = enqueue_query_impl(db, s, query, k, cutoff, metric, alpha, beta, row_base, out, caller_syncs, mode, pipe_slot);
    if (rc != GSIM_OK) s.state_dirty = true;
    return rc;
}

// Wait for a stream: poll for a short while (a query takes ~2 ms and the blocking
// wait's interrupt wake-up costs 10-20 us), then block.
int wait_event(hipEvent_t ev)
{
    for (int i = 0; i < 200000; i++) {
        hipError_t e = hipEventQuery(ev);
        if (e == hipSuccess) return GSIM_OK;
        if (e != hipErrorNotReady) return fail_hip(e, "hipEventQuery");
    }
    GSIM_HIP(hipEventSynchronize(ev));
    return GSIM_OK;
}

int wait_stream(hipStream_t st)
{
    for (int i = 0; i < 200000; i++) {
        hipError_t e = hipStreamQuery(st);
        if (e == hipSuccess) return GSIM_OK;
        if (e != hipErrorNotReady) return fail_hip(e, "hipStreamQuery");
    }
    GSIM_HIP(hipStreamSynchronize(st));
    return GSIM_OK;
}

// Wait for the result block of the last synchronous enqueue on `s` (it was given s.h_result or any
// pinned block `out`).  The single-launch path signals through the pinned epoch word -- the block
// is complete when it changes, a few microseconds before the stream reports the kernel retired;
// a query it handed back (header flag 2) is re-run by the classic kernels here.
int finish_query_sync(gsim_db* db, Shard& s, const uint32_t* query, uint32_t k, float cutoff, int metric, float alpha,
                      float beta, uint32_t row_base, void* out, uint32_t pipe_slot)
{
    Shard::PipeSlot& sl = s.slot[pipe_slot];
    sl.inflight = false;
    const bool backoff = db->knobs.fused_backoff != 0;
    auto run_again = [&]() -> int { // the four-kernel pipeline, waited for
        int rc = enqueue_query(db, s, query, k, cutoff, metric, alpha, beta, row_base, out, true, kClassic, pipe_slot);
        if (rc == GSIM_OK) rc = wait_stream(s.stream);
        sl.inflight = false;
        sl.ev_set = false;
        return rc;
    };
    if (sl.rerun) {
        // Enqueued behind a single launch that ended without closing its query (below): this one ran on per-query state
        // nobody had re-zeroed -- WHATEVER route it took (a classic or publishing query behind the failed launch
        // shared that state too) -- and a block's checksum only covers the block's own hits: run it again.  The stream had
        // drained when the flag was set, so its own kernels are over.
        sl.rerun = false;
        sl.fused = sl.publish = sl.binrank = false;
        sl.why |= kQRerunBehind;
        db->rerun_behind++;
        return run_again();
    }
    if (!sl.fused) {
        int rc = sl.ev_set ? wait_event(sl.ev) : wait_stream(s.stream);
        sl.ev_set = false;
        const bool back = rc == GSIM_OK && sl.publish && (static_cast<const gsim_result_header*>(out)->flags & 2u);
        if (rc == GSIM_OK && sl.publish && !back) {
            s.publish_streak = 0;
            if (sl.binrank) s.binrank_streak = 0;
        }
        if (back) {
            // large k, scanned by the single launch, handed back (heavy ties): the emission cleared the per-query state
            if (sl.binrank) {
                // ties in the top bins, most likely: the next ones by the radix tail -- 16, 32, ... 1024 of them while it keeps
                // happening (a fixed 16 made a tie-heavy table pay a second scan every 17th large-k query for good)
                if (backoff) s.binrank_skip = 16u << s.binrank_streak;
                s.binrank_streak = s.binrank_streak < 6 ? s.binrank_streak + 1 : 6;
            } else { // not the bin-ranked emission's doing: the launch itself cannot hold this table's queries -- back off as the single launch does
                s.publish_streak = s.publish_streak < 6 ? s.publish_streak + 1 : 6;
                if (backoff) s.publish_skip = 1u << s.publish_streak;
            }
            sl.why |= kQPublishBack;
            db->rerun_publish++;
            sl.publish = false;
            return run_again();
        }
        sl.publish = false;
        return rc;
    }
    sl.fused = false;
    volatile uint32_t* flag = &static_cast<gsim_result_header*>(out)->flags; // (flags | epoch << 8: one 16-byte store with the rest of the header)
    const uint32_t want = sl.epoch;
    bool done = false;
    for (uint64_t spins = 0;; spins++) {
        if ((*flag >> 8) == want) {
            done = true;
            break;
        }
        if ((spins & 0x3FFu) == 0x3FFu) { // now and then: did the launch fail or end without the header?
            const hipError_t e = hipStreamQuery(s.stream);
            if (e == hipSuccess) {
                done = (*flag >> 8) == want;
                break;
            }
            if (e != hipErrorNotReady) return fail_hip(e, "hipStreamQuery");
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    gsim_result_header* h = static_cast<gsim_result_header*>(out);
    bool torn = false;
    if (done && !(*flag & 2u)) {
        // The block validates itself: the closing workgroup put the sum of the words of all hits (+ epoch * kBlockCheckMul)
        // into the upper half of approx.  The hits were stored write-through at system scope and every selector waited for
        // their acknowledgements before its ticket, so they are in host memory when the header is -- on the platforms this
        // was soaked on.  Whether an acknowledgement means host visibility is the platform's business: if the
        // sums differ the hits are still on their way -- look again for a while, then let the stream drain; a block that
        // stays wrong is re-run on the four-kernel pipeline (gsim_timing.blocks_rechecked / blocks_torn count both).
        // (test hook, no production use: GSIM_TEST_TORN_EVERY=n makes every n-th block fail its check for good, so that the
        // re-run of a torn block -- with later queries of the call already enqueued behind it -- is exercised)
#ifdef GSIM_TEST_HOOKS
        static const int torn_every = env_int("GSIM_TEST_TORN_EVERY", 0);
        const uint32_t spoil = (torn_every > 0 && ++db->blocks_checked % static_cast<unsigned>(torn_every) == 0) ? 1u : 0u;
#else
        const uint32_t spoil = 0;
#endif
        auto block_ok = [&]() -> bool {
            const uint32_t n = h->count <= k ? h->count : k;
            const volatile uint32_t* w = reinterpret_cast<const volatile uint32_t*>(h + 1);
            uint32_t sum = 0;
            for (uint32_t i = 0; i < 3u * n; i++) sum += w[i];
            return static_cast<uint32_t>(h->approx >> 32) == sum + want * gsim::kBlockCheckMul + spoil;
        };
        bool good = block_ok();
        if (!good) {
            db->blocks_rechecked++;
            for (int spins = 0; spins < (spoil ? 2 : 20000) && !good; spins++) good = block_ok();
            if (!good) {
                (void) hipStreamSynchronize(s.stream);
                std::atomic_thread_fence(std::memory_order_acquire);
                good = block_ok();
            }
            if (!good) {
                db->blocks_torn++;
                torn = true; // (re-run below; the kernel itself closed the query and left the state clean)
            }
        }
    }
    if (done) {
        *flag &= 0xFFu; // the header as every other route leaves it
        h->approx &= 0xFFFFFFFFull;
    }
    if (s.d_dbg && done) dump_fused_phases(s); // phase profile of this query (instrumented runs only)
    if (done && !torn && !(h->flags & 2u)) {
        s.redo_streak = 0;
        return GSIM_OK;
    }
    if (done && !torn) {
        s.redo_streak = s.redo_streak < 6 ? s.redo_streak + 1 : 6;
        if (backoff && s.redo_streak >= 2) s.fused_skip = 1u << s.redo_streak;
        sl.why |= kQHandedBack;
        db->rerun_own++;
    } else if (torn) {
        sl.why |= kQTorn;
        db->rerun_torn++;
    }
    if (!done) {
        // The launch ended without closing the query: the state is re-zeroed, and the later queries of a pipelined call --
        // the stream has drained, so they have all run already, on that state -- are run again, whichever
        // route they took (every slot in flight, not only the single launch's)
        s.state_dirty = true;
        sl.why |= kQRerunBehind;
        db->rerun_behind++;
        for (uint32_t j = 0; j < static_cast<uint32_t>(kPipe); j++)
            if (j != pipe_slot && s.slot[j].inflight) s.slot[j].rerun = true;
    }
    // handed back: the per-query state is zero again (the last selector reset it), `redo` is set
    return run_again();
}

// One query through the single-query pipeline on every shard, host merge across shards
// (FingerprintDB::search, fingerprintdb_cuda.cu:341-381).
int search_one(gsim_db* db, const uint32_t* query, uint32_t k, float cutoff, int metric, float alpha, float beta,
               gsim_hit* hits, uint32_t* count, uint64_t* approx, std::vector<gsim_hit>& merged)
{
    const size_t nsh = db->shards.size();
    if (db->comm) return search_one_comm(db, query, k, cutoff, metric, alpha, beta, hits, count, approx);
    for (auto& s : db->shards) {
        int rc = ensure_result_capacity(s, k);
        if (rc != GSIM_OK) return rc;
        // the select kernel writes the block straight into pinned host memory
        rc = enqueue_query(db, s, query, k, cutoff, metric, alpha, beta,
                           db->row_base + static_cast<uint32_t>(s.first_row), s.h_result, true);
        if (rc != GSIM_OK) return rc;
    }
    uint64_t ap = 0;
    merged.clear();
    std::vector<size_t> ends;
    for (auto& s : db->shards) {
        GSIM_HIP(set_device(s.device));
        int rc = finish_query_sync(db, s, query, k, cutoff, metric, alpha, beta,
                                   db->row_base + static_cast<uint32_t>(s.first_row), s.h_result);
        if (rc != GSIM_OK) return rc;
        if (db->query_flags_at < db->query_flags.size()) db->query_flags[db->query_flags_at] |= s.slot[0].why;
        const gsim_result_header* h = reinterpret_cast<const gsim_result_header*>(s.h_result);
        const gsim_hit* hh = reinterpret_cast<const gsim_hit*>(h + 1);
        ap += h->approx;
        if (nsh == 1) {
            std::memcpy(hits, hh, sizeof(gsim_hit) * h->count);
            *count = h->count;
        } else {
            merged.insert(merged.end(), hh, hh + h->count);
            ends.push_back(merged.size());
        }
    }
    if (nsh > 1) *count = merge_canonical_lists(merged, ends, k, hits); // fingerprintdb_cuda.cu:363-380
    if (approx) *approx = ap;
    return GSIM_OK;
}

// Two half-grid lanes of a shard (Shard::lanes), made on first use: the same rows, half the compute units each, own
// stream / per-query state / regions / exchange buffer.
int ensure_lanes(gsim_db* db, Shard& s)
{
    const int nl = std::min(std::max(db->knobs.each_lanes, 2), 4);
    if (!s.lanes.empty()) return GSIM_OK;
    s.lanes.resize(static_cast<size_t>(nl));
    for (auto& l : s.lanes) {
        l.device = s.device;
        l.cu_share = std::max(db->knobs.each_lanes_share, 1); // (the lanes' grids: the CUs divided by this -- not necessarily by the number of lanes)
        l.first_row = s.first_row;
        l.nrows = s.nrows;
        l.W = s.W;
        l.d_rows = s.d_rows;
        l.owns_rows = false;
        const int rc = setup_shard(db, l);
        if (rc != GSIM_OK) {
            for (auto& x : s.lanes) (void) free_shard(x);
            s.lanes.clear();
            return rc;
        }
    }
    return GSIM_OK;
}

// Does a pipelined call alternate this shard's queries between its two lanes?  Only where the single launch ranks the query
// itself (k up to fused_select_max_k) on a table small enough that the chain behind the scan is a large part of the launch.
bool lanes_apply(const gsim_db* db, const Shard& s, uint32_t k, uint32_t nq)
{
    if (db->knobs.each_lanes < 2 || nq < 4 || s.stream != s.own_stream || s.d_dbg || db->knobs.fused_debug) return false;
    if (s.nrows * s.W * 4ull > static_cast<uint64_t>(db->knobs.each_lanes_max_mb) << 20) return false;
    if (s.nrows < 65536) return false; // (tiny tables: the half grid is no smaller than the whole one)
    if (fused_publish_applies(db, s, k)) return db->knobs.each_lanes_publish != 0; // (large k: the launch publishes, two small kernels rank -- per lane)
    return fused_applies(db, s, k);
}

// gsim_db_search_each on a single-shard handle: the queries still run strictly one after the other on the GPU (one
// stream, one per-query state), but up to kPipe of them are enqueued ahead of the one the host is waiting for, each with
// its own pinned result block and completion word -- the next kernel starts when the previous one retires instead of
// after a host round trip (flag seen, hits copied, next launch: ~8 us per query).
// Round 6, small tables (lanes_apply): consecutive queries alternate between the shard's two half-grid lanes, whose
// launches run side by side -- one query's scan overlaps the other's selection.
int search_each_pipelined(gsim_db* db, const uint32_t* queries, uint32_t nq, uint32_t k, uint32_t kout, float cutoff, int metric,
                          float alpha, float beta, gsim_hit* hits, uint32_t* counts, uint64_t* approx)
{
    // every shard of the handle (round 5: a multi-device handle used to pay a host round trip per query -- enqueue on all
    // shards, wait, merge, next query): up to kPipe queries are enqueued ahead on EVERY shard's stream, each with its own
    // pinned block per shard; the host merges query q's blocks while the devices run q + 1 ... q + kPipe - 1
    const size_t nsh = db->shards.size();
    const size_t blk = gsim_result_block_bytes(k);
    std::vector<char> use_lanes(nsh, 0);
    for (size_t i = 0; i < nsh; i++) {
        Shard& s = db->shards[i];
        if (s.nrows && lanes_apply(db, s, k, nq) && ensure_lanes(db, s) == GSIM_OK &&
            (fused_applies(db, s.lanes[0], k) || fused_publish_applies(db, s.lanes[0], k))) // (the half grid's own geometry must carry the route too)
            use_lanes[i] = 1; // (no lanes: the shard itself, as before)
    }
    auto pipe_blocks = [&](Shard& s) -> int {
        GSIM_HIP(set_device(s.device));
        if (blk > s.h_pipe_block) {
            if (s.h_pipe) GSIM_HIP(hipHostFree(s.h_pipe));
            s.h_pipe = nullptr;
            s.h_pipe_block = 0;
            GSIM_HIP(hipHostMalloc(reinterpret_cast<void**>(&s.h_pipe), blk * kPipe, kHostPolled));
            s.h_pipe_block = blk;
        }
        return GSIM_OK;
    };
    for (size_t i = 0; i < nsh; i++) {
        int rc = pipe_blocks(db->shards[i]);
        if (use_lanes[i])
            for (auto& l : db->shards[i].lanes)
                if (rc == GSIM_OK) rc = pipe_blocks(l);
        if (rc != GSIM_OK) return rc;
    }
    // query q of the call on shard i: which search state answers it, and in which of its pipeline slots
    auto state_of = [&](size_t i, uint32_t q) -> Shard& { return use_lanes[i] ? db->shards[i].lanes[q % db->shards[i].lanes.size()] : db->shards[i]; };
    auto slot_of = [&](size_t i, uint32_t q) -> uint32_t { return (use_lanes[i] ? q / static_cast<uint32_t>(db->shards[i].lanes.size()) : q) % kPipe; };
    std::vector<gsim_hit> merged;
    std::vector<size_t> ends;
    uint32_t issued = 0;
    for (uint32_t done = 0; done < nq; done++) {
        for (; issued < nq && issued - done < static_cast<uint32_t>(kPipe); issued++) {
            for (size_t i = 0; i < nsh; i++) {
                if (db->shards[i].nrows == 0) continue;
                Shard& s = state_of(i, issued);
                const uint32_t slot = slot_of(i, issued);
                const int rc = enqueue_query(db, s, queries + static_cast<size_t>(issued) * db->W, k, cutoff, metric, alpha, beta,
                                             db->row_base + static_cast<uint32_t>(s.first_row), s.h_pipe + slot * s.h_pipe_block, true, kAuto, slot);
                if (rc != GSIM_OK) return rc;
                if (use_lanes[i]) db->lane_queries++;
            }
        }
        uint64_t ap = 0;
        merged.clear();
        ends.clear();
        for (size_t i = 0; i < nsh; i++) {
            if (db->shards[i].nrows == 0) continue;
            Shard& s = state_of(i, done);
            const uint32_t slot = slot_of(i, done);
            GSIM_HIP(set_device(s.device));
            void* out = s.h_pipe + slot * s.h_pipe_block;
            const int rc = finish_query_sync(db, s, queries + static_cast<size_t>(done) * db->W, k, cutoff, metric, alpha, beta,
                                             db->row_base + static_cast<uint32_t>(s.first_row), out, slot);
            if (rc != GSIM_OK) return rc;
            if (done < db->query_flags.size()) db->query_flags[done] |= s.slot[slot].why;
            const gsim_result_header* h = static_cast<const gsim_result_header*>(out);
            const gsim_hit* hh = reinterpret_cast<const gsim_hit*>(h + 1);
            ap += h->approx;
            if (nsh == 1) {
                std::memcpy(hits + static_cast<size_t>(done) * kout, hh, sizeof(gsim_hit) * h->count);
                counts[done] = h->count;
            } else {
                merged.insert(merged.end(), hh, hh + h->count);
                ends.push_back(merged.size());
            }
        }
        if (nsh > 1) counts[done] = merge_canonical_lists(merged, ends, k, hits + static_cast<size_t>(done) * kout); // fingerprintdb_cuda.cu:363-380
        if (approx) approx[done] = ap;
    }
    return GSIM_OK;
}

int check_search_args(gsim_db* db, const uint32_t* queries, int metric)
{
    if (!db || !queries) return fail(GSIM_ERR_INVALID, "NULL argument");
    if (!db->finalized) return fail(GSIM_ERR_STATE, "table not finalized (no rows on a GPU)");
    if (metric != GSIM_METRIC_TANIMOTO && metric != GSIM_METRIC_TVERSKY) return fail(GSIM_ERR_INVALID, "unknown metric");
    return GSIM_OK;
}

} // namespace gsim_host

using namespace gsim_host;

extern "C" {

int gsim_db_search(gsim_db* db, const uint32_t* queries, uint32_t nq, uint32_t kout, float cutoff, int metric,
                   float alpha, float beta, gsim_hit* hits, uint32_t* counts, uint64_t* approx)
{
    int rc = check_search_args(db, queries, metric);
    if (rc != GSIM_OK) return rc;
    if ((!hits && kout && nq) || (!counts && nq)) return fail(GSIM_ERR_INVALID, "NULL output");
    std::lock_guard<std::mutex> guard(db->search_mutex);
    // kout is the stride of the caller's hits array; the search itself never asks for more hits than the
    // table has rows (result blocks, pinned buffers and the select's capacity scale with k: a wild count
    // from a client must not size them)
    const uint32_t k = static_cast<uint32_t>(std::min<uint64_t>(kout, db->nrows));
    const size_t nsh = db->shards.size();
    std::vector<gsim_hit> merged;
    db->query_flags.assign(db->timing ? nq : 0, 0); // (gsim_debug_query_flags: how each query of this call was routed / re-run)
    if (db->fold > 1) {
        if (metric != GSIM_METRIC_TANIMOTO) return fail(GSIM_ERR_INVALID, "folded tables support Tanimoto only");
        return search_folded(db, queries, nq, kout, cutoff, hits, counts, approx);
    }
    const bool batched = !g_force_each && nq >= 4 && k <= static_cast<uint32_t>(gsim::kSelectCap) && k > 0 &&
                         gsim::batch_supported(db->W) && db->knobs.batch != 0;
    if (batched)
        return db->comm ? search_batch_comm(db, queries, nq, k, kout, cutoff, metric, alpha, beta, hits, counts, approx)
                        : search_batched(db, queries, nq, k, kout, cutoff, metric, alpha, beta, hits, counts, approx);
    const int pipelined = db->knobs.each_pipeline;
    if (g_force_each && pipelined && nsh >= 1 && !db->comm && nq > 1 && k > 0 && db->nrows > 0 && !db->shards[0].d_dbg &&
        !db->knobs.fused_debug)
        return search_each_pipelined(db, queries, nq, k, kout, cutoff, metric, alpha, beta, hits, counts, approx);
    for (uint32_t q = 0; q < nq; q++) {
        const uint32_t* query = queries + static_cast<size_t>(q) * db->W;
        db->query_flags_at = q;
        rc = search_one(db, query, k, cutoff, metric, alpha, beta, hits + static_cast<size_t>(q) * kout, &counts[q],
                        approx ? &approx[q] : nullptr, merged);
        if (rc != GSIM_OK) return rc;
    }
    return GSIM_OK;
}

int gsim_db_search_timed(gsim_db* db, const uint32_t* queries, uint32_t nq, uint32_t kout, float cutoff, int metric, float alpha,
                         float beta, gsim_hit* hits, uint32_t* counts, uint64_t* approx, double* seconds)
{
    int rc = check_search_args(db, queries, metric);
    if (rc != GSIM_OK) return rc;
    if ((!hits && kout && nq) || (!counts && nq) || (!seconds && nq)) return fail(GSIM_ERR_INVALID, "NULL output");
    if (db->fold > 1) return fail(GSIM_ERR_STATE, "gsim_db_search_timed does not support folded tables");
    std::lock_guard<std::mutex> guard(db->search_mutex);
    const uint32_t k = static_cast<uint32_t>(std::min<uint64_t>(kout, db->nrows));
    std::vector<gsim_hit> merged;
    db->query_flags.assign(db->timing ? nq : 0, 0);
    for (uint32_t q = 0; q < nq; q++) {
        db->query_flags_at = q;
        const auto t0 = std::chrono::steady_clock::now();
        rc = search_one(db, queries + static_cast<size_t>(q) * db->W, k, cutoff, metric, alpha, beta, hits + static_cast<size_t>(q) * kout,
                        &counts[q], approx ? &approx[q] : nullptr, merged);
        if (rc != GSIM_OK) return rc;
        seconds[q] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    return GSIM_OK;
}

int gsim_db_search_each(gsim_db* db, const uint32_t* queries, uint32_t nq, uint32_t k, float cutoff, int metric,
                        float alpha, float beta, gsim_hit* hits, uint32_t* counts, uint64_t* approx)
{
    g_force_each = true;
    const int rc = gsim_db_search(db, queries, nq, k, cutoff, metric, alpha, beta, hits, counts, approx);
    g_force_each = false;
    return rc;
}

int gsim_db_search_device(gsim_db* db, const uint32_t* query, uint32_t k, float cutoff, int metric, float alpha,
                          float beta, void* d_result)
{
    int rc = check_search_args(db, query, metric);
    if (rc != GSIM_OK) return rc;
    if (!d_result) return fail(GSIM_ERR_INVALID, "d_result is NULL");
    if (db->shards.size() != 1) return fail(GSIM_ERR_STATE, "search_device needs a single-shard handle");
    if (db->fold > 1) return fail(GSIM_ERR_STATE, "search_device does not support folded tables");
    std::lock_guard<std::mutex> guard(db->search_mutex);
    Shard& s = db->shards[0];
    return enqueue_query(db, s, query, k, cutoff, metric, alpha, beta, db->row_base, d_result, false);
}

} // extern "C"
