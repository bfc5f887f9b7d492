// capi_debug.cpp -- instrumentation: HIP-event timing per handle, running totals, the in-kernel phase profile of
// instrumented runs (GSIM_FUSED_DEBUG), device/host tables for the parity tests.
#include "capi_internal.h"

#include <chrono>

namespace gsim_host
{

// Fold the recorded events of a shard into the handle's accumulators.
int drain_timing(gsim_db* db, Shard& s)
{
    if (s.ev_used == 0 && s.bev_used == 0) return GSIM_OK;
    GSIM_HIP(set_device(s.device));
    GSIM_HIP(hipStreamSynchronize(s.stream));
    for (uint32_t i = 0; i < s.bev_used; i++) {
        float ms = 0.f;
        GSIM_HIP(hipEventElapsedTime(&ms, s.bev[2 * i], s.bev[2 * i + 1]));
        db->acc.batch_kernel_ms_sum += ms;
        db->acc.batches++;
    }
    s.bev_used = 0;
    for (uint32_t i = 0; i < s.ev_used; i++) {
        float scan = 0.f, sel = 0.f;
        GSIM_HIP(hipEventElapsedTime(&scan, s.ev[3 * i], s.ev[3 * i + 1]));
        GSIM_HIP(hipEventElapsedTime(&sel, s.ev[3 * i + 1], s.ev[3 * i + 2]));
        db->acc.scan_ms_sum += scan;
        db->acc.select_ms_sum += sel;
        db->acc.queries++;
    }
    s.ev_used = 0;
    return GSIM_OK;
}

// Running candidate / finalist totals kept on the device by the select kernel.
int read_totals(Shard& s, unsigned long long* ncand, unsigned long long* nfinal, unsigned long long* nredo)
{
    GSIM_HIP(set_device(s.device));
    GSIM_HIP(hipMemcpyAsync(s.h_state, s.d_state, sizeof(gsim::QueryState), hipMemcpyDeviceToHost, s.stream));
    GSIM_HIP(hipStreamSynchronize(s.stream));
    *ncand = s.h_state->ncand_sum;
    *nfinal = s.h_state->nfinal_sum;
    if (nredo) *nredo = s.h_state->redo_sum;
    return GSIM_OK;
}

// Phase profile of the last single-launch query (GSIM_FUSED_DEBUG=1): per-workgroup timestamps -> stderr.
void dump_fused_phases(Shard& s)
{
        const size_t nwg = s.fgeo.nwaves / 4;
        std::vector<unsigned long long> t(nwg * 24 + 8);
        (void) hipStreamSynchronize(s.stream);
        (void) hipMemcpy(t.data(), s.d_dbg, t.size() * 8, hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull;
        for (size_t g = 0; g < nwg; g++) t0 = std::min(t0, t[g * 24]);
        auto stat = [&](int slot, int nslots, double* mn, double* av, double* mx) {
            double lo = 1e30, hi = 0, sum = 0;
            size_t n = 0;
            for (size_t g = 0; g < nwg; g++)
                for (int q = 0; q < nslots; q++) {
                    const unsigned long long v = t[g * 24 + slot + q];
                    if (v < t0 || v - t0 > 100000000ull) continue;
                    const double us = (v - t0) / 100.0;
                    lo = std::min(lo, us), hi = std::max(hi, us), sum += us, n++;
                }
            *mn = n ? lo : 0, *mx = hi, *av = n ? sum / n : 0;
        };
        double a, b, c;
        { // what the workgroups published: the headers stay as the query left them
            std::vector<uint32_t> hd(nwg * 4);
            (void) hipMemcpy(hd.data(), s.d_hdr, hd.size() * 4, hipMemcpyDeviceToHost);
            unsigned long long rows = 0;
            uint32_t unsorted = 0, most = 0;
            for (size_t g = 0; g < nwg; g++) {
                const uint32_t n = hd[4 * g] & 0x7FFFFFFFu;
                rows += n, most = std::max(most, n), unsorted += (hd[4 * g] >> 31) ? 0u : 1u;
            }
            std::fprintf(stderr, "published: %llu rows by %zu workgroups (most: %u; %u lists not in order)\n", rows, nwg, most, unsorted);
        }
        std::fprintf(stderr, "fused phases, us after the first workgroup started (min/avg/max over workgroups):\n");
        const char* names[] = {"start", "scan-end(w0)", "compacted", "published", "sel:all-arrived", "sel:filtered", "sel:ranked", "sel:fenced",
                               "tau-first-seen", "ckpt0-done", "elect-start", "elect-end"};
        for (int i = 0; i < 12; i++) {
            stat(i, 1, &a, &b, &c);
            std::fprintf(stderr, "  %-16s %8.2f %8.2f %8.2f\n", names[i], a, b, c);
        }
        stat(16, 1, &a, &b, &c);
        std::fprintf(stderr, "  %-16s %8.2f %8.2f %8.2f\n", "elect-loaded", a, b, c);
        stat(17, 1, &a, &b, &c);
        std::fprintf(stderr, "  %-16s %8.2f %8.2f %8.2f\n", "3/4-ckpt(w0)", a, b, c);
        stat(23, 1, &a, &b, &c);
        std::fprintf(stderr, "  %-16s %8.2f %8.2f %8.2f\n", "sel:elected", a, b, c);
        stat(22, 1, &a, &b, &c);
        if (c > 0) std::fprintf(stderr, "  %-16s %8.2f %8.2f %8.2f\n", "sel:elected-2nd", a, b, c);
        stat(12, 4, &a, &b, &c);
        std::fprintf(stderr, "  %-16s %8.2f %8.2f %8.2f\n  end %.2f\n", "wave scan-end", a, b, c, (t[nwg * 24] - t0) / 100.0);
        { // streaming end per workgroup class: blockIdx % 8 (the XCD a block lands on) and blockIdx / 32 (dispatch order)
            double sx[8] = {}, sq[8] = {};
            int nx[8] = {}, nqd[8] = {};
            for (size_t g = 0; g < nwg; g++) {
                const unsigned long long v = t[g * 24 + 3];
                if (v < t0 || v - t0 > 100000000ull) continue;
                sx[g % 8] += (v - t0) / 100.0, nx[g % 8]++;
                const size_t oct = g * 8 / nwg;
                sq[oct] += (v - t0) / 100.0, nqd[oct]++;
            }
            for (int slot : {18, 19, 20, 21, 22, 17, 1}) { // checkpoints after 4 ... 1024 trips, the 3/4 checkpoint, the end of streaming
                double s8[8] = {};
                int n8[8] = {};
                for (size_t g = 0; g < nwg; g++) {
                    const unsigned long long v = t[g * 24 + slot];
                    if (v < t0 || v - t0 > 100000000ull) continue;
                    s8[g % 8] += (v - t0) / 100.0, n8[g % 8]++;
                }
                static const char* const what[] = {"4 trips", "16 trips", "64 trips", "256 trips", "1024 trips"};
                std::fprintf(stderr, "  %-14s mean by blockIdx %% 8:", slot == 17 ? "3/4 checkpoint" : slot == 1 ? "scan end" : what[slot - 18]);
                for (int i = 0; i < 8; i++) std::fprintf(stderr, " %7.1f", n8[i] ? s8[i] / n8[i] : 0.0);
                std::fprintf(stderr, "\n");
            }
            std::fprintf(stderr, "  arrived, mean by blockIdx %% 8:");
            for (int i = 0; i < 8; i++) std::fprintf(stderr, " %7.1f", nx[i] ? sx[i] / nx[i] : 0.0);
            std::fprintf(stderr, "\n  arrived, mean by blockIdx octile:");
            for (int i = 0; i < 8; i++) std::fprintf(stderr, " %7.1f", nqd[i] ? sq[i] / nqd[i] : 0.0);
            std::fprintf(stderr, "\n");
        }
        (void) hipMemset(s.d_dbg, 0, t.size() * 8);
}

} // namespace gsim_host

using namespace gsim_host;

extern "C" {

int gsim_db_enable_timing(gsim_db* db, int enable)
{
    if (!db) return fail(GSIM_ERR_INVALID, "db is NULL");
    std::lock_guard<std::mutex> guard(db->search_mutex);
    db->timing = enable != 0;
    db->acc = gsim_timing{};
    auto rebase = [&](Shard& s) -> int {
        int rc = drain_timing(db, s);
        if (rc != GSIM_OK) return rc;
        unsigned long long c = 0, f = 0;
        unsigned long long r = 0;
        rc = read_totals(s, &c, &f, &r);
        if (rc != GSIM_OK) return rc;
        s.base_ncand = c;
        s.base_nfinal = f;
        s.base_nredo = r;
        return GSIM_OK;
    };
    for (auto& s : db->shards) {
        int rc = rebase(s);
        for (auto& l : s.lanes)
            if (rc == GSIM_OK) rc = rebase(l);
        if (rc != GSIM_OK) return rc;
    }
    db->acc = gsim_timing{};
    return GSIM_OK;
}

int gsim_db_get_timing(gsim_db* db, gsim_timing* out)
{
    if (!db || !out) return fail(GSIM_ERR_INVALID, "NULL argument");
    std::lock_guard<std::mutex> guard(db->search_mutex);
    db->acc.candidates_sum = 0;
    db->acc.finalists_sum = 0;
    db->acc.handed_back = 0;
    db->acc.handed_back_why = 0;
    auto fold = [&](Shard& s) -> int {
        int rc = drain_timing(db, s);
        if (rc != GSIM_OK) return rc;
        unsigned long long c = 0, f = 0;
        unsigned long long r = 0;
        rc = read_totals(s, &c, &f, &r);
        if (rc != GSIM_OK) return rc;
        db->acc.candidates_sum += c - s.base_ncand;
        db->acc.finalists_sum += f - s.base_nfinal;
        db->acc.handed_back += r - s.base_nredo;
        db->acc.handed_back_why |= s.h_state->redo_why & (31u | gsim::kRedoBinTies); // (bit 5 = "a selector saw it fail": not a reason of its own)
        return GSIM_OK;
    };
    for (auto& s : db->shards) {
        int rc = fold(s);
        for (auto& l : s.lanes) // (the half-grid lanes of gsim_db_search_each on small tables: own state, own events)
            if (rc == GSIM_OK) rc = fold(l);
        if (rc != GSIM_OK) return rc;
    }
    db->acc.lane_queries = db->lane_queries;
    db->acc.batches_dense_cutoff = db->dense_batches;
    db->acc.blocks_rechecked = db->blocks_rechecked;
    db->acc.blocks_torn = db->blocks_torn;
    db->acc.batches_regrown = db->batch_regrown;
    db->acc.large_k_single_scan = db->large_k_published.load();
    db->acc.rerun_own = db->rerun_own;
    db->acc.rerun_publish = db->rerun_publish;
    db->acc.rerun_behind = db->rerun_behind;
    db->acc.rerun_torn = db->rerun_torn;
    db->acc.backoff_skips = db->backoff_skips;
    *out = db->acc;
    return GSIM_OK;
}

int gsim_debug_query_flags(gsim_db* db, uint8_t* flags, uint32_t n, uint32_t* written)
{
    if (!db || (!flags && n)) return fail(GSIM_ERR_INVALID, "NULL argument");
    std::lock_guard<std::mutex> guard(db->search_mutex);
    const uint32_t m = static_cast<uint32_t>(std::min<size_t>(n, db->query_flags.size()));
    if (m) std::memcpy(flags, db->query_flags.data(), m);
    if (written) *written = m;
    return GSIM_OK;
}

// The litmus kernels of gsim_litmus.hip.  test 1: 16-byte sc1 stores against 16-byte sc1 loads of another workgroup; 2: 16-byte system-scope
// stores into pinned host memory against a host that polls one word and reads the rest (finish_query_sync's way); 3: entry then header,
// how often the header is visible first and whether a re-read always finds the entry (4: the entry stored by another wave, a barrier between).  `workgroups` (even; 256 = one per CU), `iterations`
// stores per slot.  stats[8]: loads, torn values, headers seen, entries behind their header at the first read, entries that never caught up,
// re-reads, readers / writers that ran out of time, stores.
int gsim_debug_litmus(int device, int test, uint32_t workgroups, uint32_t iterations, unsigned long long* stats)
{
    if (!stats || test < 1 || test > 4 || workgroups < 2 || (workgroups & 1u) || workgroups > 4096 || iterations == 0) return fail(GSIM_ERR_INVALID, "litmus: bad argument");
    int ndev = 0;
    gsim_device_count(&ndev);
    if (device < 0 || device >= ndev) return fail(GSIM_ERR_NO_DEVICE, "device index out of range");
    GSIM_HIP(set_device(device));
    const unsigned long long budget = 3000000000ull; // 30 s of the 100 MHz clock: a test that needs it has failed
    unsigned long long* d_stats = nullptr;
    GSIM_HIP(hipMalloc(reinterpret_cast<void**>(&d_stats), 8 * sizeof(unsigned long long)));
    hipError_t e = hipMemset(d_stats, 0, 8 * sizeof(unsigned long long));
    void *d_a = nullptr, *d_b = nullptr, *h_slots = nullptr;
    uint32_t* h_ack = nullptr;
    unsigned long long host_obs = 0, host_torn = 0;
    if (e == hipSuccess && test != 2) {
        const size_t bytes = static_cast<size_t>(workgroups) * 32 * 16;
        e = hipMalloc(&d_a, bytes);
        if (e == hipSuccess) e = hipMalloc(&d_b, bytes);
        if (e == hipSuccess) e = hipMemset(d_a, 0, bytes);
        if (e == hipSuccess) e = hipMemset(d_b, 0, bytes);
        if (e == hipSuccess) e = gsim::launch_litmus_pair(d_a, d_b, d_stats, workgroups, iterations, test == 3 ? 1 : (test == 4 ? 2 : 0), budget, nullptr);
        if (e == hipSuccess) e = hipDeviceSynchronize();
    } else if (e == hipSuccess) {
        e = hipHostMalloc(&h_slots, static_cast<size_t>(workgroups) * 16, kHostPolled);
        if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&h_ack), static_cast<size_t>(workgroups) * 4, kHostPolled);
        if (e == hipSuccess) {
            std::memset(h_slots, 0, static_cast<size_t>(workgroups) * 16);
            std::memset(h_ack, 0, static_cast<size_t>(workgroups) * 4);
            e = gsim::launch_litmus_host(h_slots, h_ack, d_stats, workgroups, iterations, budget, nullptr);
        }
        if (e == hipSuccess) {
            volatile uint32_t* w = static_cast<volatile uint32_t*>(h_slots);
            std::vector<uint32_t> last(workgroups, 0);
            uint32_t finished = 0;
            const auto t0 = std::chrono::steady_clock::now();
            while (finished < workgroups) {
                for (uint32_t b = 0; b < workgroups; b++) {
                    if (last[b] == iterations) continue;
                    const uint32_t it = w[4 * b + 1]; // the polled word (a header's flags | epoch)
                    if (it == last[b]) continue;
                    std::atomic_thread_fence(std::memory_order_acquire);
                    const uint32_t y = w[4 * b + 0], z = w[4 * b + 2], x3 = w[4 * b + 3];
                    host_obs++;
                    if (y != it * 0x9E3779B1u + b || z != (it ^ 0xA5A5A5A5u) + b * 0x85EBCA6Bu || x3 != ~it) host_torn++;
                    last[b] = it;
                    __atomic_store_n(&h_ack[b], it, __ATOMIC_RELEASE);
                    if (it == iterations) finished++;
                }
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 60.0) break;
            }
            e = hipDeviceSynchronize();
        }
    }
    if (e == hipSuccess) e = hipMemcpy(stats, d_stats, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    if (test == 2) {
        stats[0] = host_obs;
        stats[1] = host_torn;
    }
    if (d_a) (void) hipFree(d_a);
    if (d_b) (void) hipFree(d_b);
    if (h_slots) (void) hipHostFree(h_slots);
    if (h_ack) (void) hipHostFree(h_ack);
    (void) hipFree(d_stats);
    if (e != hipSuccess) return fail_hip(e, "litmus");
    return GSIM_OK;
}

int gsim_debug_score_table(int device, int metric, float alpha, float beta, uint32_t a, uint32_t max_b,
                           uint32_t max_c, float* out)
{
    if (!out) return fail(GSIM_ERR_INVALID, "out is NULL");
    int ndev = 0;
    gsim_device_count(&ndev);
    if (device < 0 || device >= ndev) return fail(GSIM_ERR_NO_DEVICE, "device index out of range");
    GSIM_HIP(set_device(device));
    const size_t n = static_cast<size_t>(max_b + 1) * (max_c + 1);
    float* d = nullptr;
    GSIM_HIP(hipMalloc(&d, n * sizeof(float)));
    hipError_t e = gsim::launch_score_table(metric, alpha, beta, a, max_b, max_c, d, nullptr);
    if (e == hipSuccess) e = hipMemcpy(out, d, n * sizeof(float), hipMemcpyDeviceToHost);
    (void) hipFree(d);
    if (e != hipSuccess) return fail_hip(e, "score table");
    return GSIM_OK;
}

int gsim_debug_sort_desc(int device, unsigned long long* keys, uint32_t n)
{
    if (!keys) return fail(GSIM_ERR_INVALID, "keys is NULL");
    if (n == 0 || (n & (n - 1)) != 0) return fail(GSIM_ERR_INVALID, "n must be a power of two");
    int ndev = 0;
    gsim_device_count(&ndev);
    if (device < 0 || device >= ndev) return fail(GSIM_ERR_NO_DEVICE, "device index out of range");
    GSIM_HIP(set_device(device));
    unsigned long long* d = nullptr;
    GSIM_HIP(hipMalloc(reinterpret_cast<void**>(&d), static_cast<size_t>(n) * 16));
    hipError_t e = hipMemcpy(d, keys, static_cast<size_t>(n) * 8, hipMemcpyHostToDevice);
    unsigned long long* sorted = d;
    if (e == hipSuccess) e = gsim::launch_sort_desc(d, d + n, n, nullptr, &sorted);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(keys, sorted, static_cast<size_t>(n) * 8, hipMemcpyDeviceToHost);
    (void) hipFree(d);
    if (e != hipSuccess) return fail_hip(e, "sort");
    return GSIM_OK;
}

int gsim_debug_prefilter_constants(int device, int metric, float alpha, float beta, uint32_t max_qa, int has_cutoff,
                                   float cutoff, float* out)
{
    if (!out) return fail(GSIM_ERR_INVALID, "out is NULL");
    if (max_qa > 32768) return fail(GSIM_ERR_INVALID, "max_qa too large");
    const int tv = metric == GSIM_METRIC_TVERSKY ? 1 : 0;
    if (device < 0) {
        gsim::prefilter_table_host(tv, alpha, beta, max_qa, has_cutoff, cutoff, out);
        return GSIM_OK;
    }
    int ndev = 0;
    gsim_device_count(&ndev);
    if (device >= ndev) return fail(GSIM_ERR_NO_DEVICE, "device index out of range");
    GSIM_HIP(set_device(device));
    const size_t n = static_cast<size_t>(max_qa + 1) * (has_cutoff ? 1 : gsim::kBBins) * 4;
    float* d = nullptr;
    GSIM_HIP(hipMalloc(&d, n * sizeof(float)));
    hipError_t e = gsim::launch_prefilter_table(tv, alpha, beta, max_qa, has_cutoff, cutoff, d, nullptr);
    if (e == hipSuccess) e = hipMemcpy(out, d, n * sizeof(float), hipMemcpyDeviceToHost);
    (void) hipFree(d);
    if (e != hipSuccess) return fail_hip(e, "prefilter table");
    return GSIM_OK;
}

} // extern "C"
