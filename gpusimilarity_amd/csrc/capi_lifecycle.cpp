// capi_lifecycle.cpp -- errors, device enumeration / placement, table lifecycle and accessors of the C ABI
// (include/gpusim_hip.h).  Reference: fingerprintdb_cuda.cu:33-68, 111-226, 401-413.
#include "capi_internal.h"

namespace gsim_host
{

thread_local std::string g_last_error;
thread_local bool g_force_each = false;

int fail(int code, const std::string& msg)
{
    g_last_error = msg;
    return code;
}

int fail_hip(hipError_t e, const char* what)
{
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    return GSIM_ERR_HIP;
}

namespace
{
int env_value(const char* name, int dflt)
{
    const char* v = std::getenv(name);
    if (!v || !*v) return dflt;
    return std::atoi(v);
}
} // namespace

gsim::Knobs read_knobs()
{
    gsim::Knobs k;
    k.scan_waves_per_cu = env_value("GSIM_SCAN_WAVES_PER_CU", k.scan_waves_per_cu);
    k.scan_ragged = env_value("GSIM_SCAN_RAGGED", k.scan_ragged);
    k.sample_chunks = env_value("GSIM_SAMPLE_CHUNKS", k.sample_chunks);
    k.sample_shift = env_value("GSIM_SAMPLE_SHIFT", k.sample_shift);
    k.fused = env_value("GSIM_FUSED", k.fused);
    if (const char* v = std::getenv("GSIM_FUSED_MAX_ROWS")) k.fused_max_rows = std::atoll(v);
    k.fused_debug = env_value("GSIM_FUSED_DEBUG", k.fused_debug);
    k.fused_flags = env_value("GSIM_FUSED_FLAGS", k.fused_flags);
    k.fused_seed_narrow = env_value("GSIM_FUSED_SEED_NARROW", k.fused_seed_narrow);
    k.fused_publish = env_value("GSIM_FUSED_PUBLISH", k.fused_publish);
    k.largek_binrank = env_value("GSIM_LARGEK_BINRANK", k.largek_binrank);
    k.fused_select_max_k = std::min(std::max(env_value("GSIM_FUSED_SELECT_MAX_K", k.fused_select_max_k), 2048), static_cast<int>(gsim::kFusedMaxK)); // (the large-k sort takes k > 2048)
    k.largek_one_block_max = env_value("GSIM_LARGEK_ONE_BLOCK_MAX", k.largek_one_block_max);
    k.fused_backoff = env_value("GSIM_FUSED_BACKOFF", k.fused_backoff);
    k.publish_narrow = env_value("GSIM_PUBLISH_NARROW", k.publish_narrow);
    k.largek_binrank_max_k = env_value("GSIM_LARGEK_BINRANK_MAX_K", k.largek_binrank_max_k);
    k.fused_publish_max_k = std::min(std::max(env_value("GSIM_FUSED_PUBLISH_MAX_K", k.fused_publish_max_k), 2048), static_cast<int>(gsim::kFusedPublishMaxK));
    k.publish_min_rows_per_k = std::max(env_value("GSIM_PUBLISH_MIN_ROWS_PER_K", k.publish_min_rows_per_k), 0);
    k.each_pipeline = env_value("GSIM_EACH_PIPELINE", k.each_pipeline);
    k.each_lanes = env_value("GSIM_EACH_LANES", k.each_lanes);
    k.each_lanes_share = env_value("GSIM_EACH_LANES_SHARE", k.each_lanes_share);
    k.each_lanes_publish = env_value("GSIM_EACH_LANES_PUBLISH", k.each_lanes_publish);
    k.each_lanes_max_mb = env_value("GSIM_EACH_LANES_MAX_MB", k.each_lanes_max_mb);
    k.batch = env_value("GSIM_BATCH", k.batch);
    k.batch_waves_per_cu = env_value("GSIM_BATCH_WAVES_PER_CU", k.batch_waves_per_cu);
    k.batch_seg_cap = env_value("GSIM_BATCH_SEG_CAP", k.batch_seg_cap);
    k.batch_seg_cap_init = env_value("GSIM_BATCH_SEG_CAP_INIT", k.batch_seg_cap_init);
    k.batch_sample_chunks = env_value("GSIM_BATCH_SAMPLE_CHUNKS", k.batch_sample_chunks);
    k.batch_rpl = env_value("GSIM_BATCH_RPL", k.batch_rpl);
    k.batch_mfma_min_q = env_value("GSIM_BATCH_MFMA_MIN_Q", k.batch_mfma_min_q);
    k.batch_mfma_sample = env_value("GSIM_BATCH_MFMA_SAMPLE", k.batch_mfma_sample);
    k.batch_mfma_dense = env_value("GSIM_BATCH_MFMA_DENSE", k.batch_mfma_dense);
    k.debug_batch = std::getenv("GSIM_DEBUG_BATCH") ? 1 : 0;
    k.fold_full_on_device = env_value("GSIM_FOLD_FULL_ON_DEVICE", k.fold_full_on_device);
    if (const char* v = std::getenv("GSIM_FOLD_RESCORE")) k.fold_rescore_host = std::string(v) == "host" ? 1 : 0;
    return k;
}

#ifdef GSIM_TEST_HOOKS
int env_int(const char* name, int dflt)
{
    return env_value(name, dflt);
}

int alias_devices()
{
    static const int n = env_int("GSIM_TEST_ALIAS_DEVICES", 0);
    return n > 0 ? n : 0;
}
#else
int alias_devices()
{
    return 0;
}
#endif

int phys_device(int logical)
{
    return alias_devices() ? 0 : logical;
}

hipError_t set_device(int logical)
{
    return hipSetDevice(phys_device(logical));
}

int free_shard(Shard& s)
{
    for (auto& l : s.lanes) (void) free_shard(l); // (they share s.d_rows and do not own it)
    s.lanes.clear();
    (void) set_device(s.device);
    if (s.stream) (void) hipStreamSynchronize(s.stream);
    if (s.owns_rows && s.d_rows) (void) hipFree(s.d_rows);
    if (s.d_rowpop) (void) hipFree(s.d_rowpop);
    if (s.d_query) (void) hipFree(s.d_query);
    if (s.d_state) (void) hipFree(s.d_state);
    if (s.d_cand) (void) hipFree(s.d_cand);
    if (s.d_cand_cb) (void) hipFree(s.d_cand_cb);
    if (s.d_final_cb) (void) hipFree(s.d_final_cb);
    if (s.d_seg_count) (void) hipFree(s.d_seg_count);
    if (s.d_final) (void) hipFree(s.d_final);
    if (s.d_result) (void) hipFree(s.d_result);
    if (s.d_full) (void) hipFree(s.d_full);
    if (s.d_fq) (void) hipFree(s.d_fq);
    if (s.h_fq) (void) hipHostFree(s.h_fq);
    if (s.d_key2) (void) hipFree(s.d_key2);
    if (s.d_cb2) (void) hipFree(s.d_cb2);
    if (s.d_large) (void) hipFree(s.d_large);
    if (s.d_lk) (void) hipFree(s.d_lk);
    if (s.d_bincur) (void) hipFree(s.d_bincur);
    for (auto& sl : s.slot)
        if (sl.ev) (void) hipEventDestroy(sl.ev);
    if (s.d_pub) (void) hipFree(s.d_pub);
    if (s.d_hdr) (void) hipFree(s.d_hdr);
    if (s.d_summ) (void) hipFree(s.d_summ);
    if (s.d_dbg) (void) hipFree(s.d_dbg);
    if (s.h_done) (void) hipHostFree(s.h_done);
    if (s.h_pipe) (void) hipHostFree(s.h_pipe);
    if (s.h_query) (void) hipHostFree(s.h_query);
    if (s.h_result) (void) hipHostFree(s.h_result);
    if (s.h_state) (void) hipHostFree(s.h_state);
    if (s.d_bqueries) (void) hipFree(s.d_bqueries);
    if (s.d_bqpop) (void) hipFree(s.d_bqpop);
    if (s.d_bstate) (void) hipFree(s.d_bstate);
    if (s.d_bcand) (void) hipFree(s.d_bcand);
    if (s.d_bcand_cb) (void) hipFree(s.d_bcand_cb);
    if (s.d_bcand_q) (void) hipFree(s.d_bcand_q);
    if (s.d_bseg_count) (void) hipFree(s.d_bseg_count);
    if (s.d_bfin_key) (void) hipFree(s.d_bfin_key);
    if (s.d_bfin_cb) (void) hipFree(s.d_bfin_cb);
    if (s.d_bflags) (void) hipFree(s.d_bflags);
    if (s.d_brare) (void) hipFree(s.d_brare);
    if (s.h_brare) (void) hipHostFree(s.h_brare);
    if (s.h_bflags) (void) hipHostFree(s.h_bflags);
    if (s.h_bqueries) (void) hipHostFree(s.h_bqueries);
    if (s.h_bresult) (void) hipHostFree(s.h_bresult);
    if (s.d_bresult) (void) hipFree(s.d_bresult);
    free_comm_buffers(s);
    for (auto e : s.ev) (void) hipEventDestroy(e);
    for (auto e : s.bev) (void) hipEventDestroy(e);
    for (auto e : s.q_ev) (void) hipEventDestroy(e);
    if (s.own_stream) (void) hipStreamDestroy(s.own_stream);
    s = Shard{};
    return 0;
}

// Allocate the per-shard search scratch once the rows are in place.
int setup_shard(gsim_db* db, Shard& s)
{
    if (s.nrows > 0x7FFFFFFFull) // candidate / finalist slots are 32-bit indexed, capacity a power of two <= 2^31
        return fail(GSIM_ERR_INVALID, "more than 2^31-1 rows on one device: shard the table over more devices");
    GSIM_HIP(set_device(s.device));
    hipDeviceProp_t prop;
    GSIM_HIP(hipGetDeviceProperties(&prop, phys_device(s.device)));
    s.num_cus = (prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256) / (s.cu_share > 0 ? s.cu_share : 1);
    if (s.num_cus < 1) s.num_cus = 1;
    GSIM_HIP(hipStreamCreateWithFlags(&s.own_stream, hipStreamNonBlocking));
    s.stream = s.own_stream;
    const gsim::Knobs& kn = db->knobs;
    if (s.W == 0) s.W = db->W;
    s.geo = gsim::scan_geometry(s.nrows, s.W, s.num_cus, kn.scan_waves_per_cu, 8, kn.scan_ragged != 0);
    s.sample_chunks = kn.sample_chunks;
    s.fgeo = s.geo;
    (void) gsim::fused_word_geometry(s.nrows, s.W, s.num_cus, &s.fgeo, kn.scan_ragged != 0); // (rows of 3 ... 11 or twice that many words: the single launch's own)
    if (s.fgeo.nchunks < 4ull * s.fgeo.nwaves) { // small table: threshold checkpoints need a few trips per wave
        // (narrow rows: a chunk is 512 rows and a wave's LDS store holds 2048 -- three chunks per wave, so that a store
        // cannot fill before the one threshold such a table sees, the one after the loop)
        const uint64_t per = s.fgeo.chunk_rows > 512 ? 2 : (s.fgeo.chunk_rows == 512 ? 3 : 4);
        uint64_t nw = s.fgeo.nchunks / per / 4 * 4;
        s.fgeo.nwaves = static_cast<uint32_t>(nw < 4 ? 4 : nw);
    }
    GSIM_HIP(hipMalloc(&s.d_query, static_cast<size_t>(s.W) * 4));
    GSIM_HIP(hipMalloc(&s.d_state, sizeof(gsim::QueryState)));
    GSIM_HIP(hipMemset(s.d_state, 0, sizeof(gsim::QueryState))); // the kernels keep it zero between queries
    // every workgroup of the single launch is a selector and there are at most kFusedSelectors: on a part with more CUs
    // (or with GSIM_SCAN_WAVES_PER_CU > 4) the grid is clamped -- its waves take more chunks each -- instead of losing the path
    const uint32_t fused_max_waves = static_cast<uint32_t>(gsim::kFusedSelectors) * (gsim::kScanBlock / 64);
    if (s.fgeo.nwaves > fused_max_waves) s.fgeo.nwaves = fused_max_waves;
    if (gsim::fused_supported(s.fgeo)) {
        GSIM_HIP(hipMalloc(&s.d_pub, gsim::fused_pub_bytes(s.fgeo.nwaves / 4)));
        GSIM_HIP(hipMalloc(&s.d_hdr, gsim::fused_hdr_bytes(s.fgeo.nwaves / 4)));
        // (no launch carries tag 0: a selector never takes what the regions held before their first query for a published list)
        GSIM_HIP(hipMemset(s.d_pub, 0, gsim::fused_pub_bytes(s.fgeo.nwaves / 4)));
        GSIM_HIP(hipMemset(s.d_hdr, 0, gsim::fused_hdr_bytes(s.fgeo.nwaves / 4)));
    } else {
        static std::atomic<bool> said{false};
        if (s.geo.lanes_per_row != 0 && !said.exchange(true))
            std::fprintf(stderr, "gpusimilarity_amd: the single-launch path is off for this table's scan geometry (%u waves, unroll %u): "
                                 "queries run on the four-kernel pipeline\n", s.fgeo.nwaves, s.fgeo.unroll);
    }
    GSIM_HIP(hipMalloc(reinterpret_cast<void**>(&s.d_summ), kSummBytes));
    GSIM_HIP(hipMemset(s.d_summ, 0, kSummBytes));
    GSIM_HIP(hipHostMalloc(reinterpret_cast<void**>(&s.h_done), 64, kHostPolled));
    std::memset(s.h_done, 0, 64);
    GSIM_HIP(hipHostMalloc(&s.h_query, static_cast<size_t>(s.W) * 4 * kQueryRing, kHostPinned));
    for (int i = 0; i < kQueryRing; i++) {
        hipEvent_t e;
        GSIM_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        s.q_ev.push_back(e);
    }
    GSIM_HIP(hipHostMalloc(&s.h_state, sizeof(gsim::QueryState), kHostPinned));
    return GSIM_OK;
}

// memcpy on several host threads (a single thread moves ~7 GB/s here; table loading is startup
// time, not the hot path, but a 128 GB table should not take half a minute)
void parallel_memcpy(void* dst, const void* src, size_t bytes)
{
    const size_t kMin = size_t(8) << 20;
    unsigned nt = std::min<unsigned>(8, std::max<unsigned>(1, std::thread::hardware_concurrency()));
    if (bytes < 2 * kMin || nt < 2) {
        std::memcpy(dst, src, bytes);
        return;
    }
    nt = static_cast<unsigned>(std::min<size_t>(nt, bytes / kMin));
    const size_t per = (bytes / nt + 4095) & ~size_t(4095);
    std::vector<std::thread> th;
    for (unsigned i = 0; i < nt; i++) {
        const size_t off = per * i;
        if (off >= bytes) break;
        const size_t n = std::min(per, bytes - off);
        th.emplace_back([=] { std::memcpy(static_cast<char*>(dst) + off, static_cast<const char*>(src) + off, n); });
    }
    for (auto& t : th) t.join();
}

// Pageable host memory -> device through two pinned staging buffers: the (threaded) copy into one
// buffer overlaps the DMA of the other (pageable hipMemcpy: 10 GB/s).
int upload_rows(void* d_dst, const void* h_src, size_t bytes, hipStream_t stream)
{
    const size_t kChunk = size_t(64) << 20;
    if (bytes <= kChunk) {
        GSIM_HIP(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
        return GSIM_OK;
    }
    void* stage[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    int rc = GSIM_OK;
    hipError_t e = hipSuccess;
    for (int i = 0; i < 2 && e == hipSuccess; i++) {
        e = hipHostMalloc(&stage[i], kChunk, kHostPinned);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&done[i], hipEventDisableTiming);
    }
    size_t off = 0;
    for (int i = 0; e == hipSuccess && off < bytes; i ^= 1) {
        const size_t n = std::min(kChunk, bytes - off);
        e = hipEventSynchronize(done[i]); // the previous DMA out of this buffer (no-op the first time)
        if (e != hipSuccess) break;
        parallel_memcpy(stage[i], static_cast<const char*>(h_src) + off, n);
        e = hipMemcpyAsync(static_cast<char*>(d_dst) + off, stage[i], n, hipMemcpyHostToDevice, stream);
        if (e == hipSuccess) e = hipEventRecord(done[i], stream);
        off += n;
    }
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) rc = fail_hip(e, "upload_rows");
    for (int i = 0; i < 2; i++) {
        if (done[i]) (void) hipEventDestroy(done[i]);
        if (stage[i]) (void) hipHostFree(stage[i]);
    }
    return rc;
}


std::mutex g_rr_mutex;
int g_next_device = 0;

} // namespace gsim_host

using namespace gsim_host;

extern "C" {

const char* gsim_last_error(void)
{
    return g_last_error.c_str();
}

const char* gsim_version(void)
{
    return "gpusimilarity_amd 0.1 (gfx950)";
}

int gsim_device_count(int* count)
{
    if (!count) return fail(GSIM_ERR_INVALID, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        // reference get_gpu_count() reports 0 GPUs rather than failing (fingerprintdb_cuda.cu:40-52)
        (void) hipGetLastError();
        n = 0;
    }
    if (n > 0 && alias_devices()) n = alias_devices(); // test hook, see alias_devices()
    *count = n;
    return GSIM_OK;
}

int gsim_device_free_bytes(int device, size_t* free_bytes)
{
    if (!free_bytes) return fail(GSIM_ERR_INVALID, "free_bytes is NULL");
    int n = 0;
    gsim_device_count(&n);
    if (device < 0 || device >= n) return fail(GSIM_ERR_NO_DEVICE, "device index out of range");
    GSIM_HIP(set_device(device));
    size_t fr = 0, tot = 0;
    GSIM_HIP(hipMemGetInfo(&fr, &tot));
    *free_bytes = fr;
    return GSIM_OK;
}

int gsim_available_device_bytes(size_t* total_free_bytes)
{
    if (!total_free_bytes) return fail(GSIM_ERR_INVALID, "total_free_bytes is NULL");
    int n = 0;
    gsim_device_count(&n);
    size_t sum = 0;
    for (int d = 0; d < n; d++) {
        size_t fr = 0;
        int rc = gsim_device_free_bytes(d, &fr);
        if (rc != GSIM_OK) return rc;
        sum += fr;
    }
    *total_free_bytes = sum;
    return GSIM_OK;
}

int gsim_next_device(size_t required_bytes, int* device)
{
    if (!device) return fail(GSIM_ERR_INVALID, "device is NULL");
    int n = 0;
    gsim_device_count(&n);
    if (n == 0) return fail(GSIM_ERR_NO_DEVICE, "no GPU available");
    std::lock_guard<std::mutex> lock(g_rr_mutex);
    for (int i = 0; i < n; i++) {
        const int d = g_next_device++ % n;
        size_t fr = 0;
        int rc = gsim_device_free_bytes(d, &fr);
        if (rc != GSIM_OK) return rc;
        if (fr > required_bytes) {
            *device = d;
            return GSIM_OK;
        }
    }
    return fail(GSIM_ERR_NOMEM, "Can't find a GPU with enough memory to copy data.");
}

int gsim_db_create(uint32_t fp_bits, gsim_db** out)
{
    if (!out) return fail(GSIM_ERR_INVALID, "out is NULL");
    if (fp_bits == 0 || fp_bits % 32 != 0 || fp_bits > 32768)
        return fail(GSIM_ERR_INVALID, "fp_bits must be a positive multiple of 32, <= 32768");
    gsim_db* db = new (std::nothrow) gsim_db;
    if (!db) return fail(GSIM_ERR_NOMEM, "out of host memory");
    db->fp_bits = fp_bits;
    db->W = fp_bits / 32;
    db->knobs = read_knobs(); // (once per handle; no search entry point reads the environment)
    *out = db;
    return GSIM_OK;
}

int gsim_db_add_rows(gsim_db* db, const uint32_t* rows, uint64_t nrows)
{
    if (!db || (!rows && nrows)) return fail(GSIM_ERR_INVALID, "NULL argument");
    if (db->finalized) return fail(GSIM_ERR_STATE, "table already finalized");
    if (db->nrows + nrows > 0xFFFFFFFFull) return fail(GSIM_ERR_INVALID, "more than 2^32-1 rows");
    try {
        db->slice_first.push_back(db->nrows);
        const size_t old_words = db->host_rows.size();
        db->host_rows.resize(old_words + static_cast<size_t>(nrows) * db->W);
        parallel_memcpy(db->host_rows.data() + old_words, rows, static_cast<size_t>(nrows) * db->W * 4);
    } catch (const std::bad_alloc&) {
        return fail(GSIM_ERR_NOMEM, "out of host memory");
    }
    db->nrows += nrows;
    db->has_host_copy = true;
    return GSIM_OK;
}

int gsim_db_set_fold_factor(gsim_db* db, uint32_t fold_factor)
{
    if (!db) return fail(GSIM_ERR_INVALID, "db is NULL");
    if (db->finalized) return fail(GSIM_ERR_STATE, "table already finalized");
    if (fold_factor == 0 || fold_factor > db->W) return fail(GSIM_ERR_INVALID, "fold factor out of range");
    db->fold_requested = fold_factor;
    return GSIM_OK;
}

uint32_t gsim_db_fold_factor(const gsim_db* db)
{
    return db ? db->fold : 0;
}

int gsim_db_set_fold_full_on_device(gsim_db* db, int allow)
{
    if (!db) return fail(GSIM_ERR_INVALID, "db is NULL");
    if (db->finalized) return fail(GSIM_ERR_STATE, "table already finalized");
    db->fold_full_on_device = allow != 0;
    return GSIM_OK;
}

int gsim_db_finalize(gsim_db* db, int device, int ndevices)
{
    if (!db) return fail(GSIM_ERR_INVALID, "db is NULL");
    if (db->finalized) return fail(GSIM_ERR_STATE, "table already finalized");
    int ndev = 0;
    gsim_device_count(&ndev);
    if (ndev == 0) return fail(GSIM_ERR_NO_DEVICE, "no GPU available");
    if (ndevices == 0) { // all devices from `device` on
        if (device < 0) device = 0;
        if (device >= ndev) return fail(GSIM_ERR_NO_DEVICE, "device index out of range");
        ndevices = ndev - device;
    }
    if (ndevices < 0) return fail(GSIM_ERR_INVALID, "ndevices < 0");
    if (device >= 0 && device + ndevices > ndev) return fail(GSIM_ERR_NO_DEVICE, "device range exceeds the GPUs present");
    // copyToGPU's factor adjustment, fingerprintdb_cuda.cu:170-173
    db->fold = db->fold_requested ? db->fold_requested : 1;
    while (db->W % db->fold != 0) db->fold++;
    if (db->fold > 1) {
        // Folded table (fingerprintdb_cuda.cu:184-194): every add_rows slice is one storage with
        // its own candidate list, placed round-robin like get_next_gpu (:54-68, :186-188).
        if (!db->has_host_copy) return fail(GSIM_ERR_STATE, "folding needs the host copy of the rows");
        const uint32_t Wf = db->W / db->fold;
        const size_t nsl = db->slice_first.size();
        db->shards.resize(nsl);
        std::vector<uint32_t> folded;
        for (size_t i = 0; i < nsl; i++) {
            Shard& s = db->shards[i];
            s.first_row = db->slice_first[i];
            s.nrows = (i + 1 < nsl ? db->slice_first[i + 1] : db->nrows) - s.first_row;
            s.W = Wf;
            const size_t bytes = static_cast<size_t>(s.nrows) * Wf * 4;
            if (device < 0 || ndevices != 1) {
                int d = 0;
                int rc = gsim_next_device(bytes, &d);
                if (rc != GSIM_OK) return rc;
                s.device = (device >= 0 && ndevices > 1) ? device + (d % ndevices) : d;
            } else {
                s.device = device;
            }
            folded.resize(static_cast<size_t>(s.nrows) * Wf);
            fold_rows_mt(db->host_rows.data() + s.first_row * db->W, s.nrows, db->W, db->fold, folded.data());
            GSIM_HIP(set_device(s.device));
            GSIM_HIP(hipMalloc(&s.d_rows, bytes ? bytes : 16));
            s.owns_rows = true;
            if (bytes) GSIM_HIP(hipMemcpy(s.d_rows, folded.data(), bytes, hipMemcpyHostToDevice));
            int rc = setup_shard(db, s);
            if (rc != GSIM_OK) return rc;
        }
        // The full fingerprints as well, when EVERY storage's fit next to the folded rows (on a 288 GB MI355X they
        // practically always do: folding is then a speed device, not a capacity one): the candidates are re-scored on the
        // GPU.  Decided only now, with all folded storages placed, and for all storages or none: search_folded re-scores
        // on the device only when every shard holds its full rows, and when folding is capacity-driven (the placement plan
        // picked a factor > 1 because the table does not fit) copies made storage by storage would take the memory the
        // later storages' folded rows and the search scratch were budgeted for.  Per device: the full rows
        // of its storages + 24 B per folded row (the four-kernel pipeline's scratch, allocated on first use) + 2 GB.
        // Without them the re-score runs on the host, as in the reference (fingerprintdb_cuda.cu:307-331).
        const int full_on_device = db->knobs.fold_full_on_device;
        bool keep_full = full_on_device != 0 && db->fold_full_on_device && db->nrows > 0;
        if (keep_full) {
            std::vector<size_t> need(static_cast<size_t>(ndev), 0);
            for (const auto& s : db->shards) need[static_cast<size_t>(s.device)] += static_cast<size_t>(s.nrows) * (static_cast<size_t>(db->W) * 4 + 24);
            for (int d = 0; d < ndev && keep_full; d++) {
                if (!need[static_cast<size_t>(d)]) continue;
                size_t fr = 0;
                keep_full = gsim_device_free_bytes(d, &fr) == GSIM_OK && fr > need[static_cast<size_t>(d)] + (size_t(2) << 30);
            }
        }
        for (size_t i = 0; i < nsl && keep_full; i++) {
            Shard& s = db->shards[i];
            const size_t full_bytes = static_cast<size_t>(s.nrows) * db->W * 4;
            if (!full_bytes) continue;
            GSIM_HIP(set_device(s.device));
            if (hipMalloc(reinterpret_cast<void**>(&s.d_full), full_bytes) != hipSuccess) {
                (void) hipGetLastError();
                s.d_full = nullptr;
                keep_full = false;
                break;
            }
            const int urc = upload_rows(s.d_full, db->host_rows.data() + s.first_row * db->W, full_bytes, nullptr);
            if (urc != GSIM_OK) return urc;
        }
        if (!keep_full) { // none rather than some: partial copies are never used
            for (auto& s : db->shards) {
                if (!s.d_full) continue;
                (void) set_device(s.device);
                (void) hipFree(s.d_full);
                s.d_full = nullptr;
            }
        }
        db->finalized = true;
        return GSIM_OK;
    }
    const size_t row_bytes = static_cast<size_t>(db->W) * 4;
    if (ndevices == 1 && device < 0) {
        int rc = gsim_next_device(static_cast<size_t>(db->nrows) * row_bytes, &device);
        if (rc != GSIM_OK) return rc;
    }
    if (device < 0) device = 0;
    if (device + ndevices > ndev) return fail(GSIM_ERR_NO_DEVICE, "device range exceeds the GPUs present");
    if (static_cast<uint64_t>(ndevices) > db->nrows && db->nrows > 0) ndevices = static_cast<int>(db->nrows);
    const uint64_t per = (db->nrows + ndevices - 1) / (ndevices ? ndevices : 1);
    db->shards.resize(static_cast<size_t>(ndevices));
    for (int i = 0; i < ndevices; i++) {
        Shard& s = db->shards[i];
        s.device = device + i;
        s.first_row = std::min<uint64_t>(per * i, db->nrows);
        s.nrows = std::min<uint64_t>(per, db->nrows - s.first_row);
        GSIM_HIP(set_device(s.device));
        const size_t bytes = static_cast<size_t>(s.nrows) * row_bytes;
        GSIM_HIP(hipMalloc(&s.d_rows, bytes ? bytes : 16));
        s.owns_rows = true;
        if (bytes) {
            const int urc = upload_rows(s.d_rows, db->host_rows.data() + s.first_row * db->W, bytes, nullptr);
            if (urc != GSIM_OK) return urc;
        }
        int rc = setup_shard(db, s);
        if (rc != GSIM_OK) return rc;
    }
    db->finalized = true;
    return GSIM_OK;
}

int gsim_db_generate(gsim_db* db, uint64_t seed, int kind, uint64_t first_row, uint64_t nrows, int device)
{
    if (!db) return fail(GSIM_ERR_INVALID, "db is NULL");
    if (db->finalized || db->nrows) return fail(GSIM_ERR_STATE, "table already holds rows");
    if (kind != GSIM_SYNTH_SPARSE && kind != GSIM_SYNTH_DENSE && kind != GSIM_SYNTH_MORGAN)
        return fail(GSIM_ERR_INVALID, "unknown synthetic kind");
    if (kind == GSIM_SYNTH_MORGAN && db->W > 12288) return fail(GSIM_ERR_INVALID, "Morgan-shaped rows: fp_bits too large");
    if (nrows > 0xFFFFFFFFull) return fail(GSIM_ERR_INVALID, "more than 2^32-1 rows");
    int ndev = 0;
    gsim_device_count(&ndev);
    if (ndev == 0) return fail(GSIM_ERR_NO_DEVICE, "no GPU available");
    if (device < 0 || device >= ndev) return fail(GSIM_ERR_NO_DEVICE, "device index out of range");
    db->nrows = nrows;
    db->shards.resize(1);
    Shard& s = db->shards[0];
    s.device = device;
    s.first_row = 0;
    s.nrows = nrows;
    GSIM_HIP(set_device(device));
    const size_t bytes = static_cast<size_t>(nrows) * db->W * 4;
    GSIM_HIP(hipMalloc(&s.d_rows, bytes ? bytes : 16));
    s.owns_rows = true;
    int rc = setup_shard(db, s);
    if (rc != GSIM_OK) return rc;
    GSIM_HIP(gsim::launch_generate(s.d_rows, seed, kind, first_row, nrows, db->W, s.stream));
    GSIM_HIP(hipStreamSynchronize(s.stream));
    db->finalized = true;
    return GSIM_OK;
}

int gsim_db_generate_sharded(gsim_db* db, uint64_t seed, int kind, uint64_t first_row, uint64_t nrows, int device, int ndevices)
{
    if (!db) return fail(GSIM_ERR_INVALID, "db is NULL");
    if (db->finalized || db->nrows) return fail(GSIM_ERR_STATE, "table already holds rows");
    if (kind != GSIM_SYNTH_SPARSE && kind != GSIM_SYNTH_DENSE && kind != GSIM_SYNTH_MORGAN)
        return fail(GSIM_ERR_INVALID, "unknown synthetic kind");
    if (kind == GSIM_SYNTH_MORGAN && db->W > 12288) return fail(GSIM_ERR_INVALID, "Morgan-shaped rows: fp_bits too large");
    if (nrows > 0xFFFFFFFFull) return fail(GSIM_ERR_INVALID, "more than 2^32-1 rows");
    int ndev = 0;
    gsim_device_count(&ndev);
    if (ndev == 0) return fail(GSIM_ERR_NO_DEVICE, "no GPU available");
    if (ndevices < 1 || device < 0 || device + ndevices > ndev) return fail(GSIM_ERR_NO_DEVICE, "device range exceeds the GPUs present");
    if (static_cast<uint64_t>(ndevices) > nrows && nrows > 0) ndevices = static_cast<int>(nrows);
    // the same contiguous split as gsim_db_finalize(db, device, ndevices)
    const uint64_t per = (nrows + ndevices - 1) / ndevices;
    db->nrows = nrows;
    db->shards.resize(static_cast<size_t>(ndevices));
    for (int i = 0; i < ndevices; i++) {
        Shard& s = db->shards[static_cast<size_t>(i)];
        s.device = device + i;
        s.first_row = std::min<uint64_t>(per * i, nrows);
        s.nrows = std::min<uint64_t>(per, nrows - s.first_row);
        GSIM_HIP(set_device(s.device));
        const size_t bytes = static_cast<size_t>(s.nrows) * db->W * 4;
        GSIM_HIP(hipMalloc(&s.d_rows, bytes ? bytes : 16));
        s.owns_rows = true;
        int rc = setup_shard(db, s);
        if (rc != GSIM_OK) return rc;
        if (s.nrows) GSIM_HIP(gsim::launch_generate(s.d_rows, seed, kind, first_row + s.first_row, s.nrows, db->W, s.stream));
    }
    for (auto& s : db->shards) {
        GSIM_HIP(set_device(s.device));
        GSIM_HIP(hipStreamSynchronize(s.stream));
    }
    db->finalized = true;
    return GSIM_OK;
}

int gsim_synth_row(uint64_t seed, int kind, uint64_t row, uint32_t fp_bits, uint32_t* out_words)
{
    if (!out_words) return fail(GSIM_ERR_INVALID, "out_words is NULL");
    if (fp_bits == 0 || fp_bits % 32 != 0 || fp_bits > 32768) return fail(GSIM_ERR_INVALID, "fp_bits must be a positive multiple of 32, <= 32768");
    const uint32_t W = fp_bits / 32;
    if (kind == GSIM_SYNTH_MORGAN) {
        gsim::synth_row_morgan(out_words, seed, row, W);
    } else if (kind == GSIM_SYNTH_SPARSE || kind == GSIM_SYNTH_DENSE) {
        for (uint32_t j = 0; j < W; j++) out_words[j] = gsim::synth_word_iid(seed, kind == GSIM_SYNTH_DENSE, row * W + j);
    } else {
        return fail(GSIM_ERR_INVALID, "unknown synthetic kind");
    }
    return GSIM_OK;
}

int gsim_db_attach_device_rows(gsim_db* db, const void* d_rows, uint64_t nrows, int device)
{
    if (!db || (!d_rows && nrows)) return fail(GSIM_ERR_INVALID, "NULL argument");
    if (db->finalized || db->nrows) return fail(GSIM_ERR_STATE, "table already holds rows");
    if (reinterpret_cast<uintptr_t>(d_rows) % 16 != 0) return fail(GSIM_ERR_INVALID, "device rows must be 16-byte aligned");
    if (nrows > 0xFFFFFFFFull) return fail(GSIM_ERR_INVALID, "more than 2^32-1 rows");
    int ndev = 0;
    gsim_device_count(&ndev);
    if (device < 0 || device >= ndev) return fail(GSIM_ERR_NO_DEVICE, "device index out of range");
    db->nrows = nrows;
    db->shards.resize(1);
    Shard& s = db->shards[0];
    s.device = device;
    s.nrows = nrows;
    s.d_rows = const_cast<void*>(d_rows);
    s.owns_rows = false;
    int rc = setup_shard(db, s);
    if (rc != GSIM_OK) return rc;
    db->finalized = true;
    return GSIM_OK;
}

int gsim_db_destroy(gsim_db* db)
{
    if (!db) return GSIM_OK;
    for (auto& s : db->shards) free_shard(s);
    delete db;
    return GSIM_OK;
}

uint64_t gsim_db_count(const gsim_db* db)
{
    return db ? db->nrows : 0;
}

uint32_t gsim_db_fp_bits(const gsim_db* db)
{
    return db ? db->fp_bits : 0;
}

size_t gsim_db_data_bytes(const gsim_db* db)
{
    return db ? static_cast<size_t>(db->nrows) * db->W * 4 : 0;
}

int gsim_db_shard_count(const gsim_db* db)
{
    return db ? static_cast<int>(db->shards.size()) : 0;
}

int gsim_db_shard_device(const gsim_db* db, int shard)
{
    if (!db || shard < 0 || static_cast<size_t>(shard) >= db->shards.size()) return -1;
    return db->shards[static_cast<size_t>(shard)].device;
}

int gsim_db_row(const gsim_db* db, uint64_t row, uint32_t* out_words)
{
    if (!db || !out_words) return fail(GSIM_ERR_INVALID, "NULL argument");
    if (row >= db->nrows) return fail(GSIM_ERR_INVALID, "row index out of range");
    if (db->has_host_copy) {
        std::memcpy(out_words, db->host_rows.data() + row * db->W, static_cast<size_t>(db->W) * 4);
        return GSIM_OK;
    }
    for (const auto& s : db->shards) {
        if (row >= s.first_row && row < s.first_row + s.nrows) {
            GSIM_HIP(set_device(s.device));
            const unsigned char* src = static_cast<const unsigned char*>(s.d_rows) +
                                       static_cast<size_t>(row - s.first_row) * db->W * 4;
            GSIM_HIP(hipMemcpy(out_words, src, static_cast<size_t>(db->W) * 4, hipMemcpyDeviceToHost));
            return GSIM_OK;
        }
    }
    return fail(GSIM_ERR_STATE, "row not resident");
}

int gsim_db_set_stream(gsim_db* db, void* hip_stream)
{
    if (!db || !db->finalized) return fail(GSIM_ERR_STATE, "table not finalized");
    if (db->shards.size() != 1) return fail(GSIM_ERR_STATE, "set_stream needs a single-shard handle");
    std::lock_guard<std::mutex> guard(db->search_mutex); // (not under a running search)
    Shard& s = db->shards[0];
    s.stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : s.own_stream;
    return GSIM_OK;
}

int gsim_db_set_row_base(gsim_db* db, uint32_t row_base)
{
    if (!db) return fail(GSIM_ERR_INVALID, "db is NULL");
    db->row_base = row_base;
    return GSIM_OK;
}

size_t gsim_result_block_bytes(uint32_t k)
{
    const size_t raw = sizeof(gsim_result_header) + static_cast<size_t>(k) * sizeof(gsim_hit);
    return (raw + 15) / 16 * 16;
}

} // extern "C"
