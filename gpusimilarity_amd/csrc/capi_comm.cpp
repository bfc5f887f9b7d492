// capi_comm.cpp -- the collective route of an in-process multi-device handle: "per-GPU top-k merged via a small RCCL
// gather over xGMI" without PyTorch.  FingerprintDB::search merges its storages' results on the host
// (fingerprintdb_cuda.cu:356-380); with a communicator attached (gsim_db_set_comm) every shard leaves its result block in
// its own device's HBM, ONE grouped ncclAllGather (12 016 B per device at k = 1000) brings all blocks to every device and
// merge_kernel on the first shard's device writes the merged block into pinned host memory.  Same results bit for bit as
// the host merge, which stays the default (profiles/r03_merge_cost.json: the host merge costs < 40 us and its inputs are
// already in host memory; what the collective buys on a real 8-GPU node is for the hardware to say).
#include "capi_internal.h"

#include <rccl/rccl.h>

#include <dlfcn.h>

struct gsim_comm {
    std::vector<int> devices;       // logical device per rank, in the order of the handle's shards
    std::vector<ncclComm_t> comms;  // one communicator per device (ncclCommInitAll); empty: loop-back
    bool loopback = false;          // aliased devices (test hook): all ranks live on one physical GPU, which RCCL refuses;
                                    // the gather is then device-to-device copies into the first rank's buffer
};

namespace gsim_host
{

namespace
{

int fail_nccl(ncclResult_t r, const char* what)
{
    g_last_error = std::string(what) + ": " + ncclGetErrorString(r);
    return GSIM_ERR_HIP;
}

#define GSIM_NCCL(call)                                       \
    do {                                                      \
        ncclResult_t r__ = (call);                            \
        if (r__ != ncclSuccess) return fail_nccl(r__, #call); \
    } while (0)

// every shard's gather buffer holds `per_shard` bytes for each shard of the handle
int ensure_gather(gsim_db* db, size_t per_shard)
{
    const size_t need = per_shard * db->shards.size();
    for (auto& s : db->shards) {
        GSIM_HIP(set_device(s.device));
        if (need > s.gather_bytes) {
            if (s.d_gather) GSIM_HIP(hipFree(s.d_gather));
            s.d_gather = nullptr;
            s.gather_bytes = 0;
            GSIM_HIP(hipMalloc(reinterpret_cast<void**>(&s.d_gather), need));
            GSIM_HIP(hipMemset(s.d_gather, 0, need)); // (a shard without rows never writes its slots: they read as empty blocks)
            s.gather_bytes = need;
        }
        if (!s.gather_ev) GSIM_HIP(hipEventCreateWithFlags(&s.gather_ev, hipEventDisableTiming));
    }
    Shard& s0 = db->shards[db->comm_root]; // (the shard whose device merges and answers the host)
    GSIM_HIP(set_device(s0.device));
    for (auto& e : s0.cev)
        if (!e) GSIM_HIP(hipEventCreate(&e));
    return GSIM_OK;
}

// Slot i of every shard's buffer <- shard i's slot (an in-place all-gather over the shards' own streams, so that it is
// ordered behind each shard's search kernels and before the merge on the first shard's stream).
int all_gather(gsim_db* db, size_t per_shard)
{
    const gsim_comm* c = db->comm;
    const size_t n = db->shards.size();
    if (!c->loopback) {
        ncclResult_t r = ncclGroupStart();
        for (size_t i = 0; i < n && r == ncclSuccess; i++) {
            Shard& s = db->shards[i];
            r = ncclAllGather(s.d_gather + i * per_shard, s.d_gather, per_shard, ncclChar, c->comms[i], s.stream);
        }
        const ncclResult_t e = ncclGroupEnd();
        if (r == ncclSuccess) r = e;
        if (r != ncclSuccess) {
            // the shards' kernels of this query are enqueued and their per-query state is not known to be clean: let the
            // streams drain and have the next enqueue re-zero it
            for (auto& s : db->shards) {
                (void) set_device(s.device);
                (void) hipStreamSynchronize(s.stream);
                s.state_dirty = true;
            }
            return fail_nccl(r, "ncclAllGather");
        }
        return GSIM_OK;
    }
    // Loop-back (aliased devices: RCCL refuses two ranks on one GPU): what the in-place all-gather does, with copies --
    // EVERY shard's buffer receives every other shard's slot, on the receiver's own stream, behind an event on the sender's
    // stream: the layout, the stream order and the merge input are the real route's, only the transport differs.
    for (size_t i = 0; i < n; i++) {
        Shard& s = db->shards[i];
        GSIM_HIP(set_device(s.device));
        GSIM_HIP(hipEventRecord(s.gather_ev, s.stream));
    }
    for (size_t j = 0; j < n; j++) {
        Shard& d = db->shards[j];
        GSIM_HIP(set_device(d.device));
        for (size_t i = 0; i < n; i++) {
            if (i == j) continue;
            Shard& s = db->shards[i];
            GSIM_HIP(hipStreamWaitEvent(d.stream, s.gather_ev, 0));
            GSIM_HIP(hipMemcpyAsync(d.d_gather + i * per_shard, s.d_gather + i * per_shard, per_shard, hipMemcpyDeviceToDevice, d.stream));
        }
    }
    return GSIM_OK;
}

// the shards other than the first: their part of the gather is done (the next query reuses the buffers)
int wait_others(gsim_db* db)
{
    for (size_t i = 0; i < db->shards.size(); i++) {
        if (i == db->comm_root) continue;
        Shard& s = db->shards[i];
        GSIM_HIP(set_device(s.device));
        const int rc = wait_stream(s.stream);
        if (rc != GSIM_OK) return rc;
    }
    return GSIM_OK;
}

int fold_comm_timing(gsim_db* db)
{
    Shard& s0 = db->shards[db->comm_root]; // (the shard whose device merges and answers the host)
    float g = 0.f, m = 0.f;
    GSIM_HIP(hipEventElapsedTime(&g, s0.cev[0], s0.cev[1]));
    GSIM_HIP(hipEventElapsedTime(&m, s0.cev[1], s0.cev[2]));
    db->acc.gather_ms_sum += g;
    db->acc.merge_ms_sum += m;
    db->acc.collectives++;
    return GSIM_OK;
}

} // namespace

void free_comm_buffers(Shard& s)
{
    if (s.d_gather) (void) hipFree(s.d_gather);
    if (s.d_merged) (void) hipFree(s.d_merged);
    if (s.gather_ev) (void) hipEventDestroy(s.gather_ev);
    for (auto e : s.cev)
        if (e) (void) hipEventDestroy(e);
}

// One query on every shard -> blocks in HBM -> all-gather -> merge_kernel -> pinned host block (search_one's twin).
int search_one_comm(gsim_db* db, const uint32_t* query, uint32_t k, float cutoff, int metric, float alpha, float beta, gsim_hit* hits,
                    uint32_t* count, uint64_t* approx)
{
    const size_t n = db->shards.size();
    const size_t blk = gsim_result_block_bytes(k);
    int rc = ensure_gather(db, blk);
    if (rc != GSIM_OK) return rc;
    Shard& s0 = db->shards[db->comm_root]; // (the shard whose device merges and answers the host)
    rc = ensure_result_capacity(s0, k);
    if (rc != GSIM_OK) return rc;
    for (size_t i = 0; i < n; i++) {
        Shard& s = db->shards[i];
        rc = enqueue_query(db, s, query, k, cutoff, metric, alpha, beta, db->row_base + static_cast<uint32_t>(s.first_row),
                           s.d_gather + i * blk, false);
        if (rc != GSIM_OK) return rc;
    }
    GSIM_HIP(set_device(s0.device));
    if (db->timing) GSIM_HIP(hipEventRecord(s0.cev[0], s0.stream));
    rc = all_gather(db, blk);
    if (rc != GSIM_OK) return rc;
    GSIM_HIP(set_device(s0.device));
    if (db->timing) GSIM_HIP(hipEventRecord(s0.cev[1], s0.stream));
    GSIM_HIP(gsim::launch_merge_batch(s0.d_gather, static_cast<uint32_t>(n), 1, blk, k, s0.h_result, s0.stream)); // (zero-copy into pinned memory)
    if (db->timing) GSIM_HIP(hipEventRecord(s0.cev[2], s0.stream));
    rc = wait_stream(s0.stream);
    if (rc != GSIM_OK) return rc;
    rc = wait_others(db);
    if (rc != GSIM_OK) return rc;
    if (db->timing) {
        rc = fold_comm_timing(db);
        if (rc != GSIM_OK) return rc;
    }
    const gsim_result_header* h = reinterpret_cast<const gsim_result_header*>(s0.h_result);
    std::memcpy(hits, h + 1, sizeof(gsim_hit) * h->count);
    *count = h->count;
    if (approx) *approx = h->approx;
    return GSIM_OK;
}

// nq queries through the shared table passes on every shard -> nq blocks per shard in HBM -> one all-gather per 256
// queries -> merge_kernel for all of them -> one copy to the host (search_batched's twin).
int search_batch_comm(gsim_db* db, const uint32_t* queries, uint32_t nq, uint32_t k, uint32_t kout, float cutoff, int metric, float alpha,
                      float beta, gsim_hit* hits, uint32_t* counts, uint64_t* approx)
{
    const size_t n = db->shards.size();
    const size_t blk = gsim_result_block_bytes(k);
    Shard& s0 = db->shards[db->comm_root]; // (the shard whose device merges and answers the host)
    std::vector<unsigned char> host;
    for (uint32_t base = 0; base < nq; base += kBatchMaxQ) {
        const uint32_t nb = std::min<uint32_t>(kBatchMaxQ, nq - base);
        const uint32_t* qb = queries + static_cast<size_t>(base) * db->W;
        const size_t per = blk * nb;
        int rc = ensure_gather(db, per);
        if (rc != GSIM_OK) return rc;
        GSIM_HIP(set_device(s0.device));
        if (per > s0.merged_bytes) {
            if (s0.d_merged) GSIM_HIP(hipFree(s0.d_merged));
            s0.d_merged = nullptr;
            s0.merged_bytes = 0;
            GSIM_HIP(hipMalloc(reinterpret_cast<void**>(&s0.d_merged), per));
            s0.merged_bytes = per;
        }
        // all shards scan at once; each says whether its pass could finish every query (run_batch repeats what it can)
        std::vector<void*> outs(n);
        for (size_t i = 0; i < n; i++) outs[i] = db->shards[i].d_gather + i * per;
        rc = run_batch(db, qb, nb, k, cutoff, metric, alpha, beta, outs);
        if (rc != GSIM_OK) return rc;
        for (size_t i = 0; i < n; i++) {
            Shard& s = db->shards[i];
            if (s.nrows == 0 || (s.h_bflags[0] & 5u) == 0) continue;
            // heavy ties / candidate overflow at the largest segments: this shard's chunk through the single-query pipeline
            const uint32_t row_base = db->row_base + static_cast<uint32_t>(s.first_row);
            for (uint32_t q = 0; q < nb; q++) {
                rc = enqueue_query(db, s, qb + static_cast<size_t>(q) * db->W, k, cutoff, metric, alpha, beta, row_base,
                                   s.d_gather + i * per + q * blk, false);
                if (rc != GSIM_OK) return rc;
            }
        }
        GSIM_HIP(set_device(s0.device));
        if (db->timing) GSIM_HIP(hipEventRecord(s0.cev[0], s0.stream));
        rc = all_gather(db, per);
        if (rc != GSIM_OK) return rc;
        GSIM_HIP(set_device(s0.device));
        if (db->timing) GSIM_HIP(hipEventRecord(s0.cev[1], s0.stream));
        GSIM_HIP(gsim::launch_merge_batch(s0.d_gather, static_cast<uint32_t>(n), nb, blk, k, s0.d_merged, s0.stream));
        if (db->timing) GSIM_HIP(hipEventRecord(s0.cev[2], s0.stream));
        host.resize(per);
        GSIM_HIP(hipMemcpyAsync(host.data(), s0.d_merged, per, hipMemcpyDeviceToHost, s0.stream));
        rc = wait_stream(s0.stream);
        if (rc != GSIM_OK) return rc;
        rc = wait_others(db);
        if (rc != GSIM_OK) return rc;
        if (db->timing) {
            rc = fold_comm_timing(db);
            if (rc != GSIM_OK) return rc;
        }
        for (uint32_t q = 0; q < nb; q++) {
            const gsim_result_header* h = reinterpret_cast<const gsim_result_header*>(host.data() + q * blk);
            std::memcpy(hits + static_cast<size_t>(base + q) * kout, h + 1, sizeof(gsim_hit) * h->count);
            counts[base + q] = h->count;
            if (approx) approx[base + q] = h->approx;
        }
    }
    return GSIM_OK;
}

} // namespace gsim_host

using namespace gsim_host;

extern "C" {

// Which RCCL is this process bound to?  libgsim_hip.so names librccl.so as a dependency with /opt/rocm/lib on its run path,
// but a process that loaded another librccl.so FIRST (PyTorch ships its own and loads it at import) keeps that one: the
// loader resolves a dependency by soname to what is already mapped.  The five entry points used here (ncclCommInitAll,
// ncclGroupStart/End, ncclAllGather, ncclCommDestroy + ncclGetErrorString) exist unchanged in every NCCL/RCCL 2.x, so a
// different MINOR version is reported, not refused; a different MAJOR version is refused by gsim_comm_create.
int gsim_rccl_info(int* header_version, int* runtime_version, char* path, size_t path_bytes)
{
    if (header_version) *header_version = NCCL_VERSION_CODE;
    int v = 0;
    const ncclResult_t r = ncclGetVersion(&v);
    if (r != ncclSuccess) return fail_nccl(r, "ncclGetVersion");
    if (runtime_version) *runtime_version = v;
    if (path && path_bytes) {
        Dl_info info{};
        const char* p = (dladdr(reinterpret_cast<void*>(&ncclGetVersion), &info) && info.dli_fname) ? info.dli_fname : "";
        std::snprintf(path, path_bytes, "%s", p);
    }
    return GSIM_OK;
}

int gsim_comm_create(const int* devices, int ndevices, gsim_comm** out)
{
    if (!devices || !out || ndevices < 1) return fail(GSIM_ERR_INVALID, "NULL / empty argument");
    {
        int rt = 0;
        const int rc = gsim_rccl_info(nullptr, &rt, nullptr, 0);
        if (rc != GSIM_OK) return rc;
        if (rt / 10000 != NCCL_VERSION_CODE / 10000)
            return fail(GSIM_ERR_STATE, "the librccl.so bound to this process is version " + std::to_string(rt) + ", this library was built against " +
                                            std::to_string(NCCL_VERSION_CODE) + ": a different major version (gsim_rccl_info names the file)");
    }
    int ndev = 0;
    gsim_device_count(&ndev);
    if (ndev == 0) return fail(GSIM_ERR_NO_DEVICE, "no GPU available");
    for (int i = 0; i < ndevices; i++) {
        if (devices[i] < 0 || devices[i] >= ndev) return fail(GSIM_ERR_NO_DEVICE, "device index out of range");
        for (int j = 0; j < i; j++)
            if (devices[j] == devices[i]) return fail(GSIM_ERR_INVALID, "a device appears twice in the communicator");
    }
    gsim_comm* c = new (std::nothrow) gsim_comm;
    if (!c) return fail(GSIM_ERR_NOMEM, "out of host memory");
    c->devices.assign(devices, devices + ndevices);
    if (alias_devices() && ndevices > 1) {
        c->loopback = true; // (all logical devices are physical GPU 0: RCCL refuses two ranks on one device)
    } else {
        std::vector<int> phys(ndevices);
        for (int i = 0; i < ndevices; i++) phys[i] = phys_device(devices[i]);
        c->comms.resize(static_cast<size_t>(ndevices));
        const ncclResult_t r = ncclCommInitAll(c->comms.data(), ndevices, phys.data());
        if (r != ncclSuccess) {
            delete c;
            return fail_nccl(r, "ncclCommInitAll");
        }
    }
    *out = c;
    return GSIM_OK;
}

int gsim_comm_destroy(gsim_comm* comm)
{
    if (!comm) return GSIM_OK;
    for (auto c : comm->comms) (void) ncclCommDestroy(c);
    delete comm;
    return GSIM_OK;
}

int gsim_comm_size(const gsim_comm* comm)
{
    return comm ? static_cast<int>(comm->devices.size()) : 0;
}

int gsim_db_set_comm(gsim_db* db, gsim_comm* comm)
{
    if (!db || !db->finalized) return fail(GSIM_ERR_STATE, "table not finalized");
    std::lock_guard<std::mutex> guard(db->search_mutex);
    if (comm) {
        if (db->fold > 1) return fail(GSIM_ERR_STATE, "folded tables merge on the host (their re-score needs every storage's candidates there)");
        if (comm->devices.size() != db->shards.size()) return fail(GSIM_ERR_INVALID, "the communicator's size differs from the handle's shard count");
        for (size_t i = 0; i < db->shards.size(); i++)
            if (comm->devices[i] != db->shards[i].device) return fail(GSIM_ERR_INVALID, "the communicator's devices differ from the shards' devices");
    }
    db->comm = comm;
    db->comm_root = 0;
    return GSIM_OK;
}

int gsim_db_set_comm_root(gsim_db* db, int shard)
{
    if (!db || !db->finalized) return fail(GSIM_ERR_STATE, "table not finalized");
    std::lock_guard<std::mutex> guard(db->search_mutex);
    if (shard < 0 || static_cast<size_t>(shard) >= db->shards.size()) return fail(GSIM_ERR_INVALID, "no such shard");
    db->comm_root = static_cast<size_t>(shard);
    return GSIM_OK;
}

} // extern "C"
