// gkc_dist.hip — multi-GPU under the C-ABI (include/gkc.h, "Multi-GPU"): communicators, the super-k-mer exchange, and the
// word-array reductions the distributed Bloom filter and MPHF are built from.
//
// What it replaces: the reference has no distributed path. Its hand-over between the two stages is the disk shuffle of
// SuperKmerBinFiles (tools/storage/impl/Storage.cpp:360-430: fillPartitions appends super-k-mers to one file per partition,
// SortingCountAlgorithm.cpp:1211-1344; fillSolidKmers reads the files back, :1384-1602). Here the same hand-over between N GPUs is one
// grouped point-to-point exchange of the device buckets: same canonical k-mer => same minimizer => same partition => one owner rank.
//
// MI355X shape: xGMI is point-to-point (7 links per GPU), so the all-to-all is issued as ONE ncclGroup of ncclSend / ncclRecv — every link
// busy at once, no ring — on the communicator's own HIP stream: the exchange of push i runs while Stage A scans push i+1, and
// gkc_finish_pass waits for the event. The bytes for one destination are one contiguous slice of the partition-major bucket arena per
// segment, so nothing is packed. Owner ranges are balanced by the k-mers per partition the ranks report in the first exchange of a pass.
#include "gkc_common.hpp"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <algorithm>
#include <chrono>
#include <numeric>
#include <string>
#include <unistd.h>
#include <sys/stat.h>
#include <dirent.h>
#include <map>
#include <mutex>
#include <random>

namespace {
constexpr uint64_t MSG_CHUNK = 1ull << 28;     // 256 MiB per message: through torch's all_to_all_single a 1 GiB transfer is intact, a 1.9 GiB one comes back corrupt from
                                               // byte 973 Mi on and 2 GiB and more fail outright on this stack (tools/rccl_2gib_probe.py, profiles/r02_rccl_2gib_probe.txt)
}

struct gkc_comm {
    gkc_ctx* ctx = nullptr; int world = 1, rank = 0;
    bool rccl = false; ncclComm_t nccl = nullptr;
    gkc_transport t{};
    void* owned_user = nullptr; void (*owned_free)(void*) = nullptr;   // transport state the library itself created (gkc_comm_create_files)
    hipStream_t xstream = nullptr;
    std::vector<uint32_t> first;               // owner ranges [world+1]; empty until known
    bool owners_pinned = false; uint32_t owners_pass = ~0u; uint32_t owners_P = 0;
    gkc_comm_stats stats{};
    std::vector<uint64_t> peer_sent, peer_recv;   // [world] bytes of grouped send / recv messages per peer (gkc_comm_peer_bytes)
    double init_ms = 0;                        // wall of ncclCommInitRank (RCCL) / of the session handshake (file mailbox)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> timed;      // transfer intervals not yet added to stats.ms_transfer
    DevBuf ag_send, ag_recv;                   // staging of the host all-gather (RCCL)
    uint64_t stats_bounced = 0;                // receive buffers of the IPC transport that went through a bounce block
    bool ipc = false;                          // transport communicator whose device messages go peer to peer through IPC memory handles (gkc_comm_enable_ipc)
};

// RCCL is bound at run time, when the first RCCL communicator is asked for: a single-GPU host without librccl loads libgkc_hip.so all the same
// (ADVICE r2). Inside a PyTorch process the library of that name is already mapped (torch/lib/librccl.so) and dlopen returns it.
struct RcclApi {
    void* lib = nullptr; bool tried = false; std::string why;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool load() {
        static std::mutex mu; std::lock_guard<std::mutex> lk(mu);
        if (tried) return lib != nullptr;
        tried = true;
        for (const char* name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so" }) { lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL); if (lib) break; }
        if (!lib) { const char* e = dlerror(); why = e ? e : "librccl.so not found"; return false; }
        bool ok = true;
        auto sym = [&](const char* n) -> void* { void* p = dlsym(lib, n); if (!p) { ok = false; why = std::string("symbol ") + n + " missing in librccl"; } return p; };
        GetUniqueId = (decltype(GetUniqueId))sym("ncclGetUniqueId"); CommInitRank = (decltype(CommInitRank))sym("ncclCommInitRank"); CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        AllGather = (decltype(AllGather))sym("ncclAllGather"); Send = (decltype(Send))sym("ncclSend"); Recv = (decltype(Recv))sym("ncclRecv");
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart"); GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd"); GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        if (!ok) { dlclose(lib); lib = nullptr; }
        return ok;
    }
};
static RcclApi g_rccl;

static int comm_fail(gkc_comm* m, int code, const char* what, const char* detail) { m->ctx->set_error(code, "%s: %s", what, detail); return code; }
#define NCCL_TRY(m, call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) return comm_fail((m), GKC_ERR_HIP, #call, g_rccl.GetErrorString(r_)); } while (0)

int gkc_comm_world(gkc_comm* m) { return m->world; }
int gkc_comm_rank(gkc_comm* m) { return m->rank; }

// every rank contributes n bytes of host memory; all = [world][n]
int gkc_comm_allgather_host(gkc_comm* m, const void* mine, size_t n, void* all)
{
    gkc_ctx* c = m->ctx;
    if (m->world == 1) { memcpy(all, mine, n); return GKC_OK; }
    if (!m->rccl) {
        if (m->t.allgather_host(m->t.user, mine, (uint64_t)n, all) != 0) GKC_FAIL(c, GKC_ERR_HIP, "transport all-gather failed");
        return GKC_OK;
    }
    GKC_TRY(c->ensure(m->ag_send, n + 16)); GKC_TRY(c->ensure(m->ag_recv, n * m->world + 16));
    GKC_HIP(c, hipMemcpyAsync(m->ag_send.p, mine, n, hipMemcpyHostToDevice, m->xstream));
    NCCL_TRY(m, g_rccl.AllGather(m->ag_send.p, m->ag_recv.p, n, ncclUint8, m->nccl, m->xstream));
    GKC_HIP(c, hipMemcpyAsync(all, m->ag_recv.p, n * m->world, hipMemcpyDeviceToHost, m->xstream));
    GKC_HIP(c, hipStreamSynchronize(m->xstream));
    return GKC_OK;
}

// one grouped exchange, ordered on stream st (RCCL: enqueued, not waited for; transport: st is drained first, the callback blocks)
// A rank that fails on its own just before a collective would leave the others blocked in it: the ranks exchange their local status first and
// fail TOGETHER (every rank returns an error, the first failing rank's code; the failing rank keeps its own message).
int gkc_comm_agree(gkc_comm* m, int local_rc, const char* where)
{
    if (m->world == 1) return local_rc;
    int32_t mine = local_rc; std::vector<int32_t> all((size_t)m->world, 0);
    const std::string kept = local_rc != GKC_OK ? m->ctx->err.msg : std::string();
    const int rc = gkc_comm_allgather_host(m, &mine, sizeof(mine), all.data());
    if (rc != GKC_OK) return rc;
    for (int r = 0; r < m->world; r++)
        if (all[r] != GKC_OK) {
            if (local_rc != GKC_OK) { m->ctx->set_error(local_rc, "%s", kept.c_str()); return local_rc; }
            m->ctx->set_error(all[r], "rank %d failed in %s (error %d): every rank gives up together", r, where, all[r]);
            return all[r];
        }
    return GKC_OK;
}


// ------------------------------------------------------------------------------------------------ device-to-device fallback transport (round 5)
// When RCCL refuses the communicator (gatb_core_amd/dist.py make_comm: every rank falls back together) the host-staged transport keeps the run alive at the price of
// two PCIe crossings and a TCP copy per message. This one keeps the bytes on the devices: every receive buffer is published as (hipIpcMemHandle of its allocation,
// offset, length) through the transport's HOST all-gather, the SENDER opens the handle and copies device to device (peer access over xGMI between GPUs of a node;
// a plain copy when the ranks share a GPU), a second all-gather tells everybody that what was sent to them has landed. Handles are opened per exchange and closed
// after it: the allocator of the peer may hand the block to something else later. Needs hipMalloc'ed blocks (an IPC handle names a whole allocation): the pool's
// mapped ranges are switched off for the context (gkc_comm_enable_ipc).
struct IpcEntry { int32_t src; uint32_t pad; uint64_t offset, bytes; hipIpcMemHandle_t handle; };
static int sendrecv_ipc(gkc_comm* m, const std::vector<gkc_xfer>& sends, const std::vector<gkc_xfer>& recvs, hipStream_t st)
{
    gkc_ctx* c = m->ctx;
    const int W = m->world;
    // my receive table
    std::vector<IpcEntry> mine(recvs.size());
    // A receive buffer whose allocation cannot be exported (a hipMemCreate-mapped range the context held before the communicator existed) is received into a
    // hipMalloc'ed bounce block and copied to its place once everything has landed: slower by one device copy, never a failed exchange.
    struct Bounce { void* tmp; void* dst; size_t n; };
    std::vector<Bounce> bounces;
    auto export_ptr = [&](IpcEntry& e, const void* ptr) -> bool {
        hipDeviceptr_t base = nullptr; size_t span = 0;
        if (hipMemGetAddressRange(&base, &span, (hipDeviceptr_t)ptr) != hipSuccess || hipIpcGetMemHandle(&e.handle, (void*)base) != hipSuccess) { (void)hipGetLastError(); return false; }
        e.offset = (uint64_t)((const uint8_t*)ptr - (const uint8_t*)base);
        return true;
    };
    for (size_t i = 0; i < recvs.size(); i++) {
        IpcEntry& e = mine[i]; memset(&e, 0, sizeof e);
        e.src = recvs[i].peer; e.bytes = recvs[i].n_bytes;
        if (!recvs[i].n_bytes) continue;
        const bool mapped = c->pool.is_mapped_inside(recvs[i].d_ptr);
        if (mapped || !export_ptr(e, recvs[i].d_ptr)) {
            void* tmp = nullptr;
            if (hipMalloc(&tmp, (size_t)recvs[i].n_bytes) == hipSuccess && export_ptr(e, tmp)) bounces.push_back(Bounce{ tmp, recvs[i].d_ptr, (size_t)recvs[i].n_bytes });
            else { (void)hipGetLastError(); if (tmp) (void)hipFree(tmp); e.src = -2; }      // (published all the same: the ranks fail together below)
        }
    }
    uint32_t n_mine = (uint32_t)mine.size(); std::vector<uint32_t> n_all((size_t)W, 0);
    GKC_TRY(gkc_comm_allgather_host(m, &n_mine, 4, n_all.data()));
    uint32_t n_max = 1; for (int r = 0; r < W; r++) n_max = std::max(n_max, n_all[r]);
    std::vector<IpcEntry> padded(n_max), all((size_t)W * n_max);
    memset(padded.data(), 0, padded.size() * sizeof(IpcEntry)); if (n_mine) memcpy(padded.data(), mine.data(), (size_t)n_mine * sizeof(IpcEntry));
    GKC_TRY(gkc_comm_allgather_host(m, padded.data(), (uint64_t)n_max * sizeof(IpcEntry), all.data()));
    int rc = GKC_OK; std::string why;
    for (int r = 0; r < W && rc == GKC_OK; r++) for (uint32_t i = 0; i < n_all[r]; i++) if (all[(size_t)r * n_max + i].src == -2) { rc = GKC_ERR_HIP; why = "rank " + std::to_string(r) + " could not export a receive buffer (hipIpcGetMemHandle)"; break; }
    // my sends: the j-th message to peer p goes to the j-th entry of p's table whose source is me
    std::vector<std::pair<hipIpcMemHandle_t, void*>> opened;
    auto open_handle = [&](const hipIpcMemHandle_t& h) -> void* {
        for (auto& o : opened) if (memcmp(&o.first, &h, sizeof h) == 0) return o.second;
        void* p = nullptr;
        if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        opened.push_back({h, p}); return p;
    };
    std::vector<uint32_t> cursor((size_t)W, 0);
    for (size_t j = 0; j < sends.size() && rc == GKC_OK; j++) {
        const int p = sends[j].peer;
        if (p < 0 || p >= W) { rc = GKC_ERR_ARG; why = "send to a rank outside the communicator"; break; }
        const IpcEntry* e = nullptr;
        while (cursor[p] < n_all[p]) { const IpcEntry& x = all[(size_t)p * n_max + cursor[p]++]; if (x.src == m->rank) { e = &x; break; } }
        if (!e || e->bytes != sends[j].n_bytes) { rc = GKC_ERR_HIP; why = "the receive posted by rank " + std::to_string(p) + " does not match the send"; break; }
        if (!e->bytes) continue;
        void* base = open_handle(e->handle);
        if (!base) { rc = GKC_ERR_HIP; why = "hipIpcOpenMemHandle of a buffer of rank " + std::to_string(p) + " failed"; break; }
        if (hipMemcpyAsync((uint8_t*)base + e->offset, sends[j].d_ptr, (size_t)e->bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) { (void)hipGetLastError(); rc = GKC_ERR_HIP; why = "device-to-device copy failed"; }
    }
    if (hipStreamSynchronize(st) != hipSuccess && rc == GKC_OK) { (void)hipGetLastError(); rc = GKC_ERR_HIP; why = "device-to-device copy failed"; }
    for (auto& o : opened) (void)hipIpcCloseMemHandle(o.second);
    if (rc != GKC_OK) c->set_error(rc, "IPC transport: %s", why.c_str());
    // everybody's copies have landed (or everybody learns that somebody failed)
    rc = gkc_comm_agree(m, rc, "the device-to-device exchange");
    for (const Bounce& b : bounces) if (rc == GKC_OK && hipMemcpyAsync(b.dst, b.tmp, b.n, hipMemcpyDeviceToDevice, st) != hipSuccess) { (void)hipGetLastError(); c->set_error(GKC_ERR_HIP, "IPC transport: copy out of a bounce block failed"); rc = GKC_ERR_HIP; }
    if (!bounces.empty()) { (void)hipStreamSynchronize(st); for (const Bounce& b : bounces) (void)hipFree(b.tmp); m->stats_bounced += bounces.size(); }
    return rc;
}

int gkc_comm_sendrecv(gkc_comm* m, const std::vector<gkc_xfer>& sends, const std::vector<gkc_xfer>& recvs, hipStream_t st)
{
    gkc_ctx* c = m->ctx;
    // (a rank with nothing to send or receive — it owns no partition and pushed nothing since the last exchange — still takes part in the IPC transport's
    // all-gathers: leaving here would shift every later collective of the communicator by one on this rank; found by the 4- and 8-rank tests of round 6)
    if (sends.empty() && recvs.empty() && !(m->ipc && !m->rccl && m->world > 1)) return GKC_OK;
    // split long messages the same way on both sides
    std::vector<gkc_xfer> s2, r2;
    auto split = [](const std::vector<gkc_xfer>& in, std::vector<gkc_xfer>& out) {
        for (const gkc_xfer& x : in)
            for (uint64_t o = 0; o < x.n_bytes; o += MSG_CHUNK) out.push_back(gkc_xfer{ x.peer, 0, (uint8_t*)x.d_ptr + o, std::min<uint64_t>(MSG_CHUNK, x.n_bytes - o) });
    };
    split(sends, s2); split(recvs, r2);
    if (m->peer_sent.size() != (size_t)m->world) { m->peer_sent.assign(m->world, 0); m->peer_recv.assign(m->world, 0); }
    for (const gkc_xfer& x : s2) if (x.peer >= 0 && x.peer < m->world) m->peer_sent[x.peer] += x.n_bytes;
    for (const gkc_xfer& x : r2) if (x.peer >= 0 && x.peer < m->world) m->peer_recv[x.peer] += x.n_bytes;
    if (m->rccl) {
        NCCL_TRY(m, g_rccl.GroupStart());
        for (const gkc_xfer& x : s2) NCCL_TRY(m, g_rccl.Send(x.d_ptr, (size_t)x.n_bytes, ncclUint8, x.peer, m->nccl, st));
        for (const gkc_xfer& x : r2) NCCL_TRY(m, g_rccl.Recv(x.d_ptr, (size_t)x.n_bytes, ncclUint8, x.peer, m->nccl, st));
        NCCL_TRY(m, g_rccl.GroupEnd());
        return GKC_OK;
    }
    GKC_HIP(c, hipStreamSynchronize(st));
    if (m->ipc) return sendrecv_ipc(m, s2, r2, st);
    if (m->t.sendrecv_device(m->t.user, s2.data(), (uint32_t)s2.size(), r2.data(), (uint32_t)r2.size()) != 0) GKC_FAIL(c, GKC_ERR_HIP, "transport send/recv failed");
    return GKC_OK;
}

// ------------------------------------------------------------------------------------------------ word-array reductions
// seen = OR over ranks; coll = OR over ranks | (a bit seen by two ranks). acc_* live in the rank's own slice of its arrays.
__global__ void k_or_rows(uint64_t* __restrict__ acc, const uint64_t* __restrict__ rows, uint64_t n, uint32_t n_rows, uint64_t row_stride)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t a = acc[i];
        for (uint32_t r = 0; r < n_rows; r++) a |= rows[(uint64_t)r * row_stride + i];
        acc[i] = a;
    }
}
__global__ void k_seen_coll_rows(uint64_t* __restrict__ seen, const uint64_t* __restrict__ coll, const uint64_t* __restrict__ rows_seen, const uint64_t* __restrict__ rows_coll,
                                 uint64_t n, uint32_t n_rows, uint64_t row_stride)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t s = seen[i], k = coll[i];
        for (uint32_t r = 0; r < n_rows; r++) { const uint64_t rs = rows_seen[(uint64_t)r * row_stride + i]; k |= rows_coll[(uint64_t)r * row_stride + i] | (s & rs); s |= rs; }
        seen[i] = s & ~k;                                   // clearCollisions (BooPHF.h:511-523) applied to the combined level
    }
}

// Reduce-scatter + all-gather over n_words 64-bit words, in place, ordered on st. mode 0: words |= everybody's words.
// mode 1: (seen, coll) -> seen = (OR seen) & ~(OR coll | seen-by-two), complete on every rank; coll is left undefined.
static int reduce_words(gkc_comm* m, uint64_t* d_a, uint64_t* d_b, uint64_t n_words, int mode, hipStream_t st)
{
    gkc_ctx* c = m->ctx;
    const int W = m->world, me = m->rank;
    if (W == 1) {
        if (mode == 1 && n_words) { hipLaunchKernelGGL(k_seen_coll_rows, dim3((unsigned)std::min<uint64_t>((n_words + 255) / 256, 256 * 16)), dim3(256), 0, st, d_a, d_b, d_a, d_b, n_words, 0u, 0ull); GKC_HIP(c, hipGetLastError()); }
        return GKC_OK;
    }
    const uint64_t slice = (n_words + W - 1) / W;
    auto lo = [&](int r) { return std::min<uint64_t>((uint64_t)r * slice, n_words); };
    auto len = [&](int r) { return lo(r + 1) - lo(r); };
    const uint64_t mine = len(me);
    const int arrays = mode == 1 ? 2 : 1;
    DevBuf scratch;
    GKC_TRY(c->ensure(scratch, (size_t)std::max<uint64_t>((uint64_t)(W - 1) * slice * arrays, 1) * 8));
    uint64_t* rows = (uint64_t*)scratch.p;                 // [arrays][W-1][slice]
    std::vector<gkc_xfer> sends, recvs;
    int row = 0;
    for (int r = 0; r < W; r++) {
        if (r == me) continue;
        if (len(r)) { sends.push_back(gkc_xfer{ r, 0, d_a + lo(r), len(r) * 8 }); if (mode == 1) sends.push_back(gkc_xfer{ r, 0, d_b + lo(r), len(r) * 8 }); }
        if (mine)   { recvs.push_back(gkc_xfer{ r, 0, rows + (uint64_t)row * slice, mine * 8 }); if (mode == 1) recvs.push_back(gkc_xfer{ r, 0, rows + (uint64_t)(W - 1 + row) * slice, mine * 8 }); }
        row++;
    }
    int rc = gkc_comm_sendrecv(m, sends, recvs, st);
    if (rc == GKC_OK && mine) {
        const unsigned grid = (unsigned)std::min<uint64_t>((mine + 255) / 256, 256 * 16);
        if (mode == 0) hipLaunchKernelGGL(k_or_rows, dim3(grid), dim3(256), 0, st, d_a + lo(me), (const uint64_t*)rows, mine, (uint32_t)(W - 1), slice);
        else hipLaunchKernelGGL(k_seen_coll_rows, dim3(grid), dim3(256), 0, st, d_a + lo(me), (const uint64_t*)(d_b + lo(me)), (const uint64_t*)rows, (const uint64_t*)(rows + (uint64_t)(W - 1) * slice), mine, (uint32_t)(W - 1), slice);
        if (hipGetLastError() != hipSuccess) { c->set_error(GKC_ERR_HIP, "reduction kernel failed"); rc = GKC_ERR_HIP; }
    }
    if (rc == GKC_OK) {                                     // all-gather of the reduced slices
        sends.clear(); recvs.clear();
        for (int r = 0; r < W; r++) {
            if (r == me) continue;
            if (mine) sends.push_back(gkc_xfer{ r, 0, d_a + lo(me), mine * 8 });
            if (len(r)) recvs.push_back(gkc_xfer{ r, 0, d_a + lo(r), len(r) * 8 });
        }
        rc = gkc_comm_sendrecv(m, sends, recvs, st);
    }
    (void)hipStreamSynchronize(st);                          // scratch goes back to the pool
    scratch.release();
    return rc;
}
int gkc_comm_allreduce_or_words(gkc_comm* m, uint64_t* d_words, uint64_t n_words, hipStream_t st) { return reduce_words(m, d_words, nullptr, n_words, 0, st); }
int gkc_comm_combine_seen_coll(gkc_comm* m, uint64_t* d_seen, uint64_t* d_coll, uint64_t n_words, hipStream_t st) { return reduce_words(m, d_seen, d_coll, n_words, 1, st); }

void gkc_comm_settle_timers(gkc_comm* m)
{
    for (auto& pr : m->timed) {
        if (hipEventSynchronize(pr.second) == hipSuccess) { float ms = 0; if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) m->stats.ms_transfer += ms; }
        (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second);
    }
    m->timed.clear();
}

// ------------------------------------------------------------------------------------------------ file-mailbox transport (gkc_comm_create_files)
// Collective calls happen in the same order on every rank, so a per-communicator sequence number names the files of one call. A file becomes visible
// atomically (written under a temporary name, then renamed); a reader polls for it. all-gather: rank r writes <tag>.ag.<seq>.<r>, reads the others', and removes
// its own file of call seq-1 when it enters call seq+1 (every rank has read call seq-1 before it wrote its file of call seq). send/recv: one file per
// message, named by (seq, source, destination, index of the message between the two in this call), removed by the receiver.
// <tag> = <generation>.<nonce>: every communicator has its own file names, so what a crashed run left in the directory (ag.* / p2p.* with a header of the
// expected size) is never read by a later run. generation = how many file communicators this process has made for the directory before (the ranks
// create their communicators in the same order); the nonce is drawn by rank 0 and published in session.<generation>, the others answer with join.<tag>.<rank>
// and wait for go.<tag>, which rank 0 writes once everybody has joined — a rank that picked up a stale session file waits for a go that never comes and
// fails loudly instead of exchanging with nobody. Rank 0 of generation 0 removes the handshake files earlier runs left.
namespace {
struct FileBox {
    std::string dir, tag; int world, rank; uint64_t seq_ag = 0, seq_p2p = 0;
    std::vector<uint8_t> stage;
    static bool write_file(const std::string& path, const void* data, size_t n) {
        const std::string tmp = path + ".tmp";
        FILE* f = fopen(tmp.c_str(), "wb"); if (!f) return false;
        const bool ok = n == 0 || fwrite(data, 1, n, f) == n;
        if (fclose(f) != 0 || !ok) { (void)unlink(tmp.c_str()); return false; }
        if (rename(tmp.c_str(), path.c_str()) != 0) { (void)unlink(tmp.c_str()); return false; }
        return true;
    }
    static void remove_matching(const std::string& dir, const std::vector<std::string>& prefixes) {
        DIR* d = opendir(dir.c_str()); if (!d) return;
        std::vector<std::string> gone;
        while (struct dirent* e = readdir(d)) { const std::string n = e->d_name; for (const std::string& p : prefixes) if (n.compare(0, p.size(), p) == 0) { gone.push_back(n); break; } }
        closedir(d);
        for (const std::string& n : gone) (void)unlink((dir + "/" + n).c_str());
    }
    // the handshake described above; false: the others did not show up (or a stale session file was picked up) within the time limit
    bool open_session(double timeout_s) {
        static std::mutex mu; static std::map<std::string, uint64_t> generations;
        uint64_t gen; { std::lock_guard<std::mutex> lk(mu); gen = generations[dir]++; }
        const std::string session = dir + "/session." + std::to_string(gen);
        uint64_t nonce = 0;
        if (rank == 0) {
            if (gen == 0) remove_matching(dir, { "session.", "join.", "go." });
            std::random_device rd; nonce = ((uint64_t)rd() << 32) ^ (uint64_t)rd() ^ ((uint64_t)getpid() << 20) ^ (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count();
            if (!write_file(session, &nonce, 8)) return false;
        }
        auto tag_of = [&](uint64_t nn) { char hex[32]; snprintf(hex, sizeof hex, "%016llx", (unsigned long long)nn); return std::to_string(gen) + "." + hex; };
        // join.<tag>.<r> carries a nonce of the joining PROCESS and go.<tag> echoes the nonces of everybody who joined: a go file a crashed run left behind (same
        // tag as a session file that also survived) does not hold the nonce this process has just drawn, and is ignored.
        if (rank == 0) {
            tag = tag_of(nonce);
            std::vector<uint64_t> echo((size_t)world, 0); echo[0] = nonce;
            for (int r = 1; r < world; r++) { if (!read_file(dir + "/join." + tag + "." + std::to_string(r), &echo[r], 8, timeout_s)) return false; (void)unlink((dir + "/join." + tag + "." + std::to_string(r)).c_str()); }
            return write_file(dir + "/go." + tag, echo.data(), (size_t)world * 8);
        }
        std::random_device rd_; const uint64_t my_nonce = (((uint64_t)rd_() << 32) ^ (uint64_t)rd_() ^ ((uint64_t)getpid() << 24) ^ (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count()) | 1ull;
        std::vector<uint64_t> echo((size_t)world, 0);
        // A joining rank may start before rank 0 has cleaned the directory: the session file it finds can be what a crashed run left. It therefore (i) ignores a
        // session file older than the time limit (its rank 0 has given up long ago), (ii) keeps re-reading the file while it waits for go.<tag> and joins again when the
        // nonce changes — the live rank 0 overwrites session.<generation> when it arrives.
        const auto t0 = std::chrono::steady_clock::now();
        bool joined = false; uint64_t joined_nonce = 0;
        for (;;) {
            struct stat st; uint64_t seen = 0;
            if (stat(session.c_str(), &st) == 0 && (double)(time(nullptr) - st.st_mtime) <= timeout_s && read_file(session, &seen, 8, 0.0)) {
                if (!joined || seen != joined_nonce) {
                    tag = tag_of(seen);
                    if (!write_file(dir + "/join." + tag + "." + std::to_string(rank), &my_nonce, 8)) return false;
                    joined = true; joined_nonce = seen;
                }
                if (read_file(dir + "/go." + tag, echo.data(), (size_t)world * 8, 0.0) && echo[(size_t)rank] == my_nonce) return true;
            }
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return false;
            usleep(1000);
        }
    }
    static bool read_file(const std::string& path, void* data, size_t n, double timeout_s) {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            struct stat st;
            if (stat(path.c_str(), &st) == 0 && (size_t)st.st_size == n) {
                FILE* f = fopen(path.c_str(), "rb");
                if (f) { const bool ok = n == 0 || fread(data, 1, n, f) == n; fclose(f); if (ok) return true; }
            }
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return false;
            usleep(200);
        }
    }
    std::string ag_name(uint64_t seq, int r) const { return dir + "/" + tag + ".ag." + std::to_string(seq) + "." + std::to_string(r); }
    static int allgather(void* user, const void* mine, uint64_t n, void* all) {
        FileBox* b = (FileBox*)user;
        const uint64_t seq = b->seq_ag++;
        if (seq >= 2) (void)unlink(b->ag_name(seq - 2, b->rank).c_str());
        if (!write_file(b->ag_name(seq, b->rank), mine, (size_t)n)) return 1;
        for (int r = 0; r < b->world; r++) {
            if (r == b->rank) { memcpy((uint8_t*)all + (size_t)r * n, mine, (size_t)n); continue; }
            if (!read_file(b->ag_name(seq, r), (uint8_t*)all + (size_t)r * n, (size_t)n, 600.0)) return 1;
        }
        return 0;
    }
    static int sendrecv(void* user, const gkc_xfer* sends, uint32_t n_sends, const gkc_xfer* recvs, uint32_t n_recvs) {
        FileBox* b = (FileBox*)user;
        const uint64_t seq = b->seq_p2p++;
        std::vector<uint32_t> idx(b->world, 0);
        for (uint32_t i = 0; i < n_sends; i++) {
            const gkc_xfer& x = sends[i];
            b->stage.resize((size_t)x.n_bytes);
            if (x.n_bytes && hipMemcpy(b->stage.data(), x.d_ptr, (size_t)x.n_bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
            const std::string name = b->dir + "/" + b->tag + ".p2p." + std::to_string(seq) + "." + std::to_string(b->rank) + "." + std::to_string(x.peer) + "." + std::to_string(idx[x.peer]++);
            if (!write_file(name, b->stage.data(), (size_t)x.n_bytes)) return 1;
        }
        std::fill(idx.begin(), idx.end(), 0u);
        for (uint32_t i = 0; i < n_recvs; i++) {
            const gkc_xfer& x = recvs[i];
            b->stage.resize((size_t)x.n_bytes);
            const std::string name = b->dir + "/" + b->tag + ".p2p." + std::to_string(seq) + "." + std::to_string(x.peer) + "." + std::to_string(b->rank) + "." + std::to_string(idx[x.peer]++);
            if (!read_file(name, b->stage.data(), (size_t)x.n_bytes, 600.0)) return 1;
            (void)unlink(name.c_str());
            if (x.n_bytes && hipMemcpy(x.d_ptr, b->stage.data(), (size_t)x.n_bytes, hipMemcpyHostToDevice) != hipSuccess) return 1;
        }
        return 0;
    }
    static void destroy(void* user) {
        FileBox* b = (FileBox*)user;
        for (uint64_t s = b->seq_ag >= 2 ? b->seq_ag - 2 : 0; s < b->seq_ag; s++) (void)unlink(b->ag_name(s, b->rank).c_str());
        if (b->rank == 0 && !b->tag.empty()) { (void)unlink((b->dir + "/go." + b->tag).c_str()); (void)unlink((b->dir + "/session." + b->tag.substr(0, b->tag.find('.'))).c_str()); }
        delete b;
    }
};
}

extern "C" {

int gkc_balanced_owner_ranges(const uint64_t* weights, uint32_t P, int world, uint32_t* first)
{
    if (!weights || !first || world < 1 || P < 1) return GKC_ERR_ARG;
    // contiguous ranges; boundary r is the first partition whose prefix weight reaches r/world of the total (every partition also counts
    // for one unit, so that empty tails still spread); ranges may be empty only when there are fewer partitions than ranks
    unsigned __int128 total = 0;
    for (uint32_t p = 0; p < P; p++) total += (unsigned __int128)weights[p] + 1;
    first[0] = 0;
    unsigned __int128 acc = 0; uint32_t p = 0;
    for (int r = 1; r < world; r++) {
        const unsigned __int128 want = total * (unsigned)r / (unsigned)world;
        while (p < P && acc + ((unsigned __int128)weights[p] + 1) / 2 < want) { acc += (unsigned __int128)weights[p] + 1; p++; }
        first[r] = p;
    }
    first[world] = P;
    return GKC_OK;
}

int gkc_exchange_plan(int world, int rank, uint32_t P, const uint32_t* first, const uint64_t* n_segs, uint64_t l_max, const uint64_t* counts,
                      gkc_plan_msg* sends, uint32_t* n_sends, gkc_plan_msg* recvs, uint32_t* n_recvs, uint64_t* recv_total_recs)
{
    if (world < 1 || rank < 0 || rank >= world || !first || !n_segs || !counts || !sends || !n_sends || !recvs || !n_recvs || !recv_total_recs) return GKC_ERR_ARG;
    if (first[0] != 0 || first[world] != P) return GKC_ERR_ARG;
    auto cnt = [&](int r, uint64_t j, uint32_t p) -> uint64_t { return counts[(((size_t)r * l_max + j) * 2 + 0) * P + p]; };
    uint32_t ns = 0, nr = 0; uint64_t pos = 0;
    for (int r = 0; r < world; r++) {                      // what leaves: per destination, per own segment, one contiguous slice of the arena
        if (r == rank) continue;
        for (uint64_t j = 0; j < n_segs[rank]; j++) {
            uint64_t a = 0, n = 0;
            for (uint32_t p = 0; p < first[r]; p++) a += cnt(rank, j, p);
            for (uint32_t p = first[r]; p < first[r + 1]; p++) n += cnt(rank, j, p);
            if (n) sends[ns++] = gkc_plan_msg{ r, (uint32_t)j, a, n };
        }
    }
    for (int r = 0; r < world; r++) {                      // what arrives: per source, per segment of the source, back to back
        if (r == rank) continue;
        for (uint64_t j = 0; j < n_segs[r]; j++) {
            uint64_t n = 0;
            for (uint32_t p = first[rank]; p < first[rank + 1]; p++) n += cnt(r, j, p);
            if (n) { recvs[nr++] = gkc_plan_msg{ r, (uint32_t)j, pos, n }; pos += n; }
        }
    }
    *n_sends = ns; *n_recvs = nr; *recv_total_recs = pos;
    return GKC_OK;
}

int gkc_comm_unique_id(uint8_t id[GKC_COMM_ID_BYTES])
{
    static_assert(sizeof(ncclUniqueId) == GKC_COMM_ID_BYTES, "ncclUniqueId size");
    if (!id) return GKC_ERR_ARG;
    ncclUniqueId u;
    if (!g_rccl.load() || g_rccl.GetUniqueId(&u) != ncclSuccess) return GKC_ERR_HIP;
    memcpy(id, &u, sizeof(u));
    return GKC_OK;
}

static int comm_new(gkc_ctx* c, int world, int rank, gkc_comm** out)
{
    gkc_tun_refresh();
    if (!c || !out) return GKC_ERR_ARG;
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) GKC_FAIL(c, GKC_ERR_ARG, "bad world / rank (%d / %d)", world, rank);
    GKC_HIP(c, hipSetDevice(c->device));
    gkc_comm* m = new gkc_comm();
    m->ctx = c; m->world = world; m->rank = rank;
    if (hipStreamCreateWithFlags(&m->xstream, hipStreamNonBlocking) != hipSuccess) { delete m; GKC_FAIL(c, GKC_ERR_HIP, "stream creation failed"); }
    gkc_ctx_child_add(c);
    c->comm_world = std::max(c->comm_world, world);
    *out = m;
    return GKC_OK;
}
int gkc_comm_create_rccl(gkc_ctx* c, const uint8_t id[GKC_COMM_ID_BYTES], int world, int rank, gkc_comm** out)
{
    if (!id) return GKC_ERR_ARG;
    GKC_TRY(comm_new(c, world, rank, out));
    gkc_comm* m = *out;
    ncclUniqueId u; memcpy(&u, id, sizeof(u));
    if (!g_rccl.load()) { c->set_error(GKC_ERR_HIP, "RCCL is not available on this host (%s): a communicator over several GPUs needs librccl", g_rccl.why.c_str()); gkc_comm_destroy(m); *out = nullptr; return GKC_ERR_HIP; }
    const auto t_init = std::chrono::steady_clock::now();
    const ncclResult_t r = g_rccl.CommInitRank(&m->nccl, world, u, rank);
    m->init_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_init).count();
    if (r != ncclSuccess) { c->set_error(GKC_ERR_HIP, "ncclCommInitRank failed: %s", g_rccl.GetErrorString(r)); gkc_comm_destroy(m); *out = nullptr; return GKC_ERR_HIP; }
    m->rccl = true;
    // several GPUs: the blocks RCCL sends from / receives into come from hipMalloc as in rounds 1-4 (whether it takes hipMemCreate-mapped ranges as user buffers
    // could not be checked on a one-GPU box; GKC_VMM_WITH_RCCL=1 to try). What the pool already handed out stays valid.
    if (world > 1 && !gkc_tun().vmm_with_rccl) { std::lock_guard<std::recursive_mutex> lk(c->pool.mu); c->pool.vmm_ok = false; c->pool.trim(); }      // (parked mapped ranges go: none is handed out as an exchange buffer later)
    return GKC_OK;
}
int gkc_comm_create_transport(gkc_ctx* c, const gkc_transport* t, int world, int rank, gkc_comm** out)
{
    if (!t || !t->allgather_host || !t->sendrecv_device) return GKC_ERR_ARG;
    GKC_TRY(comm_new(c, world, rank, out));
    (*out)->t = *t;
    return GKC_OK;
}
int gkc_comm_enable_ipc(gkc_comm* m, int on)
{
    if (!m) return GKC_ERR_ARG;
    if (m->rccl) { m->ctx->set_error(GKC_ERR_ARG, "gkc_comm_enable_ipc: an RCCL communicator moves its messages itself"); return GKC_ERR_ARG; }
    m->ipc = on != 0;
    if (m->ipc) { std::lock_guard<std::recursive_mutex> lk(m->ctx->pool.mu); m->ctx->pool.vmm_ok = false; m->ctx->pool.trim(); }      // an IPC handle names a hipMalloc allocation
    return GKC_OK;
}
int gkc_comm_create_files(gkc_ctx* c, const char* directory, int world, int rank, gkc_comm** out)
{
    if (!directory || !*directory) return GKC_ERR_ARG;
    GKC_TRY(comm_new(c, world, rank, out));
    FileBox* b = new FileBox();
    b->dir = directory; b->world = world; b->rank = rank;
    if (!b->open_session(gkc_tun().filebox_timeout)) {
        delete b; gkc_comm_destroy(*out); *out = nullptr;
        GKC_FAIL(c, GKC_ERR_ARG, "gkc_comm_create_files: rank %d of %d found no session with the other ranks in %s (stale files of an earlier run, or a rank that never started)", rank, world, directory);
    }
    gkc_comm* m = *out;
    m->t.user = b; m->t.allgather_host = &FileBox::allgather; m->t.sendrecv_device = &FileBox::sendrecv;
    m->owned_user = b; m->owned_free = &FileBox::destroy;
    return GKC_OK;
}
void gkc_comm_destroy(gkc_comm* m)
{
    if (!m) return;
    gkc_ctx* c = m->ctx;
    (void)hipSetDevice(c->device);
    if (m->xstream) (void)hipStreamSynchronize(m->xstream);
    gkc_comm_settle_timers(m);
    if (m->nccl) (void)g_rccl.CommDestroy(m->nccl);
    m->ag_send.release(); m->ag_recv.release();
    if (m->xstream) (void)hipStreamDestroy(m->xstream);
    if (m->owned_user && m->owned_free) m->owned_free(m->owned_user);
    delete m;
    gkc_ctx_child_release(c);
}
int gkc_comm_set_owners(gkc_comm* m, const uint32_t* first)
{
    if (!m) return GKC_ERR_ARG;
    if (!first) { m->owners_pinned = false; m->first.clear(); m->owners_pass = ~0u; return GKC_OK; }
    for (int r = 0; r < m->world; r++) if (first[r] > first[r + 1]) GKC_FAIL(m->ctx, GKC_ERR_ARG, "owner ranges must be non-decreasing");
    if (first[0] != 0) GKC_FAIL(m->ctx, GKC_ERR_ARG, "owner ranges must start at partition 0");
    m->first.assign(first, first + m->world + 1); m->owners_pinned = true; m->owners_P = first[m->world];
    return GKC_OK;
}
int gkc_comm_get_owners(gkc_comm* m, uint32_t* first)
{
    if (!m || !first) return GKC_ERR_ARG;
    if (m->first.empty()) GKC_FAIL(m->ctx, GKC_ERR_ARG, "owner ranges are not known before the first gkc_exchange of a pass (or gkc_comm_set_owners)");
    memcpy(first, m->first.data(), (size_t)(m->world + 1) * 4);
    return GKC_OK;
}
int gkc_comm_get_stats(gkc_comm* m, gkc_comm_stats* out)
{
    if (!m || !out) return GKC_ERR_ARG;
    (void)hipSetDevice(m->ctx->device);
    gkc_comm_settle_timers(m);
    *out = m->stats;
    out->reserved[0] = m->stats_bounced;
    return GKC_OK;
}

int gkc_exchange(gkc_ctx* c, gkc_comm* m)
{
    gkc_tun_refresh();
    if (!c || !m || m->ctx != c) return GKC_ERR_ARG;
    if (!c->in_pass) GKC_FAIL(c, GKC_ERR_ARG, "gkc_exchange outside a pass (gkc_begin_pass first)");
    if (c->stage_b_running || c->stage_b_thread.joinable()) GKC_FAIL(c, GKC_ERR_ARG, "gkc_exchange while gkc_finish_pass_async is in flight (gkc_finish_pass_wait first)");
    GKC_HIP(c, hipSetDevice(c->device));
    const auto t_host0 = std::chrono::steady_clock::now();
    const int W = m->world, me = m->rank;
    const uint32_t P = c->nb_partitions, rb = c->record_bytes;
    // segments pushed since the last exchange of this pass (imported ones are never forwarded)
    std::vector<size_t> mine;
    for (size_t i = c->n_exchanged_segments; i < c->segments.size(); i++) if (c->segments[i].owned && !c->segments[i].foreign) mine.push_back(i);
    // 1) how many segments does everybody bring (pushes per rank may differ)
    //    ... and does everybody count with the same model? (ranks that derived their own Configuration / Repartitor from their own share of the reads would
    //    differ: another partition count makes the tables below mismatch, another table silently splits one k-mer over two owners)
    const uint64_t hdr[2] = { (uint64_t)mine.size(), c->model_hash };
    std::vector<uint64_t> hdrs((size_t)2 * W), Ls(W);
    GKC_TRY(gkc_comm_allgather_host(m, hdr, 16, hdrs.data()));
    for (int r = 0; r < W; r++) {
        Ls[r] = hdrs[2 * (size_t)r];
        if (hdrs[2 * (size_t)r + 1] != c->model_hash)
            GKC_FAIL(c, GKC_ERR_ARG, "rank %d counts with another model than rank %d (k, minimizer size / type, partitions, passes, repartition table or frequency order differ): "
                                     "every rank of a communicator must be configured identically", r, me);
    }
    const uint64_t Lmax = *std::max_element(Ls.begin(), Ls.end());
    if (Lmax == 0) { c->n_exchanged_segments = c->segments.size(); return GKC_OK; }
    // 2) everybody's per-partition record and k-mer counts: [W][Lmax][2][P]
    std::vector<uint64_t> tab((size_t)Lmax * 2 * P, 0), all((size_t)W * Lmax * 2 * P);
    for (size_t j = 0; j < mine.size(); j++) {
        const Segment& sg = c->segments[mine[j]];
        for (uint32_t p = 0; p < P; p++) { tab[(j * 2 + 0) * P + p] = sg.rec_off[p + 1] - sg.rec_off[p]; tab[(j * 2 + 1) * P + p] = sg.nkmers[p]; }
    }
    GKC_TRY(gkc_comm_allgather_host(m, tab.data(), tab.size() * 8, all.data()));
    auto cnt = [&](int r, uint64_t j, int what, uint32_t p) -> uint64_t { return all[(((size_t)r * Lmax + j) * 2 + what) * P + p]; };
    // 3) owner ranges of this pass
    if (!m->owners_pinned && (m->owners_pass != c->pass || m->owners_P != P || m->first.empty())) {
        std::vector<uint64_t> wgt(P, 0);
        for (int r = 0; r < W; r++) for (uint64_t j = 0; j < Ls[r]; j++) for (uint32_t p = 0; p < P; p++) wgt[p] += cnt(r, j, 1, p);
        m->first.assign(W + 1, 0);
        (void)gkc_balanced_owner_ranges(wgt.data(), P, W, m->first.data());
        m->owners_pass = c->pass; m->owners_P = P;
    }
    if (m->first.size() != (size_t)W + 1 || m->first[W] != P) GKC_FAIL(c, GKC_ERR_ARG, "owner ranges do not cover the %u partitions of the context", P);
    const uint32_t lo = m->first[me], hi = m->first[me + 1];
    // 4) the messages (pure host function, shared with the CPU tests): per destination one contiguous slice of every own segment's arena;
    //    per (source, segment) the records of my partitions, back to back in one receive arena
    std::vector<gkc_plan_msg> ps((size_t)W * Lmax + 1), pr((size_t)W * Lmax + 1);
    uint32_t n_ps = 0, n_pr = 0; uint64_t recv_recs = 0;
    uint8_t* rarena = nullptr;
    std::vector<gkc_xfer> sends, recvs;
    uint64_t sent_bytes = 0;
    // (what can fail on this rank alone — the receive arena does not fit, an inconsistency — is agreed on before the transfer: a rank that
    //  simply returned here would leave the others waiting in their send / recv)
    const int local_rc = [&]() -> int {
        if (!strcmp(gkc_tun().fault, "exchange_local")) GKC_FAIL(c, GKC_ERR_NOMEM, "injected fault (GKC_FAULT=exchange_local)");
        if (gkc_exchange_plan(W, me, P, m->first.data(), Ls.data(), Lmax, all.data(), ps.data(), &n_ps, pr.data(), &n_pr, &recv_recs) != GKC_OK)
            GKC_FAIL(c, GKC_ERR_ARG, "internal error: exchange plan");
        if (recv_recs) {
            rarena = (uint8_t*)c->dalloc((size_t)recv_recs * rb);
            if (!rarena) return GKC_ERR_NOMEM;
            c->owned_arenas.push_back(rarena);
        }
        for (uint32_t i = 0; i < n_ps; i++) {
            const Segment& sg = c->segments[mine[ps[i].seg]];
            if (sg.rec_off[m->first[ps[i].peer]] != ps[i].rec_begin) GKC_FAIL(c, GKC_ERR_ARG, "internal error: segment offsets and counts disagree");
            sends.push_back(gkc_xfer{ ps[i].peer, 0, (uint8_t*)sg.d_records + ps[i].rec_begin * rb, ps[i].n_recs * rb }); sent_bytes += ps[i].n_recs * rb;
        }
        return GKC_OK;
    }();
    GKC_TRY(gkc_comm_agree(m, local_rc, "gkc_exchange"));
    std::vector<Segment> imported;
    for (uint32_t i = 0; i < n_pr; i++) {
        const int r = pr[i].peer; const uint64_t j = pr[i].seg;
        recvs.push_back(gkc_xfer{ r, 0, rarena + pr[i].rec_begin * rb, pr[i].n_recs * rb });
        Segment sg; sg.d_records = rarena + pr[i].rec_begin * rb; sg.owned = false; sg.foreign = true;
        sg.rec_off.assign(P + 1, 0); sg.nkmers.assign(P, 0);
        uint64_t run = 0;
        for (uint32_t p = 0; p < P; p++) { sg.rec_off[p] = run; if (p >= lo && p < hi) { run += cnt(r, j, 0, p); sg.nkmers[p] = cnt(r, j, 1, p); } }
        sg.rec_off[P] = run;
        imported.push_back(std::move(sg));
    }
    // 5) the transfer: after Stage A of the pushes, on the communicator's stream; gkc_finish_pass waits for `done`
    hipEvent_t ready = nullptr, t0 = nullptr, done = nullptr;
    GKC_HIP(c, hipEventCreateWithFlags(&ready, hipEventDisableTiming));
    GKC_HIP(c, hipEventRecord(ready, c->stream));
    GKC_HIP(c, hipStreamWaitEvent(m->xstream, ready, 0));
    GKC_HIP(c, hipEventCreate(&t0)); GKC_HIP(c, hipEventCreate(&done));
    GKC_HIP(c, hipEventRecord(t0, m->xstream));
    const auto t_x0 = std::chrono::steady_clock::now();
    int rc = gkc_comm_sendrecv(m, sends, recvs, m->xstream);
    if (!m->rccl) m->stats.ms_transfer += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_x0).count();
    GKC_HIP(c, hipEventRecord(done, m->xstream));
    (void)hipEventDestroy(ready);
    if (rc != GKC_OK) { (void)hipEventDestroy(t0); (void)hipEventDestroy(done); return rc; }
    if (m->rccl) m->timed.push_back({ t0, done }); else (void)hipEventDestroy(t0);
    {   hipEvent_t wait_ev = nullptr;                        // Stage B waits for this one (the timing pair is owned by the communicator)
        GKC_HIP(c, hipEventCreateWithFlags(&wait_ev, hipEventDisableTiming));
        GKC_HIP(c, hipEventRecord(wait_ev, m->xstream));
        c->pending_events.push_back(wait_ev);
        if (!m->rccl) (void)hipEventDestroy(done);
    }
    // 6) my own segments keep the partitions I own; what arrived joins the pass
    for (size_t i : mine) {
        Segment& sg = c->segments[i];
        const uint64_t a = sg.rec_off[lo], b = sg.rec_off[hi];
        for (uint32_t p = 0; p <= P; p++) sg.rec_off[p] = p < lo ? a : (p > hi ? b : sg.rec_off[p]);
        for (uint32_t p = 0; p < P; p++) if (p < lo || p >= hi) sg.nkmers[p] = 0;
    }
    for (Segment& sg : imported) c->segments.push_back(std::move(sg));
    c->n_exchanged_segments = c->segments.size();
    m->stats.n_exchanges++; m->stats.bytes_sent += sent_bytes; m->stats.bytes_received += recv_recs * rb;
    m->stats.ms_host += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_host0).count();
    return GKC_OK;
}

__global__ void k_loop_fill(uint64_t* __restrict__ a, uint64_t n, uint64_t seed)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) a[i] = (i + seed) * 0x9E3779B97F4A7C15ULL;
}
__global__ void k_loop_check(const uint64_t* __restrict__ a, uint64_t n, uint64_t seed, unsigned long long* __restrict__ bad)
{
    unsigned long long b = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) b += a[i] != (i + seed) * 0x9E3779B97F4A7C15ULL;
    if (b) atomicAdd(bad, b);
}
int gkc_comm_loopback(gkc_ctx* c, gkc_comm* m, uint64_t n_bytes, uint64_t* mismatches, double* ms)
{
    if (!c || !m || m->ctx != c || !mismatches) return GKC_ERR_ARG;
    GKC_HIP(c, hipSetDevice(c->device));
    const uint64_t n = (n_bytes + 7) / 8;
    DevBuf src, dst, bad;
    GKC_TRY(c->ensure(src, (size_t)n * 8 + 8));
    int rc = c->ensure(dst, (size_t)n * 8 + 8); if (rc == GKC_OK) rc = c->ensure(bad, 8);
    if (rc != GKC_OK) { src.release(); dst.release(); return rc; }
    (void)hipMemsetAsync(bad.p, 0, 8, m->xstream); (void)hipMemsetAsync(dst.p, 0, (size_t)n * 8, m->xstream);
    hipLaunchKernelGGL(k_loop_fill, dim3(1024), dim3(256), 0, m->xstream, (uint64_t*)src.p, n, 12345ull);
    (void)hipStreamSynchronize(m->xstream);
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<gkc_xfer> sends{ gkc_xfer{ m->rank, 0, src.p, n * 8 } }, recvs{ gkc_xfer{ m->rank, 0, dst.p, n * 8 } };
    rc = gkc_comm_sendrecv(m, sends, recvs, m->xstream);
    (void)hipStreamSynchronize(m->xstream);
    if (ms) *ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    unsigned long long h = 0;
    if (rc == GKC_OK) {
        hipLaunchKernelGGL(k_loop_check, dim3(1024), dim3(256), 0, m->xstream, (const uint64_t*)dst.p, n, 12345ull, (unsigned long long*)bad.p);
        if (hipMemcpyAsync(&h, bad.p, 8, hipMemcpyDeviceToHost, m->xstream) != hipSuccess || hipStreamSynchronize(m->xstream) != hipSuccess) { c->set_error(GKC_ERR_HIP, "loopback check failed"); rc = GKC_ERR_HIP; }
    }
    src.release(); dst.release(); bad.release();
    *mismatches = h;
    return rc;
}
// Start-up self-test over REAL peers (gkc_comm_loopback reaches only the rank itself): one grouped exchange in which every rank sends n_bytes of a pattern keyed by
// (source, destination) to every other rank and checks what arrives from each — the grouped ncclSend / ncclRecv path of gkc_exchange with its 256 MiB chunking between
// every pair of GPUs, before any real record travels. Collective. mismatches: 8-byte words that differ, summed over the peers; ms: wall time of the exchange.
int gkc_comm_selftest(gkc_ctx* c, gkc_comm* m, uint64_t n_bytes, uint64_t* mismatches, double* ms)
{
    if (!c || !m || m->ctx != c || !mismatches) return GKC_ERR_ARG;
    GKC_HIP(c, hipSetDevice(c->device));
    const int W = m->world, me = m->rank;
    *mismatches = 0; if (ms) *ms = 0;
    if (W == 1) return gkc_comm_loopback(c, m, n_bytes, mismatches, ms);
    const uint64_t n = (n_bytes + 7) / 8;
    DevBuf src, dst, bad;
    int rc = c->ensure(src, (size_t)n * 8 * (W - 1));
    if (rc == GKC_OK) rc = c->ensure(dst, (size_t)n * 8 * (W - 1));
    if (rc == GKC_OK) rc = c->ensure(bad, 8);
    GKC_TRY(gkc_comm_agree(m, rc, "gkc_comm_selftest"));
    (void)hipMemsetAsync(bad.p, 0, 8, m->xstream); (void)hipMemsetAsync(dst.p, 0, (size_t)n * 8 * (W - 1), m->xstream);
    std::vector<gkc_xfer> sends, recvs;
    int slot = 0;
    for (int r = 0; r < W; r++) {
        if (r == me) continue;
        uint64_t* sp = (uint64_t*)src.p + (size_t)slot * n; uint64_t* dp = (uint64_t*)dst.p + (size_t)slot * n;
        hipLaunchKernelGGL(k_loop_fill, dim3(1024), dim3(256), 0, m->xstream, sp, n, (uint64_t)(1000003ull * (uint64_t)me + (uint64_t)r));
        sends.push_back(gkc_xfer{ r, 0, sp, n * 8 }); recvs.push_back(gkc_xfer{ r, 0, dp, n * 8 });
        slot++;
    }
    (void)hipStreamSynchronize(m->xstream);
    const auto t0 = std::chrono::steady_clock::now();
    rc = gkc_comm_sendrecv(m, sends, recvs, m->xstream);
    (void)hipStreamSynchronize(m->xstream);
    if (ms) *ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    unsigned long long h = 0;
    if (rc == GKC_OK) {
        slot = 0;
        for (int r = 0; r < W; r++) {
            if (r == me) continue;
            hipLaunchKernelGGL(k_loop_check, dim3(1024), dim3(256), 0, m->xstream, (const uint64_t*)dst.p + (size_t)slot * n, n, (uint64_t)(1000003ull * (uint64_t)r + (uint64_t)me), (unsigned long long*)bad.p);
            slot++;
        }
        if (hipMemcpyAsync(&h, bad.p, 8, hipMemcpyDeviceToHost, m->xstream) != hipSuccess || hipStreamSynchronize(m->xstream) != hipSuccess) { c->set_error(GKC_ERR_HIP, "self-test check failed"); rc = GKC_ERR_HIP; }
    }
    src.release(); dst.release(); bad.release();
    *mismatches = h;
    return gkc_comm_agree(m, rc != GKC_OK ? rc : (h ? GKC_ERR_HIP : GKC_OK), "gkc_comm_selftest");      // one rank seeing garbage fails every rank
}
int gkc_comm_peer_bytes(gkc_comm* m, uint64_t* sent, uint64_t* received, double* init_ms)
{
    if (!m) return GKC_ERR_ARG;
    for (int r = 0; r < m->world; r++) {
        if (sent) sent[r] = (size_t)r < m->peer_sent.size() ? m->peer_sent[r] : 0;
        if (received) received[r] = (size_t)r < m->peer_recv.size() ? m->peer_recv[r] : 0;
    }
    if (init_ms) *init_ms = m->init_ms;
    return GKC_OK;
}
int gkc_gather_results(gkc_ctx* c, gkc_comm* m, int root)
{
    if (!c || !m || m->ctx != c) return GKC_ERR_ARG;
    const int W = m->world, me = m->rank;
    if (root < 0 || root >= W) GKC_FAIL(c, GKC_ERR_ARG, "bad root rank %d", root);
    if (c->stage_b_running || c->in_pass) GKC_FAIL(c, GKC_ERR_ARG, "gkc_gather_results before the pass has finished (gkc_finish_pass / gkc_finish_pass_wait first)");
    GKC_HIP(c, hipSetDevice(c->device));
    if (W == 1) return GKC_OK;
    if (m->first.size() != (size_t)W + 1) GKC_FAIL(c, GKC_ERR_ARG, "owner ranges unknown: no gkc_exchange happened in this pass");
    const uint32_t P = c->nb_partitions, pass = c->pass, rb = c->key_words == 1 ? 16 : 32;
    // a pass is gathered once: a second call would add the other ranks' histograms and counters to the root's again (every rank keeps the mark for itself,
    // so the ranks agree without talking; gkc_begin_pass of the pass clears it with the pass's statistics)
    if (c->pass_stats[pass].reserved[1] == GKC_PASS_GATHERED) return GKC_OK;
    Dataset* DS = c->datasets.data() + (size_t)pass * P;
    // 1) what every owner holds: per partition (solid, distinct, k-mers, starts a new contiguous array on its owner)
    std::vector<uint64_t> tab((size_t)P * 4, 0), all((size_t)W * P * 4);
    const uint32_t lo = m->first[me], hi = m->first[me + 1];
    {   int rc0 = GKC_OK;
        for (uint32_t p = lo; p < hi && rc0 == GKC_OK; p++) if (!DS[p].done) { c->set_error(GKC_ERR_ARG, "partition %u of pass %u has not been counted", p, pass); rc0 = GKC_ERR_ARG; }
        GKC_TRY(gkc_comm_agree(m, rc0, "gkc_gather_results"));
    }
    for (uint32_t p = lo; p < hi; p++) {
        const Dataset& D = DS[p];
        tab[(size_t)p * 4 + 0] = D.n_solid; tab[(size_t)p * 4 + 1] = D.n_distinct; tab[(size_t)p * 4 + 2] = D.n_kmers;
        const bool joins = p > lo && DS[p - 1].n_solid && D.n_solid && (const uint8_t*)D.d_counts == (const uint8_t*)DS[p - 1].d_counts + DS[p - 1].n_solid * rb;
        tab[(size_t)p * 4 + 3] = joins ? 0 : 1;
    }
    GKC_TRY(gkc_comm_allgather_host(m, tab.data(), tab.size() * 8, all.data()));
    // 2) the arrays travel, one message per contiguous array of the owner (a Stage-B batch), into one buffer per owner on the root
    std::vector<gkc_xfer> sends, recvs;
    int root_rc = GKC_OK;                                          // the root's buffers may not fit: agreed on before anybody sends
    if (me != root) {
        for (uint32_t p = lo; p < hi; ) {
            if (!DS[p].n_solid) { p++; continue; }
            uint32_t q = p + 1; uint64_t n = DS[p].n_solid;
            while (q < hi && (DS[q].n_solid == 0 || tab[(size_t)q * 4 + 3] == 0)) { n += DS[q].n_solid; q++; }
            sends.push_back(gkc_xfer{ root, 0, (void*)DS[p].d_counts, n * rb });
            p = q;
        }
    } else {
        for (int r = 0; r < W; r++) {
            if (r == root) continue;
            const uint64_t* T = all.data() + (size_t)r * P * 4;
            uint64_t total = 0;
            for (uint32_t p = m->first[r]; p < m->first[r + 1]; p++) total += T[(size_t)p * 4];
            uint8_t* buf = nullptr;
            if (total) {
                buf = (uint8_t*)c->dalloc((size_t)total * rb);
                if (!buf) { root_rc = GKC_ERR_NOMEM; break; }
                c->pass_outputs[pass].push_back(buf);
            }
            uint64_t off = 0;
            for (uint32_t p = m->first[r]; p < m->first[r + 1]; ) {
                if (!T[(size_t)p * 4]) { Dataset& D = DS[p]; D = Dataset(); D.n_distinct = T[(size_t)p * 4 + 1]; D.n_kmers = T[(size_t)p * 4 + 2]; D.done = true; p++; continue; }
                uint32_t q = p; uint64_t n = 0;
                do {
                    Dataset& D = DS[q]; D = Dataset();
                    D.n_solid = T[(size_t)q * 4]; D.n_distinct = T[(size_t)q * 4 + 1]; D.n_kmers = T[(size_t)q * 4 + 2]; D.done = true;
                    D.d_counts = D.n_solid ? buf + (off + n) * rb : nullptr;
                    n += D.n_solid; q++;
                } while (q < m->first[r + 1] && (T[(size_t)q * 4] == 0 || T[(size_t)q * 4 + 3] == 0));
                recvs.push_back(gkc_xfer{ r, 0, buf + off * rb, n * rb });
                off += n; p = q;
            }
        }
    }
    GKC_TRY(gkc_comm_agree(m, root_rc, "gkc_gather_results"));
    GKC_HIP(c, hipStreamSynchronize(c->stream));
    GKC_TRY(gkc_comm_sendrecv(m, sends, recvs, m->xstream));
    GKC_HIP(c, hipStreamSynchronize(m->xstream));
    // 3) the abundance histogram and the statistics of the pass: summed on the root
    const size_t hn = (size_t)c->histo_max + 1;
    std::vector<uint64_t> h(hn), hall(hn * W);
    GKC_HIP(c, hipMemcpy(h.data(), c->histo_of(pass), hn * 8, hipMemcpyDeviceToHost));
    GKC_TRY(gkc_comm_allgather_host(m, h.data(), hn * 8, hall.data()));
    std::vector<gkc_stats> sall(W);
    GKC_TRY(gkc_comm_allgather_host(m, &c->pass_stats[pass], sizeof(gkc_stats), sall.data()));
    if (me == root) {
        for (size_t i = 0; i < hn; i++) { uint64_t v = 0; for (int r = 0; r < W; r++) v += hall[(size_t)r * hn + i]; h[i] = v; }
        GKC_HIP(c, hipMemcpy(c->histo_of(pass), h.data(), hn * 8, hipMemcpyHostToDevice));
        gkc_stats& S = c->pass_stats[pass];
        for (int r = 0; r < W; r++) {
            if (r == root) continue;
            const gkc_stats& o = sall[r];
            S.kmers_nb_valid += o.kmers_nb_valid; S.kmers_nb_invalid += o.kmers_nb_invalid; S.kmers_nb_distinct += o.kmers_nb_distinct; S.kmers_nb_solid += o.kmers_nb_solid;
            S.nb_superkmers += o.nb_superkmers; S.nb_sequences += o.nb_sequences; S.nb_bases += o.nb_bases; S.superkmer_bytes += o.superkmer_bytes;
            S.oversize_buckets += o.oversize_buckets; S.dedupe_kmers_in += o.dedupe_kmers_in; S.dedupe_keys_out += o.dedupe_keys_out;
            // what BankStats::update keeps per sequence (BankKmers.hpp:176-186): shortest (0 = that rank saw no read) / longest read, sum of the squared lengths
            if (o.seq_len_min != 0 && (S.seq_len_min == 0 || o.seq_len_min < S.seq_len_min)) S.seq_len_min = o.seq_len_min;
            S.seq_len_max = std::max(S.seq_len_max, o.seq_len_max); S.seq_len_sq_sum += o.seq_len_sq_sum;
        }
    }
    c->pass_stats[pass].reserved[1] = GKC_PASS_GATHERED;
    return GKC_OK;
}

}  // extern "C"
