// gkc_common.hpp — shared host-side state and device helpers of libgkc_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdarg.h>
#include <string.h>
#include <string>
#include <vector>
#include <chrono>
#include <map>
#include <atomic>
#include <mutex>
#include <condition_variable>
#include <thread>
#include <algorithm>
#include "../../include/gkc.h"

// ------------------------------------------------------------------------------------------------ errors
struct gkc_error { int code; std::string msg; };

#define GKC_FAIL(ctx, code_, ...) do { (ctx)->set_error((code_), __VA_ARGS__); return (code_); } while (0)
#define GKC_HIP(ctx, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    (ctx)->set_error(e_ == hipErrorOutOfMemory ? GKC_ERR_NOMEM : GKC_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    return e_ == hipErrorOutOfMemory ? GKC_ERR_NOMEM : GKC_ERR_HIP; } } while (0)
#define GKC_TRY(expr) do { int rc_ = (expr); if (rc_ != GKC_OK) return rc_; } while (0)

// ------------------------------------------------------------------------------------------------ geometry
// Stage A tile: one 512-thread workgroup scans TILE k-mer start positions (16 per thread) plus a halo. 512 threads share one set of
// per-partition LDS counters (16 KB at 4096 partitions): 2 workgroups = 16 waves per CU fit in LDS, with 256 threads only 12 did
// (measured: scan 64 -> 54 ms).
constexpr uint64_t GKC_PASS_GATHERED = 0x6761746865726564ull;   // pass_stats[pass].reserved[1] once gkc_gather_results has run for the pass
constexpr int SCAN_THREADS = 512;
constexpr int SCAN_PER_THREAD = 16;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_PER_THREAD;      // 8192 positions
constexpr int DESC_START_BITS = 13, DESC_NBK_BITS = 6;         // record descriptor: [partition : 13][nbK-1 : 6][start in tile : 13]
static_assert(SCAN_TILE == (1 << DESC_START_BITS), "descriptor start field");
constexpr uint32_t DESC_PARTS_MAX = 1u << (32 - DESC_START_BITS - DESC_NBK_BITS);
// Stage A buckets at most this many ways in one pass (LDS counters / cursors per workgroup); more partitions are reached in two levels:
// scan into groups of 2^coarse_shift consecutive partitions, then split every group by recomputing each record's minimizer (k_refine_*)
constexpr uint32_t SCAN_COARSE_MAX = 4096;
constexpr int SCAN_HALO_WORDS = 4;                             // 64 bases of look-ahead (k<=63)
constexpr int SCAN_WORDS = SCAN_TILE / 16 + SCAN_HALO_WORDS;   // 16-base words per tile

// Stage B: a partition is split by key range into <= MAX_SUB sub-buckets, each sorted in registers by one wave (or workgroup).
constexpr int MAX_SUB_BITS = 13;
constexpr int MAX_SUB = 1 << MAX_SUB_BITS;                     // LDS histogram / cursors: 32 KB
constexpr int SUB_TARGET = 512;                                // mean keys per level-1 bucket: a wave sorts <= 1024 / 512 straight from HBM (<= 2048 / 1024 in
                                                               // the double-size tier), a workgroup <= 4096 / 2048, larger buckets are split in HBM

// ------------------------------------------------------------------------------------------------ developer / test switches
// Every environment variable the library looks at, in ONE place. None selects a CPU path (there is none): they pick between HIP code paths so that test-size inputs reach
// the tiers real inputs reach (split levels, slices, region builds, wire widths ...), or print diagnostics. Read into a process-wide snapshot by gkc_tun_refresh(): at
// gkc_create, at the start of a pass and at the creation / entry points of the Bloom, MPHF and communicator objects — never inside a kernel-launch loop.
struct GkcTun {
    // Stage A
    uint32_t scan_coarse_max; uint64_t push_split, push_chunk; bool scan_global_atomics, refine_recompute, scan_no_desc;
    // Stage B
    int weight_bits, dedupe, max_sub_bits, lanes; uint64_t slice_min, batch_keys; bool slices, batch_lpt, no_f64; uint32_t wg_max, scatter_wgs, deep_bits, sink_first_div;
    // result sink
    bool sink_packed, sink_packed2, sink_width6, sink_debug; int sink_adaptive; uint64_t sink_dense; int unpack_threads;
    // Bloom / MPHF
    bool bloom_atomic, bloom_gather, mphf_regions, mphf_ordered; uint64_t bloom_query_regions_min, mphf_regions_min;
    // allocator, communicators, diagnostics
    int vmm; uint32_t vmm_chunk_mb, vmm_min_mb; bool vmm_with_rccl, pool_debug, pool_trace, verbose; double filebox_timeout; char fault[32];       // (plain data: concurrent refreshes of equal values are harmless)
    static const char* raw(const char* name) { return getenv(name); }
    static long long num(const char* name, long long dflt) { const char* e = raw(name); return e && *e ? atoll(e) : dflt; }
    static bool on(const char* name) { return raw(name) != nullptr; }                       // set (to anything) = on
    static bool not0(const char* name) { const char* e = raw(name); return !(e && atoi(e) == 0); }      // on unless set to 0
    void read() {
        scan_coarse_max = (uint32_t)std::max<long long>(1, num("GKC_SCAN_COARSE_MAX", SCAN_COARSE_MAX));
        push_split = (uint64_t)std::max<long long>(1024, num("GKC_PUSH_SPLIT", 16000000000ll)); push_chunk = (uint64_t)std::max<long long>(64, num("GKC_PUSH_CHUNK", 1ll << 30));
        scan_global_atomics = on("GKC_SCAN_GLOBAL_ATOMICS"); refine_recompute = on("GKC_REFINE_RECOMPUTE"); scan_no_desc = on("GKC_SCAN_NO_DESC");
        weight_bits = (int)num("GKC_WEIGHT_BITS", 0); dedupe = (int)num("GKC_DEDUPE", -1); max_sub_bits = (int)num("GKC_MAX_SUB_BITS", -1); lanes = (int)num("GKC_STAGEB_LANES", 2);
        slice_min = (uint64_t)std::max<long long>(1, num("GKC_SLICE_MIN", 8000000)); batch_keys = (uint64_t)num("GKC_BATCH_KEYS", 0);
        slices = not0("GKC_SLICES"); batch_lpt = not0("GKC_BATCH_LPT"); no_f64 = on("GKC_NO_F64");
        wg_max = (uint32_t)num("GKC_WG_MAX", 0); scatter_wgs = (uint32_t)std::max<long long>(1, num("GKC_SCATTER_WGS", 176)); deep_bits = (uint32_t)std::max<long long>(1, num("GKC_DEEP_BITS", MAX_SUB_BITS));
        sink_first_div = (uint32_t)std::max<long long>(1, num("GKC_SINK_FIRST_DIV", 4));
        sink_packed = not0("GKC_SINK_PACKED"); sink_packed2 = not0("GKC_SINK_PACKED2"); sink_width6 = not0("GKC_SINK_WIDTH6"); sink_debug = on("GKC_SINK_DEBUG"); sink_adaptive = (int)num("GKC_SINK_ADAPTIVE", 1);      /* 0: never raw on its own, 1: when the host is behind, 2 (tests): every other batch */
        sink_dense = (uint64_t)num("GKC_SINK_DENSE", 0); unpack_threads = (int)num("GKC_UNPACK_THREADS", 0);
        bloom_atomic = on("GKC_BLOOM_ATOMIC"); bloom_gather = on("GKC_BLOOM_GATHER"); mphf_regions = not0("GKC_MPHF_REGIONS"); mphf_ordered = on("GKC_MPHF_ORDERED");
        bloom_query_regions_min = (uint64_t)num("GKC_BLOOM_QUERY_REGIONS_MIN", 2000000); mphf_regions_min = (uint64_t)num("GKC_MPHF_REGIONS_MIN", 1ll << 21);
        vmm = (int)num("GKC_VMM", 1); vmm_chunk_mb = (uint32_t)std::max<long long>(2, num("GKC_VMM_CHUNK_MB", 1024)); vmm_min_mb = (uint32_t)std::max<long long>(1, num("GKC_VMM_MIN_MB", 64));
        vmm_with_rccl = num("GKC_VMM_WITH_RCCL", 0) == 1; pool_debug = on("GKC_POOL_DEBUG"); pool_trace = on("GKC_POOL_TRACE"); verbose = on("GKC_VERBOSE");
        filebox_timeout = raw("GKC_FILEBOX_TIMEOUT") ? atof(raw("GKC_FILEBOX_TIMEOUT")) : 600.0; memset(fault, 0, sizeof fault); if (raw("GKC_FAULT")) strncpy(fault, raw("GKC_FAULT"), sizeof fault - 1);
    }
};
inline GkcTun& gkc_tun() { static GkcTun t = [] { GkcTun x; x.read(); return x; }(); return t; }
inline void gkc_tun_refresh() { static std::mutex mu; std::lock_guard<std::mutex> lk(mu); GkcTun x; x.read(); gkc_tun() = x; }

// Stage B runs batches on two host threads, each with its own stream: the thread-local override routes every launch / copy / timer of
// that thread to its lane's stream (cur_stream() below)
inline thread_local hipStream_t tl_stream_ = nullptr;

// ------------------------------------------------------------------------------------------------ device buffer
// Caching device allocator: hipMalloc/hipFree of multi-GB buffers cost tens of ms per GB on this platform (far more than
// the kernels that use them), so freed blocks are kept and reused by later passes of the same shape.
// Where the blocks come from (round 5): the first pass of a PROCESS used to spend 2-5 s in hipMalloc (10^8 reads: ~150 GB of buffers at ~22 ms per GB, erratic:
// tools/malloc_bench) — more than ten times the kernels, and a dbgh5 run is one process, one pass. The HIP virtual-memory API maps physical memory ~20x cheaper
// on these boxes (tools/vmm_probe: 512 MiB chunks created + mapped in 0.1-0.5 ms): a block of 64 MiB or more is an address range reserved with
// hipMemAddressReserve and backed by hipMemCreate chunks of <= 1 GiB mapped read-write for this device; kernels, hipMemcpy and hipMemset see an ordinary
// device pointer. GKC_VMM=0 keeps hipMalloc for everything; a context with a communicator (RCCL / IPC transports register their buffers) keeps hipMalloc as well
// (DevPool::vmm_ok cleared by gkc_comm_create*): whether RCCL accepts VMM ranges as send / receive buffers cannot be checked on a one-GPU box.
struct VmmBlock { size_t bytes = 0; std::vector<hipMemGenericAllocationHandle_t> handles; std::vector<size_t> sizes; };
struct DevPool {
    std::recursive_mutex mu;                  // Stage B drives two host threads (two streams) through one pool
    std::multimap<size_t, void*> cache;       // free blocks by size
    std::map<void*, size_t> live;             // blocks handed out
    std::map<void*, VmmBlock> vmm;            // blocks that are mapped ranges (handed out or parked)
    int device = 0; bool vmm_ok = true; size_t vmm_gran = 0; int vmm_state = 0;      // 0: not probed, 1: usable, -1: not
    // (mapped ranges from GkcTun::vmm_min_mb = 64 MiB up, chunks of vmm_chunk_mb = 1 GiB)
    hipError_t raw_alloc(void** out, size_t want) {
        if (vmm_state == 0) {
            vmm_state = -1;
            int sup = 0;
            if (gkc_tun().vmm != 0 && hipDeviceGetAttribute(&sup, hipDeviceAttributeVirtualMemoryManagementSupported, device) == hipSuccess && sup) {
                hipMemAllocationProp prop{}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = device;
                if (hipMemGetAllocationGranularity(&vmm_gran, &prop, hipMemAllocationGranularityRecommended) == hipSuccess && vmm_gran) vmm_state = 1;
            }
            (void)hipGetLastError();
        }
        if (vmm_state != 1 || !vmm_ok || want < ((size_t)gkc_tun().vmm_min_mb << 20)) return hipMalloc(out, want);
        const size_t total = (want + vmm_gran - 1) / vmm_gran * vmm_gran;
        void* base = nullptr;
        hipError_t e = hipMemAddressReserve(&base, total, 0, nullptr, 0);
        if (e != hipSuccess) { (void)hipGetLastError(); return hipMalloc(out, want); }
        hipMemAllocationProp prop{}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = device;
        VmmBlock B; B.bytes = total;
        size_t off = 0;
        const size_t chunk_bytes = (size_t)gkc_tun().vmm_chunk_mb << 20;
        const bool dbg = gkc_tun().pool_debug;
        double t_create = 0, t_map = 0, t_access = 0;
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        while (off < total && e == hipSuccess) {
            const size_t n = std::min(std::max(chunk_bytes / vmm_gran, (size_t)1) * vmm_gran, total - off);
            hipMemGenericAllocationHandle_t h;
            const auto t0 = now();
            e = hipMemCreate(&h, n, &prop, 0);
            const auto t1 = now();
            if (e == hipSuccess) { e = hipMemMap((char*)base + off, n, 0, h, 0); if (e != hipSuccess) (void)hipMemRelease(h); }
            t_create += ms(t0, t1); t_map += ms(t1, now());
            if (e == hipSuccess) { B.handles.push_back(h); B.sizes.push_back(n); off += n; }
        }
        if (e == hipSuccess) { const auto t0 = now(); hipMemAccessDesc acc{}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite; e = hipMemSetAccess(base, total, &acc, 1); t_access = ms(t0, now()); }
        if (dbg && t_create + t_map + t_access > 20.0) fprintf(stderr, "[gkc pool] %.2f GB in %zu chunks: hipMemCreate %.1f ms, hipMemMap %.1f ms, hipMemSetAccess %.1f ms\n", (double)total / 1e9, B.handles.size(), t_create, t_map, t_access);
        if (e != hipSuccess) {                                   // (out of memory, mostly: the caller trims the parked blocks and asks again)
            (void)hipGetLastError();
            size_t o = 0; for (size_t i = 0; i < B.handles.size(); i++) { (void)hipMemUnmap((char*)base + o, B.sizes[i]); (void)hipMemRelease(B.handles[i]); o += B.sizes[i]; }
            (void)hipMemAddressFree(base, total);
            return e == hipErrorOutOfMemory ? e : hipErrorOutOfMemory;
        }
        vmm[base] = std::move(B);
        *out = base;
        return hipSuccess;
    }
    void raw_free(void* p) {
        auto it = vmm.find(p);
        if (it == vmm.end()) { (void)hipFree(p); return; }
        (void)hipDeviceSynchronize();                            // (hipFree waits for the device as well: nothing may still be using the range)
        size_t o = 0; for (size_t i = 0; i < it->second.handles.size(); i++) { (void)hipMemUnmap((char*)p + o, it->second.sizes[i]); (void)hipMemRelease(it->second.handles[i]); o += it->second.sizes[i]; }
        (void)hipMemAddressFree(p, it->second.bytes);
        vmm.erase(it);
    }
    size_t cached_bytes = 0;
    uint64_t n_malloc = 0, n_fail = 0, n_trim = 0; double malloc_ms = 0;   // diagnostics (GKC_POOL_DEBUG)
    // size classes: 256 B granules below 2 MB, 2 MB granules up to 64 MB, then 16 classes per octave (<= 6.25 % slack) so that the
    // slightly different buffer sizes of successive Stage-B batches / passes land in the same class and are reused
    static size_t round(size_t b) {
        if (b < ((size_t)2 << 20)) return (b + 255) / 256 * 256 ? (b + 255) / 256 * 256 : 256;
        if (b < ((size_t)64 << 20)) { const size_t g = (size_t)2 << 20; return (b + g - 1) / g * g; }
        int lg = 63 - __builtin_clzll((unsigned long long)b);
        const size_t g = (size_t)1 << (lg - 4);
        return (b + g - 1) / g * g;
    }
    void* alloc(size_t bytes, hipError_t* err) {
        std::lock_guard<std::recursive_mutex> lk(mu);
        const size_t want = round(bytes ? bytes : 1);
        auto it = cache.lower_bound(want);
        if (it != cache.end() && it->first <= want + want / 4 + ((size_t)1 << 20)) {       // close enough: reuse
            void* p = it->second; live[p] = it->first; cached_bytes -= it->first; cache.erase(it); *err = hipSuccess; return p;
        }
        void* p = nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        hipError_t e = raw_alloc(&p, want);
        n_malloc++;
        if (e != hipSuccess) {
            // out of memory: give parked blocks back, largest first, until the request fits. (Handing out a parked block that is merely
            // large enough would keep exactly the memory that ran out: a 23 GB result block of the previous pass serving a 10 GB request.)
            (void)hipGetLastError();
            n_fail++; n_trim++;
            if (gkc_tun().pool_debug) fprintf(stderr, "[gkc pool] hipMalloc of %.2f GB failed; parked %.2f GB in %zu blocks\n", (double)want / 1e9, (double)cached_bytes / 1e9, cache.size());
            while (e != hipSuccess && !cache.empty()) {
                auto last = std::prev(cache.end());
                raw_free(last->second); cached_bytes -= last->first; cache.erase(last);
                e = raw_alloc(&p, want);
                if (e != hipSuccess) (void)hipGetLastError();
            }
        }
        const double ms_ = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        malloc_ms += ms_;
        if (ms_ > 20.0 && gkc_tun().pool_debug) fprintf(stderr, "[gkc pool] %.2f GB took %.1f ms (%s)\n", (double)want / 1e9, ms_, vmm.count(p) ? "mapped chunks" : "hipMalloc");
        if (gkc_tun().pool_trace && want >= ((size_t)256 << 20)) {
            size_t live_b = 0; for (auto& kv : live) live_b += kv.second;
            size_t fr = 0, tot = 0; (void)hipMemGetInfo(&fr, &tot);
            fprintf(stderr, "[gkc pool trace] +%.2f GB in %.1f ms; before it: live %.1f GB, parked %.1f GB, device free %.1f of %.1f GB\n", (double)want / 1e9, ms_, (double)live_b / 1e9, (double)cached_bytes / 1e9, (double)fr / 1e9, (double)tot / 1e9);
        }
        *err = e;
        if (e != hipSuccess) return nullptr;
        live[p] = want;
        return p;
    }
    void free(void* p) {
        if (!p) return;
        // a block freed by one lane may be handed to the other lane (another stream) at once: its last user must have finished
        if (tl_stream_) (void)hipStreamSynchronize(tl_stream_);
        std::lock_guard<std::recursive_mutex> lk(mu);
        auto it = live.find(p);
        if (it == live.end()) { raw_free(p); return; }
        // a context that got a communicator after its first allocations (vmm_ok cleared): its mapped ranges are not recycled — the next user could be an exchange buffer
        if (!vmm_ok && vmm.count(p)) { live.erase(it); raw_free(p); return; }
        cache.insert({it->second, p}); cached_bytes += it->second; live.erase(it);
    }
    bool is_mapped_inside(const void* q) {       // q points into a mapped range
        std::lock_guard<std::recursive_mutex> lk(mu);
        auto it = vmm.upper_bound((void*)q); if (it == vmm.begin()) return false; --it;
        return (const char*)q < (const char*)it->first + it->second.bytes;
    }
    bool is_mapped(void* p) { std::lock_guard<std::recursive_mutex> lk(mu); return vmm.count(p) != 0; }
    void trim() { std::lock_guard<std::recursive_mutex> lk(mu); for (auto& kv : cache) raw_free(kv.second); cache.clear(); cached_bytes = 0; }
    void destroy() { std::lock_guard<std::recursive_mutex> lk(mu); trim(); for (auto& kv : live) raw_free(kv.first); live.clear(); }
};

struct DevBuf {
    void* p = nullptr; size_t bytes = 0; DevPool* pool = nullptr;
    void release() { if (p) { if (pool) pool->free(p); else (void)hipFree(p); p = nullptr; bytes = 0; } }
};

// one bucketed batch of super-k-mer records (device analogue of SuperKmerBinFiles for one push)
struct Segment {
    const void* d_records = nullptr;     // arena, partition-major
    bool owned = false;
    bool foreign = false;                // arrived through gkc_segment_import / gkc_exchange: never forwarded by a later exchange
    std::vector<uint64_t> rec_off;       // [P+1] in records
    std::vector<uint64_t> nkmers;        // [P]
};

struct Dataset {                          // result of (pass, part)
    const void* d_counts = nullptr;       // Count records (points into a pass-level output buffer)
    uint64_t n_solid = 0, n_distinct = 0, n_kmers = 0;
    bool done = false;
    const void* h_counts = nullptr;       // the same records in the host sink (gkc_set_host_sink), valid once `landed` has completed
    hipEvent_t landed = nullptr;          // D2H copy of the Stage-B batch this dataset belongs to (owned by the context's landed_events list)
    const void* sink_batch = nullptr;     // ... or the packed batch it travelled in (gkc_sink.hip): in the sink once the host threads have expanded it
};

struct Timing { double ms = 0; uint64_t launches = 0; };

struct gkc_unpacker;                      // gkc_sink.hip: staging buffer + host threads that expand packed result batches into the sink
struct gkc_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    gkc_error err{0, ""};
    // model
    bool configured = false;
    uint32_t k = 0, m = 0, nb_partitions = 0, nb_passes = 1;
    int minimizer_type = 0;
    uint32_t maxs = 0;
    uint32_t key_words = 1, record_bytes = 16;
    size_t last_plan_budget = 0;                // Stage B batch budget of the last pass (see gkc_count_pass)
    bool dedupe_off = false; unsigned long long dedupe_in = 0, dedupe_out = 0;   // super-k-mer deduplication in Stage B: switched off for the rest of a run when it gives < 1.18x fewer keys
    uint64_t model_hash = 0;               // of (k, m, partitions, passes, minimizer type, repartition table, frequency order): every rank of a communicator must hold the same
    int32_t amin = 1, amax = 2147483647; uint32_t histo_max = 10000;
    // device tables
    DevBuf d_mkey_lut;      // u32[4^m] : m-mer (forward strand) -> order key (freq mode only)
    DevBuf d_key2val;       // u32[4^m] : order key -> minimizer value (freq mode only)
    DevBuf d_repart;        // u16[4^m] : minimizer value -> partition
    DevBuf d_repart_coarse; // u16[4^m] : minimizer value -> partition >> coarse_shift (only when nb_partitions > SCAN_COARSE_MAX)
    uint32_t coarse_shift = 0;
    uint32_t default_key = 0;   // order key of the default minimizer 4^m-1
    // pass state
    bool in_pass = false; uint32_t pass = 0;
    std::vector<Segment> segments;
    std::vector<void*> owned_arenas;
    size_t n_exchanged_segments = 0;       // segments [0, n) went through gkc_exchange already (multi-GPU)
    std::vector<hipEvent_t> pending_events; // transfers of gkc_exchange still in flight on a communicator's stream: Stage B (and anything that frees arenas) waits for them
    void drain_pending() { for (hipEvent_t e : pending_events) { (void)hipEventSynchronize(e); (void)hipEventDestroy(e); } pending_events.clear(); }
    // results
    std::vector<Dataset> datasets;                   // nb_passes * nb_partitions
    std::map<uint32_t, std::vector<void*>> pass_outputs;   // pass -> output buffers
    DevBuf d_histo;                                  // u64[nb_passes][histo_max+1]: one histogram per pass (a pass that is run again starts from zero), summed by gkc_histogram
    unsigned long long* histo_of(uint32_t pass_) { return (unsigned long long*)d_histo.p + (size_t)pass_ * ((size_t)histo_max + 1); }
    // objects built from this context (gkc_bloom, gkc_mphf) hold device memory of its allocator: gkc_destroy with children alive only
    // marks the context closed; the last child to be destroyed frees it (gkc_ctx_child_release)
    int children = 0; bool closed = false;
    int comm_world = 1;                    // world size of the communicator built on this context (multi-GPU): Stage A then leaves a few CUs to the exchange kernels
    std::vector<gkc_stats> pass_stats;               // one per pass; gkc_get_stats sums them (reserved[1] == GKC_PASS_GATHERED: gkc_gather_results has run for the pass)
    gkc_stats& stats_now() { return pass_stats[pass]; }
    // streamed results (gkc_set_host_sink): every Stage-B batch is copied to page-locked host memory on a copy stream as soon as it is compacted
    void* sink = nullptr; uint64_t sink_cap = 0, sink_used = 0; bool sink_overflow = false; bool sink_raw = false;      // sink_raw: gkc_set_sink_mode(GKC_SINK_RAW)
    std::chrono::steady_clock::time_point t_stage_b0;   // when the current gkc_count_pass began (GKC_SINK_DEBUG: how long the link waited for the first batch)
    uint64_t sink_wire_bytes = 0;         // bytes the packed batches of the pass took on the link (gkc_stats.reserved[1])
    std::atomic<bool> sink_no6{false};                // the packed transfer found too few abundances of 1 for its 6-byte entries to pay (gkc_sink.hip)
    hipStream_t copy_stream = nullptr;
    gkc_unpacker* unpacker = nullptr;
    hipStream_t fetch_stream = nullptr;    // gkc_partition_counts: D2H of finished datasets, beside (not inside) the Stage-B lanes
    std::vector<hipEvent_t> landed_events;
    std::condition_variable cv_done;       // a dataset finished / the pass ended (gkc_wait_partition)
    std::thread stage_b_thread; bool stage_b_running = false; int stage_b_rc = GKC_OK;
    // gkc_finish_pass_async DETACHES the pass: its number, segment list, arenas and pending exchange events move here and Stage B works on them on a stream of its
    // own, so that gkc_begin_pass / gkc_push_reads* of the NEXT pass may run meanwhile (Stage A is issue-bound, Stage B memory-bound: they share the chip well).
    // gkc_finish_pass_wait gives the lists back when no other pass has been begun in between (gkc_partition_superkmers / gkc_segment_export keep working).
    uint32_t b_pass = 0; std::vector<Segment> b_segments; std::vector<void*> b_arenas; std::vector<hipEvent_t> b_pending; bool b_detached = false; bool b_moved_on = false;
    hipStream_t bg_stream = nullptr;
    // double-buffered host -> device staging of gkc_push_reads (H2D of chunk j+1 overlaps the scan of chunk j)
    DevBuf h2d_bases[2], h2d_offs[2]; hipEvent_t h2d_copied[2] = {nullptr, nullptr}, h2d_scanned[2] = {nullptr, nullptr};
    std::mutex mu;                         // shared bookkeeping (timing, stats, outputs, error text) when Stage B runs two lanes
    hipStream_t lane_streams[3] = {nullptr, nullptr, nullptr};   // extra Stage-B lanes (created on first use)
    std::vector<uint8_t> pass_released;       // gkc_release_pass was called for the pass: whole-context consumers of the results refuse to run
    uint64_t slots_hint = 0;                  // key slots the big working buffers of a Stage-B batch are sized for (the pass's batch budget)
    double d_hint = 0;                        // solid records per key of the last Stage-B pass (0 = none yet): sizes the next pass's batches
    std::map<std::string, Timing> timing;
    // scratch reused across calls
    DevBuf d_scan_counters;    // u64[2P + 8]
    DevBuf d_rsbits;           // read-start bitmask
    DevBuf d_scan_matrix;      // [grid][P] per-workgroup partition bases (u64) + counts (u32) (LDS-counter scan)
    DevBuf d_desc, d_desc_tile; // record descriptors of the count pass, per-tile (offset,count)
    size_t key_budget = 0;     // max keys per Stage-B batch (0 = auto)
    size_t batch_cap = 0;      // gkc_set_batch_keys: upper bound of the planned batch size (0 = the library's own: few, large batches)

    void set_error(int code, const char* fmt, ...) {
        char buf[1024]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
        std::lock_guard<std::mutex> lk(err_mu); err.code = code; err.msg = buf;
    }
    std::mutex err_mu;
    DevPool pool;
    int ensure(DevBuf& b, size_t bytes) {
        if (b.bytes >= bytes && b.p && (pool.vmm_ok || !pool.is_mapped(b.p))) return GKC_OK;      // (a mapped range kept from before the communicator: replaced by a hipMalloc block)
        b.release();
        hipError_t e;
        b.p = pool.alloc(bytes ? bytes : 16, &e); b.pool = &pool;
        if (!b.p) { set_error(GKC_ERR_NOMEM, "device allocation of %zu bytes failed: %s", bytes, hipGetErrorString(e)); return GKC_ERR_NOMEM; }
        b.bytes = bytes;
        return GKC_OK;
    }
    // data-sized buffers (bucket arenas, result arrays). In a multi-pass run the passes ask for slightly different sizes: one size class (1/16) of
    // slack lets the block a released pass parked serve the next pass's request (otherwise: hipMalloc, out of memory, parked blocks given back, seconds)
    void* dalloc(size_t bytes) { if (nb_passes > 1 && bytes > ((size_t)64 << 20)) bytes += bytes / 16; hipError_t e; void* p = pool.alloc(bytes, &e); if (!p) set_error(GKC_ERR_NOMEM, "device allocation of %zu bytes failed: %s", bytes, hipGetErrorString(e)); return p; }
    void dfree(void* p) { pool.free(p); }
};

inline hipStream_t cur_stream(gkc_ctx* c) { return tl_stream_ ? tl_stream_ : c->stream; }

// RAII event timer accumulating into ctx->timing[name]
struct ScopedTimer {
    gkc_ctx* c; const char* name; hipEvent_t a, b; bool on; uint64_t counts;
    ScopedTimer(gkc_ctx* c_, const char* n, uint64_t counts_ = 1 /* launches this interval stands for (0: a further piece of an interval already counted) */) : c(c_), name(n), on(true), counts(counts_) {
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { on = false; return; }
        (void)hipEventRecord(a, cur_stream(c));
    }
    ~ScopedTimer() {
        if (!on) return;
        (void)hipEventRecord(b, cur_stream(c)); (void)hipEventSynchronize(b);
        float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
        { std::lock_guard<std::mutex> lk(c->mu); Timing& t = c->timing[name]; t.ms += ms; t.launches += counts; }
        (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    }
};

// ------------------------------------------------------------------------------------------------ launchers (defined in the .hip files)
int gkc_scan_push(gkc_ctx* c, const char* d_bases, const uint64_t* d_offsets, uint64_t n_reads, uint64_t n_bases);
int gkc_count_pass(gkc_ctx* c, uint32_t pass, const std::vector<Segment>& segments, hipStream_t lane0, double reserve_bytes);
int gkc_scan_sample(gkc_ctx* c, const char* d_bases, const uint64_t* d_offsets, uint64_t n_reads, uint64_t n_bases, uint64_t* h_superkmers, uint64_t* h_kmers);
int gkc_scan_sample_exact(gkc_ctx* c, const char* bases, const uint64_t* offsets, uint64_t n_reads, uint64_t max_superkmers, uint64_t* h_nsk, uint64_t* h_nk, uint64_t* h_nkx, uint64_t* reads_used);
int gkc_scan_count_mmers(gkc_ctx* c, uint32_t m, const char* d_bases, const uint64_t* d_offsets, uint64_t n_reads, uint32_t* h_counts);
int gkc_export_superkmers(gkc_ctx* c, uint32_t part, uint8_t* out, uint64_t cap, uint64_t* nb, uint64_t* nsk, uint64_t* nk);

int gkc_require_resident(gkc_ctx* c, const char* who);
// packed result batches (gkc_sink.hip)
bool  gkc_sink_packed(gkc_ctx* c);
bool  gkc_sink_host_behind(gkc_ctx* c, uint64_t n_records);   // several ranks on one host: landed, unexpanded records beyond 1.5 batches -> this batch travels raw
int   gkc_sink_prepare(gkc_ctx* c);
void  gkc_sink_reset(gkc_ctx* c);
void  gkc_sink_drain(gkc_ctx* c);
void  gkc_sink_shutdown(gkc_ctx* c);
void  gkc_sink_wait_batch(gkc_ctx* c, const void* batch);
void* gkc_sink_send_packed(gkc_ctx* c, const void* d_out, const uint64_t* d_ptot, const std::vector<uint64_t>& solid_prefix, uint8_t* h_dest);
const char* gkc_sink_last_refusal();                          // why this thread's last gkc_sink_send_packed returned nullptr
struct gkc_comm;                              // gkc_dist.hip
int gkc_comm_world(gkc_comm* m);
int gkc_comm_rank(gkc_comm* m);
int gkc_comm_allgather_host(gkc_comm* m, const void* mine, size_t n, void* all);
int gkc_comm_agree(gkc_comm* m, int local_rc, const char* where);          // all ranks return an error if one of them has one (gkc_dist.hip)
int gkc_comm_sendrecv(gkc_comm* m, const std::vector<gkc_xfer>& sends, const std::vector<gkc_xfer>& recvs, hipStream_t st);
int gkc_comm_allreduce_or_words(gkc_comm* m, uint64_t* d_words, uint64_t n_words, hipStream_t st);
int gkc_comm_combine_seen_coll(gkc_comm* m, uint64_t* d_seen, uint64_t* d_coll, uint64_t n_words, hipStream_t st);
void gkc_ctx_child_add(gkc_ctx* c);          // gkc_api.hip
void gkc_ctx_child_release(gkc_ctx* c);      // destroys a closed context when its last child goes
int gkc_alloc_histo(gkc_ctx* c);      // gkc_api.hip: fails when a pass of the context was released
