// gkc_api.hip — the C-ABI of libgkc_hip.so (include/gkc.h): context, configuration, pass control, results,
// plus the synthetic-read generator and the independent k-mer checksum kernel used for full-size parity properties.
#include "gkc_common.hpp"
#include "gkc_device.hpp"
#include <algorithm>
#include <numeric>

int gkc_result_checksum_impl(gkc_ctx* c, uint64_t* checksum, uint64_t* sum_abundance);

static thread_local std::string g_create_error;

// ------------------------------------------------------------------------------------------------ synthetic reads
// rnd(seed, stream, idx) — counter-based; twin: gatb-core_amd/gkc.py:synth_rnd / tests/util.py
__device__ __host__ __forceinline__ uint64_t synth_rnd(uint64_t seed, uint64_t stream, uint64_t idx)
{
    return mix64(mix64(seed ^ (stream << 56)) + idx);
}
// base at a position of the synthetic genome. profile 0: uniform. profile 1 (GKC_SYNTH_SKEWED): repeat families — the genome is cut into slots of 8192 bases, a quarter of
// the slots start with a copy of one of 50 family sequences of 1000..5000 bases (about 300 copies per family at 5e8 bases, ~9 % of the genome), every copy diverged
// from its family by 0.5 % substitutions; the rest is uniform
__device__ __host__ __forceinline__ uint32_t synth_genome_code(uint64_t seed, uint64_t pos, uint32_t profile)
{
    if (profile == 1) {
        const uint64_t slot = pos >> 13, in = pos & 8191;
        const uint64_t h = synth_rnd(seed, 4, slot);
        if ((h & 3) == 0) {
            const uint64_t f = (h >> 8) % 50, len = 1000 + synth_rnd(seed, 5, f) % 4001;
            if (in < len) {
                uint32_t code = (uint32_t)(synth_rnd(seed, 6, f * 8192 + in) & 3);
                const uint64_t d = synth_rnd(seed, 7, pos);
                if (d % 1000 < 5) code = (code + 1 + (uint32_t)((d >> 32) % 3)) & 3;
                return code;
            }
        }
    }
    return (uint32_t)(synth_rnd(seed, 1, pos) & 3);
}
__global__ void k_synth_reads(uint64_t seed, uint64_t first_read, uint64_t n_reads, uint32_t read_len, uint64_t genome_len, uint32_t sub_ppm, uint32_t profile,
                              uint8_t* __restrict__ bases, uint64_t* __restrict__ offsets)
{
    const uint64_t total = n_reads * read_len;
    const uint64_t span = genome_len - read_len + 1;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i = g / read_len; const uint32_t j = (uint32_t)(g - i * read_len);
        const uint64_t u = synth_rnd(seed, 2, first_read + i);
        const uint64_t start = (u >> 1) % span;
        uint32_t code;
        if (u & 1) code = synth_genome_code(seed, start + read_len - 1 - j, profile) ^ 2u;   // reverse complement
        else       code = synth_genome_code(seed, start + j, profile);
        if (profile == 1) {                                         // 1 % of the reads are low complexity: a unit of 1..3 bases repeated (poly-A, (AC)n, (ACG)n ...)
            const uint64_t hl = synth_rnd(seed, 8, first_read + i);
            if (hl % 100 == 0) { const uint32_t ul = 1 + (uint32_t)((hl >> 8) % 3); code = (uint32_t)(hl >> (16 + 2 * (j % ul))) & 3; }
        }
        const uint64_t v = synth_rnd(seed, 3, first_read * read_len + g);
        if ((v % 1000000ULL) < sub_ppm) code = (code + 1 + (uint32_t)((v >> 32) % 3)) & 3;
        bases[g] = (uint8_t)("ACTG"[code]);
    }
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= n_reads; i += (uint64_t)gridDim.x * blockDim.x)
        offsets[i] = i * read_len;
}

// ------------------------------------------------------------------------------------------------ independent checksum
// one thread per read, plain rolling canonical k-mer (no minimizers, no buckets, no sort)
template <int KW>
__global__ void k_kmer_checksum(const uint8_t* __restrict__ bases, const uint64_t* __restrict__ offsets, uint64_t n_reads, uint32_t k,
                                unsigned long long* __restrict__ out)
{
    typedef typename KeyT<KW>::type key_t;
    uint64_t cs = 0, nv = 0;
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_reads; r += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t b = offsets[r], e = offsets[r + 1];
        const key_t mask = KeyT<KW>::mask(k);
        key_t fw = 0, rv = 0; uint32_t good = 0;
        for (uint64_t g = b; g < e; g++) {
            const uint32_t c = bases[g];
            if (nt_valid(c)) {
                const uint32_t code = nt_code(c);
                fw = ((fw << 2) | (key_t)code) & mask;
                rv = (rv >> 2) | ((key_t)(code ^ 2u) << (2 * (k - 1)));
                good++;
            } else { good = 0; fw = 0; rv = 0; }
            if (good >= k) { cs += KeyT<KW>::mixv(fw < rv ? fw : rv); nv++; }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { cs += __shfl_down(cs, d, 64); nv += __shfl_down(nv, d, 64); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&out[0], (unsigned long long)cs); atomicAdd(&out[1], (unsigned long long)nv); }
}

// offsets of a slice of the caller's CSR table -> offsets relative to the slice (gkc_push_reads sends the reads in chunks)
__global__ void k_rebase_offsets(uint64_t* __restrict__ off, uint64_t n, uint64_t base)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) off[i] -= base;
}

void gkc_ctx_child_add(gkc_ctx* c) { std::lock_guard<std::mutex> lk(c->mu); c->children++; }
int gkc_alloc_histo(gkc_ctx* c)
{
    const size_t bytes = (size_t)std::max<uint32_t>(c->nb_passes, 1) * ((size_t)c->histo_max + 1) * 8;
    c->d_histo.release();
    GKC_TRY(c->ensure(c->d_histo, bytes));
    GKC_HIP(c, hipMemset(c->d_histo.p, 0, bytes));
    return GKC_OK;
}
int gkc_require_resident(gkc_ctx* c, const char* who)
{
    for (size_t p = 0; p < c->pass_released.size(); p++)
        if (c->pass_released[p]) GKC_FAIL(c, GKC_ERR_ARG, "%s needs the results of every pass on the device, pass %zu was released (gkc_release_pass)", who, p);
    return GKC_OK;
}
static void free_pass_outputs(gkc_ctx* c, uint32_t pass)
{
    auto it = c->pass_outputs.find(pass);
    if (it == c->pass_outputs.end()) return;
    for (void* p : it->second) c->dfree(p);
    c->pass_outputs.erase(it);
}
static void clear_segments(gkc_ctx* c)
{
    c->drain_pending();                       // an exchange may still be reading / filling arenas on a communicator's stream
    for (void* p : c->owned_arenas) c->dfree(p);
    c->owned_arenas.clear(); c->segments.clear(); c->n_exchanged_segments = 0;
}

// Stage B of a detached pass (gkc_finish_pass_async) is in flight or not yet joined
static bool bg_active(const gkc_ctx* c) { return c->stage_b_thread.joinable(); }
// ... and the caller may go on with the NEXT pass meanwhile: one GPU (the exchange works on the context's own segment list), no host sink (it holds one pass)
static bool bg_overlap_ok(const gkc_ctx* c) { return c->b_detached && c->comm_world == 1 && c->sink == nullptr; }
// the detached pass is through: its records are no longer needed (another pass has been begun), or the lists go back to the context
static int bg_join(gkc_ctx* c)
{
    if (!c->stage_b_thread.joinable()) return GKC_OK;
    c->stage_b_thread.join();
    if (c->b_detached) {
        if (c->b_moved_on || !c->segments.empty() || !c->owned_arenas.empty() || c->in_pass) {      // the context went on to another pass: the detached records go
            for (hipEvent_t e : c->b_pending) { (void)hipEventSynchronize(e); (void)hipEventDestroy(e); }
            for (void* p : c->b_arenas) c->dfree(p);
        } else {                                                                                   // ... or come back (gkc_segment_export, gkc_partition_superkmers after the pass)
            c->segments = std::move(c->b_segments); c->owned_arenas = std::move(c->b_arenas);
            for (hipEvent_t e : c->b_pending) c->pending_events.push_back(e);
            // Stage B failed (GKC_ERR_NOMEM ...) and the context has not moved on: the pass is in progress again, exactly as after a failed gkc_finish_pass —
            // gkc_finish_pass / gkc_finish_pass_async may be called again (gkc_count_pass: "a pass counted again")
            if (c->stage_b_rc != GKC_OK) { c->pass = c->b_pass; c->in_pass = true; }
        }
        c->b_segments.clear(); c->b_arenas.clear(); c->b_pending.clear(); c->b_detached = false; c->b_moved_on = false;
    }
    return c->stage_b_rc;
}

static void ctx_destroy_now(gkc_ctx* c);
void gkc_ctx_child_release(gkc_ctx* c)
{
    bool last;
    { std::lock_guard<std::mutex> lk(c->mu); c->children--; last = c->closed && c->children == 0; }
    if (last) ctx_destroy_now(c);
}

extern "C" {

const char* gkc_version(void) { return "gkc-hip 0.1 (gfx950)"; }

int gkc_create(int device, gkc_ctx** out)
{
    gkc_tun_refresh();
    if (!out) return GKC_ERR_ARG;
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) { g_create_error = std::string("no HIP device: ") + hipGetErrorString(e); return GKC_ERR_NODEVICE; }
    if (device < 0 || device >= n) { g_create_error = "device index out of range"; return GKC_ERR_ARG; }
    if ((e = hipSetDevice(device)) != hipSuccess) { g_create_error = std::string("hipSetDevice: ") + hipGetErrorString(e); return GKC_ERR_HIP; }
    gkc_ctx* c = new gkc_ctx();
    c->device = device; c->pool.device = device;
    if ((e = hipStreamCreate(&c->stream)) != hipSuccess) { g_create_error = std::string("hipStreamCreate: ") + hipGetErrorString(e); delete c; return GKC_ERR_HIP; }
    *out = c;
    return GKC_OK;
}

static void ctx_destroy_now(gkc_ctx* c);
void gkc_destroy(gkc_ctx* c)
{
    if (!c) return;
    {   std::lock_guard<std::mutex> lk(c->mu);
        if (c->closed) return;                  // destroyed twice while children keep it alive
        c->closed = true;
        if (c->children > 0) return;            // a gkc_bloom / gkc_mphf still holds memory of this context: the last one frees it
    }
    ctx_destroy_now(c);
}
static void ctx_destroy_now(gkc_ctx* c)
{
    (void)hipSetDevice(c->device);
    (void)bg_join(c);
    (void)hipStreamSynchronize(c->stream);
    if (c->bg_stream) { (void)hipStreamSynchronize(c->bg_stream); (void)hipStreamDestroy(c->bg_stream); c->bg_stream = nullptr; }
    c->drain_pending();
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    gkc_sink_shutdown(c);                                                // the unpack threads and their page-locked staging buffer
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->fetch_stream) { (void)hipStreamSynchronize(c->fetch_stream); (void)hipStreamDestroy(c->fetch_stream); }
    for (hipEvent_t e : c->landed_events) (void)hipEventDestroy(e);
    for (int i = 0; i < 2; i++) { c->h2d_bases[i].release(); c->h2d_offs[i].release(); if (c->h2d_copied[i]) (void)hipEventDestroy(c->h2d_copied[i]); if (c->h2d_scanned[i]) (void)hipEventDestroy(c->h2d_scanned[i]); }
    clear_segments(c);
    std::vector<uint32_t> passes; for (auto& kv : c->pass_outputs) passes.push_back(kv.first);
    for (uint32_t p : passes) free_pass_outputs(c, p);
    c->d_mkey_lut.release(); c->d_key2val.release(); c->d_repart.release(); c->d_repart_coarse.release(); c->d_histo.release();
    c->d_scan_counters.release(); c->d_rsbits.release(); c->d_scan_matrix.release(); c->d_desc.release(); c->d_desc_tile.release();
    c->pool.destroy();
    (void)hipStreamDestroy(c->stream);
    for (hipStream_t st : c->lane_streams) if (st) (void)hipStreamDestroy(st);
    delete c;
}

const char* gkc_last_error(const gkc_ctx* c) { return c ? c->err.msg.c_str() : g_create_error.c_str(); }

int gkc_configure(gkc_ctx* c, uint32_t k, uint32_t m, uint32_t nb_partitions, uint32_t nb_passes,
                  int minimizer_type, const uint16_t* repart, const uint32_t* freq_order)
{
    if (!c) return GKC_ERR_ARG;
    if (c->stage_b_running || c->stage_b_thread.joinable()) GKC_FAIL(c, GKC_ERR_ARG, "%s while gkc_finish_pass_async is in flight (gkc_finish_pass_wait first)", __func__);
    GKC_HIP(c, hipSetDevice(c->device));
    if (k <= 2) GKC_FAIL(c, GKC_ERR_ARG, "kmer size %u too small (SortingCountAlgorithm.cpp:662-666 refuses k<=2)", k);
    if (k > 63) GKC_FAIL(c, GKC_ERR_ARG, "kmer size %u not supported (this build covers spans 32 and 64: k<=63)", k);
    if (m < 2 || m >= k || m > 14) GKC_FAIL(c, GKC_ERR_ARG, "minimizer size %u invalid for k=%u (need 2 <= m <= min(k-1,14))", m, k);
    if (nb_partitions < 1 || nb_partitions > 65535) GKC_FAIL(c, GKC_ERR_ARG, "nb_partitions must be in [1,65535] (Repartitor::Value is u16, PartiInfo.hpp:297)");
    if (nb_passes < 1) GKC_FAIL(c, GKC_ERR_ARG, "nb_passes must be >= 1");
    if (!repart) GKC_FAIL(c, GKC_ERR_ARG, "repart table is required");
    if (minimizer_type != GKC_MINIMIZER_LEXI && minimizer_type != GKC_MINIMIZER_FREQ) GKC_FAIL(c, GKC_ERR_ARG, "bad minimizer_type");
    if (minimizer_type == GKC_MINIMIZER_FREQ && !freq_order) GKC_FAIL(c, GKC_ERR_ARG, "frequency order requires freq_order[4^m]");
    const uint64_t nm = 1ULL << (2 * m);
    for (uint64_t i = 0; i < nm; i++) if (repart[i] >= nb_partitions) GKC_FAIL(c, GKC_ERR_ARG, "repart[%llu]=%u >= nb_partitions", (unsigned long long)i, repart[i]);

    (void)hipStreamSynchronize(c->stream);
    clear_segments(c);
    { std::vector<uint32_t> passes; for (auto& kv : c->pass_outputs) passes.push_back(kv.first); for (uint32_t p : passes) free_pass_outputs(c, p); }
    c->d_hint = 0; c->dedupe_off = false; c->dedupe_in = 0; c->dedupe_out = 0;
    c->k = k; c->m = m; c->nb_partitions = nb_partitions; c->nb_passes = nb_passes; c->minimizer_type = minimizer_type;
    c->key_words = k <= 31 ? 1 : 2; c->record_bytes = k <= 31 ? 16 : 32;
    const uint32_t def = k <= 31 ? 28 : 60;                      // min((8*sizeof(Type)-8)/2, 255), Sequence2SuperKmer.hpp:147
    if (c->maxs == 0 || c->maxs > def) c->maxs = def;
    c->datasets.assign((size_t)nb_partitions * nb_passes, Dataset());
    c->pass_stats.assign(nb_passes, gkc_stats{}); c->pass_released.assign(nb_passes, 0); c->pass = 0; c->timing.clear(); c->in_pass = false;

    GKC_TRY(c->ensure(c->d_repart, nm * 2));
    GKC_HIP(c, hipMemcpy(c->d_repart.p, repart, nm * 2, hipMemcpyHostToDevice));
    {   // two-level Stage A above SCAN_COARSE_MAX partitions: groups of 2^coarse_shift consecutive partitions
        const uint32_t cmax = gkc_tun().scan_coarse_max;
        c->coarse_shift = 0;
        while (((nb_partitions - 1) >> c->coarse_shift) + 1 > cmax) c->coarse_shift++;
        if (c->coarse_shift) {
            std::vector<uint16_t> coarse(nm);
            for (uint64_t i = 0; i < nm; i++) coarse[i] = (uint16_t)(repart[i] >> c->coarse_shift);
            GKC_TRY(c->ensure(c->d_repart_coarse, nm * 2));
            GKC_HIP(c, hipMemcpy(c->d_repart_coarse.p, coarse.data(), nm * 2, hipMemcpyHostToDevice));
        }
    }
    if (minimizer_type == GKC_MINIMIZER_FREQ) {
        // order keys: dense rank of canonical m-mers (and of the default 4^m-1) under (freq_order[c], c)   (Model.hpp:957-973)
        auto revm = [&](uint32_t x) { uint32_t r = 0; for (uint32_t i = 0; i < m; i++) { r = (r << 2) | ((x & 3) ^ 2); x >>= 2; } return r; };
        std::vector<uint32_t> canon(nm), cand;
        cand.reserve(nm / 2 + 2);
        for (uint64_t x = 0; x < nm; x++) { uint32_t r = revm((uint32_t)x); canon[x] = r < x ? r : (uint32_t)x; if (canon[x] == x) cand.push_back((uint32_t)x); }
        const uint32_t defv = (uint32_t)(nm - 1);
        if (canon[defv] != defv) cand.push_back(defv);
        std::sort(cand.begin(), cand.end(), [&](uint32_t a, uint32_t b) { return freq_order[a] != freq_order[b] ? freq_order[a] < freq_order[b] : a < b; });
        std::vector<uint32_t> rank(nm, 0), key2val(nm, defv), lut(nm);
        for (uint32_t i = 0; i < cand.size(); i++) { rank[cand[i]] = i; key2val[i] = cand[i]; }
        for (uint64_t x = 0; x < nm; x++) lut[x] = rank[canon[x]];
        c->default_key = rank[defv];
        GKC_TRY(c->ensure(c->d_mkey_lut, nm * 4)); GKC_TRY(c->ensure(c->d_key2val, nm * 4));
        GKC_HIP(c, hipMemcpy(c->d_mkey_lut.p, lut.data(), nm * 4, hipMemcpyHostToDevice));
        GKC_HIP(c, hipMemcpy(c->d_key2val.p, key2val.data(), nm * 4, hipMemcpyHostToDevice));
    } else {
        c->default_key = (uint32_t)(nm - 1);
    }
    {   // fingerprint of the model: gkc_exchange compares it across the ranks (different tables would silently split one k-mer over two owners)
        uint64_t h = 0x9E3779B97F4A7C15ULL;
        auto mixin = [&](uint64_t v) { h ^= v + 0x9E3779B97F4A7C15ULL + (h << 6) + (h >> 2); h *= 0xBF58476D1CE4E5B9ULL; h ^= h >> 31; };
        mixin(k); mixin(m); mixin(nb_partitions); mixin(nb_passes); mixin((uint64_t)minimizer_type);
        for (uint64_t i = 0; i < nm; i++) mixin(repart[i]);
        if (minimizer_type == GKC_MINIMIZER_FREQ) for (uint64_t i = 0; i < nm; i++) mixin(freq_order[i]);
        c->model_hash = h;
    }
    GKC_TRY(gkc_alloc_histo(c));
    c->configured = true;
    return GKC_OK;
}

int gkc_set_solidity(gkc_ctx* c, int32_t amin, int32_t amax, uint32_t histo_max)
{
    if (!c) return GKC_ERR_ARG;
    if (c->stage_b_running || c->stage_b_thread.joinable()) GKC_FAIL(c, GKC_ERR_ARG, "%s while gkc_finish_pass_async is in flight (gkc_finish_pass_wait first)", __func__);
    if (amin > amax) GKC_FAIL(c, GKC_ERR_ARG, "abundance_min > abundance_max");
    if (histo_max < 1 || histo_max > (1u << 24)) GKC_FAIL(c, GKC_ERR_ARG, "histo_max out of range");
    c->amin = amin; c->amax = amax; c->d_hint = 0;
    if (histo_max != c->histo_max || !c->d_histo.p) {
        c->histo_max = histo_max;
        GKC_TRY(gkc_alloc_histo(c));
    }
    return GKC_OK;
}

int gkc_set_max_superkmer(gkc_ctx* c, uint32_t maxs)
{
    if (!c) return GKC_ERR_ARG;
    c->maxs = maxs == 1 ? 2 : maxs;            // the cap division uses a multiply-high reciprocal that needs maxs >= 2
    if (c->configured) { const uint32_t def = c->k <= 31 ? 28 : 60; if (c->maxs == 0 || c->maxs > def) c->maxs = def; }
    return GKC_OK;
}

int gkc_set_batch_keys(gkc_ctx* c, uint64_t max_keys)
{
    if (!c) return GKC_ERR_ARG;
    if (bg_active(c)) GKC_FAIL(c, GKC_ERR_ARG, "gkc_set_batch_keys while gkc_finish_pass_async is in flight");
    c->batch_cap = (size_t)max_keys;
    return GKC_OK;
}

int gkc_begin_pass(gkc_ctx* c, uint32_t pass)
{
    gkc_tun_refresh();
    if (!c) return GKC_ERR_ARG;
    if (!c->configured) GKC_FAIL(c, GKC_ERR_ARG, "gkc_configure must be called first");
    if (pass >= c->nb_passes) GKC_FAIL(c, GKC_ERR_ARG, "pass %u >= nb_passes %u", pass, c->nb_passes);
    GKC_HIP(c, hipSetDevice(c->device));
    // Stage B of the previous pass may still be counting (gkc_finish_pass_async): the NEXT pass may begin beside it — its Stage A is issue-bound, that Stage B
    // memory-bound — as long as nothing the two share is touched: another pass number (pass 0 clears every histogram: a new run), one GPU, no host sink
    const bool beside = bg_active(c);
    if (beside && !(bg_overlap_ok(c) && pass != c->b_pass && pass != 0))
        GKC_FAIL(c, GKC_ERR_ARG, "gkc_finish_pass_async of pass %u is in flight (gkc_finish_pass_wait first; only another pass > 0 of a one-GPU context without a host sink may begin beside it)", c->b_pass);
    (void)hipStreamSynchronize(c->stream);
    clear_segments(c);
    if (beside) c->b_moved_on = true;
    {   std::lock_guard<std::mutex> lk(c->mu);
        free_pass_outputs(c, pass);
        if (!beside) {
            if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
            for (hipEvent_t e : c->landed_events) (void)hipEventDestroy(e);
            gkc_sink_reset(c);
            c->landed_events.clear(); c->sink_used = 0; c->sink_overflow = false;        // the host sink holds ONE pass: the previous pass's records are overwritten from here on
            for (Dataset& D : c->datasets) { D.h_counts = nullptr; D.landed = nullptr; D.sink_batch = nullptr; }
            if (c->sink) (void)gkc_sink_prepare(c);                      // a sink set before gkc_configure / a context configured again (key width, partitions): the staging buffer is re-sized here
        }
        for (uint32_t p = 0; p < c->nb_partitions; p++) c->datasets[(size_t)pass * c->nb_partitions + p] = Dataset();
        c->pass_stats[pass] = gkc_stats{}; c->pass_released[pass] = 0;
    }
    if (pass == 0) GKC_HIP(c, hipMemsetAsync(c->d_histo.p, 0, (size_t)c->nb_passes * ((size_t)c->histo_max + 1) * 8, c->stream));   // pass 0 starts a new run
    else GKC_HIP(c, hipMemsetAsync(c->histo_of(pass), 0, ((size_t)c->histo_max + 1) * 8, c->stream));                         // a pass that is run again starts from zero
    c->pass = pass; c->in_pass = true;
    return GKC_OK;
}

int gkc_push_reads_device(gkc_ctx* c, const char* d_bases, const uint64_t* d_offsets, uint64_t n_reads, uint64_t n_bases)
{
    if (!c) return GKC_ERR_ARG;
    if (bg_active(c) && !(bg_overlap_ok(c) && c->in_pass)) GKC_FAIL(c, GKC_ERR_ARG, "%s while gkc_finish_pass_async is in flight (gkc_finish_pass_wait first)", __func__);
    if (!c->in_pass) GKC_FAIL(c, GKC_ERR_ARG, "gkc_begin_pass must be called first");
    if (((uintptr_t)d_bases & 15) != 0) GKC_FAIL(c, GKC_ERR_ARG, "d_bases must be 16-byte aligned");
    GKC_HIP(c, hipSetDevice(c->device));
    ScopedTimer tm(c, "total_stage_a");
    // A push is one segment with its own per-push buffers (0.8 bytes of descriptors per base, the record arena, above 4096 partitions its refined copy): those of ONE
    // push of 2e8 reads (3e10 bases) are 23 + 20 + 20 GB of fresh hipMallocs beside 100+ GB of parked blocks of other sizes — 2.6 s per step (DESIGN r3 §14). A push beyond
    // PUSH_SPLIT_BASES is therefore scanned in slices of about that many bases, cut at a read whose first base is 16-byte aligned (the scan loads 16 bytes at a time): every
    // slice is a segment of its own, and the slices ask the allocator for the same blocks one after the other. Same records, same counts (a read is never cut).
    const uint64_t PUSH_SPLIT_BASES = gkc_tun().push_split;
    if (n_bases <= PUSH_SPLIT_BASES + PUSH_SPLIT_BASES / 4 || n_reads < 2) return gkc_scan_push(c, d_bases, d_offsets, n_reads, n_bases);
    auto off_at = [&](uint64_t r, uint64_t* v) -> int { GKC_HIP(c, hipMemcpy(v, d_offsets + r, 8, hipMemcpyDeviceToHost)); return GKC_OK; };
    DevBuf d_off;
    uint64_t r0 = 0, base0 = 0;
    const uint64_t n_slices = (n_bases + PUSH_SPLIT_BASES - 1) / PUSH_SPLIT_BASES;
    int rc = GKC_OK;
    for (uint64_t j = 1; j <= n_slices && rc == GKC_OK && r0 < n_reads; j++) {
        uint64_t r1 = n_reads, base1 = n_bases;
        if (j < n_slices) {
            const uint64_t target = n_bases / n_slices * j;
            uint64_t lo = r0 + 1, hi = n_reads;                               // first read at or behind the target (binary search in the device table: 8-byte fetches)
            while (lo < hi && rc == GKC_OK) { const uint64_t mid = (lo + hi) / 2; uint64_t v = 0; rc = off_at(mid, &v); if (v < target) lo = mid + 1; else hi = mid; }
            if (rc != GKC_OK) break;
            std::vector<uint64_t> win((size_t)std::min<uint64_t>(4096, n_reads - lo + 1));        // ... and from there the first one that starts on a multiple of 16
            GKC_HIP(c, hipMemcpy(win.data(), d_offsets + lo, win.size() * 8, hipMemcpyDeviceToHost));
            size_t w = 0; while (w < win.size() && ((win[w] & 15) != 0 || lo + w >= n_reads)) w++;
            if (w == win.size()) continue;                                    // none in reach: this slice grows into the next one
            r1 = lo + w; base1 = win[w];
        }
        const uint64_t nr = r1 - r0;
        rc = c->ensure(d_off, (size_t)(nr + 1) * 8);
        if (rc != GKC_OK) break;
        GKC_HIP(c, hipMemcpyAsync(d_off.p, d_offsets + r0, (size_t)(nr + 1) * 8, hipMemcpyDeviceToDevice, c->stream));
        if (base0) hipLaunchKernelGGL(k_rebase_offsets, dim3((unsigned)((nr + 1 + 255) / 256)), dim3(256), 0, c->stream, (uint64_t*)d_off.p, nr + 1, base0);
        rc = gkc_scan_push(c, d_bases + base0, (const uint64_t*)d_off.p, nr, base1 - base0);
        r0 = r1; base0 = base1;
    }
    (void)hipStreamSynchronize(c->stream);
    d_off.release();
    return rc;
}

// Host buffers: the reads go to the device in chunks of about PUSH_CHUNK_BASES through two staging buffers — the H2D copy of chunk j+1
// (copy stream, DMA engine) runs while Stage A scans chunk j, so a push costs max(PCIe, scan) instead of their sum. Page-locked caller
// memory (gkc_host_alloc) makes the copies truly asynchronous; pageable memory works at the driver's staging rate. Each chunk is one
// segment of the pass. The call returns when the caller's buffers have been read completely.
#define PUSH_CHUNK_BASES (gkc_tun().push_chunk)      /* 1 GiB; GKC_PUSH_CHUNK: tests force many small chunks */
int gkc_push_reads(gkc_ctx* c, const char* bases, const uint64_t* offsets, uint64_t n_reads)
{
    if (!c) return GKC_ERR_ARG;
    if (bg_active(c) && !(bg_overlap_ok(c) && c->in_pass)) GKC_FAIL(c, GKC_ERR_ARG, "%s while gkc_finish_pass_async is in flight (gkc_finish_pass_wait first)", __func__);
    if (!c->in_pass) GKC_FAIL(c, GKC_ERR_ARG, "gkc_begin_pass must be called first");
    if (!offsets) GKC_FAIL(c, GKC_ERR_ARG, "offsets is required");
    if (offsets[0] != 0) GKC_FAIL(c, GKC_ERR_ARG, "offsets[0] must be 0");
    GKC_HIP(c, hipSetDevice(c->device));
    if (!c->copy_stream) GKC_HIP(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    for (int i = 0; i < 2; i++) {
        if (!c->h2d_copied[i]) GKC_HIP(c, hipEventCreateWithFlags(&c->h2d_copied[i], hipEventDisableTiming));
        if (!c->h2d_scanned[i]) GKC_HIP(c, hipEventCreateWithFlags(&c->h2d_scanned[i], hipEventDisableTiming));
    }
    // chunk boundaries on read boundaries
    std::vector<uint64_t> cut{0};
    while (cut.back() < n_reads) {
        const uint64_t r0 = cut.back();
        const uint64_t* e = std::upper_bound(offsets + r0 + 1, offsets + n_reads + 1, offsets[r0] + PUSH_CHUNK_BASES);
        uint64_t r1 = (uint64_t)(e - offsets) - 1;                // last read that still ends inside the budget
        if (r1 <= r0) r1 = r0 + 1;                                // one read longer than the budget: alone
        cut.push_back(std::min(r1, n_reads));
    }
    if (n_reads == 0) cut.push_back(0);
    const size_t n_chunks = cut.size() - 1;
    auto issue_copy = [&](size_t j) -> int {                      // chunk j -> staging buffer j & 1, on the copy stream
        const int b = (int)(j & 1);
        const uint64_t r0 = cut[j], r1 = cut[j + 1], nb = offsets[r1] - offsets[r0];
        GKC_HIP(c, hipStreamWaitEvent(c->copy_stream, c->h2d_scanned[b], 0));      // the scan that read this buffer two chunks ago is over
        GKC_TRY(c->ensure(c->h2d_bases[b], (size_t)std::max<uint64_t>(nb, PUSH_CHUNK_BASES / 4) + 64));
        GKC_TRY(c->ensure(c->h2d_offs[b], (size_t)(r1 - r0 + 1) * 8));
        if (nb) GKC_HIP(c, hipMemcpyAsync(c->h2d_bases[b].p, bases + offsets[r0], (size_t)nb, hipMemcpyHostToDevice, c->copy_stream));
        GKC_HIP(c, hipMemcpyAsync(c->h2d_offs[b].p, offsets + r0, (size_t)(r1 - r0 + 1) * 8, hipMemcpyHostToDevice, c->copy_stream));
        GKC_HIP(c, hipEventRecord(c->h2d_copied[b], c->copy_stream));
        return GKC_OK;
    };
    int rc = n_chunks ? issue_copy(0) : GKC_OK;
    for (size_t j = 0; rc == GKC_OK && j < n_chunks; j++) {
        const int b = (int)(j & 1);
        const uint64_t r0 = cut[j], r1 = cut[j + 1], nb = offsets[r1] - offsets[r0];
        if (j + 1 < n_chunks) {                                   // the next chunk's copy must not overwrite a buffer whose scan is still running:
            rc = issue_copy(j + 1);                               // it waits for h2d_scanned[(j+1)&1] (recorded after chunk j-1's scan) on the copy stream
            if (rc != GKC_OK) break;
        }
        GKC_HIP(c, hipStreamWaitEvent(c->stream, c->h2d_copied[b], 0));
        if (offsets[r0]) {
            hipLaunchKernelGGL(k_rebase_offsets, dim3((unsigned)((r1 - r0 + 1 + 255) / 256)), dim3(256), 0, c->stream, (uint64_t*)c->h2d_offs[b].p, r1 - r0 + 1, offsets[r0]);
            GKC_HIP(c, hipGetLastError());
        }
        rc = gkc_push_reads_device(c, (const char*)c->h2d_bases[b].p, (const uint64_t*)c->h2d_offs[b].p, r1 - r0, nb);
        if (hipEventRecord(c->h2d_scanned[b], c->stream) != hipSuccess) { (void)hipGetLastError(); }
    }
    (void)hipStreamSynchronize(c->copy_stream);                  // the caller's buffers are free again
    if (rc != GKC_OK) (void)hipStreamSynchronize(c->stream);
    return rc;
}

// host-buffer helper shared by the sampling entry points
static int upload_reads(gkc_ctx* c, const char* bases, const uint64_t* offsets, uint64_t n_reads, DevBuf& db, DevBuf& dof, uint64_t* n_bases)
{
    if (!offsets || offsets[0] != 0) GKC_FAIL(c, GKC_ERR_ARG, "offsets[0] must be 0");
    *n_bases = offsets[n_reads];
    GKC_TRY(c->ensure(db, (size_t)*n_bases + 64));
    int rc = c->ensure(dof, (size_t)(n_reads + 1) * 8);
    if (rc != GKC_OK) { db.release(); return rc; }
    hipError_t e = hipSuccess;
    if (*n_bases) e = hipMemcpyAsync(db.p, bases, (size_t)*n_bases, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dof.p, offsets, (size_t)(n_reads + 1) * 8, hipMemcpyHostToDevice, c->stream);
    if (e != hipSuccess) { db.release(); dof.release(); GKC_FAIL(c, GKC_ERR_HIP, "H2D copy failed: %s", hipGetErrorString(e)); }
    return GKC_OK;
}

int gkc_sample_minimizers(gkc_ctx* c, const char* bases, const uint64_t* offsets, uint64_t n_reads, uint64_t* superkmers_per_minim, uint64_t* kmers_per_minim)
{
    if (!c || !superkmers_per_minim || !kmers_per_minim) return GKC_ERR_ARG;
    if (!c->configured) GKC_FAIL(c, GKC_ERR_ARG, "gkc_configure must be called first (any repartition table)");
    GKC_HIP(c, hipSetDevice(c->device));
    DevBuf db, dof; uint64_t nb = 0;
    GKC_TRY(upload_reads(c, bases, offsets, n_reads, db, dof, &nb));
    int rc = gkc_scan_sample(c, (const char*)db.p, (const uint64_t*)dof.p, n_reads, nb, superkmers_per_minim, kmers_per_minim);
    (void)hipStreamSynchronize(c->stream);
    db.release(); dof.release();
    return rc;
}

int gkc_sample_exact(gkc_ctx* c, const char* bases, const uint64_t* offsets, uint64_t n_reads, uint64_t max_superkmers,
                     uint64_t* superkmers_per_minim, uint64_t* kmers_per_minim, uint64_t* kxmers_per_minim, uint64_t* reads_used)
{
    if (!c || !offsets || (!bases && n_reads)) return GKC_ERR_ARG;
    if (!c->configured) GKC_FAIL(c, GKC_ERR_ARG, "gkc_configure must be called first (any repartition table)");
    if (offsets[0] != 0) GKC_FAIL(c, GKC_ERR_ARG, "offsets[0] must be 0");
    GKC_HIP(c, hipSetDevice(c->device));
    return gkc_scan_sample_exact(c, bases, offsets, n_reads, max_superkmers, superkmers_per_minim, kmers_per_minim, kxmers_per_minim, reads_used);
}

int gkc_count_mmers(gkc_ctx* c, uint32_t m, const char* bases, const uint64_t* offsets, uint64_t n_reads, uint32_t* counts)
{
    if (!c || !counts) return GKC_ERR_ARG;
    if (m < 2 || m > 14) GKC_FAIL(c, GKC_ERR_ARG, "m must be in [2,14]");
    GKC_HIP(c, hipSetDevice(c->device));
    DevBuf db, dof; uint64_t nb = 0;
    GKC_TRY(upload_reads(c, bases, offsets, n_reads, db, dof, &nb));
    int rc = gkc_scan_count_mmers(c, m, (const char*)db.p, (const uint64_t*)dof.p, n_reads, counts);
    (void)hipStreamSynchronize(c->stream);
    db.release(); dof.release();
    return rc;
}

// detached: the pass is counted from c->b_* on the background stream (see gkc_finish_pass_async); otherwise in line from the context's own lists and stream
static int finish_pass_body(gkc_ctx* c, bool detached)
{
    (void)hipSetDevice(c->device);
    int rc;
    if (detached) {
        tl_stream_ = c->bg_stream;                                    // everything this thread launches, times or frees goes to the background stream
        for (hipEvent_t e : c->b_pending) { (void)hipEventSynchronize(e); (void)hipEventDestroy(e); }      // multi-GPU: the records other ranks sent must have arrived
        c->b_pending.clear();
        double reserve = 0;                                              // what Stage A of the next pass may allocate beside this Stage B: as much as this pass's records took
        if (bg_overlap_ok(c)) for (const Segment& sg : c->b_segments) reserve += (double)sg.rec_off.back() * (double)c->record_bytes * 1.6;
        {   ScopedTimer tm(c, "total_stage_b");
            rc = gkc_count_pass(c, c->b_pass, c->b_segments, c->bg_stream, reserve);
        }
        tl_stream_ = nullptr;
    } else {
        c->drain_pending();                                              // multi-GPU: the records other ranks sent must have arrived
        ScopedTimer tm(c, "total_stage_b");
        rc = gkc_count_pass(c, c->pass, c->segments, c->stream, 0.0);
    }
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);      // streamed results have landed ...
    gkc_sink_drain(c);                                                   // ... and the packed batches have been expanded into the sink
    if (rc == GKC_OK && c->sink_overflow) { c->set_error(GKC_ERR_CAPACITY, "the host sink (%llu bytes) is too small for the pass; the records that did not fit stay on the device (gkc_partition_counts)", (unsigned long long)c->sink_cap); }
    return rc;
}
int gkc_finish_pass(gkc_ctx* c)
{
    gkc_tun_refresh();
    if (!c) return GKC_ERR_ARG;
    if (!c->in_pass) GKC_FAIL(c, GKC_ERR_ARG, "no pass in progress");
    GKC_HIP(c, hipSetDevice(c->device));
    if (bg_active(c)) GKC_TRY(bg_join(c));                               // the pass before this one was detached: it finishes first (its lanes own the chip's memory plan)
    const int rc = finish_pass_body(c, false);
    if (rc != GKC_OK) return rc;
    c->in_pass = false;
    return GKC_OK;
}
int gkc_finish_pass_async(gkc_ctx* c)
{
    gkc_tun_refresh();
    if (!c) return GKC_ERR_ARG;
    if (!c->in_pass) GKC_FAIL(c, GKC_ERR_ARG, "no pass in progress");
    if (bg_active(c) && !bg_overlap_ok(c)) GKC_FAIL(c, GKC_ERR_ARG, "gkc_finish_pass_async is already in flight");
    GKC_HIP(c, hipSetDevice(c->device));
    if (bg_active(c)) GKC_TRY(bg_join(c));                               // Stage B of the pass before (its results stay): one Stage B at a time
    (void)hipStreamSynchronize(c->stream);                               // Stage A of this pass is complete
    if (!c->bg_stream) GKC_HIP(c, hipStreamCreateWithFlags(&c->bg_stream, hipStreamNonBlocking));
    // the pass leaves the context: from here on c->pass / c->segments belong to whatever the caller does next
    c->b_pass = c->pass; c->b_segments = std::move(c->segments); c->b_arenas = std::move(c->owned_arenas); c->b_pending = std::move(c->pending_events);
    c->segments.clear(); c->owned_arenas.clear(); c->pending_events.clear(); c->n_exchanged_segments = 0;
    c->b_detached = true; c->b_moved_on = false; c->in_pass = false;
    { std::lock_guard<std::mutex> lk(c->mu); c->stage_b_running = true; c->stage_b_rc = GKC_OK; }
    c->stage_b_thread = std::thread([c] {
        const int rc = finish_pass_body(c, true);
        { std::lock_guard<std::mutex> lk(c->mu); c->stage_b_rc = rc; c->stage_b_running = false; }
        c->cv_done.notify_all();
    });
    return GKC_OK;
}
int gkc_finish_pass_wait(gkc_ctx* c)
{
    if (!c) return GKC_ERR_ARG;
    if (!c->stage_b_thread.joinable()) GKC_FAIL(c, GKC_ERR_ARG, "no gkc_finish_pass_async in flight");
    GKC_HIP(c, hipSetDevice(c->device));
    return bg_join(c);
}
int gkc_set_host_sink(gkc_ctx* c, void* pinned, uint64_t cap_bytes)
{
    gkc_tun_refresh();
    if (!c) return GKC_ERR_ARG;
    if (c->stage_b_running) GKC_FAIL(c, GKC_ERR_ARG, "Stage B is running");
    GKC_HIP(c, hipSetDevice(c->device));
    if (pinned && !c->copy_stream) GKC_HIP(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    {   std::lock_guard<std::mutex> lk(c->mu);
        for (hipEvent_t e : c->landed_events) (void)hipEventDestroy(e);
        c->landed_events.clear();
        gkc_sink_reset(c);                                               // (deletes the packed batches: no dataset may keep a handle of one)
        for (Dataset& D : c->datasets) { D.h_counts = nullptr; D.landed = nullptr; D.sink_batch = nullptr; }
    }
    c->sink = pinned; c->sink_cap = pinned ? cap_bytes : 0; c->sink_used = 0; c->sink_overflow = false; c->sink_no6 = false;
    return gkc_sink_prepare(c);                                          // 8-byte keys: the page-locked staging buffer of the packed transfer (7/16 of the sink)
}
int gkc_set_sink_mode(gkc_ctx* c, int mode)
{
    if (!c) return GKC_ERR_ARG;
    if (mode != GKC_SINK_PACKED && mode != GKC_SINK_RAW) GKC_FAIL(c, GKC_ERR_ARG, "gkc_set_sink_mode: unknown mode %d", mode);
    if (c->stage_b_running) GKC_FAIL(c, GKC_ERR_ARG, "Stage B is running");
    c->sink_raw = mode == GKC_SINK_RAW;
    return c->sink ? gkc_set_host_sink(c, c->sink, c->sink_cap) : GKC_OK;      // (a sink already set: its staging buffer is made / kept for the new mode)
}
int gkc_wait_partition(gkc_ctx* c, uint32_t pass, uint32_t part, const void** host_records, uint64_t* n_solid)
{
    if (!c) return GKC_ERR_ARG;
    if (!c->configured || pass >= c->nb_passes || part >= c->nb_partitions) GKC_FAIL(c, GKC_ERR_ARG, "dataset (%u,%u) out of range", pass, part);
    Dataset* D = &c->datasets[(size_t)pass * c->nb_partitions + part];
    hipEvent_t ev = nullptr;
    {   std::unique_lock<std::mutex> lk(c->mu);
        c->cv_done.wait(lk, [&] { return D->done || !c->stage_b_running; });
        if (!D->done) { lk.unlock(); GKC_FAIL(c, GKC_ERR_ARG, "dataset (%u,%u) was not counted (pass not run, or Stage B failed: %s)", pass, part, c->err.msg.c_str()); }
        ev = D->landed;
    }
    if (ev) GKC_HIP(c, hipEventSynchronize(ev));
    if (D->sink_batch) gkc_sink_wait_batch(c, D->sink_batch);           // packed on the wire: in the sink once the host threads have expanded the batch
    if (host_records) *host_records = D->h_counts;
    if (n_solid) *n_solid = D->n_solid;
    return GKC_OK;
}

static int dataset_of(gkc_ctx* c, uint32_t pass, uint32_t part, Dataset** D)
{
    if (!c->configured) GKC_FAIL(c, GKC_ERR_ARG, "not configured");
    if (pass >= c->nb_passes || part >= c->nb_partitions) GKC_FAIL(c, GKC_ERR_ARG, "dataset (%u,%u) out of range", pass, part);
    *D = &c->datasets[(size_t)pass * c->nb_partitions + part];
    if (!(*D)->done) GKC_FAIL(c, GKC_ERR_ARG, "dataset (%u,%u) not counted yet", pass, part);
    return GKC_OK;
}

int gkc_partition_info(gkc_ctx* c, uint32_t pass, uint32_t part, uint64_t* n_solid, uint64_t* n_distinct, uint64_t* n_kmers)
{
    if (!c) return GKC_ERR_ARG;
    Dataset* D; GKC_TRY(dataset_of(c, pass, part, &D));
    if (n_solid) *n_solid = D->n_solid;  if (n_distinct) *n_distinct = D->n_distinct;  if (n_kmers) *n_kmers = D->n_kmers;
    return GKC_OK;
}
static int fetch_records(gkc_ctx* c, Dataset* D, uint32_t pass, uint32_t part, uint64_t first, uint64_t n, void* out);
int gkc_partition_counts(gkc_ctx* c, uint32_t pass, uint32_t part, void* out, uint64_t cap, uint64_t* n_solid)
{
    if (!c) return GKC_ERR_ARG;
    Dataset* D; GKC_TRY(dataset_of(c, pass, part, &D));
    if (n_solid) *n_solid = D->n_solid;
    if (cap < D->n_solid) GKC_FAIL(c, GKC_ERR_CAPACITY, "dataset holds %llu records, buffer %llu", (unsigned long long)D->n_solid, (unsigned long long)cap);
    return fetch_records(c, D, pass, part, 0, D->n_solid, out);
}
int gkc_partition_counts_range(gkc_ctx* c, uint32_t pass, uint32_t part, uint64_t first, uint64_t n, void* out)
{
    if (!c) return GKC_ERR_ARG;
    Dataset* D; GKC_TRY(dataset_of(c, pass, part, &D));
    if (first > D->n_solid || n > D->n_solid - first) GKC_FAIL(c, GKC_ERR_ARG, "records [%llu, +%llu) of a dataset of %llu", (unsigned long long)first, (unsigned long long)n, (unsigned long long)D->n_solid);
    if (n && !out) GKC_FAIL(c, GKC_ERR_ARG, "no output buffer");
    return fetch_records(c, D, pass, part, first, n, out);
}
static int fetch_records(gkc_ctx* c, Dataset* D, uint32_t pass, uint32_t part, uint64_t first, uint64_t n, void* out)
{
    const size_t rb = c->key_words == 1 ? 16 : 32;
    if (n) {
        // On a stream of its own: a finished dataset's records are complete (the batch's stream was synchronized before it was marked done), and consumers fetch
        // partitions WHILE Stage B counts the later batches (gkc_finish_pass_async + gkc_wait_partition). On the context's stream — Stage B's first lane — every fetch
        // queued behind, and between, that lane's kernels: measured inside the patched dbgh5 at 1e8 reads, Stage B 0.23 s -> 1.9 s with 3884 partition commands fetching.
        hipStream_t fs;
        {   std::lock_guard<std::mutex> lk(c->mu);
            if (!c->fetch_stream) GKC_HIP(c, hipStreamCreateWithFlags(&c->fetch_stream, hipStreamNonBlocking));
            fs = c->fetch_stream;
        }
        hipEvent_t ev = nullptr;
        GKC_HIP(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        hipError_t e = hipMemcpyAsync(out, (const uint8_t*)D->d_counts + (size_t)first * rb, (size_t)n * rb, hipMemcpyDeviceToHost, fs);
        if (e == hipSuccess) e = hipEventRecord(ev, fs);
        if (e == hipSuccess) e = hipEventSynchronize(ev);               // (this call's copy, not what other threads queued behind it)
        (void)hipEventDestroy(ev);
        if (e != hipSuccess) GKC_FAIL(c, GKC_ERR_HIP, "fetching dataset (%u,%u) failed: %s", pass, part, hipGetErrorString(e));
    }
    return GKC_OK;
}
int gkc_partition_counts_device(gkc_ctx* c, uint32_t pass, uint32_t part, const void** d_counts, uint64_t* n_solid)
{
    if (!c) return GKC_ERR_ARG;
    Dataset* D; GKC_TRY(dataset_of(c, pass, part, &D));
    if (d_counts) *d_counts = D->d_counts;  if (n_solid) *n_solid = D->n_solid;
    return GKC_OK;
}
int gkc_histogram(gkc_ctx* c, uint64_t* out, uint32_t n_bins)
{
    if (!c) return GKC_ERR_ARG;
    if (n_bins < c->histo_max + 1) GKC_FAIL(c, GKC_ERR_CAPACITY, "histogram has %u bins", c->histo_max + 1);
    const size_t nb = (size_t)c->histo_max + 1;
    std::vector<uint64_t> all((size_t)c->nb_passes * nb);
    GKC_HIP(c, hipMemcpyAsync(all.data(), c->d_histo.p, all.size() * 8, hipMemcpyDeviceToHost, c->stream));
    GKC_HIP(c, hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < nb; i++) { uint64_t s = 0; for (uint32_t p = 0; p < c->nb_passes; p++) s += all[(size_t)p * nb + i]; out[i] = s; }
    return GKC_OK;
}
int gkc_get_stats(gkc_ctx* c, gkc_stats* out)
{
    if (!c || !out) return GKC_ERR_ARG;
    gkc_stats t{};
    for (size_t p = 0; p < c->pass_stats.size(); p++) {
        const gkc_stats& S = c->pass_stats[p];
        if (p == 0) { t.kmers_nb_valid = S.kmers_nb_valid; t.kmers_nb_invalid = S.kmers_nb_invalid; t.nb_sequences = S.nb_sequences; t.nb_bases = S.nb_bases;
                      t.seq_len_min = S.seq_len_min; t.seq_len_max = S.seq_len_max; t.seq_len_sq_sum = S.seq_len_sq_sum; }
        t.kmers_nb_distinct += S.kmers_nb_distinct; t.kmers_nb_solid += S.kmers_nb_solid; t.nb_superkmers += S.nb_superkmers;
        t.superkmer_bytes += S.superkmer_bytes; t.oversize_buckets += S.oversize_buckets; t.dedupe_kmers_in += S.dedupe_kmers_in; t.dedupe_keys_out += S.dedupe_keys_out;
    }
    if (c->pass < c->pass_stats.size()) t.reserved[0] = c->pass_stats[c->pass].nb_sequences;      // pass_nb_sequences: reads pushed in the CURRENT pass (nb_sequences is pass 0's)
    t.reserved[1] = c->sink_wire_bytes;                                                            // bytes the packed result batches of the last counted pass took on the link
    *out = t;
    return GKC_OK;
}
int gkc_get_timing(gkc_ctx* c, const char* name, double* ms, uint64_t* launches)
{
    if (!c || !name) return GKC_ERR_ARG;
    auto it = c->timing.find(name);
    if (ms) *ms = it == c->timing.end() ? 0.0 : it->second.ms;
    if (launches) *launches = it == c->timing.end() ? 0 : it->second.launches;
    return GKC_OK;
}

int gkc_partition_superkmers(gkc_ctx* c, uint32_t part, uint8_t* out, uint64_t cap, uint64_t* n_bytes, uint64_t* n_sk, uint64_t* n_k)
{
    if (!c) return GKC_ERR_ARG;
    if (!c->configured || part >= c->nb_partitions) GKC_FAIL(c, GKC_ERR_ARG, "bad partition");
    GKC_HIP(c, hipStreamSynchronize(c->stream));
    uint64_t a = 0, b = 0, d = 0;
    GKC_TRY(gkc_export_superkmers(c, part, out, cap, &a, &b, &d));
    if (n_bytes) *n_bytes = a;  if (n_sk) *n_sk = b;  if (n_k) *n_k = d;
    return GKC_OK;
}

int gkc_segment_count(gkc_ctx* c, uint32_t* n) { if (!c || !n) return GKC_ERR_ARG; *n = (uint32_t)c->segments.size(); return GKC_OK; }
int gkc_segment_export(gkc_ctx* c, uint32_t seg, const void** d_records, uint32_t* record_bytes, uint64_t* rec_offsets, uint64_t* kmers)
{
    if (!c) return GKC_ERR_ARG;
    if (seg >= c->segments.size()) GKC_FAIL(c, GKC_ERR_ARG, "segment %u out of range", seg);
    GKC_HIP(c, hipStreamSynchronize(c->stream));
    const Segment& s = c->segments[seg];
    if (d_records) *d_records = s.d_records;  if (record_bytes) *record_bytes = c->record_bytes;
    if (rec_offsets) memcpy(rec_offsets, s.rec_off.data(), (size_t)(c->nb_partitions + 1) * 8);
    if (kmers) memcpy(kmers, s.nkmers.data(), (size_t)c->nb_partitions * 8);
    return GKC_OK;
}
int gkc_segment_import(gkc_ctx* c, const void* d_records, const uint64_t* rec_offsets, const uint64_t* kmers)
{
    if (!c) return GKC_ERR_ARG;
    if (!c->in_pass) GKC_FAIL(c, GKC_ERR_ARG, "gkc_begin_pass must be called first");
    if (!rec_offsets || !kmers) GKC_FAIL(c, GKC_ERR_ARG, "offset / k-mer tables are required");
    Segment s; s.d_records = d_records; s.owned = false; s.foreign = true;
    s.rec_off.assign(rec_offsets, rec_offsets + c->nb_partitions + 1);
    s.nkmers.assign(kmers, kmers + c->nb_partitions);
    c->segments.push_back(std::move(s));
    return GKC_OK;
}
int gkc_segments_clear(gkc_ctx* c)
{
    if (!c) return GKC_ERR_ARG;
    if (c->stage_b_running || c->stage_b_thread.joinable()) GKC_FAIL(c, GKC_ERR_ARG, "%s while gkc_finish_pass_async is in flight (gkc_finish_pass_wait first)", __func__);
    (void)hipStreamSynchronize(c->stream); clear_segments(c); return GKC_OK;
}

int gkc_synth_reads_device(gkc_ctx* c, uint64_t seed, uint64_t first_read, uint64_t n_reads, uint32_t read_len, uint64_t genome_len,
                           uint32_t sub_ppm, char** d_bases, uint64_t** d_offsets)
{
    return gkc_synth_reads_profile_device(c, seed, first_read, n_reads, read_len, genome_len, sub_ppm, GKC_SYNTH_UNIFORM, d_bases, d_offsets);
}
int gkc_synth_reads_profile_device(gkc_ctx* c, uint64_t seed, uint64_t first_read, uint64_t n_reads, uint32_t read_len, uint64_t genome_len,
                                   uint32_t sub_ppm, uint32_t profile, char** d_bases, uint64_t** d_offsets)
{
    if (!c || !d_bases || !d_offsets) return GKC_ERR_ARG;
    if (profile > GKC_SYNTH_SKEWED) GKC_FAIL(c, GKC_ERR_ARG, "unknown generator profile %u", profile);
    if (read_len == 0 || genome_len < read_len) GKC_FAIL(c, GKC_ERR_ARG, "genome_len must be >= read_len > 0");
    GKC_HIP(c, hipSetDevice(c->device));
    void *b = nullptr, *o = nullptr;
    const size_t nb = (size_t)n_reads * read_len;
    if (hipMalloc(&b, nb + 64) != hipSuccess) GKC_FAIL(c, GKC_ERR_NOMEM, "hipMalloc of %zu bases failed", nb);
    if (hipMalloc(&o, (size_t)(n_reads + 1) * 8) != hipSuccess) { (void)hipFree(b); GKC_FAIL(c, GKC_ERR_NOMEM, "hipMalloc of offsets failed"); }
    const unsigned grid = (unsigned)std::min<uint64_t>((nb + 255) / 256 + 1, 256 * 32);
    hipLaunchKernelGGL(k_synth_reads, dim3(grid), dim3(256), 0, c->stream, seed, first_read, n_reads, read_len, genome_len, sub_ppm, profile, (uint8_t*)b, (uint64_t*)o);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { (void)hipFree(b); (void)hipFree(o); GKC_FAIL(c, GKC_ERR_HIP, "synth kernel failed: %s", hipGetErrorString(e)); }
    *d_bases = (char*)b; *d_offsets = (uint64_t*)o;
    return GKC_OK;
}
int gkc_device_free(gkc_ctx* c, void* p) { if (!c) return GKC_ERR_ARG; if (p) { GKC_HIP(c, hipSetDevice(c->device)); c->dfree(p); } return GKC_OK; }      // back to the context's allocator (a foreign pointer: hipFree)
int gkc_release_pass(gkc_ctx* c, uint32_t pass)
{
    if (!c) return GKC_ERR_ARG;
    if (bg_active(c) && (!bg_overlap_ok(c) || pass == c->b_pass))       // another pass than the one Stage B is counting may go (one GPU, no sink: the conditions of overlapped passes)
        GKC_FAIL(c, GKC_ERR_ARG, "%s while gkc_finish_pass_async is in flight (gkc_finish_pass_wait first)", __func__);
    if (!c->configured || pass >= c->nb_passes) GKC_FAIL(c, GKC_ERR_ARG, "no such pass %u", pass);
    if (c->in_pass && c->pass == pass) GKC_FAIL(c, GKC_ERR_ARG, "pass %u is still open (gkc_finish_pass first)", pass);
    GKC_HIP(c, hipSetDevice(c->device));
    (void)hipStreamSynchronize(c->stream);
    std::lock_guard<std::mutex> lk(c->mu);                                // (Stage B of another pass may be adding its own results beside this)
    free_pass_outputs(c, pass);
    for (uint32_t p = 0; p < c->nb_partitions; p++) c->datasets[(size_t)pass * c->nb_partitions + p] = Dataset();   // statistics of the pass stay
    c->pass_released[pass] = 1;
    return GKC_OK;
}
int gkc_device_memory(gkc_ctx* c, uint64_t* usable_bytes, uint64_t* total_bytes)
{
    if (!c) return GKC_ERR_ARG;
    GKC_HIP(c, hipSetDevice(c->device));
    size_t free_b = 0, total_b = 0;
    GKC_HIP(c, hipMemGetInfo(&free_b, &total_b));
    if (usable_bytes) *usable_bytes = (uint64_t)free_b + (uint64_t)c->pool.cached_bytes;
    if (total_bytes) *total_bytes = (uint64_t)total_b;
    return GKC_OK;
}
int gkc_host_alloc(void** p, uint64_t n_bytes)
{
    if (!p) return GKC_ERR_ARG;
    *p = nullptr;
    if (hipHostMalloc(p, n_bytes ? (size_t)n_bytes : 1, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); *p = nullptr; return GKC_ERR_NOMEM; }
    return GKC_OK;
}
int gkc_host_free(void* p) { if (p && hipHostFree(p) != hipSuccess) { (void)hipGetLastError(); return GKC_ERR_HIP; } return GKC_OK; }
int gkc_device_to_host(gkc_ctx* c, void* dst, const void* src, uint64_t n)
{
    if (!c) return GKC_ERR_ARG;
    GKC_HIP(c, hipMemcpy(dst, src, (size_t)n, hipMemcpyDeviceToHost));
    return GKC_OK;
}
int gkc_host_to_device(gkc_ctx* c, void* dst, const void* src, uint64_t n)
{
    if (!c) return GKC_ERR_ARG;
    GKC_HIP(c, hipMemcpy(dst, src, (size_t)n, hipMemcpyHostToDevice));
    return GKC_OK;
}

int gkc_kmer_checksum_device(gkc_ctx* c, const char* d_bases, const uint64_t* d_offsets, uint64_t n_reads, uint64_t n_bases,
                             uint64_t* checksum, uint64_t* n_valid)
{
    if (!c) return GKC_ERR_ARG;
    if (!c->configured) GKC_FAIL(c, GKC_ERR_ARG, "not configured");
    (void)n_bases;
    DevBuf d; GKC_TRY(c->ensure(d, 16));
    GKC_HIP(c, hipMemsetAsync(d.p, 0, 16, c->stream));
    const unsigned grid = (unsigned)std::min<uint64_t>((n_reads + 255) / 256 + 1, 256 * 64);
    if (c->key_words == 1) hipLaunchKernelGGL((k_kmer_checksum<1>), dim3(grid), dim3(256), 0, c->stream, (const uint8_t*)d_bases, d_offsets, n_reads, c->k, (unsigned long long*)d.p);
    else                   hipLaunchKernelGGL((k_kmer_checksum<2>), dim3(grid), dim3(256), 0, c->stream, (const uint8_t*)d_bases, d_offsets, n_reads, c->k, (unsigned long long*)d.p);
    uint64_t h[2] = {0, 0};
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(h, d.p, 16, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    d.release();
    if (e != hipSuccess) GKC_FAIL(c, GKC_ERR_HIP, "checksum kernel failed: %s", hipGetErrorString(e));
    if (checksum) *checksum = h[0];  if (n_valid) *n_valid = h[1];
    return GKC_OK;
}
int gkc_result_checksum(gkc_ctx* c, uint64_t* checksum, uint64_t* sum_abundance)
{
    if (!c || !checksum || !sum_abundance) return GKC_ERR_ARG;
    GKC_TRY(gkc_require_resident(c, "gkc_result_checksum"));
    return gkc_result_checksum_impl(c, checksum, sum_abundance);
}

}  // extern "C"
