// gkc_mphf.hip — minimal perfect hash of the solid k-mers + abundance map on gfx950.
//
// Replaces (reference, under /root/reference/gatb-core/):
//   boomphf::mphf ctor / processLevel / getLevel / lookup / save     thirdparty/BooPHF/BooPHF.h:734-1108
//   bitVector build_ranks / rank / save                              thirdparty/BooPHF/BooPHF.h:596-635
//   BooPHF<Key> hasher (jenkins64, seed = std::mt19937_64(37)())      src/gatb/tools/collections/impl/BooPHF.hpp:73-258
//   MPHFAlgorithm::execute / populate                                src/gatb/kmer/impl/MPHFAlgorithm.cpp:150-275
//   MapMPHF::initDiscretizationScheme                                src/gatb/tools/collections/impl/MapMPHF.hpp:96-145
//
// BooPHF is level-wise collision filtering: at level i every key still alive hashes into a bit array of gamma * (expected survivors)
// bits; a slot hit exactly once keeps its bit and places the key, slots hit more than once are cleared and their keys go on to level
// i + 1 with the next hash of their xorshift sequence. Nothing in it is sequential except the level loop, so per level:
//   k_mphf_insert   every alive key: bit = atomic OR into the level's bit array; if it was already set, OR into the collision array
//   k_mphf_clear    bits &= ~collisions, popcount per 512-bit block (the rank samples)
//   scan            block popcounts -> ranks (BooPHF's _ranks entries: one per 512 bits, running over all levels)
//   k_mphf_filter   alive keys whose slot ended up cleared -> stable compaction into the next level's key list
// The level bit arrays and rank samples are the ones BooPHF builds for the same key set (they do not depend on the processing order), so
// gkc_mphf_save writes the byte stream of mphf::save. Keys that survive all 24 filtering levels (probability ~1e-13 per key at gamma 3)
// get consecutive codes after the last rank in key order, as a single-threaded BooPHF assigns them.
#include "gkc_common.hpp"
#include "gkc_device.hpp"
#include <cmath>
#include <vector>

namespace {

constexpr int MPHF_LEVELS = 25;                                   // BooPHF.h:1029
constexpr uint64_t MPHF_SEED = 18006821046139946489ULL;           // std::mt19937_64 rng(37); rng()   (BooPHF.hpp:246-249)
constexpr int MPHF_THREADS = 256;
constexpr int GKC_MPHF_REBUILD_ORDERED = -1000;                    // internal: mphf_build_from_list -> mphf_build_arrays

struct MphfLevels {                                               // by value into the kernels
    uint64_t domain[MPHF_LEVELS];                                 // bits of level i (multiple of 64)
    uint64_t word0[MPHF_LEVELS];                                  // first 64-bit word of level i in the concatenated bit array (nchar = domain/64 + 1 words each)
    uint64_t rank0[MPHF_LEVELS];                                  // first rank sample of level i in the concatenated rank array
};

__device__ __forceinline__ void jenkins_mix(uint64_t& a, uint64_t& b, uint64_t& c)     // BooPHF.hpp:186-201
{
    a -= b; a -= c; a ^= (c >> 43);  b -= c; b -= a; b ^= (a << 9);   c -= a; c -= b; c ^= (b >> 8);
    a -= b; a -= c; a ^= (c >> 38);  b -= c; b -= a; b ^= (a << 23);  c -= a; c -= b; c ^= (b >> 5);
    a -= b; a -= c; a ^= (c >> 35);  b -= c; b -= a; b ^= (a << 49);  c -= a; c -= b; c ^= (b >> 11);
    a -= b; a -= c; a ^= (c >> 12);  b -= c; b -= a; b ^= (a << 18);  c -= a; c -= b; c ^= (b >> 22);
}
// the two base hashes of a key: get<0> and get<2> of jenkins64 over its 8 / 16 bytes (BooPHF.hpp:93-146, 254-258)
__device__ __forceinline__ void mphf_hash_pair(uint64_t lo, uint64_t hi, int wide, uint64_t& h0, uint64_t& h1)
{
    uint64_t a = MPHF_SEED, b = MPHF_SEED, c = 0x9e3779b97f4a7c13ULL;
    c += wide ? 16 : 8;
    if (wide) b += hi;
    a += lo;
    jenkins_mix(a, b, c);
    h0 = a; h1 = c;
}
__device__ __forceinline__ uint64_t xs_next(uint64_t& s0_, uint64_t& s1_)             // BooPHF.h:350-358, state s[0], s[1]
{
    uint64_t s1 = s0_; const uint64_t s0 = s1_;
    s0_ = s0; s1 ^= s1 << 23;
    s1_ = s1 ^ s0 ^ (s1 >> 17) ^ (s0 >> 26);
    return s1_ + s0;
}
// hash of `level` for a key (h0, h1, then the xorshift sequence seeded with them)
__device__ __forceinline__ uint64_t mphf_level_hash(uint64_t lo, uint64_t hi, int wide, int level)
{
    uint64_t h0, h1; mphf_hash_pair(lo, hi, wide, h0, h1);
    if (level == 0) return h0;
    if (level == 1) return h1;
    uint64_t s0 = h0, s1 = h1, h = 0;
    for (int i = 2; i <= level; i++) h = xs_next(s0, s1);
    return h;
}
__device__ __forceinline__ void load_key2(const uint8_t* p, int wide, uint64_t& lo, uint64_t& hi)
{
    lo = *reinterpret_cast<const uint64_t*>(p); hi = wide ? *reinterpret_cast<const uint64_t*>(p + 8) : 0;
}

// keys of Count records (stride 16 / 32) or plain key arrays -> compact key list (8 / 16 bytes per key)
__global__ void k_mphf_gather(const uint8_t* __restrict__ src, uint64_t n, uint32_t stride, int wide, uint64_t* __restrict__ dst)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t lo, hi; load_key2(src + i * stride, wide, lo, hi);
        if (wide) { dst[2 * i] = lo; dst[2 * i + 1] = hi; } else dst[i] = lo;
    }
}
__global__ void k_mphf_insert(const uint64_t* __restrict__ keys, uint64_t n, int wide, int level, uint64_t domain, uint32_t* __restrict__ bits, uint32_t* __restrict__ coll)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t lo = wide ? keys[2 * i] : keys[i], hi = wide ? keys[2 * i + 1] : 0;
        const uint64_t pos = mphf_level_hash(lo, hi, wide, level) % domain;              // insertIntoLevel (BooPHF.h:1098-1108)
        const uint32_t bit = 1u << (pos & 31);
        if (atomicOr(&bits[pos >> 5], bit) & bit) atomicOr(&coll[pos >> 5], bit);
    }
}
// clearCollisions (BooPHF.h:511-523) + popcount of every 512-bit block (8 words) -> block_pop
__global__ void k_mphf_clear(uint64_t* __restrict__ bits, const uint64_t* __restrict__ coll, uint64_t nchar, uint64_t* __restrict__ block_pop)
{
    const uint64_t blk = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (blk * 8 >= nchar) return;
    uint64_t pc = 0;
    for (int j = 0; j < 8; j++) { const uint64_t w = blk * 8 + j; if (w < nchar) { const uint64_t v = bits[w] & ~coll[w]; bits[w] = v; pc += __popcll(v); } }
    block_pop[blk] = pc;
}
__global__ void k_mphf_add_offset(uint64_t* __restrict__ ranks, uint64_t n, uint64_t offset)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) ranks[i] += offset;
}
// alive after this level = the slot of the key was cleared (collision)
__global__ void k_mphf_flag(const uint64_t* __restrict__ keys, uint64_t n, int wide, int level, uint64_t domain, const uint32_t* __restrict__ bits, uint64_t* __restrict__ flag)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t lo = wide ? keys[2 * i] : keys[i], hi = wide ? keys[2 * i + 1] : 0;
    const uint64_t pos = mphf_level_hash(lo, hi, wide, level) % domain;
    flag[i] = ((bits[pos >> 5] >> (pos & 31)) & 1u) ? 0 : 1;
}
__global__ void k_mphf_compact(const uint64_t* __restrict__ keys, uint64_t n, int wide, const uint64_t* __restrict__ flag_excl, const uint64_t* __restrict__ flag_total,
                               uint64_t* __restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t e = flag_excl[i], nx = (i + 1 < n) ? flag_excl[i + 1] : *flag_total;
    if (nx != e) { if (wide) { out[2 * e] = keys[2 * i]; out[2 * e + 1] = keys[2 * i + 1]; } else out[e] = keys[i]; }
}

// ------------------------------------------------------------------------------------------------ region build of a level (round 4, second session)
// k_mphf_insert / k_mphf_flag / the flag scan / k_mphf_compact cost two global atomics, one random 4-byte gather and ~50 bytes of flag traffic per alive key and level: at the
// chip's global-atomic rate (~2e10 /s) level 0 of 5.8e8 keys alone is ~37 ms of atomics. The level's bit array depends on the keys' slots only, not on their order, so the
// keys are first bucketed by REGION of their slot (2^19 bits = 64 KB of the level's array; LDS histogram / cursors per workgroup, static key -> workgroup assignment,
// like the Bloom build), then one workgroup per region builds its piece in LDS — `seen` and `collided`, LDS atomics — writes both out with plain stores (regions are
// disjoint word ranges) and appends the region's keys whose slot collided to the next level's list (one global atomic per wave and round). k_mphf_clear and the rank
// scan then run as before. The next list comes out in region order instead of key order — nothing depends on it except the codes of keys that survive all 24
// filtering levels, for which the caller rebuilds with the ordered path (probability ~1e-13 per key).
constexpr uint32_t MR_BITS = 19, MR_WORDS = 1u << (MR_BITS - 5), MR_MAX_REGIONS = 16384, MR_WGS = 1024, MR_THREADS = 1024;
template <bool SCATTER>
__global__ __launch_bounds__(MR_THREADS) void k_mphf_regions(const uint64_t* __restrict__ keys, uint64_t n, int wide, int level, uint64_t domain, uint64_t chunk, uint32_t n_regions,
                                                              uint32_t* __restrict__ wg_cnt, const uint64_t* __restrict__ region_off, uint64_t* __restrict__ items)
{
    extern __shared__ uint32_t s_r[];                              // [n_regions] count / cursor (absolute slot: fewer than 2^32 keys per level here)
    for (uint32_t r = threadIdx.x; r < n_regions; r += MR_THREADS) s_r[r] = SCATTER ? (uint32_t)region_off[r] + wg_cnt[(uint64_t)blockIdx.x * n_regions + r] : 0u;
    __syncthreads();
    const uint64_t i0 = (uint64_t)blockIdx.x * chunk, i1 = min(n, i0 + chunk);
    for (uint64_t g = i0 + threadIdx.x; g < i1; g += MR_THREADS) {
        const uint64_t lo = wide ? keys[2 * g] : keys[g], hi = wide ? keys[2 * g + 1] : 0;
        const uint64_t pos = mphf_level_hash(lo, hi, wide, level) % domain;
        const uint32_t r = (uint32_t)(pos >> MR_BITS);
        const uint32_t slot = atomicAdd(&s_r[r], 1u);
        if (SCATTER) {
            const uint64_t at = slot;
            if (wide) { items[2 * at] = lo; items[2 * at + 1] = hi; } else items[at] = lo;
        }
    }
    if (!SCATTER) {
        __syncthreads();
        for (uint32_t r = threadIdx.x; r < n_regions; r += MR_THREADS) wg_cnt[(uint64_t)blockIdx.x * n_regions + r] = s_r[r];
    }
}
__global__ void k_mr_wg_prefix(uint32_t* __restrict__ wg_cnt, uint32_t n_wgs, uint32_t n_regions, uint64_t* __restrict__ region_tot)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_regions) return;
    uint32_t run = 0;
    for (uint32_t w = 0; w < n_wgs; w++) { const uint32_t t = wg_cnt[(uint64_t)w * n_regions + r]; wg_cnt[(uint64_t)w * n_regions + r] = run; run += t; }
    region_tot[r] = run;
}
__global__ __launch_bounds__(MR_THREADS) void k_mphf_region_build(const uint64_t* __restrict__ items, const uint64_t* __restrict__ region_off, int wide, int level, uint64_t domain,
                                                                   uint32_t* __restrict__ lbits, uint32_t* __restrict__ coll, uint64_t n_words32 /* of the level's array */,
                                                                   uint64_t* __restrict__ next_keys, unsigned long long* __restrict__ next_n)
{
    extern __shared__ uint32_t s_b[];                              // [MR_WORDS] seen, [MR_WORDS] collided
    uint32_t* s_seen = s_b; uint32_t* s_coll = s_b + MR_WORDS;
    for (uint32_t w = threadIdx.x; w < 2 * MR_WORDS; w += MR_THREADS) s_b[w] = 0u;
    __syncthreads();
    const uint32_t r = blockIdx.x;
    const uint64_t i0 = region_off[r], i1 = region_off[r + 1];
    for (uint64_t i = i0 + threadIdx.x; i < i1; i += MR_THREADS) {
        const uint64_t lo = wide ? items[2 * i] : items[i], hi = wide ? items[2 * i + 1] : 0;
        const uint32_t rel = (uint32_t)((mphf_level_hash(lo, hi, wide, level) % domain) & ((1u << MR_BITS) - 1u));
        const uint32_t bit = 1u << (rel & 31);
        if (atomicOr(&s_seen[rel >> 5], bit) & bit) atomicOr(&s_coll[rel >> 5], bit);
    }
    __syncthreads();
    const uint64_t w0 = (uint64_t)r * MR_WORDS;
    for (uint32_t w = threadIdx.x; w < MR_WORDS; w += MR_THREADS) if (w0 + w < n_words32) { lbits[w0 + w] = s_seen[w]; coll[w0 + w] = s_coll[w]; }
    // the keys whose slot collided go on to the next level: counted first (one reservation on the next list's cursor per WORKGROUP — a reservation per wave and
    // round was 9e6 atomics on one address: 109 ms for level 0 of 5.8e8 keys), then written wave by wave through an LDS cursor
    __shared__ uint32_t s_cnt[MR_THREADS / 64], s_n;
    __shared__ unsigned long long s_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    auto alive_of = [&](uint64_t i, uint64_t& lo, uint64_t& hi) -> bool {
        lo = wide ? items[2 * i] : items[i]; hi = wide ? items[2 * i + 1] : 0;
        const uint32_t rel = (uint32_t)((mphf_level_hash(lo, hi, wide, level) % domain) & ((1u << MR_BITS) - 1u));
        return (s_coll[rel >> 5] >> (rel & 31)) & 1u;
    };
    uint32_t mine = 0;
    for (uint64_t i = i0 + threadIdx.x; i < i1; i += MR_THREADS) { uint64_t lo, hi; mine += alive_of(i, lo, hi) ? 1u : 0u; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d, 64);
    if (lane == 0) s_cnt[wave] = mine;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t tot = 0; for (int w = 0; w < MR_THREADS / 64; w++) tot += s_cnt[w]; s_base = tot ? atomicAdd(next_n, (unsigned long long)tot) : 0ull; }
    __syncthreads();
    const unsigned long long base0 = s_base;
    for (uint64_t ib = i0; ib < i1; ib += MR_THREADS) {
        const uint64_t i = ib + threadIdx.x;
        uint64_t lo = 0, hi = 0; bool alive = false;
        if (i < i1) alive = alive_of(i, lo, hi);
        const unsigned long long bal = __ballot(alive);
        if (bal) {
            uint32_t wb = 0;
            if (lane == 0) wb = atomicAdd(&s_n, (uint32_t)__popcll(bal));
            wb = __shfl(wb, 0, 64);
            if (alive) {
                const uint64_t at = base0 + wb + (uint64_t)__popcll(bal & ((1ull << lane) - 1ull));
                if (wide) { next_keys[2 * at] = lo; next_keys[2 * at + 1] = hi; } else next_keys[at] = lo;
            }
        }
    }
}

// lookup (BooPHF.h:787-822): first level whose bit is set at the key's slot -> rank; all 24 filtering levels miss -> the final list
__device__ __forceinline__ uint64_t mphf_lookup_dev(const MphfLevels& L, const uint64_t* __restrict__ bits, const uint64_t* __restrict__ ranks, int wide,
                                                    const uint64_t* __restrict__ final_keys, uint64_t n_final, uint64_t lastrank, uint64_t lo, uint64_t hi)
{
    uint64_t h0, h1; mphf_hash_pair(lo, hi, wide, h0, h1);
    uint64_t s0 = h0, s1 = h1;
    for (int lv = 0; lv < MPHF_LEVELS - 1; lv++) {
        const uint64_t h = lv == 0 ? h0 : (lv == 1 ? h1 : xs_next(s0, s1));
        const uint64_t pos = h % L.domain[lv];
        const uint64_t* b = bits + L.word0[lv];
        if ((b[pos >> 6] >> (pos & 63)) & 1ULL) {
            const uint64_t word = pos >> 6, block = pos >> 9;
            uint64_t r = ranks[L.rank0[lv] + block];                                      // bitVector::rank (BooPHF.h:611-624)
            for (uint64_t w = block * 8; w < word; w++) r += __popcll(b[w]);
            r += __popcll(b[word] & ((1ULL << (pos & 63)) - 1));
            return r;
        }
    }
    for (uint64_t i = 0; i < n_final; i++) {
        const uint64_t flo = wide ? final_keys[2 * i] : final_keys[i], fhi = wide ? final_keys[2 * i + 1] : 0;
        if (flo == lo && fhi == hi) return lastrank + i;
    }
    return ~0ULL;
}
__global__ void k_mphf_lookup(MphfLevels L, const uint64_t* __restrict__ bits, const uint64_t* __restrict__ ranks, int wide, const uint64_t* __restrict__ final_keys,
                              uint64_t n_final, uint64_t lastrank, const uint8_t* __restrict__ keys, uint64_t n, uint32_t stride, uint64_t* __restrict__ codes)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t lo, hi; load_key2(keys + i * stride, wide, lo, hi);
        codes[i] = mphf_lookup_dev(L, bits, ranks, wide, final_keys, n_final, lastrank, lo, hi);
    }
}
// MPHFAlgorithm::populate (MPHFAlgorithm.cpp:240-270): map[code(kmer)] = index of the abundance in the discretization table
__constant__ int c_abund_disc[257];
__global__ void k_mphf_populate(MphfLevels L, const uint64_t* __restrict__ bits, const uint64_t* __restrict__ ranks, int wide, const uint64_t* __restrict__ final_keys,
                                uint64_t n_final, uint64_t lastrank, const uint8_t* __restrict__ recs, uint64_t n, uint32_t stride, uint64_t n_keys,
                                uint8_t* __restrict__ map, unsigned long long* __restrict__ stats /* [0] above precision, [1] codes out of bounds */)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t lo, hi; load_key2(recs + i * stride, wide, lo, hi);
        const uint64_t code = mphf_lookup_dev(L, bits, ranks, wide, final_keys, n_final, lastrank, lo, hi);
        if (code >= n_keys) { atomicAdd(&stats[1], 1ULL); continue; }
        const int abundance = *reinterpret_cast<const int32_t*>(recs + i * stride + (wide ? 16 : 8));
        int idx;
        if (abundance >= c_abund_disc[255]) { atomicAdd(&stats[0], 1ULL); idx = 255; }
        else { int a = 0, b = 257; while (a < b) { const int mid = (a + b) >> 1; if (c_abund_disc[mid] <= abundance) a = mid + 1; else b = mid; } idx = a - 1; }
        map[code] = (uint8_t)idx;
    }
}

// exclusive scan of a u64 array (n <= 2^26) with the workgroup-scan kernels below
constexpr int MS_ITEMS = 8, MS_CHUNK = 1024 * MS_ITEMS;
__device__ __forceinline__ void ms_wg_scan(uint64_t tv, uint64_t& excl, uint64_t& total, uint64_t* s_w)
{
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    uint64_t x = tv;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint64_t y = __shfl_up((unsigned long long)x, d, 64); if (lane >= d) x += y; }
    if (lane == 63) s_w[wave] = x;
    __syncthreads();
    uint64_t p = 0; total = 0;
    for (int w = 0; w < 16; w++) { if (w < wave) p += s_w[w]; total += s_w[w]; }
    excl = p + x - tv;
}
__global__ __launch_bounds__(1024) void k_ms_chunks(uint64_t* __restrict__ a, uint64_t n, uint64_t* __restrict__ ca)
{
    __shared__ uint64_t s_w[16];
    const uint64_t i0 = (uint64_t)blockIdx.x * MS_CHUNK + (uint64_t)threadIdx.x * MS_ITEMS;
    uint64_t v[MS_ITEMS], tv = 0;
#pragma unroll
    for (int j = 0; j < MS_ITEMS; j++) { v[j] = i0 + j < n ? a[i0 + j] : 0; tv += v[j]; }
    uint64_t r, tot; ms_wg_scan(tv, r, tot, s_w);
#pragma unroll
    for (int j = 0; j < MS_ITEMS; j++) if (i0 + j < n) { a[i0 + j] = r; r += v[j]; }
    if (threadIdx.x == 0) ca[blockIdx.x] = tot;
}
__global__ __launch_bounds__(1024) void k_ms_totals(uint64_t* __restrict__ ca, uint32_t n_chunks, uint64_t* __restrict__ total)
{
    __shared__ uint64_t s_w[16];
    const uint32_t i0 = threadIdx.x * MS_ITEMS;
    uint64_t v[MS_ITEMS], tv = 0;
#pragma unroll
    for (int j = 0; j < MS_ITEMS; j++) { v[j] = i0 + j < n_chunks ? ca[i0 + j] : 0; tv += v[j]; }
    uint64_t r, tot; ms_wg_scan(tv, r, tot, s_w);
#pragma unroll
    for (int j = 0; j < MS_ITEMS; j++) if (i0 + j < n_chunks) { ca[i0 + j] = r; r += v[j]; }
    if (threadIdx.x == 0) *total = tot;
}
__global__ __launch_bounds__(1024) void k_ms_add(uint64_t* __restrict__ a, uint64_t n, const uint64_t* __restrict__ ca)
{
    const uint64_t o = ca[blockIdx.x];
    const uint64_t i0 = (uint64_t)blockIdx.x * MS_CHUNK + (uint64_t)threadIdx.x * MS_ITEMS;
#pragma unroll
    for (int j = 0; j < MS_ITEMS; j++) if (i0 + j < n) a[i0 + j] += o;
}
// in-place exclusive scan of a[0..n), *d_total = sum: chunks of 8192, the chunk totals scanned the same way (two levels reach 5.5e11 items)
int ms_scan(gkc_ctx* c, uint64_t* a, uint64_t n, uint64_t* d_total, DevBuf& scratch)
{
    const uint64_t n_chunks = (n + MS_CHUNK - 1) / MS_CHUNK;
    GKC_TRY(c->ensure(scratch, (size_t)std::max<uint64_t>(n_chunks, 1) * 8));
    if (n_chunks) hipLaunchKernelGGL(k_ms_chunks, dim3((unsigned)n_chunks), dim3(1024), 0, c->stream, a, n, (uint64_t*)scratch.p);
    if (n_chunks <= (uint64_t)MS_CHUNK) hipLaunchKernelGGL(k_ms_totals, dim3(1), dim3(1024), 0, c->stream, (uint64_t*)scratch.p, (uint32_t)n_chunks, d_total);
    else { DevBuf deeper; int rc = ms_scan(c, (uint64_t*)scratch.p, n_chunks, d_total, deeper); (void)hipStreamSynchronize(c->stream); deeper.release(); if (rc != GKC_OK) return rc; }
    if (n_chunks) hipLaunchKernelGGL(k_ms_add, dim3((unsigned)n_chunks), dim3(1024), 0, c->stream, a, n, (const uint64_t*)scratch.p);
    GKC_HIP(c, hipGetLastError());
    return GKC_OK;
}

}   // namespace

struct gkc_mphf {
    gkc_ctx* ctx; int wide; uint32_t k;
    double gamma; uint64_t nelem, lastbitsetrank;
    MphfLevels L; uint64_t nchar[MPHF_LEVELS], nranks[MPHF_LEVELS]; uint64_t total_words, total_ranks;
    DevBuf bits, ranks, final_keys; uint64_t n_final;
};

// comm != nullptr: the keys are this rank's share of a key set spread over the communicator's ranks (in rank order); every rank builds the
// complete function (level arrays combined per level, see gkc_mphf_build_solid_dist in gkc.h)
static int mphf_build_from_list(gkc_ctx* c, gkc_mphf* m, DevBuf& keysA, uint64_t n_local, gkc_comm* comm = nullptr, bool ordered = false)
{
    // ordered: every level through the atomic / flag / stable-compaction kernels (the next lists stay in key order). Default: large levels of a single-rank build go
    // through the region build (above), whose next lists come out in region order; gkc_mphf_build* rebuild `ordered` in the one case where that order would show.
    const bool regions_env = gkc_tun().mphf_regions;
    const uint64_t regions_min = gkc_tun().mphf_regions_min;      // keys of a level from which it pays (tests lower it)
    uint64_t n = n_local;
    if (comm) {
        std::vector<uint64_t> ns(gkc_comm_world(comm));
        GKC_TRY(gkc_comm_allgather_host(comm, &n_local, 8, ns.data()));
        n = 0; for (uint64_t v : ns) n += v;
        if (n == 0) GKC_FAIL(c, GKC_ERR_ARG, "MPHF of an empty key set (the reference leaves the object unbuilt)");
    }
    const int wide = m->wide; const size_t kb = wide ? 16 : 8;
    m->gamma = 3.0; m->nelem = n;                                                                       // BooPHF.hpp:300 (gamma 3)
    const uint64_t hash_domain = (uint64_t)std::ceil((double)n * m->gamma);                               // BooPHF.h:735
    const double proba = 1.0 - std::pow(((m->gamma * (double)n - 1) / (m->gamma * (double)n)), (double)(n - 1));   // :1024
    uint64_t words = 0, nr = 0;
    for (int i = 0; i < MPHF_LEVELS; i++) {                                                               // :1034-1046
        uint64_t d = (((uint64_t)(hash_domain * std::pow(proba, i)) + 63) / 64) * 64;
        if (d == 0) d = 64;
        m->L.domain[i] = d; m->nchar[i] = 1 + d / 64; m->nranks[i] = (m->nchar[i] + 7) / 8;
        m->L.word0[i] = words; m->L.rank0[i] = nr; words += m->nchar[i]; nr += m->nranks[i];
    }
    m->total_words = words; m->total_ranks = nr;
    GKC_TRY(c->ensure(m->bits, (size_t)words * 8)); GKC_TRY(c->ensure(m->ranks, (size_t)nr * 8));
    GKC_HIP(c, hipMemsetAsync(m->bits.p, 0, (size_t)words * 8, c->stream));
    DevBuf keysB, coll, flag, scratch, d_tot, r_wg, r_off, r_items;
    struct Guard { std::vector<DevBuf*> v; ~Guard() { for (DevBuf* b : v) b->release(); } } guard; guard.v = { &keysB, &coll, &flag, &scratch, &d_tot, &r_wg, &r_off, &r_items };
    GKC_TRY(c->ensure(coll, (size_t)m->nchar[0] * 8)); GKC_TRY(c->ensure(d_tot, 64));
    DevBuf* cur = &keysA; DevBuf* nxt = &keysB;
    bool used_regions = false;
    uint64_t alive = n_local, offset = 0;      // alive: keys of THIS rank still unplaced; n - offset: keys of all ranks still unplaced
    for (int lv = 0; lv < MPHF_LEVELS; lv++) {
        uint64_t* lbits = (uint64_t*)m->bits.p + m->L.word0[lv];
        uint64_t* lranks = (uint64_t*)m->ranks.p + m->L.rank0[lv];
        const uint64_t global_alive = n - offset;
        const uint64_t n_regions64 = (m->L.domain[lv] + (1ull << MR_BITS) - 1) >> MR_BITS;
        const bool by_region = !comm && !ordered && regions_env && lv < MPHF_LEVELS - 1 && alive >= regions_min && n_regions64 <= MR_MAX_REGIONS && alive < (1ull << 32);
        if (by_region) {
            used_regions = true;
            const uint32_t n_regions = (uint32_t)n_regions64;
            const uint32_t n_wgs = (uint32_t)std::min<uint64_t>(MR_WGS, (alive + MR_THREADS - 1) / MR_THREADS);
            const uint64_t chunk = (alive + n_wgs - 1) / n_wgs;
            GKC_TRY(c->ensure(r_wg, (size_t)n_wgs * n_regions * 4)); GKC_TRY(c->ensure(r_off, ((size_t)n_regions + 1) * 8)); GKC_TRY(c->ensure(r_items, (size_t)alive * kb));
            GKC_TRY(c->ensure(*nxt, (size_t)alive * kb));                   // (the survivors are written before their number is known: ~28 % of `alive`)
            static std::once_flag once;
            std::call_once(once, [] {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_mphf_regions<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(MR_MAX_REGIONS * 4));
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_mphf_regions<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(MR_MAX_REGIONS * 4));
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_mphf_region_build), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * MR_WORDS * 4));
            });
            GKC_HIP(c, hipMemsetAsync(coll.p, 0, (size_t)m->nchar[lv] * 8, c->stream));
            GKC_HIP(c, hipMemsetAsync((uint64_t*)d_tot.p + 2, 0, 8, c->stream));
            const bool dbg = gkc_tun().verbose;
            hipEvent_t ev[5]; if (dbg) { for (auto& e : ev) (void)hipEventCreate(&e); (void)hipEventRecord(ev[0], c->stream); }
            hipLaunchKernelGGL((k_mphf_regions<false>), dim3(n_wgs), dim3(MR_THREADS), (size_t)n_regions * 4, c->stream, (const uint64_t*)cur->p, alive, wide, lv, m->L.domain[lv], chunk, n_regions,
                               (uint32_t*)r_wg.p, (const uint64_t*)nullptr, (uint64_t*)nullptr);
            if (dbg) (void)hipEventRecord(ev[1], c->stream);
            hipLaunchKernelGGL(k_mr_wg_prefix, dim3((n_regions + 255) / 256), dim3(256), 0, c->stream, (uint32_t*)r_wg.p, n_wgs, n_regions, (uint64_t*)r_off.p);
            GKC_TRY(ms_scan(c, (uint64_t*)r_off.p, n_regions, (uint64_t*)r_off.p + n_regions, scratch));
            if (dbg) (void)hipEventRecord(ev[2], c->stream);
            hipLaunchKernelGGL((k_mphf_regions<true>), dim3(n_wgs), dim3(MR_THREADS), (size_t)n_regions * 4, c->stream, (const uint64_t*)cur->p, alive, wide, lv, m->L.domain[lv], chunk, n_regions,
                               (uint32_t*)r_wg.p, (const uint64_t*)r_off.p, (uint64_t*)r_items.p);
            if (dbg) (void)hipEventRecord(ev[3], c->stream);
            hipLaunchKernelGGL(k_mphf_region_build, dim3(n_regions), dim3(MR_THREADS), (size_t)2 * MR_WORDS * 4, c->stream, (const uint64_t*)r_items.p, (const uint64_t*)r_off.p, wide, lv, m->L.domain[lv],
                               (uint32_t*)lbits, (uint32_t*)coll.p, (uint64_t)m->nchar[lv] * 2, (uint64_t*)nxt->p, (unsigned long long*)((uint64_t*)d_tot.p + 2));
            GKC_HIP(c, hipGetLastError());
            if (dbg) {
                (void)hipEventRecord(ev[4], c->stream); (void)hipEventSynchronize(ev[4]);
                float t[4]; for (int i = 0; i < 4; i++) (void)hipEventElapsedTime(&t[i], ev[i], ev[i + 1]);
                fprintf(stderr, "[gkc mphf] level %d: %llu keys, %u regions: count %.2f ms, prefix + scan %.2f, scatter %.2f, build %.2f\n", lv, (unsigned long long)alive, n_regions, t[0], t[1], t[2], t[3]);
                for (auto& e : ev) (void)hipEventDestroy(e);
            }
        } else if (lv < MPHF_LEVELS - 1 && global_alive) {
            GKC_HIP(c, hipMemsetAsync(coll.p, 0, (size_t)m->nchar[lv] * 8, c->stream));
            if (alive) {
                const unsigned grid = (unsigned)std::min<uint64_t>((alive + MPHF_THREADS - 1) / MPHF_THREADS, 256 * 32);
                hipLaunchKernelGGL(k_mphf_insert, dim3(grid), dim3(MPHF_THREADS), 0, c->stream, (const uint64_t*)cur->p, alive, wide, lv, m->L.domain[lv], (uint32_t*)lbits, (uint32_t*)coll.p);
            }
            if (comm) {     // the level over all ranks: seen = OR, collided = OR | seen by two ranks; comes back with the collisions cleared
                GKC_TRY(gkc_comm_combine_seen_coll(comm, lbits, (uint64_t*)coll.p, m->nchar[lv], c->stream));
                GKC_HIP(c, hipMemsetAsync(coll.p, 0, (size_t)m->nchar[lv] * 8, c->stream));
            }
        } else GKC_HIP(c, hipMemsetAsync(coll.p, 0, (size_t)m->nchar[lv] * 8, c->stream));
        // clear collisions, block popcounts -> rank samples (exclusive scan + running offset)
        hipLaunchKernelGGL(k_mphf_clear, dim3((unsigned)((m->nranks[lv] + 255) / 256)), dim3(256), 0, c->stream, lbits, (const uint64_t*)coll.p, m->nchar[lv], lranks);
        GKC_TRY(ms_scan(c, lranks, m->nranks[lv], (uint64_t*)d_tot.p, scratch));
        hipLaunchKernelGGL(k_mphf_add_offset, dim3((unsigned)((m->nranks[lv] + 255) / 256)), dim3(256), 0, c->stream, lranks, m->nranks[lv], offset);
        uint64_t placed = 0;
        GKC_HIP(c, hipMemcpyAsync(&placed, d_tot.p, 8, hipMemcpyDeviceToHost, c->stream));
        GKC_HIP(c, hipStreamSynchronize(c->stream));
        offset += placed;
        if (lv == MPHF_LEVELS - 1 || !global_alive) continue;
        // survivors of this level -> next list (stable)
        if (by_region) {                                                  // the survivors are in the next list already
            uint64_t left = 0;
            GKC_HIP(c, hipMemcpyAsync(&left, (uint64_t*)d_tot.p + 2, 8, hipMemcpyDeviceToHost, c->stream));
            GKC_HIP(c, hipStreamSynchronize(c->stream));
            if (left != alive - placed) GKC_FAIL(c, GKC_ERR_HIP, "internal error: MPHF level %d (region build) placed %llu of %llu keys but %llu are left (duplicate keys?)", lv,
                                                 (unsigned long long)placed, (unsigned long long)alive, (unsigned long long)left);
            alive = left; std::swap(cur, nxt);
            continue;
        }
        if (placed == global_alive) { alive = 0; continue; }
        if (!alive) continue;
        GKC_TRY(c->ensure(flag, (size_t)(alive + 1) * 8));
        GKC_TRY(c->ensure(*nxt, (size_t)std::max<uint64_t>(comm ? alive : alive - placed, 1) * kb));     // `placed` counts all ranks' keys: with a communicator only `alive` bounds this rank's survivors
        const unsigned g1 = (unsigned)((alive + MPHF_THREADS - 1) / MPHF_THREADS);
        hipLaunchKernelGGL(k_mphf_flag, dim3(g1), dim3(MPHF_THREADS), 0, c->stream, (const uint64_t*)cur->p, alive, wide, lv, m->L.domain[lv], (const uint32_t*)lbits, (uint64_t*)flag.p);
        GKC_TRY(ms_scan(c, (uint64_t*)flag.p, alive, (uint64_t*)d_tot.p + 1, scratch));
        hipLaunchKernelGGL(k_mphf_compact, dim3(g1), dim3(MPHF_THREADS), 0, c->stream, (const uint64_t*)cur->p, alive, wide, (const uint64_t*)flag.p, (const uint64_t*)d_tot.p + 1, (uint64_t*)nxt->p);
        uint64_t left = 0;
        GKC_HIP(c, hipMemcpyAsync(&left, (uint64_t*)d_tot.p + 1, 8, hipMemcpyDeviceToHost, c->stream));
        GKC_HIP(c, hipStreamSynchronize(c->stream));
        if (!comm && left != alive - placed) GKC_FAIL(c, GKC_ERR_HIP, "internal error: MPHF level %d placed %llu of %llu keys but %llu are left (duplicate keys?)", lv,
                                             (unsigned long long)placed, (unsigned long long)alive, (unsigned long long)left);
        alive = left; std::swap(cur, nxt);
    }
    m->lastbitsetrank = offset;
    // what survived all filtering levels: exact list, codes lastbitsetrank + i in key order (processLevel :896-903)
    if (comm) {     // the survivors of all ranks in rank order (= key order of the whole set); a handful of keys at most
        const int W = gkc_comm_world(comm);
        std::vector<uint64_t> cnts(W);
        GKC_TRY(gkc_comm_allgather_host(comm, &alive, 8, cnts.data()));
        const uint64_t mx = *std::max_element(cnts.begin(), cnts.end());
        uint64_t tot = 0; for (uint64_t v : cnts) tot += v;
        m->n_final = tot;
        if (tot) {
            std::vector<uint8_t> mine((size_t)mx * kb, 0), all((size_t)mx * kb * W), packed((size_t)tot * kb);
            if (alive) { GKC_HIP(c, hipMemcpyAsync(mine.data(), cur->p, (size_t)alive * kb, hipMemcpyDeviceToHost, c->stream)); GKC_HIP(c, hipStreamSynchronize(c->stream)); }
            GKC_TRY(gkc_comm_allgather_host(comm, mine.data(), mine.size(), all.data()));
            size_t o = 0;
            for (int r = 0; r < W; r++) { memcpy(packed.data() + o, all.data() + (size_t)r * mx * kb, (size_t)cnts[r] * kb); o += (size_t)cnts[r] * kb; }
            GKC_TRY(c->ensure(m->final_keys, (size_t)tot * kb));
            GKC_HIP(c, hipMemcpy(m->final_keys.p, packed.data(), (size_t)tot * kb, hipMemcpyHostToDevice));
        }
    } else {
        if (alive && used_regions) return GKC_MPHF_REBUILD_ORDERED;          // keys survived all filtering levels: their codes follow KEY order (see mphf_build_arrays)
        m->n_final = alive;
        if (alive) { GKC_TRY(c->ensure(m->final_keys, (size_t)alive * kb)); GKC_HIP(c, hipMemcpyAsync(m->final_keys.p, cur->p, (size_t)alive * kb, hipMemcpyDeviceToDevice, c->stream)); }
    }
    GKC_HIP(c, hipStreamSynchronize(c->stream));
    if (m->lastbitsetrank + m->n_final != n) GKC_FAIL(c, GKC_ERR_ARG, "MPHF: %llu keys placed out of %llu: the key set holds duplicates", (unsigned long long)(m->lastbitsetrank + m->n_final), (unsigned long long)n);
    return GKC_OK;
}

extern "C" {

void gkc_mphf_destroy(gkc_mphf* m) { if (m) { gkc_ctx* c = m->ctx; m->bits.release(); m->ranks.release(); m->final_keys.release(); delete m; gkc_ctx_child_release(c); } }
uint64_t gkc_mphf_size(const gkc_mphf* m) { return m ? m->nelem : 0; }

static int mphf_build_arrays(gkc_ctx* c, const std::vector<std::pair<const uint8_t*, uint64_t>>& segs, uint32_t stride, uint32_t k, bool on_host, gkc_mphf** out, gkc_comm* comm = nullptr)
{
    uint64_t n = 0; for (auto& s : segs) n += s.second;
    if (n == 0 && !comm) GKC_FAIL(c, GKC_ERR_ARG, "MPHF of an empty key set (the reference leaves the object unbuilt)");
    if (n >= (1ULL << 34)) GKC_FAIL(c, GKC_ERR_ARG, "MPHF: too many keys for one device");
    GKC_HIP(c, hipSetDevice(c->device));
    ScopedTimer tm(c, "mphf_build");
    gkc_mphf* m = new gkc_mphf(); m->ctx = c; m->wide = k > 31; m->k = k; m->n_final = 0; gkc_ctx_child_add(c);
    const size_t kb = m->wide ? 16 : 8;
    DevBuf keys, tmp;
    int rc = GKC_OK;
  for (int attempt = 0; attempt < 2; attempt++) {                        // (a second time only for keys that survived all levels of a region build: ordered rebuild)
    rc = c->ensure(keys, (size_t)std::max<uint64_t>(n, 1) * kb);
    uint64_t done = 0;
    for (size_t i = 0; rc == GKC_OK && i < segs.size(); i++) {
        const uint8_t* src = segs[i].first; const uint64_t ni = segs[i].second;
        if (!ni) continue;
        if (on_host) {
            rc = c->ensure(tmp, (size_t)ni * stride);
            if (rc == GKC_OK && hipMemcpyAsync(tmp.p, src, (size_t)ni * stride, hipMemcpyHostToDevice, c->stream) != hipSuccess) { c->set_error(GKC_ERR_HIP, "H2D copy failed"); rc = GKC_ERR_HIP; }
            src = (const uint8_t*)tmp.p;
        }
        if (rc != GKC_OK) break;
        const unsigned grid = (unsigned)std::min<uint64_t>((ni + 255) / 256, 256 * 32);
        hipLaunchKernelGGL(k_mphf_gather, dim3(grid), dim3(256), 0, c->stream, src, ni, stride, m->wide, (uint64_t*)keys.p + done * (kb / 8));
        if (on_host) (void)hipStreamSynchronize(c->stream);
        done += ni;
    }
    if (rc == GKC_OK) rc = mphf_build_from_list(c, m, keys, n, comm, attempt == 1 || gkc_tun().mphf_ordered);
    if (rc != GKC_MPHF_REBUILD_ORDERED) break;
  }
    (void)hipStreamSynchronize(c->stream);
    keys.release(); tmp.release();
    if (rc != GKC_OK) { gkc_mphf_destroy(m); return rc; }
    *out = m;
    return GKC_OK;
}

int gkc_mphf_build(gkc_ctx* c, const void* keys, uint64_t n, uint32_t stride, uint32_t k, gkc_mphf** out)
{
    gkc_tun_refresh();
    if (!c || !out || (!keys && n)) return GKC_ERR_ARG;
    if (k < 1 || k > 63) GKC_FAIL(c, GKC_ERR_ARG, "k must be in [1,63]");
    const uint32_t need = k > 31 ? 16 : 8;
    if (stride < need || stride % 8) GKC_FAIL(c, GKC_ERR_ARG, "stride %u invalid for k=%u", stride, k);
    std::vector<std::pair<const uint8_t*, uint64_t>> segs{ { (const uint8_t*)keys, n } };
    return mphf_build_arrays(c, segs, stride, k, true, out);
}
int gkc_mphf_build_solid(gkc_ctx* c, gkc_mphf** out)
{
    gkc_tun_refresh();
    if (!c || !out) return GKC_ERR_ARG;
    if (!c->configured) GKC_FAIL(c, GKC_ERR_ARG, "context not configured");
    GKC_TRY(gkc_require_resident(c, "gkc_mphf_build_solid"));
    const uint32_t stride = c->key_words == 1 ? 16 : 32;
    std::vector<std::pair<const uint8_t*, uint64_t>> segs;
    for (const Dataset& D : c->datasets) {                      // iteration order of getSolidKmers(): dataset by dataset, ascending inside
        if (!D.done || !D.n_solid) continue;
        if (!segs.empty() && (const uint8_t*)D.d_counts == segs.back().first + segs.back().second * stride) { segs.back().second += D.n_solid; continue; }
        segs.push_back({ (const uint8_t*)D.d_counts, D.n_solid });
    }
    return mphf_build_arrays(c, segs, stride, c->k, false, out);
}

static void solid_segments(gkc_ctx* c, uint32_t stride, std::vector<std::pair<const uint8_t*, uint64_t>>& segs)
{
    for (const Dataset& D : c->datasets) {                      // iteration order of getSolidKmers(): dataset by dataset, ascending inside
        if (!D.done || !D.n_solid) continue;
        if (!segs.empty() && (const uint8_t*)D.d_counts == segs.back().first + segs.back().second * stride) { segs.back().second += D.n_solid; continue; }
        segs.push_back({ (const uint8_t*)D.d_counts, D.n_solid });
    }
}
int gkc_mphf_build_solid_dist(gkc_ctx* c, gkc_comm* comm, gkc_mphf** out)
{
    gkc_tun_refresh();
    if (!c || !comm || !out) return GKC_ERR_ARG;
    if (!c->configured) GKC_FAIL(c, GKC_ERR_ARG, "context not configured");
    GKC_TRY(gkc_require_resident(c, "gkc_mphf_build_solid_dist"));
    const uint32_t stride = c->key_words == 1 ? 16 : 32;
    std::vector<std::pair<const uint8_t*, uint64_t>> segs;
    solid_segments(c, stride, segs);
    return mphf_build_arrays(c, segs, stride, c->k, false, out, comm);
}

int gkc_mphf_lookup(gkc_mphf* m, const void* keys, uint64_t n, uint32_t stride, uint64_t* codes)
{
    if (!m || (!keys && n) || (!codes && n)) return GKC_ERR_ARG;
    gkc_ctx* c = m->ctx;
    if (!n) return GKC_OK;
    const uint32_t need = m->wide ? 16 : 8;
    if (stride < need || stride % 8) GKC_FAIL(c, GKC_ERR_ARG, "stride %u invalid", stride);
    DevBuf dk, dc; GKC_TRY(c->ensure(dk, (size_t)n * stride));
    int rc = c->ensure(dc, (size_t)n * 8);
    if (rc != GKC_OK) { dk.release(); return rc; }
    hipError_t e = hipMemcpyAsync(dk.p, keys, (size_t)n * stride, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        ScopedTimer tm(c, "mphf_lookup");
        const unsigned grid = (unsigned)std::min<uint64_t>((n + 255) / 256, 256 * 32);
        hipLaunchKernelGGL(k_mphf_lookup, dim3(grid), dim3(256), 0, c->stream, m->L, (const uint64_t*)m->bits.p, (const uint64_t*)m->ranks.p, m->wide, (const uint64_t*)m->final_keys.p,
                           m->n_final, m->lastbitsetrank, (const uint8_t*)dk.p, n, stride, (uint64_t*)dc.p);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(codes, dc.p, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    dk.release(); dc.release();
    if (e != hipSuccess) GKC_FAIL(c, GKC_ERR_HIP, "MPHF lookup failed: %s", hipGetErrorString(e));
    return GKC_OK;
}

// byte stream of boomphf::mphf::save (BooPHF.h:933-958; bitVector::save :627-635)
uint64_t gkc_mphf_save_size(const gkc_mphf* m)
{
    if (!m) return 0;
    uint64_t s = 8 + 4 + 8 + 8;
    for (int i = 0; i < MPHF_LEVELS; i++) s += 8 + 8 + m->nchar[i] * 8 + 8 + m->nranks[i] * 8;
    s += 8 + m->n_final * ((m->wide ? 16 : 8) + 8);
    return s;
}
int gkc_mphf_save(gkc_mphf* m, uint8_t* out, uint64_t cap)
{
    if (!m || !out) return GKC_ERR_ARG;
    gkc_ctx* c = m->ctx;
    if (cap < gkc_mphf_save_size(m)) GKC_FAIL(c, GKC_ERR_CAPACITY, "MPHF stream needs %llu bytes", (unsigned long long)gkc_mphf_save_size(m));
    std::vector<uint64_t> hb(m->total_words), hr(m->total_ranks), hf((size_t)m->n_final * (m->wide ? 2 : 1));
    GKC_HIP(c, hipMemcpyAsync(hb.data(), m->bits.p, hb.size() * 8, hipMemcpyDeviceToHost, c->stream));
    GKC_HIP(c, hipMemcpyAsync(hr.data(), m->ranks.p, hr.size() * 8, hipMemcpyDeviceToHost, c->stream));
    if (m->n_final) GKC_HIP(c, hipMemcpyAsync(hf.data(), m->final_keys.p, hf.size() * 8, hipMemcpyDeviceToHost, c->stream));
    GKC_HIP(c, hipStreamSynchronize(c->stream));
    uint64_t pos = 0;
    auto put = [&](const void* p, size_t nb) { memcpy(out + pos, p, nb); pos += nb; };
    const int nbl = MPHF_LEVELS;
    put(&m->gamma, 8); put(&nbl, 4); put(&m->lastbitsetrank, 8); put(&m->nelem, 8);
    for (int i = 0; i < MPHF_LEVELS; i++) {
        put(&m->L.domain[i], 8); put(&m->nchar[i], 8); put(hb.data() + m->L.word0[i], m->nchar[i] * 8);
        put(&m->nranks[i], 8); put(hr.data() + m->L.rank0[i], m->nranks[i] * 8);
    }
    put(&m->n_final, 8);
    for (uint64_t i = 0; i < m->n_final; i++) { put(hf.data() + i * (m->wide ? 2 : 1), m->wide ? 16 : 8); put(&i, 8); }
    return GKC_OK;
}

// MPHFAlgorithm::populate: abundance map (one byte per key, index into MapMPHF's discretization table) of the context's solid k-mers
static int abundance_map_impl(gkc_mphf* m, gkc_ctx* c, gkc_comm* comm, uint8_t* out, uint64_t cap, uint64_t* nb_above_precision);
int gkc_mphf_abundance_map(gkc_mphf* m, gkc_ctx* c, uint8_t* out, uint64_t cap, uint64_t* nb_above_precision) { return abundance_map_impl(m, c, nullptr, out, cap, nb_above_precision); }
int gkc_mphf_abundance_map_dist(gkc_mphf* m, gkc_ctx* c, gkc_comm* comm, uint8_t* out, uint64_t cap, uint64_t* nb_above_precision)
{
    if (!comm) return GKC_ERR_ARG;
    return abundance_map_impl(m, c, comm, out, cap, nb_above_precision);
}
static int abundance_map_impl(gkc_mphf* m, gkc_ctx* c, gkc_comm* comm, uint8_t* out, uint64_t cap, uint64_t* nb_above_precision)
{
    if (!m || !c || !out) return GKC_ERR_ARG;
    if (cap < m->nelem) GKC_FAIL(c, GKC_ERR_CAPACITY, "abundance map needs %llu bytes", (unsigned long long)m->nelem);
    if ((c->k > 31) != (m->wide != 0)) GKC_FAIL(c, GKC_ERR_ARG, "MPHF key width differs from the context's");
    GKC_TRY(gkc_require_resident(c, "gkc_mphf_abundance_map"));
    static int disc[257]; static bool init = false;
    if (!init) {                                                  // MapMPHF.hpp:96-145
        int total = 0, idx = 1; disc[0] = 0;
        auto run = [&](int cnt, int step) { for (int i = 1; i <= cnt; i++, idx++) { total += step; disc[idx] = total; } };
        run(70, 1); run(15, 2); run(40, 10); run(25, 20); run(40, 100); run(25, 200); run(40, 1000);
        disc[256] = total; init = true;
    }
    GKC_HIP(c, hipSetDevice(c->device));
    GKC_HIP(c, hipMemcpyToSymbol(HIP_SYMBOL(c_abund_disc), disc, sizeof(disc)));
    const size_t map_bytes = ((size_t)m->nelem + 15) / 8 * 8;      // whole 8-byte words (the cross-rank OR works on words)
    DevBuf dmap, dst; GKC_TRY(c->ensure(dmap, map_bytes));
    int rc = c->ensure(dst, 16);
    if (rc != GKC_OK) { dmap.release(); return rc; }
    hipError_t e = hipMemsetAsync(dmap.p, 0, map_bytes, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(dst.p, 0, 16, c->stream);
    const uint32_t stride = c->key_words == 1 ? 16 : 32;
    {   ScopedTimer tm(c, "mphf_populate");
        std::vector<std::pair<const uint8_t*, uint64_t>> segs;        // datasets of one Stage-B batch are one array
        for (const Dataset& D : c->datasets) {
            if (!D.done || !D.n_solid) continue;
            if (!segs.empty() && (const uint8_t*)D.d_counts == segs.back().first + segs.back().second * stride) { segs.back().second += D.n_solid; continue; }
            segs.push_back({ (const uint8_t*)D.d_counts, D.n_solid });
        }
        for (auto& sg : segs) {
            if (e != hipSuccess) break;
            const unsigned grid = (unsigned)std::min<uint64_t>((sg.second + 255) / 256, 256 * 32);
            hipLaunchKernelGGL(k_mphf_populate, dim3(grid), dim3(256), 0, c->stream, m->L, (const uint64_t*)m->bits.p, (const uint64_t*)m->ranks.p, m->wide, (const uint64_t*)m->final_keys.p,
                               m->n_final, m->lastbitsetrank, sg.first, sg.second, stride, m->nelem, (uint8_t*)dmap.p, (unsigned long long*)dst.p);
            e = hipGetLastError();
        }
    }
    unsigned long long st[2] = {0, 0};
    if (e == hipSuccess && comm) {          // every cell is written by exactly one rank (zero elsewhere): OR over the ranks = the whole map
        rc = gkc_comm_allreduce_or_words(comm, (uint64_t*)dmap.p, (uint64_t)(map_bytes / 8), c->stream);
        if (rc != GKC_OK) { dmap.release(); dst.release(); return rc; }
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, dmap.p, (size_t)m->nelem, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(st, dst.p, 16, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    dmap.release(); dst.release();
    if (e != hipSuccess) GKC_FAIL(c, GKC_ERR_HIP, "abundance map failed: %s", hipGetErrorString(e));
    if (!comm && st[1]) GKC_FAIL(c, GKC_ERR_ARG, "MPHF check: value out of bounds (%llu k-mers are not keys of this MPHF)", st[1]);      // MPHFAlgorithm.cpp:247
    if (comm) {                                                    // (several ranks: the counters are summed first, so that all ranks fail together)                                                    // counters over all ranks
        std::vector<unsigned long long> alls((size_t)2 * gkc_comm_world(comm));
        GKC_TRY(gkc_comm_allgather_host(comm, st, 16, alls.data()));
        st[0] = st[1] = 0; for (size_t i = 0; i < alls.size(); i += 2) { st[0] += alls[i]; st[1] += alls[i + 1]; }
        if (st[1]) GKC_FAIL(c, GKC_ERR_ARG, "MPHF check: value out of bounds (%llu k-mers are not keys of this MPHF)", st[1]);
    }
    if (nb_above_precision) *nb_above_precision = st[0];
    return GKC_OK;
}

}   // extern "C"
