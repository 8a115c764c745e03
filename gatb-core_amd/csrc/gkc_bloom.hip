// gkc_bloom.hip — Bloom filter of solid k-mers on gfx950 (insert / contains / contains8), bit-identical arrays.
//
// Replaces (reference, /root/reference/gatb-core/src/gatb/tools/collections/impl/Bloom.hpp):
//   C1 HashFunctors seeds + hash1                 :59-98   (+ tools/math/LargeInt1.pri:157-170, LargeInt2.pri:200-206)
//      simplehash16 (3-term 64-bit / 2-term 128-bit)        tools/math/LargeInt1.pri:190-211, NativeInt64.hpp:210-221
//   C2 BloomSynchronized::insert / BloomContainer::contains :394-412, 211-234   ("basic")
//   C3 BloomCacheCoherent::insert/contains                   :437-490            ("cache")
//   C4 BloomNeighborCoherent::insert/contains/contains4/8    :555-828            ("neighbor")
// Bits are set with 32-bit atomic OR on the little-endian word holding the byte (same bytes as __sync_fetch_and_or on
// the u8 array). One thread per k-mer, grid-stride; the bitset (~11 bits per solid k-mer) is the only HBM traffic.
#include "gkc_common.hpp"
#include "gkc_device.hpp"

__constant__ uint64_t c_random_values[256] = {
#include "../../include/gkc_random_values.inc"
};

struct gkc_bloom {
    gkc_ctx* ctx;
    int kind; uint32_t nb_hash, k; int wide;
    uint64_t tai, nchar, reduced_tai; int pow2;
    uint64_t seeds[10];
    DevBuf bits;
};

struct BloomParams {
    uint32_t* words; int kind; uint32_t nb_hash, k; int wide, pow2;
    uint64_t tai, reduced_tai; uint64_t seeds[10];
};

__device__ __forceinline__ uint64_t simplehash16_dev(uint64_t key_lo, int shift, int wide)
{
    uint64_t in = key_lo >> shift;
    uint64_t r = c_random_values[in & 255];
    in >>= 8;
    r ^= c_random_values[in & 255];
    if (!wide) r ^= c_random_values[key_lo & 255];      // LargeInt<1> adds the low byte; NativeInt64/LargeInt<2> do not
    return r;
}
__device__ __forceinline__ uint64_t hash1_dev(u128 x, uint64_t seed, int wide)
{
    return wide ? (hash64((uint64_t)(x >> 64), seed) ^ hash64((uint64_t)x, seed)) : hash64((uint64_t)x, seed);
}
__device__ __forceinline__ u128 load_key(const uint8_t* p, int wide)
{
    uint64_t lo = *reinterpret_cast<const uint64_t*>(p);
    uint64_t hi = wide ? *reinterpret_cast<const uint64_t*>(p + 8) : 0;
    return ((u128)hi << 64) | lo;
}
__device__ __forceinline__ u128 kmask128(uint32_t k) { return (((u128)1) << (2 * k)) - 1; }

__device__ const uint32_t d_cano2[16] = { 0, 1, 2, 3, 4, 5, 3, 7, 8, 9, 0, 4, 9, 13, 1, 5 };

// canonical (k-2)-mer core + base bit position of the neighbor-coherent layout
__device__ __forceinline__ void neighbor_root(const BloomParams& B, u128 x, u128& core, uint64_t& racine)
{
    const uint32_t k = B.k;
    core = (x >> 2) & kmask128(k - 2);
    u128 rv = revcomp128(core, k - 2);
    if (rv < core) core = rv;
    racine = hash1_dev(core, B.seeds[0], B.wide) % B.reduced_tai;
}

// bit positions of item x -> pos[0..nb_hash)
__device__ __forceinline__ void bloom_positions(const BloomParams& B, u128 x, uint64_t* pos)
{
    if (B.kind == 0) {
        for (uint32_t i = 0; i < B.nb_hash; i++) {
            uint64_t h = hash1_dev(x, B.seeds[i], B.wide);
            pos[i] = B.pow2 ? (h & B.tai) : (h % B.tai);
        }
    } else if (B.kind == 1) {
        uint64_t h0 = hash1_dev(x, B.seeds[0], B.wide) % B.reduced_tai;
        pos[0] = h0;
        for (uint32_t i = 1; i < B.nb_hash; i++) pos[i] = h0 + (simplehash16_dev((uint64_t)x, (int)i, B.wide) & 4095);
    } else {
        const uint32_t k = B.k;
        const uint32_t suffix = (uint32_t)x & 3u, prefix = ((uint32_t)(x >> (2 * (k - 1))) & 3u) << 2;
        u128 core; uint64_t racine;
        neighbor_root(B, x, core, racine);
        uint64_t h0 = racine + d_cano2[(prefix + suffix) & 15];
        pos[0] = h0;
        for (uint32_t i = 1; i < B.nb_hash; i++) pos[i] = h0 + (simplehash16_dev((uint64_t)core, (int)i, B.wide) & 4095);
    }
}
__device__ __forceinline__ bool test_bit(const uint32_t* w, uint64_t h) { return (w[h >> 5] >> (h & 31)) & 1u; }

__global__ void k_bloom_insert(BloomParams B, const uint8_t* __restrict__ keys, uint64_t n, uint32_t stride)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        u128 x = load_key(keys + i * stride, B.wide);
        uint64_t pos[20];
        bloom_positions(B, x, pos);
        for (uint32_t j = 0; j < B.nb_hash; j++) atomicOr(&B.words[pos[j] >> 5], 1u << (pos[j] & 31));
    }
}
__global__ void k_bloom_contains(BloomParams B, const uint8_t* __restrict__ keys, uint64_t n, uint32_t stride, uint8_t* __restrict__ out)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        u128 x = load_key(keys + i * stride, B.wide);
        uint64_t pos[20];
        bloom_positions(B, x, pos);
        bool ok = true;
        for (uint32_t j = 0; j < B.nb_hash; j++) ok = ok && test_bit(B.words, pos[j]);
        out[i] = ok;
    }
}
// contains8 (Bloom.hpp:645-811): bits 0-3 = right extensions ((x<<2)|j)&mask, bits 4-7 = left extensions (x>>2)|(j<<2(k-1)),
// j = A,C,T,G. The four neighbours of one side share the canonical (k-2)-mer core, hence one hash1 and one block.
__global__ void k_bloom_contains8(BloomParams B, const uint8_t* __restrict__ keys, uint64_t n, uint32_t stride, uint8_t* __restrict__ out)
{
    const uint32_t k = B.k;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        u128 x = load_key(keys + i * stride, B.wide);
        uint32_t res = 0;
        for (int side = 0; side < 2; side++) {
            const u128 elem = side == 0 ? ((x << 2) & kmask128(k)) : (x >> 2);
            u128 core; uint64_t racine;
            neighbor_root(B, elem, core, racine);
            uint64_t tab[20];
            for (uint32_t h = 1; h < B.nb_hash; h++) tab[h] = simplehash16_dev((uint64_t)core, (int)h, B.wide) & 4095;
            for (uint32_t j = 0; j < 4; j++) {
                uint32_t pre, suf;
                if (side == 0) { pre = ((uint32_t)(elem >> (2 * (k - 1))) & 3u) << 2; suf = j; }
                else { pre = j << 2; suf = (uint32_t)elem & 3u; }
                const uint64_t h0 = racine + d_cano2[(pre + suf) & 15];
                bool ok = test_bit(B.words, h0);
                for (uint32_t h = 1; ok && h < B.nb_hash; h++) ok = test_bit(B.words, h0 + tab[h]);
                res |= (uint32_t)ok << (4 * side + j);
            }
        }
        out[i] = (uint8_t)res;
    }
}

// ------------------------------------------------------------------------------------------------ region build (cache / neighbor kinds)
// The atomic kernel above is bound by the device's global-atomic rate (~2e10 /s: 7 bits x 5.8e8 k-mers = 190 ms). In the two
// block-coherent kinds every bit of an item lies in [h0, h0 + 4096) — so the items are first bucketed by REGION of h0 (2^20 bits of the
// array = 128 KB), then one workgroup per region builds its piece of the array in LDS (LDS atomics) and ORs it into HBM once:
// plain read-modify-write for the words only this region can touch, atomic OR for the two 4 KB-bit fringes it shares with its
// neighbours. Same bits, any order: the array stays byte-identical to the reference's.
constexpr uint32_t BR_BITS = 20, BR_WORDS = 1u << (BR_BITS - 5), BR_FRINGE_WORDS = (4096 + 64) / 32 + 1;      // 131 words may spill over
constexpr uint32_t BR_MAX_REGIONS = 16384, BR_WGS = 1024, BR_THREADS = 1024;
struct BloomItem { uint64_t key64; uint32_t rel; uint32_t pad; };                 // what the build needs: hash input of the offsets + h0 inside the region
struct BSeg { const uint8_t* p; uint64_t n, first; };
struct BSegTable { BSeg s[16]; uint32_t n; uint32_t stride; uint64_t total; };

__device__ __forceinline__ void bloom_root(const BloomParams& B, u128 x, uint64_t& h0, uint64_t& key64)
{
    if (B.kind == 1) { h0 = hash1_dev(x, B.seeds[0], B.wide) % B.reduced_tai; key64 = (uint64_t)x; }
    else {
        const uint32_t k = B.k;
        const uint32_t suffix = (uint32_t)x & 3u, prefix = ((uint32_t)(x >> (2 * (k - 1))) & 3u) << 2;
        u128 core; uint64_t racine;
        neighbor_root(B, x, core, racine);
        h0 = racine + d_cano2[(prefix + suffix) & 15]; key64 = (uint64_t)core;
    }
}
__device__ __forceinline__ const uint8_t* bseg_item(const BSegTable& T, uint64_t g)
{
    uint32_t i = 0;
    while (i + 1 < T.n && g >= T.s[i + 1].first) i++;
    return T.s[i].p + (g - T.s[i].first) * T.stride;
}
// pass 1 / 2 over the items with a STATIC item -> workgroup assignment (identical in both launches): LDS histogram of regions, then LDS cursors
template <bool SCATTER>
__global__ __launch_bounds__(BR_THREADS) void k_bloom_regions(BloomParams B, BSegTable T, uint64_t chunk, uint32_t n_regions, uint32_t* __restrict__ wg_cnt,
                                                               const uint32_t* __restrict__ region_off, BloomItem* __restrict__ items)
{
    extern __shared__ uint32_t s_r[];                              // [n_regions] count / cursor
    for (uint32_t r = threadIdx.x; r < n_regions; r += BR_THREADS)
        s_r[r] = SCATTER ? region_off[r] + wg_cnt[(uint64_t)blockIdx.x * n_regions + r] : 0u;
    __syncthreads();
    const uint64_t i0 = (uint64_t)blockIdx.x * chunk, i1 = min(T.total, i0 + chunk);
    for (uint64_t g = i0 + threadIdx.x; g < i1; g += BR_THREADS) {
        const u128 x = load_key(bseg_item(T, g), B.wide);
        uint64_t h0, key64; bloom_root(B, x, h0, key64);
        const uint32_t slot = atomicAdd(&s_r[(uint32_t)(h0 >> BR_BITS)], 1u);
        if (SCATTER) { BloomItem it; it.key64 = key64; it.rel = (uint32_t)(h0 & ((1u << BR_BITS) - 1)); it.pad = 0; items[slot] = it; }
    }
    if (!SCATTER) {
        __syncthreads();
        for (uint32_t r = threadIdx.x; r < n_regions; r += BR_THREADS) wg_cnt[(uint64_t)blockIdx.x * n_regions + r] = s_r[r];
    }
}
// basic kind: the nb_hash positions of an item are independent -> every POSITION is bucketed by region (4 bytes each)
template <bool SCATTER>
__global__ __launch_bounds__(BR_THREADS) void k_bloom_regions_basic(BloomParams B, BSegTable T, uint64_t chunk, uint32_t n_regions, uint32_t* __restrict__ wg_cnt,
                                                                     const uint32_t* __restrict__ region_off, uint32_t* __restrict__ rels)
{
    extern __shared__ uint32_t s_r[];
    for (uint32_t r = threadIdx.x; r < n_regions; r += BR_THREADS)
        s_r[r] = SCATTER ? region_off[r] + wg_cnt[(uint64_t)blockIdx.x * n_regions + r] : 0u;
    __syncthreads();
    const uint64_t i0 = (uint64_t)blockIdx.x * chunk, i1 = min(T.total, i0 + chunk);
    for (uint64_t g = i0 + threadIdx.x; g < i1; g += BR_THREADS) {
        const u128 x = load_key(bseg_item(T, g), B.wide);
        for (uint32_t j = 0; j < B.nb_hash; j++) {
            const uint64_t h = hash1_dev(x, B.seeds[j], B.wide);
            const uint64_t pos = B.pow2 ? (h & B.tai) : (h % B.tai);
            const uint32_t slot = atomicAdd(&s_r[(uint32_t)(pos >> BR_BITS)], 1u);
            if (SCATTER) rels[slot] = (uint32_t)(pos & ((1u << BR_BITS) - 1));
        }
    }
    if (!SCATTER) {
        __syncthreads();
        for (uint32_t r = threadIdx.x; r < n_regions; r += BR_THREADS) wg_cnt[(uint64_t)blockIdx.x * n_regions + r] = s_r[r];
    }
}
__global__ __launch_bounds__(BR_THREADS) void k_bloom_region_build_basic(BloomParams B, const uint32_t* __restrict__ rels, const uint32_t* __restrict__ region_off)
{
    extern __shared__ uint32_t s_img[];                            // [BR_WORDS]: positions never leave their region
    for (uint32_t i = threadIdx.x; i < BR_WORDS; i += BR_THREADS) s_img[i] = 0;
    __syncthreads();
    const uint32_t r = blockIdx.x, i0 = region_off[r], i1 = region_off[r + 1];
    for (uint32_t i = i0 + threadIdx.x; i < i1; i += BR_THREADS) { const uint32_t h = rels[i]; atomicOr(&s_img[h >> 5], 1u << (h & 31)); }
    __syncthreads();
    uint32_t* g = B.words + (uint64_t)r * BR_WORDS;
    for (uint32_t i = threadIdx.x; i < BR_WORDS; i += BR_THREADS) { const uint32_t v = s_img[i]; if (v) g[i] |= v; }
}

// per region: exclusive prefix of the workgroup counts (in place) and the region total
__global__ void k_bloom_wg_prefix(uint32_t* __restrict__ wg_cnt, uint32_t n_wgs, uint32_t n_regions, uint32_t* __restrict__ region_tot)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_regions) return;
    uint32_t run = 0;
    for (uint32_t w = 0; w < n_wgs; w++) { const uint32_t t = wg_cnt[(uint64_t)w * n_regions + r]; wg_cnt[(uint64_t)w * n_regions + r] = run; run += t; }
    region_tot[r] = run;
}
__global__ __launch_bounds__(1024) void k_bloom_region_scan(const uint32_t* __restrict__ tot, uint32_t n_regions, uint32_t* __restrict__ off)
{   // exclusive scan of <= 16384 totals by one workgroup (16 per thread); off[n_regions] = sum
    __shared__ uint32_t s_w[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    uint32_t v[16], tv = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) { const uint32_t i = t * 16 + j; v[j] = i < n_regions ? tot[i] : 0; tv += v[j]; }
    uint32_t x = tv;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
    if (lane == 63) s_w[wave] = x;
    __syncthreads();
    uint32_t p = 0, total = 0;
    for (int w = 0; w < 16; w++) { if (w < wave) p += s_w[w]; total += s_w[w]; }
    uint32_t r = p + x - tv;
#pragma unroll
    for (int j = 0; j < 16; j++) { const uint32_t i = t * 16 + j; if (i < n_regions) off[i] = r; r += v[j]; }
    if (t == 0) off[n_regions] = total;
}
__global__ __launch_bounds__(BR_THREADS) void k_bloom_region_build(BloomParams B, const BloomItem* __restrict__ items, const uint32_t* __restrict__ region_off)
{
    extern __shared__ uint32_t s_img[];                            // [BR_WORDS + BR_FRINGE_WORDS] this region's bits + what spills into the next one
    constexpr uint32_t NW = BR_WORDS + BR_FRINGE_WORDS;
    for (uint32_t i = threadIdx.x; i < NW; i += BR_THREADS) s_img[i] = 0;
    __syncthreads();
    const uint32_t r = blockIdx.x, i0 = region_off[r], i1 = region_off[r + 1];
    for (uint32_t i = i0 + threadIdx.x; i < i1; i += BR_THREADS) {
        const BloomItem it = items[i];
        atomicOr(&s_img[it.rel >> 5], 1u << (it.rel & 31));
        for (uint32_t j = 1; j < B.nb_hash; j++) {
            const uint32_t h = it.rel + (uint32_t)(simplehash16_dev(it.key64, (int)j, B.wide) & 4095);
            atomicOr(&s_img[h >> 5], 1u << (h & 31));
        }
    }
    __syncthreads();
    uint32_t* g = B.words + (uint64_t)r * BR_WORDS;
    for (uint32_t i = threadIdx.x; i < NW; i += BR_THREADS) {
        const uint32_t v = s_img[i];
        if (!v) continue;
        if (i < BR_FRINGE_WORDS || i >= BR_WORDS) atomicOr(&g[i], v);            // words the neighbouring regions' workgroups may touch too
        else g[i] |= v;
    }
}


// ------------------------------------------------------------------------------------------------ contains8 by region (round 5)
// k_bloom_contains8 gathers: per k-mer 2 sides x (7 dependent bit tests of the present neighbour + ~2 of each absent one) = ~25 scattered 4-byte reads of an array
// far beyond the caches — 5.8e8 solid k-mers: 273 ms against 28 ms for their insert. The insert's trick works for the query as well: the eight neighbours of a k-mer
// are two groups of four that share a canonical (k-2)-mer core, hence one root position each, and every bit of a group lies in [root, root + 16 + 4096): a QUERY ITEM per
// (k-mer, side) is bucketed by the 2^20-bit region of its root (same count / prefix / scatter passes as the insert), one workgroup per region copies its 128 KB of the
// array (+ the fringe) into LDS with coalesced loads, answers its items from there, and writes each item's four answers as one byte at (side, k-mer index); a last pass puts
// the two nibbles of a k-mer together. Replaces the same reference loop (DebloomMinimizerAlgorithm.cpp:201: contains8 of every solid k-mer; Bloom.hpp:645-811).
struct BloomQ { uint64_t key64; uint32_t rel; uint32_t idx; };      // core (hash input of the offsets); root inside the region | the side's own end nucleotide << 20; 2 * k-mer index + side
template <bool SCATTER>
__global__ __launch_bounds__(BR_THREADS) void k_bloom_q_regions(BloomParams B, BSegTable T, uint64_t chunk, uint32_t n_regions, uint32_t* __restrict__ wg_cnt,
                                                                 const uint32_t* __restrict__ region_off, BloomQ* __restrict__ items)
{
    extern __shared__ uint32_t s_r[];                              // [n_regions] count / cursor
    for (uint32_t r = threadIdx.x; r < n_regions; r += BR_THREADS)
        s_r[r] = SCATTER ? region_off[r] + wg_cnt[(uint64_t)blockIdx.x * n_regions + r] : 0u;
    __syncthreads();
    const uint32_t k = B.k;
    const uint64_t i0 = (uint64_t)blockIdx.x * chunk, i1 = min(T.total, i0 + chunk);
    for (uint64_t g = i0 + threadIdx.x; g < i1; g += BR_THREADS) {
        const u128 x = load_key(bseg_item(T, g), B.wide);
#pragma unroll
        for (int side = 0; side < 2; side++) {
            const u128 elem = side == 0 ? ((x << 2) & kmask128(k)) : (x >> 2);      // (the neighbour without its new nucleotide: Bloom.hpp:660, :735)
            u128 core; uint64_t racine;
            neighbor_root(B, elem, core, racine);
            const uint32_t slot = atomicAdd(&s_r[(uint32_t)(racine >> BR_BITS)], 1u);
            if (SCATTER) {
                const uint32_t nt = side == 0 ? ((uint32_t)(elem >> (2 * (k - 1))) & 3u) : ((uint32_t)elem & 3u);
                BloomQ it; it.key64 = (uint64_t)core; it.rel = (uint32_t)(racine & ((1u << BR_BITS) - 1)) | (nt << BR_BITS); it.idx = (uint32_t)(2 * g + side);
                items[slot] = it;
            }
        }
    }
    if (!SCATTER) {
        __syncthreads();
        for (uint32_t r = threadIdx.x; r < n_regions; r += BR_THREADS) wg_cnt[(uint64_t)blockIdx.x * n_regions + r] = s_r[r];
    }
}
__global__ __launch_bounds__(BR_THREADS) void k_bloom_region_query8(BloomParams B, uint64_t n_words, const BloomQ* __restrict__ items, const uint32_t* __restrict__ region_off,
                                                                     uint8_t* __restrict__ sides /* [2][total] */, uint64_t total)
{
    extern __shared__ uint32_t s_img[];                            // [BR_WORDS + BR_FRINGE_WORDS] this region's bits + what the offsets reach of the next one
    constexpr uint32_t NW = BR_WORDS + BR_FRINGE_WORDS;
    const uint32_t r = blockIdx.x, i0 = region_off[r], i1 = region_off[r + 1];
    if (i0 == i1) return;
    const uint64_t w0 = (uint64_t)r * BR_WORDS;
    for (uint32_t i = threadIdx.x; i < NW; i += BR_THREADS) s_img[i] = w0 + i < n_words ? B.words[w0 + i] : 0u;
    __syncthreads();
    for (uint32_t i = i0 + threadIdx.x; i < i1; i += BR_THREADS) {
        const BloomQ it = items[i];
        const uint32_t rel = it.rel & ((1u << BR_BITS) - 1), nt = it.rel >> BR_BITS, side = it.idx & 1u;
        uint32_t tab[10];
        for (uint32_t h = 1; h < B.nb_hash; h++) tab[h] = (uint32_t)(simplehash16_dev(it.key64, (int)h, B.wide) & 4095);
        uint32_t res = 0;
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) {
            const uint32_t pre = side == 0 ? nt << 2 : j << 2, suf = side == 0 ? j : nt;
            const uint32_t h0 = rel + d_cano2[(pre + suf) & 15];
            bool ok = (s_img[h0 >> 5] >> (h0 & 31)) & 1u;
            for (uint32_t h = 1; ok && h < B.nb_hash; h++) { const uint32_t q = h0 + tab[h]; ok = (s_img[q >> 5] >> (q & 31)) & 1u; }
            res |= (uint32_t)ok << j;
        }
        sides[(uint64_t)side * total + (it.idx >> 1)] = (uint8_t)res;
    }
}
__global__ void k_bloom_join_sides(const uint8_t* __restrict__ sides, uint64_t total, uint8_t* __restrict__ out)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) out[i] = (uint8_t)(sides[i] | (sides[total + i] << 4));
}

// ------------------------------------------------------------------------------------------------ C-ABI
static BloomParams params_of(const gkc_bloom* b)
{
    BloomParams P{};
    P.words = (uint32_t*)b->bits.p; P.kind = b->kind; P.nb_hash = b->nb_hash; P.k = b->k; P.wide = b->wide; P.pow2 = b->pow2;
    P.tai = b->tai; P.reduced_tai = b->reduced_tai; memcpy(P.seeds, b->seeds, sizeof(P.seeds));
    return P;
}

extern "C" {

int gkc_bloom_create(gkc_ctx* c, int kind, uint64_t tai_bits, uint32_t nb_hash, uint32_t k, gkc_bloom** out)
{
    gkc_tun_refresh();
    if (!c || !out) return GKC_ERR_ARG;
    if (kind < 0 || kind > 2) GKC_FAIL(c, GKC_ERR_ARG, "bloom kind must be 0 (basic), 1 (cache) or 2 (neighbor)");
    if (nb_hash < 1 || nb_hash > 10) GKC_FAIL(c, GKC_ERR_ARG, "nb_hash must be in [1,10] (HashFunctors holds 10 seeds, Bloom.hpp:94)");
    if (k < 3 || k > 63) GKC_FAIL(c, GKC_ERR_ARG, "k must be in [3,63]");
    if (tai_bits == 0) GKC_FAIL(c, GKC_ERR_ARG, "bloom size must be > 0");
    gkc_bloom* b = new gkc_bloom();
    b->ctx = c; b->kind = kind; b->nb_hash = nb_hash; b->k = k; b->wide = k > 31;
    uint64_t tai = tai_bits;
    if (kind != 0) tai += 2 * 4096;                        // BloomCacheCoherent ctor (Bloom.hpp:437-441)
    b->nchar = 1 + tai / 8;                                // BloomContainer ctor (Bloom.hpp:185-198)
    b->pow2 = (tai && !(tai & (tai - 1)));
    if (b->pow2) tai--;
    b->tai = tai;
    b->reduced_tai = kind != 0 ? tai - 2 * 4096 : tai;
    static const uint64_t rbase[10] = {
        0xAAAAAAAA55555555ULL, 0x33333333CCCCCCCCULL, 0x6666666699999999ULL, 0xB5B5B5B54B4B4B4BULL, 0xAA55AA5555335533ULL,
        0x33CC33CCCC66CC66ULL, 0x6699669999B599B5ULL, 0xB54BB54B4BAA4BAAULL, 0xAA33AA3355CC55CCULL, 0x33663366CC99CC99ULL };
    for (int i = 0; i < 10; i++) b->seeds[i] = rbase[i];
    for (int i = 0; i < 10; i++) b->seeds[i] = b->seeds[i] * b->seeds[(i + 3) % 10] + 0;   // user_seed = 0, in place (Bloom.hpp:80-91)
    const size_t bytes = (size_t)((b->nchar + 3) / 4 * 4 + 8);
    int rc = c->ensure(b->bits, bytes);
    if (rc != GKC_OK) { delete b; return rc; }
    if (hipMemsetAsync(b->bits.p, 0, bytes, c->stream) != hipSuccess) { b->bits.release(); delete b; GKC_FAIL(c, GKC_ERR_HIP, "memset failed"); }
    gkc_ctx_child_add(c);
    *out = b;
    return GKC_OK;
}
void gkc_bloom_destroy(gkc_bloom* b) { if (b) { gkc_ctx* c = b->ctx; b->bits.release(); delete b; gkc_ctx_child_release(c); } }
int gkc_bloom_allreduce_or(gkc_bloom* b, gkc_comm* m)
{
    if (!b || !m) return GKC_ERR_ARG;
    gkc_ctx* c = b->ctx;
    GKC_HIP(c, hipSetDevice(c->device));
    ScopedTimer tm(c, "bloom_allreduce");
    // whole 8-byte words of the allocation ((nchar + 3) / 4 * 4 + 8 bytes, zero beyond nchar): covers every byte of the array
    GKC_TRY(gkc_comm_allreduce_or_words(m, (uint64_t*)b->bits.p, (uint64_t)(b->bits.bytes / 8), c->stream));
    GKC_HIP(c, hipStreamSynchronize(c->stream));
    return GKC_OK;
}
uint64_t gkc_bloom_nbytes(const gkc_bloom* b) { return b ? b->nchar : 0; }
uint64_t gkc_bloom_bitsize(const gkc_bloom* b) { return b ? (b->kind == 0 ? b->tai : b->reduced_tai) : 0; }

static int check_stride(gkc_bloom* b, uint32_t stride)
{
    const uint32_t need = b->wide ? 16 : 8;
    if (stride < need || (stride % 8) != 0) { b->ctx->set_error(GKC_ERR_ARG, "stride %u invalid for k=%u (need a multiple of 8, >= %u)", stride, b->k, need); return GKC_ERR_ARG; }
    return GKC_OK;
}

// inserts the items of up to 16 device arrays; block-coherent kinds go through the region build when the array has <= BR_MAX_REGIONS regions
static int bloom_insert_arrays(gkc_bloom* b, const BSeg* segs, uint32_t n_segs, uint32_t stride)
{
    gkc_ctx* c = b->ctx;
    uint64_t total = 0; for (uint32_t i = 0; i < n_segs; i++) total += segs[i].n;
    if (!total) return GKC_OK;
    ScopedTimer tm(c, "bloom_insert");
    const uint64_t n_bits = b->tai + 1;
    const uint32_t n_regions = (uint32_t)std::min<uint64_t>((n_bits + (1u << BR_BITS) - 1) >> BR_BITS, 0xffffffffu);
    const uint64_t n_virtual = b->kind == 0 ? total * b->nb_hash : total;        // basic: one bucketed entry per position
    const bool regions = n_regions <= BR_MAX_REGIONS && n_virtual < (1ULL << 32) && n_segs <= 16 && !gkc_tun().bloom_atomic;
    if (!regions) {
        for (uint32_t i = 0; i < n_segs; i++) if (segs[i].n) {
            const unsigned grid = (unsigned)std::min<uint64_t>((segs[i].n + 255) / 256, 256 * 16);
            hipLaunchKernelGGL(k_bloom_insert, dim3(grid), dim3(256), 0, c->stream, params_of(b), segs[i].p, segs[i].n, stride);
        }
        GKC_HIP(c, hipGetLastError());
        return GKC_OK;
    }
    BSegTable T{}; T.n = n_segs; T.stride = stride; T.total = total;
    { uint64_t first = 0; for (uint32_t i = 0; i < n_segs; i++) { T.s[i] = segs[i]; T.s[i].first = first; first += segs[i].n; } }
    const uint32_t n_wgs = (uint32_t)std::min<uint64_t>(BR_WGS, (total + BR_THREADS - 1) / BR_THREADS);
    const uint64_t chunk = (total + n_wgs - 1) / n_wgs;
    DevBuf d_wg, d_tot, d_off, d_items;
    struct Guard { DevBuf *a, *b2, *c2, *d; ~Guard() { a->release(); b2->release(); c2->release(); d->release(); } } guard{&d_wg, &d_tot, &d_off, &d_items};
    GKC_TRY(c->ensure(d_wg, (size_t)n_wgs * n_regions * 4)); GKC_TRY(c->ensure(d_tot, (size_t)n_regions * 4)); GKC_TRY(c->ensure(d_off, ((size_t)n_regions + 1) * 4));
    GKC_TRY(c->ensure(d_items, b->kind == 0 ? (size_t)n_virtual * 4 : (size_t)total * sizeof(BloomItem)));
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_bloom_regions<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(BR_MAX_REGIONS * 4));
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_bloom_regions<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(BR_MAX_REGIONS * 4));
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_bloom_region_build), hipFuncAttributeMaxDynamicSharedMemorySize, (int)((BR_WORDS + BR_FRINGE_WORDS) * 4));
        attr_set = true;
    }
    const BloomParams P = params_of(b);
    if (b->kind == 0) {
        static bool attr_basic = false;
        if (!attr_basic) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_bloom_regions_basic<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(BR_MAX_REGIONS * 4));
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_bloom_regions_basic<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(BR_MAX_REGIONS * 4));
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_bloom_region_build_basic), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(BR_WORDS * 4));
            attr_basic = true;
        }
        hipLaunchKernelGGL((k_bloom_regions_basic<false>), dim3(n_wgs), dim3(BR_THREADS), (size_t)n_regions * 4, c->stream, P, T, chunk, n_regions, (uint32_t*)d_wg.p, (const uint32_t*)nullptr, (uint32_t*)nullptr);
        hipLaunchKernelGGL(k_bloom_wg_prefix, dim3((n_regions + 255) / 256), dim3(256), 0, c->stream, (uint32_t*)d_wg.p, n_wgs, n_regions, (uint32_t*)d_tot.p);
        hipLaunchKernelGGL(k_bloom_region_scan, dim3(1), dim3(1024), 0, c->stream, (const uint32_t*)d_tot.p, n_regions, (uint32_t*)d_off.p);
        hipLaunchKernelGGL((k_bloom_regions_basic<true>), dim3(n_wgs), dim3(BR_THREADS), (size_t)n_regions * 4, c->stream, P, T, chunk, n_regions, (uint32_t*)d_wg.p, (const uint32_t*)d_off.p, (uint32_t*)d_items.p);
        hipLaunchKernelGGL(k_bloom_region_build_basic, dim3(n_regions), dim3(BR_THREADS), (size_t)BR_WORDS * 4, c->stream, P, (const uint32_t*)d_items.p, (const uint32_t*)d_off.p);
    } else {
    hipLaunchKernelGGL((k_bloom_regions<false>), dim3(n_wgs), dim3(BR_THREADS), (size_t)n_regions * 4, c->stream, P, T, chunk, n_regions, (uint32_t*)d_wg.p, (const uint32_t*)nullptr, (BloomItem*)nullptr);
        hipLaunchKernelGGL(k_bloom_wg_prefix, dim3((n_regions + 255) / 256), dim3(256), 0, c->stream, (uint32_t*)d_wg.p, n_wgs, n_regions, (uint32_t*)d_tot.p);
        hipLaunchKernelGGL(k_bloom_region_scan, dim3(1), dim3(1024), 0, c->stream, (const uint32_t*)d_tot.p, n_regions, (uint32_t*)d_off.p);
        hipLaunchKernelGGL((k_bloom_regions<true>), dim3(n_wgs), dim3(BR_THREADS), (size_t)n_regions * 4, c->stream, P, T, chunk, n_regions, (uint32_t*)d_wg.p, (const uint32_t*)d_off.p, (BloomItem*)d_items.p);
        hipLaunchKernelGGL(k_bloom_region_build, dim3(n_regions), dim3(BR_THREADS), (size_t)(BR_WORDS + BR_FRINGE_WORDS) * 4, c->stream, P, (const BloomItem*)d_items.p, (const uint32_t*)d_off.p);
    }
    GKC_HIP(c, hipGetLastError());
    GKC_HIP(c, hipStreamSynchronize(c->stream));                   // the scratch buffers go back to the pool
    return GKC_OK;
}

int gkc_bloom_insert_device(gkc_bloom* b, const void* d_keys, uint64_t n, uint32_t stride)
{
    gkc_tun_refresh();
    if (!b) return GKC_ERR_ARG;
    GKC_TRY(check_stride(b, stride));
    if (!n) return GKC_OK;
    BSeg sg{ (const uint8_t*)d_keys, n, 0 };
    return bloom_insert_arrays(b, &sg, 1, stride);
}
int gkc_bloom_insert(gkc_bloom* b, const void* keys, uint64_t n, uint32_t stride)
{
    gkc_tun_refresh();
    if (!b) return GKC_ERR_ARG;
    gkc_ctx* c = b->ctx;
    GKC_TRY(check_stride(b, stride));
    if (!n) return GKC_OK;
    DevBuf d; GKC_TRY(c->ensure(d, (size_t)n * stride));
    hipError_t e = hipMemcpyAsync(d.p, keys, (size_t)n * stride, hipMemcpyHostToDevice, c->stream);
    int rc = (e == hipSuccess) ? gkc_bloom_insert_device(b, d.p, n, stride) : GKC_ERR_HIP;
    (void)hipStreamSynchronize(c->stream);
    d.release();
    if (e != hipSuccess) GKC_FAIL(c, GKC_ERR_HIP, "H2D copy failed: %s", hipGetErrorString(e));
    return rc;
}
int gkc_bloom_insert_solid(gkc_bloom* b, gkc_ctx* c)
{
    gkc_tun_refresh();
    if (!b || !c) return GKC_ERR_ARG;
    if (c->k != b->k) GKC_FAIL(c, GKC_ERR_ARG, "bloom k (%u) differs from the context's k (%u)", b->k, c->k);
    GKC_TRY(gkc_require_resident(c, "gkc_bloom_insert_solid"));
    // datasets of one Stage-B batch are consecutive in one output buffer: a handful of arrays in all
    const uint32_t stride = c->key_words == 1 ? 16 : 32;
    std::vector<BSeg> segs;
    for (const Dataset& D : c->datasets) {
        if (!D.done || !D.n_solid) continue;
        if (!segs.empty() && (const uint8_t*)D.d_counts == segs.back().p + segs.back().n * stride) { segs.back().n += D.n_solid; continue; }
        segs.push_back(BSeg{ (const uint8_t*)D.d_counts, D.n_solid, 0 });
    }
    for (size_t i = 0; i < segs.size(); i += 16) GKC_TRY(bloom_insert_arrays(b, segs.data() + i, (uint32_t)std::min<size_t>(16, segs.size() - i), stride));
    GKC_HIP(c, hipStreamSynchronize(c->stream));
    return GKC_OK;
}

// contains8 of up to 16 device arrays of k-mers (total < 2^31), answers in array order into d_out[total]: by region when the array has few enough regions, else false
static int bloom_contains8_regions(gkc_bloom* b, const BSeg* segs, uint32_t n_segs, uint32_t stride, uint8_t* d_out, bool* handled)
{
    gkc_ctx* c = b->ctx;
    *handled = false;
    uint64_t total = 0; for (uint32_t i = 0; i < n_segs; i++) total += segs[i].n;
    const uint64_t n_bits = b->tai + 1;
    const uint32_t n_regions = (uint32_t)std::min<uint64_t>((n_bits + (1u << BR_BITS) - 1) >> BR_BITS, 0xffffffffu);
    const uint64_t min_items = gkc_tun().bloom_query_regions_min;      // (tests lower it: the bucketing passes do not pay for a few k-mers)
    if (!total || total < min_items || n_regions > BR_MAX_REGIONS || total >= (1ULL << 31) || n_segs > 16 || b->kind != 2 || gkc_tun().bloom_gather) return GKC_OK;
    BSegTable T{}; T.n = n_segs; T.stride = stride; T.total = total;
    { uint64_t first = 0; for (uint32_t i = 0; i < n_segs; i++) { T.s[i] = segs[i]; T.s[i].first = first; first += segs[i].n; } }
    const uint32_t n_wgs = (uint32_t)std::min<uint64_t>(BR_WGS, (total + BR_THREADS - 1) / BR_THREADS);
    const uint64_t chunk = (total + n_wgs - 1) / n_wgs;
    DevBuf d_wg, d_tot, d_off, d_items, d_sides;
    struct Guard { DevBuf *a, *b2, *c2, *d, *e; ~Guard() { a->release(); b2->release(); c2->release(); d->release(); e->release(); } } guard{&d_wg, &d_tot, &d_off, &d_items, &d_sides};
    GKC_TRY(c->ensure(d_wg, (size_t)n_wgs * n_regions * 4)); GKC_TRY(c->ensure(d_tot, (size_t)n_regions * 4)); GKC_TRY(c->ensure(d_off, ((size_t)n_regions + 1) * 4));
    GKC_TRY(c->ensure(d_items, (size_t)total * 2 * sizeof(BloomQ))); GKC_TRY(c->ensure(d_sides, (size_t)total * 2));
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_bloom_q_regions<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(BR_MAX_REGIONS * 4));
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_bloom_q_regions<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(BR_MAX_REGIONS * 4));
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_bloom_region_query8), hipFuncAttributeMaxDynamicSharedMemorySize, (int)((BR_WORDS + BR_FRINGE_WORDS) * 4));
        attr_set = true;
    }
    const BloomParams P = params_of(b);
    hipLaunchKernelGGL((k_bloom_q_regions<false>), dim3(n_wgs), dim3(BR_THREADS), (size_t)n_regions * 4, c->stream, P, T, chunk, n_regions, (uint32_t*)d_wg.p, (const uint32_t*)nullptr, (BloomQ*)nullptr);
    hipLaunchKernelGGL(k_bloom_wg_prefix, dim3((n_regions + 255) / 256), dim3(256), 0, c->stream, (uint32_t*)d_wg.p, n_wgs, n_regions, (uint32_t*)d_tot.p);
    hipLaunchKernelGGL(k_bloom_region_scan, dim3(1), dim3(1024), 0, c->stream, (const uint32_t*)d_tot.p, n_regions, (uint32_t*)d_off.p);
    hipLaunchKernelGGL((k_bloom_q_regions<true>), dim3(n_wgs), dim3(BR_THREADS), (size_t)n_regions * 4, c->stream, P, T, chunk, n_regions, (uint32_t*)d_wg.p, (const uint32_t*)d_off.p, (BloomQ*)d_items.p);
    hipLaunchKernelGGL(k_bloom_region_query8, dim3(n_regions), dim3(BR_THREADS), (size_t)(BR_WORDS + BR_FRINGE_WORDS) * 4, c->stream, P, (uint64_t)(b->bits.bytes / 4), (const BloomQ*)d_items.p,
                       (const uint32_t*)d_off.p, (uint8_t*)d_sides.p, total);
    hipLaunchKernelGGL(k_bloom_join_sides, dim3((unsigned)std::min<uint64_t>((total + 255) / 256, 256 * 16)), dim3(256), 0, c->stream, (const uint8_t*)d_sides.p, total, d_out);
    GKC_HIP(c, hipGetLastError());
    GKC_HIP(c, hipStreamSynchronize(c->stream));                   // the scratch buffers go back to the pool
    *handled = true;
    return GKC_OK;
}
static int bloom_query(gkc_bloom* b, const void* keys, uint64_t n, uint32_t stride, uint8_t* out, bool c8)
{
    gkc_tun_refresh();
    if (!b) return GKC_ERR_ARG;
    gkc_ctx* c = b->ctx;
    GKC_TRY(check_stride(b, stride));
    if (c8 && b->kind != 2) GKC_FAIL(c, GKC_ERR_ARG, "contains8 is implemented by the neighbor kind only (Bloom.hpp:245-250 throws ExceptionNotImplemented)");
    if (!n) return GKC_OK;
    DevBuf d, o; GKC_TRY(c->ensure(d, (size_t)n * stride));
    int rc = c->ensure(o, (size_t)n);
    if (rc != GKC_OK) { d.release(); return rc; }
    hipError_t e = hipMemcpyAsync(d.p, keys, (size_t)n * stride, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        ScopedTimer tm(c, c8 ? "bloom_contains8" : "bloom_contains");
        const unsigned grid = (unsigned)std::min<uint64_t>((n + 255) / 256, 256 * 16);
        bool by_region = false;
        if (c8) { BSeg sg{ (const uint8_t*)d.p, n, 0 }; (void)hipStreamSynchronize(c->stream); if (bloom_contains8_regions(b, &sg, 1, stride, (uint8_t*)o.p, &by_region) != GKC_OK) by_region = false; }
        if (by_region) {}
        else if (c8) hipLaunchKernelGGL(k_bloom_contains8, dim3(grid), dim3(256), 0, c->stream, params_of(b), (const uint8_t*)d.p, n, stride, (uint8_t*)o.p);
        else    hipLaunchKernelGGL(k_bloom_contains, dim3(grid), dim3(256), 0, c->stream, params_of(b), (const uint8_t*)d.p, n, stride, (uint8_t*)o.p);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, o.p, (size_t)n, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    d.release(); o.release();
    if (e != hipSuccess) GKC_FAIL(c, GKC_ERR_HIP, "bloom query failed: %s", hipGetErrorString(e));
    return GKC_OK;
}
__global__ void k_sum_bits8(const uint8_t* __restrict__ a, uint64_t n, unsigned long long* __restrict__ out)
{
    unsigned long long s = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) s += (unsigned)__popc((unsigned)a[i]);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_down(s, d, 64);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(out, s);
}
int gkc_bloom_query_solid(gkc_bloom* b, gkc_ctx* c, int neighbors8, uint8_t* d_out, uint64_t* n_queried, uint64_t* n_positive)
{
    gkc_tun_refresh();
    if (!b || !c) return GKC_ERR_ARG;
    if (c->k != b->k) GKC_FAIL(c, GKC_ERR_ARG, "bloom k (%u) differs from the context's k (%u)", b->k, c->k);
    if (neighbors8 && b->kind != 2) GKC_FAIL(c, GKC_ERR_ARG, "contains8 is implemented by the neighbor kind only (Bloom.hpp:245-250 throws ExceptionNotImplemented)");
    GKC_TRY(gkc_require_resident(c, "gkc_bloom_query_solid"));
    const uint32_t stride = c->key_words == 1 ? 16 : 32;
    uint64_t total = 0;
    for (const Dataset& D : c->datasets) if (D.done) total += D.n_solid;
    if (n_queried) *n_queried = total;
    if (n_positive) *n_positive = 0;
    if (!total) return GKC_OK;
    DevBuf tmp, acc;
    if (!d_out) { GKC_TRY(c->ensure(tmp, (size_t)total)); d_out = (uint8_t*)tmp.p; }
    int rc = c->ensure(acc, 8);
    if (rc != GKC_OK) { tmp.release(); return rc; }
    hipError_t e = hipMemsetAsync(acc.p, 0, 8, c->stream);
    {   ScopedTimer tm(c, neighbors8 ? "bloom_contains8" : "bloom_contains");
        uint64_t done = 0;
        const uint8_t* run_p = nullptr; uint64_t run_n = 0;                // consecutive datasets of one Stage-B batch form one array
        bool by_region = false;
        if (neighbors8) {                                                  // all arrays at once, bucketed by region of the Bloom array (<= 16 arrays, < 2^31 k-mers: else the gathers below)
            std::vector<BSeg> segs;
            for (const Dataset& D : c->datasets) {
                if (!D.done || !D.n_solid) continue;
                if (!segs.empty() && (const uint8_t*)D.d_counts == segs.back().p + segs.back().n * stride) { segs.back().n += D.n_solid; continue; }
                segs.push_back(BSeg{ (const uint8_t*)D.d_counts, D.n_solid, 0 });
            }
            if (segs.size() <= 16 && bloom_contains8_regions(b, segs.data(), (uint32_t)segs.size(), stride, d_out, &by_region) != GKC_OK) by_region = false;
        }
        auto flush = [&]() {
            if (!run_n || by_region) return;
            const unsigned grid = (unsigned)std::min<uint64_t>((run_n + 255) / 256, 256 * 16);
            if (neighbors8) hipLaunchKernelGGL(k_bloom_contains8, dim3(grid), dim3(256), 0, c->stream, params_of(b), run_p, run_n, stride, d_out + done);
            else            hipLaunchKernelGGL(k_bloom_contains, dim3(grid), dim3(256), 0, c->stream, params_of(b), run_p, run_n, stride, d_out + done);
            done += run_n; run_n = 0;
        };
        for (const Dataset& D : c->datasets) {
            if (!D.done || !D.n_solid) continue;
            if (run_n && (const uint8_t*)D.d_counts == run_p + run_n * stride) { run_n += D.n_solid; continue; }
            flush();
            run_p = (const uint8_t*)D.d_counts; run_n = D.n_solid;
        }
        flush();
        if (e == hipSuccess) e = hipGetLastError();
    }
    unsigned long long h = 0;
    if (e == hipSuccess) { hipLaunchKernelGGL(k_sum_bits8, dim3(1024), dim3(256), 0, c->stream, (const uint8_t*)d_out, total, (unsigned long long*)acc.p); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipMemcpyAsync(&h, acc.p, 8, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    tmp.release(); acc.release();
    if (e != hipSuccess) GKC_FAIL(c, GKC_ERR_HIP, "bloom query failed: %s", hipGetErrorString(e));
    if (n_positive) *n_positive = h;
    return GKC_OK;
}
int gkc_bloom_contains(gkc_bloom* b, const void* keys, uint64_t n, uint32_t stride, uint8_t* out) { return bloom_query(b, keys, n, stride, out, false); }
int gkc_bloom_contains8(gkc_bloom* b, const void* keys, uint64_t n, uint32_t stride, uint8_t* out) { return bloom_query(b, keys, n, stride, out, true); }

int gkc_bloom_get_array(gkc_bloom* b, uint8_t* out, uint64_t cap)
{
    if (!b) return GKC_ERR_ARG;
    gkc_ctx* c = b->ctx;
    if (cap < b->nchar) GKC_FAIL(c, GKC_ERR_CAPACITY, "bloom array needs %llu bytes", (unsigned long long)b->nchar);
    GKC_HIP(c, hipMemcpyAsync(out, b->bits.p, (size_t)b->nchar, hipMemcpyDeviceToHost, c->stream));
    GKC_HIP(c, hipStreamSynchronize(c->stream));
    return GKC_OK;
}
int gkc_bloom_device_array(gkc_bloom* b, void** d_bits, uint64_t* n_bytes)
{
    if (!b || !d_bits || !n_bytes) return GKC_ERR_ARG;
    (void)hipStreamSynchronize(b->ctx->stream);
    *d_bits = b->bits.p; *n_bytes = (b->nchar + 3) / 4 * 4;
    return GKC_OK;
}
int gkc_bloom_set_array(gkc_bloom* b, const uint8_t* in, uint64_t n_bytes)
{
    if (!b) return GKC_ERR_ARG;
    gkc_ctx* c = b->ctx;
    if (n_bytes != b->nchar) GKC_FAIL(c, GKC_ERR_ARG, "bloom array must be %llu bytes", (unsigned long long)b->nchar);
    GKC_HIP(c, hipMemcpyAsync(b->bits.p, in, (size_t)n_bytes, hipMemcpyHostToDevice, c->stream));
    GKC_HIP(c, hipStreamSynchronize(c->stream));
    return GKC_OK;
}

}  // extern "C"
