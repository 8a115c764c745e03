// gkc_count.hip — Stage B on gfx950: super-k-mer buckets -> ascending (canonical k-mer, abundance) records per partition.
//
// Replaces (reference, under /root/reference/gatb-core/src/gatb/kmer/impl/):
//   B1 ReadSuperKCommand::execute            PartitionsCommand.cpp:944-1128   (decode, regenerate canonical k-mers)
//   B2 SortCommand::execute / executeSort    PartitionsCommand.cpp:1400-1504  (sort)
//   B3 KxmerPointer + executeDump            PartitionsCommand.cpp:1515-1805  (merge + run-length count)
//   B5 CountProcessorChain::process          CountProcessorChain.hpp:128-135  (histogram -> solidity(sum) -> dump)
//   B4 PartitionsByHashCommand (fallback when a partition does not fit): here = the oversize path.
//
// MI355X design: a key never makes more than one round trip through HBM.
//   expand_count   one workgroup per partition streams its 16/32-byte records (coalesced), regenerates the canonical
//                  k-mers with a rolling forward/reverse pair and histograms their top bits in LDS (<=4096 key-range
//                  sub-buckets per partition, sized so a sub-bucket fits one LDS sort); the same workgroup scans the
//                  histogram into exact sub-bucket offsets.
//   expand_scatter same stream again, LDS cursors (no global atomics), keys written once to their sub-bucket.
//   bucket_sort    persistent workgroups: load one sub-bucket into LDS, bitonic sort, run-length count, abundance
//                  histogram (LDS-aggregated), write distinct keys + counts back in place.
//   compact        exclusive scan of solid counts, then records {value, abundance} in the reference's Count layout,
//                  contiguous and ascending per partition (sub-buckets are key ranges, so concatenation is sorted).
//   oversize path  sub-buckets larger than the LDS capacity (massively repeated k-mers, or too few partitions):
//                  global-memory bitonic sort + a chunked single-workgroup run-length pass. Slow but exact.
#include "gkc_common.hpp"
#include "gkc_device.hpp"
#include <algorithm>

// ------------------------------------------------------------------------------------------------ tables
struct PartDesc {          // one per partition of the current Stage-B batch
    uint32_t part;         // partition id
    uint32_t sub_bits;     // log2(#sub-buckets)
    uint32_t shift;        // canonical >> shift = sub-bucket id  (2k - sub_bits)
    uint32_t pad;
    uint64_t key_base;     // first key of the partition in the batch key buffer
    uint64_t sub_base;     // first sub-bucket of the partition in the batch sub-bucket tables
};

struct SegTable {          // device copy of the segment list
    const uint8_t* const* rec;      // [n_seg] arena pointers
    const uint64_t* rec_off;        // [n_seg][P+1]
    uint32_t n_seg, P;
};

// nucleotide i of a device record (see RecT in gkc_device.hpp)
template <int RW> __device__ __forceinline__ uint32_t rec_nt(const uint64_t (&R)[RW], uint32_t i)
{
    if (i < 28) return (uint32_t)(R[0] >> (54 - 2 * i)) & 3u;
    uint32_t j = i - 28;
    return (uint32_t)(R[1 + (j >> 5)] >> (62 - 2 * (j & 31))) & 3u;
}

// calls f(canonical) for every k-mer of the record (B1: temp=((temp<<2)|nt)&mask, rev=((rev>>2)|(comp(nt)<<shift))&mask)
template <int KW, int RW, class F>
__device__ __forceinline__ void for_each_kmer(const uint64_t (&R)[RW], uint32_t k, F f)
{
    typedef typename KeyT<KW>::type key_t;
    const uint32_t nbk = (uint32_t)(R[0] >> 56);
    const key_t mask = KeyT<KW>::mask(k);
    key_t fw = 0;
    for (uint32_t i = 0; i < k; i++) fw = (fw << 2) | (key_t)rec_nt<RW>(R, i);
    key_t rv = KeyT<KW>::revcomp(fw, k);
    const uint32_t sh = 2 * (k - 1);
    for (uint32_t i = 0; i < nbk; i++) {
        f(fw < rv ? fw : rv);
        if (i + 1 < nbk) {
            const uint32_t c = rec_nt<RW>(R, k + i);
            fw = ((fw << 2) | (key_t)c) & mask;
            rv = (rv >> 2) | ((key_t)(c ^ 2u) << sh);
        }
    }
}

template <int RW> __device__ __forceinline__ void load_rec(const uint8_t* base, uint64_t idx, uint64_t (&R)[RW])
{
    const ulonglong2* p = reinterpret_cast<const ulonglong2*>(base + idx * (RW * 8));
#pragma unroll
    for (int i = 0; i < RW; i += 2) { ulonglong2 v = p[i / 2]; R[i] = v.x; R[i + 1] = v.y; }
}

constexpr int EXPAND_THREADS = 512;

// ------------------------------------------------------------------------------------------------ B1 expand_count
template <int KW, int RW>
__global__ __launch_bounds__(EXPAND_THREADS) void k_expand_count(const PartDesc* __restrict__ parts, SegTable segs, uint32_t k,
                                                                  uint64_t* __restrict__ sub_off)
{
    typedef typename KeyT<KW>::type key_t;
    __shared__ uint32_t s_hist[MAX_SUB];
    __shared__ uint32_t s_wsum[EXPAND_THREADS / 64];
    const PartDesc pd = parts[blockIdx.x];
    const uint32_t nsub = 1u << pd.sub_bits;
    for (uint32_t i = threadIdx.x; i < nsub; i += EXPAND_THREADS) s_hist[i] = 0;
    __syncthreads();
    for (uint32_t s = 0; s < segs.n_seg; s++) {
        const uint64_t r0 = segs.rec_off[(uint64_t)s * (segs.P + 1) + pd.part], r1 = segs.rec_off[(uint64_t)s * (segs.P + 1) + pd.part + 1];
        const uint8_t* base = segs.rec[s];
        for (uint64_t r = r0 + threadIdx.x; r < r1; r += EXPAND_THREADS) {
            uint64_t R[RW]; load_rec<RW>(base, r, R);
            for_each_kmer<KW, RW>(R, k, [&](key_t c) { atomicAdd(&s_hist[(uint32_t)(c >> pd.shift)], 1u); });
        }
    }
    __syncthreads();
    // exclusive scan of the nsub counters -> absolute key offsets of the sub-buckets
    const uint32_t per = (nsub + EXPAND_THREADS - 1) / EXPAND_THREADS;       // <= 8
    const uint32_t b = threadIdx.x * per;
    uint32_t loc = 0;
    for (uint32_t i = 0; i < per; i++) if (b + i < nsub) loc += s_hist[b + i];
    uint32_t x = loc;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { uint32_t y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
    if (lane == 63) s_wsum[wave] = x;
    __syncthreads();
    uint32_t wpre = 0;
    for (int w = 0; w < wave; w++) wpre += s_wsum[w];
    uint32_t run = wpre + x - loc;
    for (uint32_t i = 0; i < per; i++) if (b + i < nsub) { sub_off[pd.sub_base + b + i] = pd.key_base + run; run += s_hist[b + i]; }
}

// ------------------------------------------------------------------------------------------------ B3 expand_scatter
template <int KW, int RW>
__global__ __launch_bounds__(EXPAND_THREADS) void k_expand_scatter(const PartDesc* __restrict__ parts, SegTable segs, uint32_t k,
                                                                    const uint64_t* __restrict__ sub_off,
                                                                    typename KeyT<KW>::type* __restrict__ keys)
{
    typedef typename KeyT<KW>::type key_t;
    __shared__ uint32_t s_cur[MAX_SUB];
    const PartDesc pd = parts[blockIdx.x];
    const uint32_t nsub = 1u << pd.sub_bits;
    for (uint32_t i = threadIdx.x; i < nsub; i += EXPAND_THREADS) s_cur[i] = (uint32_t)(sub_off[pd.sub_base + i] - pd.key_base);
    __syncthreads();
    key_t* out = keys + pd.key_base;
    for (uint32_t s = 0; s < segs.n_seg; s++) {
        const uint64_t r0 = segs.rec_off[(uint64_t)s * (segs.P + 1) + pd.part], r1 = segs.rec_off[(uint64_t)s * (segs.P + 1) + pd.part + 1];
        const uint8_t* base = segs.rec[s];
        for (uint64_t r = r0 + threadIdx.x; r < r1; r += EXPAND_THREADS) {
            uint64_t R[RW]; load_rec<RW>(base, r, R);
            for_each_kmer<KW, RW>(R, k, [&](key_t c) {
                const uint32_t slot = atomicAdd(&s_cur[(uint32_t)(c >> pd.shift)], 1u);
                out[slot] = c;
            });
        }
    }
}

// ------------------------------------------------------------------------------------------------ B4 bucket_sort + RLE
constexpr int SORT_THREADS = 256;
constexpr int HIST_LDS = 64;

struct SortOut {
    uint32_t* cnt;            // [n_keys] abundance of distinct key j of sub-bucket g at cnt[start_g + j]
    uint32_t* n_distinct;     // [n_sub]
    uint32_t* n_solid;        // [n_sub]
    unsigned long long* histo; uint32_t histo_max;
    int32_t amin, amax;
    uint32_t* oversize_list; uint32_t* oversize_count; uint32_t oversize_cap;
};

template <int KW>
__global__ __launch_bounds__(SORT_THREADS) void k_bucket_sort(typename KeyT<KW>::type* __restrict__ keys, const uint64_t* __restrict__ sub_off,
                                                               uint32_t n_sub, SortOut O)
{
    typedef typename KeyT<KW>::type key_t;
    constexpr int CAP = (KW == 1) ? SORT_CAP_W1 : SORT_CAP_W2;
    __shared__ key_t s_k[CAP];
    __shared__ uint16_t s_head[CAP + 1];
    __shared__ uint32_t s_hc[HIST_LDS];
    __shared__ uint32_t s_wsum[SORT_THREADS / 64];
    __shared__ uint32_t s_nd, s_ns;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t < HIST_LDS) s_hc[t] = 0;
    for (uint32_t g = blockIdx.x; g < n_sub; g += gridDim.x) {
        const uint64_t start = sub_off[g];
        const uint64_t n64 = sub_off[g + 1] - start;
        __syncthreads();                                   // previous iteration fully drained (s_k, s_head reuse)
        if (n64 == 0) { if (t == 0) { O.n_distinct[g] = 0; O.n_solid[g] = 0; } continue; }
        if (n64 > (uint64_t)CAP) {
            if (t == 0) {
                O.n_distinct[g] = 0; O.n_solid[g] = 0;
                uint32_t slot = atomicAdd(O.oversize_count, 1u);
                if (slot < O.oversize_cap) O.oversize_list[slot] = g;
            }
            continue;
        }
        const uint32_t n = (uint32_t)n64;
        uint32_t N = 2; while (N < n) N <<= 1;
        for (uint32_t i = t; i < N; i += SORT_THREADS) s_k[i] = (i < n) ? keys[start + i] : KeyT<KW>::max();
        __syncthreads();
        // bitonic sort, all comparators ascending (mirror step then half-cleaners)
        for (uint32_t size = 2; size <= N; size <<= 1) {
            const uint32_t half = size >> 1;
            for (uint32_t q = t; q < (N >> 1); q += SORT_THREADS) {
                const uint32_t blk = q / half, off = q % half;
                const uint32_t i = blk * size + off, j = blk * size + size - 1 - off;
                key_t a = s_k[i], b = s_k[j];
                if (b < a) { s_k[i] = b; s_k[j] = a; }
            }
            __syncthreads();
            for (uint32_t stride = half >> 1; stride >= 1; stride >>= 1) {
                for (uint32_t q = t; q < (N >> 1); q += SORT_THREADS) {
                    const uint32_t i = 2 * stride * (q / stride) + (q % stride), j = i + stride;
                    key_t a = s_k[i], b = s_k[j];
                    if (b < a) { s_k[i] = b; s_k[j] = a; }
                }
                __syncthreads();
            }
        }
        // run-length count (B3): heads of runs, their positions compacted into s_head
        const uint32_t per = (n + SORT_THREADS - 1) / SORT_THREADS;
        const uint32_t b0 = t * per;
        uint32_t loc = 0;
        for (uint32_t i = b0; i < b0 + per && i < n; i++) loc += (i == 0 || s_k[i] != s_k[i - 1]);
        uint32_t x = loc;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { uint32_t y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
        if (lane == 63) s_wsum[wave] = x;
        __syncthreads();
        uint32_t wpre = 0;
        for (int w = 0; w < wave; w++) wpre += s_wsum[w];
        uint32_t idx = wpre + x - loc;
        for (uint32_t i = b0; i < b0 + per && i < n; i++) if (i == 0 || s_k[i] != s_k[i - 1]) s_head[idx++] = (uint16_t)i;
        if (t == SORT_THREADS - 1) { s_nd = idx; s_head[idx] = (uint16_t)n; s_ns = 0; }
        __syncthreads();
        const uint32_t nd = s_nd;
        uint32_t solid = 0;
        for (uint32_t j = t; j < nd; j += SORT_THREADS) {
            const uint32_t h = s_head[j];
            const uint32_t c = (uint32_t)s_head[j + 1] - h;
            keys[start + j] = s_k[h];                       // in place: the whole sub-bucket already sits in LDS
            O.cnt[start + j] = c;
            const uint32_t hb = c >= O.histo_max ? O.histo_max : c;          // Histogram::inc (Histogram.hpp:92)
            if (hb < HIST_LDS) atomicAdd(&s_hc[hb], 1u); else atomicAdd(&O.histo[hb], 1ULL);
            solid += ((int32_t)c >= O.amin && (int32_t)c <= O.amax);        // CountRange::includes, closed interval
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) solid += __shfl_down(solid, d, 64);
        if (lane == 0 && solid) atomicAdd(&s_ns, solid);
        __syncthreads();
        if (t == 0) { O.n_distinct[g] = nd; O.n_solid[g] = s_ns; }
    }
    __syncthreads();
    if (t < HIST_LDS && s_hc[t]) atomicAdd(&O.histo[t], (unsigned long long)s_hc[t]);
}

// ------------------------------------------------------------------------------------------------ oversize path
template <int KW>
__global__ void k_gbitonic_step(typename KeyT<KW>::type* __restrict__ a, uint64_t n, uint64_t N, uint64_t size, uint64_t stride, int mirror)
{
    typedef typename KeyT<KW>::type key_t;
    const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (N >> 1)) return;
    uint64_t i, j;
    if (mirror) { const uint64_t half = size >> 1, blk = q / half, off = q % half; i = blk * size + off; j = blk * size + size - 1 - off; }
    else { i = 2 * stride * (q / stride) + (q % stride); j = i + stride; }
    if (j >= n) return;                                   // virtual +inf padding never moves (all comparators ascending)
    key_t x = a[i], y = a[j];
    if (y < x) { a[i] = y; a[j] = x; }
}

// chunked run-length pass over a sorted segment by ONE workgroup; writes the same outputs as k_bucket_sort
template <int KW>
__global__ __launch_bounds__(1024) void k_rle_big(typename KeyT<KW>::type* __restrict__ keys, uint64_t start, uint64_t n, uint32_t g, SortOut O)
{
    typedef typename KeyT<KW>::type key_t;
    __shared__ uint64_t s_w[16]; __shared__ long long s_wh[16];
    __shared__ uint64_t s_carry_out; __shared__ long long s_carry_head; __shared__ uint32_t s_solid;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) { s_carry_out = 0; s_carry_head = -1; s_solid = 0; }
    __syncthreads();
    key_t* a = keys + start;
    for (uint64_t c0 = 0; c0 < n; c0 += 1024) {
        const uint64_t i = c0 + t;
        const bool in = i < n;
        key_t me = in ? a[i] : KeyT<KW>::max();
        const bool head = in && (i == 0 || a[i - 1] != me);
        const bool tail = in && (i == n - 1 || a[i + 1] != me);
        // inclusive max-scan of head positions, exclusive sum-scan of tails
        long long hp = head ? (long long)i : -1; uint64_t tc = tail ? 1 : 0;
        long long x = hp; uint64_t y = tc;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { long long x2 = __shfl_up(x, d, 64); uint64_t y2 = __shfl_up(y, d, 64); if (lane >= d) { x = x2 > x ? x2 : x; y += y2; } }
        if (lane == 63) { s_wh[wave] = x; s_w[wave] = y; }
        __syncthreads();
        long long ch = s_carry_head; uint64_t co = s_carry_out;
        for (int w = 0; w < wave; w++) { ch = s_wh[w] > ch ? s_wh[w] : ch; co += s_w[w]; }
        const long long myhead = x > ch ? x : ch;
        const uint64_t myidx = co + y - tc;
        __syncthreads();                                  // every a[i-1]/a[i+1] read of this chunk is done
        if (tail) {
            const uint64_t cnt64 = i - (uint64_t)myhead + 1;
            const uint32_t c = cnt64 > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)cnt64;   // CountNumber is int32
            a[myidx] = me; O.cnt[start + myidx] = c;
            const uint32_t hb = c >= O.histo_max ? O.histo_max : c;
            atomicAdd(&O.histo[hb], 1ULL);
            if ((int32_t)c >= O.amin && (int32_t)c <= O.amax) atomicAdd(&s_solid, 1u);
        }
        __syncthreads();
        if (t == 1023) { s_carry_head = myhead; s_carry_out = myidx + tc; }
        __syncthreads();
    }
    if (t == 0) { O.n_distinct[g] = (uint32_t)s_carry_out; O.n_solid[g] = s_solid; }
}

// ------------------------------------------------------------------------------------------------ scans / gather / compact
// out[i] = sum_{j<i} in[j]  (u32 -> u64), n+1 outputs; three small kernels
constexpr int SCAN_BLK = 1024, SCAN_ITEMS = 4;
__global__ __launch_bounds__(SCAN_BLK) void k_scan_block_sums(const uint32_t* __restrict__ in, uint64_t n, uint64_t* __restrict__ block_sums)
{
    __shared__ uint64_t s[SCAN_BLK / 64];
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_BLK * SCAN_ITEMS + (uint64_t)threadIdx.x * SCAN_ITEMS;
    uint64_t v = 0;
    for (int i = 0; i < SCAN_ITEMS; i++) if (base + i < n) v += in[base + i];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d, 64);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) { uint64_t tot = 0; for (int w = 0; w < SCAN_BLK / 64; w++) tot += s[w]; block_sums[blockIdx.x] = tot; }
}
__global__ void k_scan_serial(uint64_t* __restrict__ a, uint64_t n)     // exclusive, in place, tiny n (#blocks)
{
    if (threadIdx.x || blockIdx.x) return;
    uint64_t run = 0;
    for (uint64_t i = 0; i < n; i++) { uint64_t v = a[i]; a[i] = run; run += v; }
    a[n] = run;
}
__global__ __launch_bounds__(SCAN_BLK) void k_scan_final(const uint32_t* __restrict__ in, uint64_t n, const uint64_t* __restrict__ block_sums,
                                                          uint64_t* __restrict__ out)
{
    __shared__ uint64_t s[SCAN_BLK / 64];
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_BLK * SCAN_ITEMS + (uint64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS]; uint64_t loc = 0;
    for (int i = 0; i < SCAN_ITEMS; i++) { v[i] = (base + i < n) ? in[base + i] : 0; loc += v[i]; }
    uint64_t x = loc; const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { uint64_t y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
    if (lane == 63) s[wave] = x;
    __syncthreads();
    uint64_t run = block_sums[blockIdx.x];
    for (int w = 0; w < wave; w++) run += s[w];
    run += x - loc;
    for (int i = 0; i < SCAN_ITEMS; i++) { if (base + i < n) out[base + i] = run; run += v[i]; }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == SCAN_BLK - 1) out[n] = block_sums[gridDim.x];
}

// per-partition totals: distinct (sum over its sub-buckets) and the solid offset of its first sub-bucket
__global__ void k_part_totals(const PartDesc* __restrict__ parts, uint32_t n_parts, const uint32_t* __restrict__ n_distinct,
                              const uint64_t* __restrict__ solid_off, uint64_t* __restrict__ out /* [n_parts][2] */)
{
    __shared__ uint64_t s[4];
    const PartDesc pd = parts[blockIdx.x];
    const uint32_t nsub = 1u << pd.sub_bits;
    uint64_t v = 0;
    for (uint32_t i = threadIdx.x; i < nsub; i += blockDim.x) v += n_distinct[pd.sub_base + i];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d, 64);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = s[0] + s[1] + s[2] + s[3]; out[2 * blockIdx.x + 1] = solid_off[pd.sub_base]; }
}

// B5 dump: Count records {value, abundance} (Abundance.hpp:68-129), solid only, ascending
template <int KW>
__global__ __launch_bounds__(SORT_THREADS) void k_compact(const typename KeyT<KW>::type* __restrict__ keys, const uint32_t* __restrict__ cnt,
                                                           const uint64_t* __restrict__ sub_off, const uint32_t* __restrict__ n_distinct,
                                                           const uint64_t* __restrict__ solid_off, uint32_t n_sub,
                                                           int32_t amin, int32_t amax, uint64_t* __restrict__ out)
{
    __shared__ uint32_t s_wsum[SORT_THREADS / 64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    constexpr int OW = (KW == 1) ? 2 : 4;                   // u64 words per Count record
    for (uint32_t g = blockIdx.x; g < n_sub; g += gridDim.x) {
        const uint32_t nd = n_distinct[g];
        const uint64_t start = sub_off[g];
        uint64_t obase = solid_off[g];
        for (uint32_t c0 = 0; c0 < nd; c0 += SORT_THREADS) {
            const uint32_t j = c0 + t;
            uint32_t c = 0; typename KeyT<KW>::type key = 0;
            bool ok = false;
            if (j < nd) { c = cnt[start + j]; key = keys[start + j]; ok = ((int32_t)c >= amin && (int32_t)c <= amax); }
            uint32_t x = ok;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { uint32_t y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
            __syncthreads();
            if (lane == 63) s_wsum[wave] = x;
            __syncthreads();
            uint32_t pre = 0, tot = 0;
            for (int w = 0; w < SORT_THREADS / 64; w++) { if (w < wave) pre += s_wsum[w]; tot += s_wsum[w]; }
            if (ok) {
                uint64_t* o = out + (obase + pre + x - 1) * OW;
                if (KW == 1) { *reinterpret_cast<ulonglong2*>(o) = make_ulonglong2((uint64_t)key, (uint64_t)c); }
                else {
                    *reinterpret_cast<ulonglong2*>(o) = make_ulonglong2((uint64_t)key, (uint64_t)((u128)key >> 64));
                    *reinterpret_cast<ulonglong2*>(o + 2) = make_ulonglong2((uint64_t)c, 0ULL);
                }
            }
            obase += tot;
        }
    }
}

// checksum of a Count-record array: sum abundance * mix(value), sum abundance
template <int KW>
__global__ void k_result_checksum(const uint64_t* __restrict__ recs, uint64_t n, unsigned long long* __restrict__ out)
{
    constexpr int OW = (KW == 1) ? 2 : 4;
    uint64_t cs = 0, sa = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t* r = recs + i * OW;
        uint64_t c; uint64_t mv;
        if (KW == 1) { mv = mix64(r[0]); c = (uint32_t)r[1]; }
        else { mv = mix64(r[0]) ^ mix64(~r[1]); c = (uint32_t)r[2]; }
        cs += c * mv; sa += c;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { cs += __shfl_down(cs, d, 64); sa += __shfl_down(sa, d, 64); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&out[0], (unsigned long long)cs); atomicAdd(&out[1], (unsigned long long)sa); }
}

// ------------------------------------------------------------------------------------------------ host orchestration
template <int KW, int RW>
static int count_batch(gkc_ctx* c, const std::vector<uint32_t>& batch_parts, const std::vector<uint64_t>& part_keys,
                       const SegTable& segs, std::vector<void*>& outputs)
{
    typedef typename KeyT<KW>::type key_t;
    constexpr int CAP = (KW == 1) ? SORT_CAP_W1 : SORT_CAP_W2;
    const uint32_t nb = (uint32_t)batch_parts.size();
    const uint32_t k = c->k;
    // --- host-built tables (sizes are known exactly from Stage A)
    std::vector<PartDesc> pd(nb);
    uint64_t n_keys = 0, n_sub = 0;
    for (uint32_t i = 0; i < nb; i++) {
        const uint64_t np = part_keys[batch_parts[i]];
        if (np >= (1ULL << 32)) GKC_FAIL(c, GKC_ERR_ARG, "partition %u holds %llu k-mers (>= 2^32): use more partitions", batch_parts[i], (unsigned long long)np);
        uint32_t bits = 0;
        const uint32_t target = (KW == 1) ? SUB_TARGET : SUB_TARGET / 2;
        while (bits < (uint32_t)MAX_SUB_BITS && bits < 2 * k && (np >> bits) > target) bits++;
        pd[i].part = batch_parts[i]; pd[i].sub_bits = bits; pd[i].shift = 2 * k - bits; pd[i].pad = 0;
        pd[i].key_base = n_keys; pd[i].sub_base = n_sub;
        n_keys += np; n_sub += (1ull << bits);
    }
    if (n_sub >= (1ULL << 31)) GKC_FAIL(c, GKC_ERR_ARG, "too many sub-buckets in one batch");
    DevBuf d_pd, d_keys, d_cnt, d_suboff, d_nd, d_ns, d_soloff, d_bsum, d_over, d_ptot;
    auto cleanup = [&]() { d_pd.release(); d_keys.release(); d_cnt.release(); d_suboff.release(); d_nd.release(); d_ns.release();
                           d_soloff.release(); d_bsum.release(); d_over.release(); d_ptot.release(); };
#define CB_TRY(expr) do { int rc__ = (expr); if (rc__ != GKC_OK) { cleanup(); return rc__; } } while (0)
#define CB_HIP(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { cleanup(); c->set_error(GKC_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); return GKC_ERR_HIP; } } while (0)
    CB_TRY(c->ensure(d_pd, nb * sizeof(PartDesc)));
    CB_TRY(c->ensure(d_keys, (size_t)std::max<uint64_t>(n_keys, 1) * sizeof(key_t)));
    CB_TRY(c->ensure(d_cnt, (size_t)std::max<uint64_t>(n_keys, 1) * 4));
    CB_TRY(c->ensure(d_suboff, (size_t)(n_sub + 1) * 8));
    CB_TRY(c->ensure(d_nd, (size_t)n_sub * 4));
    CB_TRY(c->ensure(d_ns, (size_t)n_sub * 4));
    CB_TRY(c->ensure(d_soloff, (size_t)(n_sub + 1) * 8));
    const uint64_t n_scan_blocks = (n_sub + (uint64_t)SCAN_BLK * SCAN_ITEMS - 1) / ((uint64_t)SCAN_BLK * SCAN_ITEMS);
    CB_TRY(c->ensure(d_bsum, (size_t)(n_scan_blocks + 1) * 8));
    const uint32_t over_cap = 1u << 20;
    CB_TRY(c->ensure(d_over, (size_t)(over_cap + 1) * 4));
    CB_TRY(c->ensure(d_ptot, (size_t)nb * 16));
    CB_HIP(hipMemcpyAsync(d_pd.p, pd.data(), nb * sizeof(PartDesc), hipMemcpyHostToDevice, c->stream));
    CB_HIP(hipMemcpyAsync((uint64_t*)d_suboff.p + n_sub, &n_keys, 8, hipMemcpyHostToDevice, c->stream));
    CB_HIP(hipMemsetAsync(d_over.p, 0, (size_t)(over_cap + 1) * 4, c->stream));
    uint32_t* over_list = (uint32_t*)d_over.p + 1; uint32_t* over_count = (uint32_t*)d_over.p;

    {   ScopedTimer tm(c, "expand_count");
        hipLaunchKernelGGL((k_expand_count<KW, RW>), dim3(nb), dim3(EXPAND_THREADS), 0, c->stream, (const PartDesc*)d_pd.p, segs, k, (uint64_t*)d_suboff.p);
        CB_HIP(hipGetLastError());
    }
    {   ScopedTimer tm(c, "expand_scatter");
        hipLaunchKernelGGL((k_expand_scatter<KW, RW>), dim3(nb), dim3(EXPAND_THREADS), 0, c->stream, (const PartDesc*)d_pd.p, segs, k,
                           (const uint64_t*)d_suboff.p, (key_t*)d_keys.p);
        CB_HIP(hipGetLastError());
    }
    SortOut O{};
    O.cnt = (uint32_t*)d_cnt.p; O.n_distinct = (uint32_t*)d_nd.p; O.n_solid = (uint32_t*)d_ns.p;
    O.histo = (unsigned long long*)c->d_histo.p; O.histo_max = c->histo_max; O.amin = c->amin; O.amax = c->amax;
    O.oversize_list = over_list; O.oversize_count = over_count; O.oversize_cap = over_cap;
    const uint32_t sort_grid = (uint32_t)std::min<uint64_t>(n_sub, 256 * 8);
    {   ScopedTimer tm(c, "bucket_sort");
        hipLaunchKernelGGL((k_bucket_sort<KW>), dim3(sort_grid), dim3(SORT_THREADS), 0, c->stream, (key_t*)d_keys.p, (const uint64_t*)d_suboff.p, (uint32_t)n_sub, O);
        CB_HIP(hipGetLastError());
    }
    // --- oversize sub-buckets (rare): global bitonic sort + chunked RLE
    uint32_t n_over = 0;
    CB_HIP(hipMemcpyAsync(&n_over, over_count, 4, hipMemcpyDeviceToHost, c->stream));
    CB_HIP(hipStreamSynchronize(c->stream));
    if (n_over > over_cap) { cleanup(); GKC_FAIL(c, GKC_ERR_ARG, "more than %u oversize sub-buckets: use more partitions", over_cap); }
    if (n_over) {
        ScopedTimer tm(c, "oversize_sort");
        std::vector<uint32_t> ol(n_over);
        CB_HIP(hipMemcpy(ol.data(), over_list, (size_t)n_over * 4, hipMemcpyDeviceToHost));
        std::vector<uint64_t> so(2);
        for (uint32_t g : ol) {
            CB_HIP(hipMemcpy(so.data(), (uint64_t*)d_suboff.p + g, 16, hipMemcpyDeviceToHost));
            const uint64_t start = so[0], n = so[1] - so[0];
            uint64_t N = 2; while (N < n) N <<= 1;
            key_t* a = (key_t*)d_keys.p + start;
            const unsigned blocks = (unsigned)(((N >> 1) + 255) / 256);
            for (uint64_t size = 2; size <= N; size <<= 1) {
                hipLaunchKernelGGL((k_gbitonic_step<KW>), dim3(blocks), dim3(256), 0, c->stream, a, n, N, size, (uint64_t)0, 1);
                for (uint64_t stride = size >> 2; stride >= 1; stride >>= 1)
                    hipLaunchKernelGGL((k_gbitonic_step<KW>), dim3(blocks), dim3(256), 0, c->stream, a, n, N, size, stride, 0);
            }
            hipLaunchKernelGGL((k_rle_big<KW>), dim3(1), dim3(1024), 0, c->stream, (key_t*)d_keys.p, start, n, g, O);
            CB_HIP(hipGetLastError());
        }
        c->stats_now().oversize_buckets += n_over;
    }
    // --- solid offsets, per-partition totals, output allocation, compaction
    uint64_t total_solid = 0;
    std::vector<uint64_t> ptot((size_t)nb * 2);
    {   ScopedTimer tm(c, "compact");
        hipLaunchKernelGGL(k_scan_block_sums, dim3((unsigned)n_scan_blocks), dim3(SCAN_BLK), 0, c->stream, (const uint32_t*)d_ns.p, n_sub, (uint64_t*)d_bsum.p);
        hipLaunchKernelGGL(k_scan_serial, dim3(1), dim3(1), 0, c->stream, (uint64_t*)d_bsum.p, n_scan_blocks);
        hipLaunchKernelGGL(k_scan_final, dim3((unsigned)n_scan_blocks), dim3(SCAN_BLK), 0, c->stream, (const uint32_t*)d_ns.p, n_sub, (const uint64_t*)d_bsum.p, (uint64_t*)d_soloff.p);
        hipLaunchKernelGGL(k_part_totals, dim3(nb), dim3(256), 0, c->stream, (const PartDesc*)d_pd.p, nb, (const uint32_t*)d_nd.p, (const uint64_t*)d_soloff.p, (uint64_t*)d_ptot.p);
        CB_HIP(hipGetLastError());
        CB_HIP(hipMemcpyAsync(&total_solid, (uint64_t*)d_soloff.p + n_sub, 8, hipMemcpyDeviceToHost, c->stream));
        CB_HIP(hipMemcpyAsync(ptot.data(), d_ptot.p, (size_t)nb * 16, hipMemcpyDeviceToHost, c->stream));
        CB_HIP(hipStreamSynchronize(c->stream));
        constexpr int OW = (KW == 1) ? 2 : 4;
        void* out = nullptr;
        hipError_t e = hipMalloc(&out, (size_t)std::max<uint64_t>(total_solid, 1) * OW * 8);
        if (e != hipSuccess) { cleanup(); GKC_FAIL(c, GKC_ERR_NOMEM, "hipMalloc of %llu Count records failed: %s", (unsigned long long)total_solid, hipGetErrorString(e)); }
        outputs.push_back(out);
        hipLaunchKernelGGL((k_compact<KW>), dim3(sort_grid), dim3(SORT_THREADS), 0, c->stream, (const key_t*)d_keys.p, (const uint32_t*)d_cnt.p,
                           (const uint64_t*)d_suboff.p, (const uint32_t*)d_nd.p, (const uint64_t*)d_soloff.p, (uint32_t)n_sub, c->amin, c->amax, (uint64_t*)out);
        CB_HIP(hipGetLastError());
        CB_HIP(hipStreamSynchronize(c->stream));
        for (uint32_t i = 0; i < nb; i++) {
            Dataset& D = c->datasets[(size_t)c->pass * c->nb_partitions + batch_parts[i]];
            const uint64_t s0 = ptot[2 * i + 1], s1 = (i + 1 < nb) ? ptot[2 * (i + 1) + 1] : total_solid;
            D.d_counts = (const uint8_t*)out + s0 * OW * 8;
            D.n_solid = s1 - s0; D.n_distinct = ptot[2 * i]; D.n_kmers = part_keys[batch_parts[i]]; D.done = true;
            c->stats_now().kmers_nb_distinct += D.n_distinct; c->stats_now().kmers_nb_solid += D.n_solid;
        }
    }
    cleanup();
#undef CB_TRY
#undef CB_HIP
    (void)CAP;
    return GKC_OK;
}

int gkc_count_pass(gkc_ctx* c)
{
    const uint32_t Pn = c->nb_partitions;
    const uint32_t n_seg = (uint32_t)c->segments.size();
    std::vector<uint64_t> part_keys(Pn, 0);
    for (const Segment& s : c->segments) for (uint32_t p = 0; p < Pn; p++) part_keys[p] += s.nkmers[p];
    // device copy of the segment table
    DevBuf d_recptr, d_recoff;
    std::vector<const uint8_t*> ptrs(std::max<uint32_t>(n_seg, 1), nullptr);
    std::vector<uint64_t> offs((size_t)std::max<uint32_t>(n_seg, 1) * (Pn + 1), 0);
    for (uint32_t s = 0; s < n_seg; s++) {
        ptrs[s] = (const uint8_t*)c->segments[s].d_records;
        memcpy(&offs[(size_t)s * (Pn + 1)], c->segments[s].rec_off.data(), (size_t)(Pn + 1) * 8);
    }
    GKC_TRY(c->ensure(d_recptr, ptrs.size() * sizeof(void*)));
    int rc = c->ensure(d_recoff, offs.size() * 8);
    if (rc != GKC_OK) { d_recptr.release(); return rc; }
    hipError_t e1 = hipMemcpy(d_recptr.p, ptrs.data(), ptrs.size() * sizeof(void*), hipMemcpyHostToDevice);
    hipError_t e2 = hipMemcpy(d_recoff.p, offs.data(), offs.size() * 8, hipMemcpyHostToDevice);
    if (e1 != hipSuccess || e2 != hipSuccess) { d_recptr.release(); d_recoff.release(); GKC_FAIL(c, GKC_ERR_HIP, "segment table upload failed"); }
    SegTable segs{ (const uint8_t* const*)d_recptr.p, (const uint64_t*)d_recoff.p, n_seg, Pn };

    // batches of consecutive partitions bounded by the key budget
    size_t budget = c->key_budget;
    if (!budget) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = (size_t)8 << 30;
        // per key: key + 4 B count + (solid) Count record; keep a third of the free memory for outputs
        const size_t per_key = (c->key_words == 1 ? 8 : 16) + 4;
        budget = std::max<size_t>((free_b / 3) / per_key, (size_t)1 << 20);
        budget = std::min<size_t>(budget, (size_t)3 << 30);
    }
    std::vector<void*>& outputs = c->pass_outputs[c->pass];
    std::vector<uint32_t> batch; uint64_t acc = 0;
    rc = GKC_OK;
    auto flush = [&]() -> int {
        if (batch.empty()) return GKC_OK;
        int r = (c->key_words == 1) ? count_batch<1, 2>(c, batch, part_keys, segs, outputs) : count_batch<2, 4>(c, batch, part_keys, segs, outputs);
        batch.clear(); acc = 0; return r;
    };
    for (uint32_t p = 0; p < Pn && rc == GKC_OK; p++) {
        if (!batch.empty() && acc + part_keys[p] > budget) rc = flush();
        if (rc != GKC_OK) break;
        batch.push_back(p); acc += part_keys[p];
    }
    if (rc == GKC_OK) rc = flush();
    d_recptr.release(); d_recoff.release();
    return rc;
}

// explicit result checksum entry (used by the C-ABI)
int gkc_result_checksum_impl(gkc_ctx* c, uint64_t* checksum, uint64_t* sum_abundance)
{
    DevBuf d; GKC_TRY(c->ensure(d, 16));
    GKC_HIP(c, hipMemsetAsync(d.p, 0, 16, c->stream));
    for (const Dataset& D : c->datasets) {
        if (!D.done || !D.n_solid) continue;
        if (c->key_words == 1) hipLaunchKernelGGL((k_result_checksum<1>), dim3(1024), dim3(256), 0, c->stream, (const uint64_t*)D.d_counts, D.n_solid, (unsigned long long*)d.p);
        else                   hipLaunchKernelGGL((k_result_checksum<2>), dim3(1024), dim3(256), 0, c->stream, (const uint64_t*)D.d_counts, D.n_solid, (unsigned long long*)d.p);
    }
    uint64_t h[2];
    hipError_t e = hipMemcpyAsync(h, d.p, 16, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    d.release();
    if (e != hipSuccess) GKC_FAIL(c, GKC_ERR_HIP, "result checksum failed: %s", hipGetErrorString(e));
    *checksum = h[0]; *sum_abundance = h[1];
    return GKC_OK;
}
