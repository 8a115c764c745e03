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
//   expand_count   one workgroup per partition streams its 16/32-byte records (coalesced), cuts the canonical k-mers out of the
//                  record's bit string and histograms their top bits in LDS (<= 8192 key-range sub-buckets per partition, sized
//                  so a sub-bucket fits one wave's registers); the same workgroup scans the histogram into exact sub-bucket offsets.
//   expand_scatter same stream again; keys leave in aligned PAIRS (one parking slot per sub-bucket in LDS, exchange-only protocol, no
//                  global atomics): 8-byte keys as 16-byte stores, 16-byte keys as 32-byte stores (128-bit LDS exchange, ds_wrxchg2).
//   bucket_sort    one WAVE per sub-bucket: keys in registers, bitonic network (in-lane steps as v_min/max_f64 on double-tagged keys,
//                  cross-lane steps as DPP / bpermute exchanges, no LDS, no barrier), run-length count, abundance histogram
//                  (LDS-aggregated); distinct keys and abundances are written back at the head of the sub-bucket's own slot range.
//                  Tiers: <= 1024 keys (k_wave_sort), <= 2048 (k_wave_sort_big), <= 4096 (k_wg_sort, 4 waves merged through LDS).
//   split levels   sub-buckets larger than that (k-mers that start with their minimizer share their top bits; repeats; too few
//                  partitions) are split again on their next informative key bits, keys -> keys, and go through the tiers again;
//                  when no key bit is left all keys are one k-mer. Any skew terminates in <= ceil(2k/13)+1 levels.
//   compact        slot flags (abundance != 0) -> block sums -> prefix -> records {value, abundance} in the reference's
//                  Count layout, contiguous and ascending per partition (slot order is key order).
//   Batches of partitions (equal key budgets planned once per pass, so the caching allocator hands the same blocks out again) are
//   taken from one queue by two host threads, each on its own stream (gkc_count_pass).
#include "gkc_common.hpp"
#include "gkc_device.hpp"
#include <algorithm>
#include <mutex>
#include <thread>

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
    const uint32_t j = i - 28;
    uint64_t w = R[1];                                    // selects instead of a dynamic register-array index (no scratch)
    if (RW == 4) { const uint32_t q = j >> 5; w = q == 0 ? R[1] : (q == 1 ? R[RW > 2 ? 2 : 1] : R[RW > 2 ? 3 : 1]); }
    return (uint32_t)(w >> (62 - 2 * (j & 31))) & 3u;
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

// 16-byte record as a left-aligned bit string (120 bits = 60 nt): the 64-bit window that starts at nucleotide i
__device__ __forceinline__ uint64_t rec_window(uint64_t s_hi, uint64_t s_lo, uint32_t i)
{
    const uint32_t s = 2 * i;                                   // 0..54
    return (s_hi << s) | ((s_lo >> 1) >> (63 - s));
}

// k <= 31, 16-byte records: k-mer i is a window of the record's bit string (no per-nucleotide setup loop); the reverse
// complement rolls: the nucleotide entering at the right is the low 2 bits of the new window
template <class F>
__device__ __forceinline__ void for_each_kmer16(const uint64_t (&R)[2], uint32_t k, F f)
{
    const uint64_t s_hi = (R[0] << 8) | (R[1] >> 56), s_lo = R[1] << 8;
    const uint32_t nbk = (uint32_t)(R[0] >> 56), down = 64 - 2 * k, sh = 2 * (k - 1);
    uint64_t fw = s_hi >> down, rv = revcomp64(fw, k);
    for (uint32_t i = 0; i < nbk; i++) {
        f(fw < rv ? fw : rv);
        fw = rec_window(s_hi, s_lo, i + 1) >> down;
        rv = (rv >> 2) | ((uint64_t)(((uint32_t)fw & 3u) ^ 2u) << sh);
    }
}

// 32 <= k <= 63, 32-byte records (248-bit string, 124 nt): the same with a 128-bit window, written on 64-bit halves (the compiler's
// variable 128-bit shifts cost several times the few funnel shifts that are needed: `down` is in [2, 64], `sh` in [62, 124])
template <class F>
__device__ __forceinline__ void for_each_kmer32(const uint64_t (&R)[4], uint32_t k, F f)
{
    const uint64_t S0 = (R[0] << 8) | (R[1] >> 56), S1 = (R[1] << 8) | (R[2] >> 56), S2 = (R[2] << 8) | (R[3] >> 56), S3 = R[3] << 8;
    const uint32_t nbk = (uint32_t)(R[0] >> 56), down = 128 - 2 * k, sh = 2 * (k - 1);
    uint64_t fh, fl;                                              // forward k-mer (2k bits) = window >> down
    auto window = [&](uint32_t i) {                               // bits [2i, 2i+128) of the string (2i <= 118), shifted right by `down`
        const uint32_t s = 2 * i, t = s & 63;
        const bool j = s >= 64;
        const uint64_t A = j ? S1 : S0, B = j ? S2 : S1, C = j ? S3 : S2;
        const uint64_t hi = (A << t) | ((B >> 1) >> (63 - t)), lo = (B << t) | ((C >> 1) >> (63 - t));
        if (down == 64) { fh = 0; fl = hi; }
        else { fh = hi >> down; fl = (lo >> down) | (hi << (64 - down)); }
    };
    window(0);
    const u128 rv0 = revcomp128(((u128)fh << 64) | fl, k);
    uint64_t rh = (uint64_t)(rv0 >> 64), rl = (uint64_t)rv0;
    for (uint32_t i = 0; i < nbk; i++) {
        const bool fwd = fh < rh || (fh == rh && fl < rl);
        f(fwd ? (((u128)fh << 64) | fl) : (((u128)rh << 64) | rl));
        window(i + 1);
        const uint64_t c = (uint64_t)(((uint32_t)fl & 3u) ^ 2u);
        rl = (rl >> 2) | (rh << 62); rh >>= 2;
        if (sh >= 64) rh |= c << (sh - 64); else rl |= c << sh;
    }
}
// Counting pass for 16-byte keys: only the sub-bucket of every k-mer is needed = the top `bits` (<= 13) bits of min(forward, reverse
// complement). Those are decided by the TOP 64 bits of the two (the first 32 nt of the k-mer / the reverse complement of its last 32):
// when the top words tie, both give the same sub-bucket. So the walk keeps two 64-bit words only: one funnel shift for the forward
// top word, a rolling reverse complement of the last 32 nt. calls f(sub-bucket) for every k-mer of the record (32 <= k <= 63).
template <class F>
__device__ __forceinline__ void for_each_sub32(const uint64_t (&R)[4], uint32_t k, uint32_t bits, F f)
{
    const uint64_t S0 = (R[0] << 8) | (R[1] >> 56), S1 = (R[1] << 8) | (R[2] >> 56), S2 = (R[2] << 8) | (R[3] >> 56), S3 = R[3] << 8;
    const uint32_t nbk = (uint32_t)(R[0] >> 56), down = 128 - 2 * k;
    // first k-mer in full (once per record): its reverse complement, left-aligned, gives the initial top word
    uint64_t fh, fl;
    if (down == 64) { fh = 0; fl = S0; } else { fh = S0 >> down; fl = (S1 >> down) | (S0 << (64 - down)); }
    const u128 rv0 = revcomp128(((u128)fh << 64) | fl, k);
    const uint64_t rh = (uint64_t)(rv0 >> 64), rl = (uint64_t)rv0;
    uint64_t rtop = down == 64 ? rl : (rh << down) | (rl >> (64 - down));
    const uint32_t idx_sh = 64 - bits;
    for (uint32_t i = 0; i < nbk; i++) {
        const uint32_t s = 2 * i, t = s & 63;
        const bool j = s >= 64;
        const uint64_t A = j ? S1 : S0, B = j ? S2 : S1;
        const uint64_t ftop = (A << t) | ((B >> 1) >> (63 - t));
        const uint64_t m = ftop < rtop ? ftop : rtop;
        f(bits ? (uint32_t)(m >> idx_sh) : 0u);
        const uint32_t bit = 2 * (i + k), w = bit >> 6, sh = 62 - (bit & 63);       // nucleotide i + k enters the next k-mer at the right
        const uint64_t W = w == 0 ? S0 : (w == 1 ? S1 : (w == 2 ? S2 : S3));
        rtop = (rtop >> 2) | ((((W >> sh) & 3ull) ^ 2ull) << 62);
    }
}
// record width -> fastest k-mer walk (the generic per-nucleotide for_each_kmer stays as the reference restatement for other widths)
template <int KW, int RW, class F>
__device__ __forceinline__ void for_each_kmer_fast(const uint64_t (&R)[RW], uint32_t k, F f)
{
    if constexpr (KW == 1 && RW == 2) for_each_kmer16(R, k, f);
    else if constexpr (KW == 2 && RW == 4) { if (k >= 32) for_each_kmer32(R, k, f); else for_each_kmer<KW, RW>(R, k, f); }
    else for_each_kmer<KW, RW>(R, k, f);
}

// sub-bucket of a key = its top bits: key >> shift (the result fits 13 bits). For 16-byte keys a variable 128-bit shift is several times the
// work of the one or two 64-bit shifts that are needed
template <int KW> __device__ __forceinline__ uint32_t sub_index(typename KeyT<KW>::type c, uint32_t shift)
{
    if constexpr (KW == 1) return (uint32_t)(c >> shift);
    else {
        const uint64_t hi = (uint64_t)(c >> 64), lo = (uint64_t)c;
        if (shift >= 64) return (uint32_t)(hi >> (shift - 64));
        return (uint32_t)((hi << (64 - shift)) | (lo >> shift));            // shift in [1, 63] here (2k - 13 >= 51 for k >= 32)
    }
}

constexpr int EXPAND_THREADS = 512;

// bin tables of a batch (written by k_expand_count, read by k_expand_coarse and k_bin_sort)
constexpr uint32_t BIN_WINDOW_LOG = 12, BIN_WINDOW = 1u << BIN_WINDOW_LOG;     // slots
constexpr uint32_t BIN_ISO = 4096;                                               // a sub-bucket beyond this many keys is a bin of its own
constexpr uint32_t BIN_SUBS_MAX = 1024;                                          // sub-buckets of a bin (LDS cursors of k_bin_sort)
constexpr uint32_t BIN_NBMAX = 2048;                                             // bins of a partition (LDS counters / cursors of k_expand_coarse)
struct BinTables {
    uint16_t* of_sub;          // [n_sub] bin of the sub-bucket inside its partition
    uint32_t* first;           // [nb][nbmax + 1] first sub-bucket of every bin, then the sentinel nsub
    uint32_t* n;               // [nb] bins of the partition
    uint32_t* bad;             // set when a partition has more than nbmax bins: the batch takes the pair scatter + wave sort instead
    uint32_t nbmax;
};

// ------------------------------------------------------------------------------------------------ B1 expand_count
template <int KW, int RW>
__global__ __launch_bounds__(EXPAND_THREADS) void k_expand_count(const PartDesc* __restrict__ parts, SegTable segs, uint32_t k,
                                                                  uint64_t* __restrict__ b_start, uint32_t* __restrict__ b_n, uint8_t* __restrict__ b_consumed,
                                                                  uint32_t line_slots /* 0: every sub-bucket starts on a multiple of 4 slots (pair scatter);
                                                                                         n: SUPER-buckets of 4 sub-buckets start on multiples of n slots, sub-buckets packed inside */,
                                                                  BinTables bins /* nbmax == 0: no bin tables (see k_expand_coarse) */,
                                                                  const uint32_t* __restrict__ order /* workgroup -> partition of the batch (largest first), or nullptr */)
{
    typedef typename KeyT<KW>::type key_t;
    __shared__ uint32_t s_hist[MAX_SUB];
    __shared__ uint32_t s_wsum[EXPAND_THREADS / 64];
    const uint32_t bi = order ? order[blockIdx.x] : blockIdx.x;
    const PartDesc pd = parts[bi];
    const uint32_t nsub = 1u << pd.sub_bits;
    for (uint32_t i = threadIdx.x; i < nsub; i += EXPAND_THREADS) s_hist[i] = 0;
    __syncthreads();
    for (uint32_t s = 0; s < segs.n_seg; s++) {
        const uint64_t r0 = segs.rec_off[(uint64_t)s * (segs.P + 1) + pd.part], r1 = segs.rec_off[(uint64_t)s * (segs.P + 1) + pd.part + 1];
        const uint8_t* base = segs.rec[s];
        for (uint64_t r = r0 + threadIdx.x; r < r1; r += EXPAND_THREADS) {
            uint64_t R[RW]; load_rec<RW>(base, r, R);
            if constexpr (KW == 2 && RW == 4) {
                if (k >= 32) for_each_sub32(R, k, pd.sub_bits, [&](uint32_t sb) { atomicAdd(&s_hist[sb], 1u); });
                else for_each_kmer_fast<KW, RW>(R, k, [&](key_t c) { atomicAdd(&s_hist[sub_index<KW>(c, pd.shift)], 1u); });
            } else for_each_kmer_fast<KW, RW>(R, k, [&](key_t c) { atomicAdd(&s_hist[sub_index<KW>(c, pd.shift)], 1u); });
        }
    }
    __syncthreads();
    // exclusive scan of the counters -> absolute key offsets of the sub-buckets. Units of the scan: single sub-buckets padded to 4 slots
    // (pair scatter), or super-buckets of 4 consecutive sub-buckets padded to one 64-byte line (line scatter; sub_bits >= 2 there)
    const uint32_t grp = line_slots ? 4u : 1u, pad = line_slots ? line_slots - 1u : 3u;
    const uint32_t nunit = nsub / grp;
    const uint32_t per = (nunit + EXPAND_THREADS - 1) / EXPAND_THREADS;      // <= 16
    const uint32_t b = threadIdx.x * per;
    auto unit_size = [&](uint32_t u) { uint32_t z = 0; for (uint32_t d = 0; d < grp; d++) z += s_hist[u * grp + d]; return z; };
    uint32_t loc = 0;
    for (uint32_t i = 0; i < per; i++) if (b + i < nunit) loc += (unit_size(b + i) + pad) & ~pad;
    uint32_t x = loc;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { uint32_t y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
    if (lane == 63) s_wsum[wave] = x;
    __syncthreads();
    uint32_t wpre = 0;
    for (int w = 0; w < wave; w++) wpre += s_wsum[w];
    uint32_t run = wpre + x - loc;
    const uint32_t run0 = run;
    for (uint32_t i = 0; i < per; i++) if (b + i < nunit) {
        uint32_t o = run;
        for (uint32_t d = 0; d < grp; d++) {
            const uint32_t j = (b + i) * grp + d;
            b_start[pd.sub_base + j] = pd.key_base + o; b_n[pd.sub_base + j] = s_hist[j]; b_consumed[pd.sub_base + j] = (uint8_t)pd.sub_bits;
            o += s_hist[j];
        }
        run += (unit_size(b + i) + pad) & ~pad;
    }
    if (bins.nbmax == 0 || line_slots) return;
    // BINS (coarse scatter + in-LDS split, k_expand_coarse / k_bin_sort): consecutive sub-buckets whose first slots lie in the same window of BIN_WINDOW slots
    // form a bin; a sub-bucket beyond BIN_ISO keys is a bin of its own (it goes to the deeper tiers as it is), and bins do not cross multiples of
    // BIN_SUBS_MAX sub-buckets. So a bin spans < BIN_WINDOW + BIN_ISO slots and <= BIN_SUBS_MAX sub-buckets: it fits k_bin_sort's LDS whatever the skew.
    auto new_bin = [&](uint32_t j, uint32_t start_j) -> bool {
        if (j == 0 || (j & (BIN_SUBS_MAX - 1)) == 0 || s_hist[j] > BIN_ISO) return true;
        const uint32_t np = s_hist[j - 1];
        if (np > BIN_ISO) return true;
        const uint32_t start_p = start_j - ((np + 3u) & ~3u);
        return (start_j >> BIN_WINDOW_LOG) != (start_p >> BIN_WINDOW_LOG);
    };
    uint32_t floc = 0;
    { uint32_t o = run0; for (uint32_t i = 0; i < per; i++) if (b + i < nsub) { floc += new_bin(b + i, o) ? 1u : 0u; o += (s_hist[b + i] + 3u) & ~3u; } }
    uint32_t fx = floc;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { uint32_t y = __shfl_up(fx, d, 64); if (lane >= d) fx += y; }
    __syncthreads();                                             // s_wsum is reused
    if (lane == 63) s_wsum[wave] = fx;
    __syncthreads();
    uint32_t fpre = 0, ftot = 0;
    for (int w = 0; w < EXPAND_THREADS / 64; w++) { if (w < wave) fpre += s_wsum[w]; ftot += s_wsum[w]; }
    uint32_t bid = fpre + fx - floc;                             // bins that start before this thread's first sub-bucket
    uint32_t* first = bins.first + (uint64_t)bi * (bins.nbmax + 1);
    { uint32_t o = run0;
      for (uint32_t i = 0; i < per; i++) if (b + i < nsub) {
          const uint32_t j = b + i;
          if (new_bin(j, o)) { if (bid < bins.nbmax) first[bid] = j; bid++; }
          bins.of_sub[pd.sub_base + j] = (uint16_t)(bid - 1);
          o += (s_hist[j] + 3u) & ~3u;
      } }
    if (threadIdx.x == 0) {
        bins.n[bi] = ftot;
        if (ftot <= bins.nbmax) first[ftot] = nsub; else atomicOr(bins.bad, 1u);
    }
}

// ------------------------------------------------------------------------------------------------ B3 expand_scatter
template <int KW, int RW>
__global__ __launch_bounds__(EXPAND_THREADS) void k_expand_scatter(const PartDesc* __restrict__ parts, SegTable segs, uint32_t k,
                                                                    const uint64_t* __restrict__ b_start,
                                                                    typename KeyT<KW>::type* __restrict__ keys)
{
    typedef typename KeyT<KW>::type key_t;
    __shared__ uint32_t s_cur[MAX_SUB];
    const PartDesc pd = parts[blockIdx.x];
    const uint32_t nsub = 1u << pd.sub_bits;
    for (uint32_t i = threadIdx.x; i < nsub; i += EXPAND_THREADS) s_cur[i] = (uint32_t)(b_start[pd.sub_base + i] - pd.key_base);
    __syncthreads();
    key_t* out = keys + pd.key_base;
    for (uint32_t s = 0; s < segs.n_seg; s++) {
        const uint64_t r0 = segs.rec_off[(uint64_t)s * (segs.P + 1) + pd.part], r1 = segs.rec_off[(uint64_t)s * (segs.P + 1) + pd.part + 1];
        const uint8_t* base = segs.rec[s];
        for (uint64_t r = r0 + threadIdx.x; r < r1; r += EXPAND_THREADS) {
            uint64_t R[RW]; load_rec<RW>(base, r, R);
            for_each_kmer_fast<KW, RW>(R, k, [&](key_t c) {
                const uint32_t slot = atomicAdd(&s_cur[sub_index<KW>(c, pd.shift)], 1u);
                out[slot] = c;
            });
        }
    }
}

__device__ __forceinline__ uint32_t count_at(const uint8_t* cnt8, const uint32_t* cnt32, uint64_t slot, uint32_t b) { return b == 255u ? cnt32[slot] : b; }

// B1 (8-byte keys): same stream, but keys leave the workgroup in 16-byte PAIRS. A single 8-byte store to one of 8192 open
// sub-buckets never combines in L2 (measured: 3.3x write amplification, ~1 fabric write transaction of 32 B per key), so each
// sub-bucket has a one-key parking slot in LDS: a key either parks (CAS EMPTY -> key) or takes the parked key out (CAS key ->
// EMPTY) and the two are written with one aligned 16-byte store -> half the write transactions. Lock-free: every attempt
// either succeeds or lost to another thread's success, nobody waits on anybody. Leftover parked keys are flushed at the end.
constexpr int PAIR_THREADS = 1024;

// LDS cost model (tools/lds_bench, random slots in a 64 KB table, per wave instruction): read64 18 clk, cas64 21 clk, exch64 12 clk,
// add32 with return 10 clk, add32 without 6 clk — random 8-byte LDS accesses run at ~3-5 lanes/clk, so the protocol uses the cheapest
// primitive only: EXCHANGES. A thread holding key h first swaps EMPTY into the slot: a key came out -> the two leave as a pair.
// Nothing came out -> it swaps h in: EMPTY came out -> parked; a key came out (someone parked in between) -> it now holds
// that key instead and starts over. Keys are conserved by every exchange, nobody waits on anybody; 1.5 exchanges per key.
__global__ __launch_bounds__(PAIR_THREADS) void k_expand_scatter_pair(const PartDesc* __restrict__ parts, SegTable segs, uint32_t k,
                                                                       const uint64_t* __restrict__ b_start, uint64_t* __restrict__ keys,
                                                                       const uint32_t* __restrict__ only_if /* nullptr, or: run only when this word is set */,
                                                                       const uint32_t* __restrict__ order /* workgroup -> partition of the batch (largest first), or nullptr */)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long s_pend[];       // [nsub] parked key or EMPTY
    if (only_if && !*only_if) return;
    const PartDesc pd = parts[order ? order[blockIdx.x] : blockIdx.x];
    const uint32_t nsub = 1u << pd.sub_bits;
    uint32_t* s_cur = reinterpret_cast<uint32_t*>(s_pend + nsub);                     // [nsub] next free slot of the sub-bucket
    constexpr unsigned long long EMPTY = ~0ULL;
    for (uint32_t i = threadIdx.x; i < nsub; i += PAIR_THREADS) { s_pend[i] = EMPTY; s_cur[i] = (uint32_t)(b_start[pd.sub_base + i] - pd.key_base); }
    __syncthreads();
    uint64_t* out = keys + pd.key_base;
    for (uint32_t s = 0; s < segs.n_seg; s++) {
        const uint64_t r0 = segs.rec_off[(uint64_t)s * (segs.P + 1) + pd.part], r1 = segs.rec_off[(uint64_t)s * (segs.P + 1) + pd.part + 1];
        const ulonglong2* recs = reinterpret_cast<const ulonglong2*>(segs.rec[s]);
        uint64_t r = r0 + threadIdx.x;
        ulonglong2 nx = r < r1 ? recs[r] : make_ulonglong2(0, 0);
        for (; r < r1; r += PAIR_THREADS) {
            const uint64_t R[2] = {nx.x, nx.y};
            if (r + PAIR_THREADS < r1) nx = recs[r + PAIR_THREADS];                  // next record in flight while this one is expanded
            for_each_kmer16(R, k, [&](uint64_t c) {
                const uint32_t q = (uint32_t)(c >> pd.shift);
                unsigned long long h = c;
                for (;;) {
                    const unsigned long long y = atomicExch(&s_pend[q], EMPTY);
                    if (y != EMPTY) {
                        const uint32_t p = atomicAdd(&s_cur[q], 2u);
#ifdef GKC_EXP_NOSTORE
                        if (c == 0x123456789ULL)
#endif
                        *reinterpret_cast<ulonglong2*>(out + p) = make_ulonglong2(y, h);
                        break;
                    }
                    const unsigned long long z = atomicExch(&s_pend[q], h);
                    if (z == EMPTY) break;
                    h = z;
                }
            });
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nsub; i += PAIR_THREADS) { const unsigned long long v = s_pend[i]; if (v != EMPTY) out[s_cur[i]] = v; }
}

// B1, COARSE scatter (8-byte keys): the keys of a partition leave the expansion grouped by BIN (a run of consecutive sub-buckets, ~4096 slots:
// BinTables) instead of by sub-bucket, and in whole runs: every bin owns S = STAGE_KEYS / nbins staging slots in LDS; one round = every thread
// expands one record into the staging area (one LDS counter add per key), then the workgroup writes every bin's staged keys as ONE contiguous
// run at the bin's cursor (lanes <-> consecutive slots: 64..256-byte runs instead of 16-byte pairs — the scattered-store ceiling of this chip is
// set by the number of store requests, profiles/r02_scatter_store_calibration.txt). A key that finds its bin's slots full this round goes to its
// final slot directly (cursor + rank, the cursor only moves in the flush). Inside a bin the keys are in no particular order: k_bin_sort
// finishes the split by sub-bucket inside LDS, where a random access costs no HBM transaction.
constexpr int COARSE_THREADS = 1024;
constexpr uint32_t COARSE_STAGE_KEYS = 16384;        // 128 KB of staging
__global__ __launch_bounds__(COARSE_THREADS) void k_expand_coarse(const PartDesc* __restrict__ parts, SegTable segs, uint32_t k,
                                                                   const uint64_t* __restrict__ b_start, BinTables bins, uint64_t* __restrict__ keys,
                                                                   const uint32_t* __restrict__ order /* workgroup -> partition of the batch (largest first), or nullptr */)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long s_stage[];      // [COARSE_STAGE_KEYS]
    if (*bins.bad) return;
    const uint32_t bi = order ? order[blockIdx.x] : blockIdx.x;
    const PartDesc pd = parts[bi];
    const uint32_t nsub = 1u << pd.sub_bits;
    uint32_t* s_cnt = reinterpret_cast<uint32_t*>(s_stage + COARSE_STAGE_KEYS);        // [BIN_NBMAX] keys of the bin this round
    uint32_t* s_gcur = s_cnt + BIN_NBMAX;                                              // [BIN_NBMAX] next free slot of the bin (relative to key_base)
    uint16_t* s_map = reinterpret_cast<uint16_t*>(s_gcur + BIN_NBMAX);                 // [nsub] sub-bucket -> bin
    const uint32_t nbins = bins.n[bi];
    const uint32_t* first = bins.first + (uint64_t)bi * (bins.nbmax + 1);
    for (uint32_t i = threadIdx.x; i < nsub; i += COARSE_THREADS) s_map[i] = bins.of_sub[pd.sub_base + i];
    for (uint32_t i = threadIdx.x; i < nbins; i += COARSE_THREADS) { s_cnt[i] = 0; s_gcur[i] = (uint32_t)(b_start[pd.sub_base + first[i]] - pd.key_base); }
    const uint32_t S = nbins ? COARSE_STAGE_KEYS / nbins : COARSE_STAGE_KEYS;          // >= 16
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // flush geometry: S < 64: a wave instruction serves G = 64 / S bins (lane -> bin gi, slot fj); S >= 64: one bin, strided
    const uint32_t G = S < 64 ? 64u / S : 1u;
    const uint32_t gi = S < 64 ? (uint32_t)lane / S : 0u, fj = S < 64 ? (uint32_t)lane % S : (uint32_t)lane;
    __syncthreads();
    uint64_t* out = keys + pd.key_base;
    for (uint32_t s = 0; s < segs.n_seg; s++) {
        const uint64_t r0 = segs.rec_off[(uint64_t)s * (segs.P + 1) + pd.part], r1 = segs.rec_off[(uint64_t)s * (segs.P + 1) + pd.part + 1];
        const ulonglong2* recs = reinterpret_cast<const ulonglong2*>(segs.rec[s]);
        uint64_t r = r0 + threadIdx.x;
        ulonglong2 nx = r < r1 ? recs[r] : make_ulonglong2(0, 0);
        for (uint64_t base = r0; base < r1; base += COARSE_THREADS, r += COARSE_THREADS) {
            if (r < r1) {
                const uint64_t R[2] = {nx.x, nx.y};
                if (r + COARSE_THREADS < r1) nx = recs[r + COARSE_THREADS];            // next record in flight while this one is expanded
                for_each_kmer16(R, k, [&](uint64_t c) {
                    const uint32_t bn = s_map[(uint32_t)(c >> pd.shift)];
                    const uint32_t pos = atomicAdd(&s_cnt[bn], 1u);
                    if (pos < S) s_stage[bn * S + pos] = c;
                    else out[s_gcur[bn] + pos] = c;                                   // staging slots of the bin are full this round: straight to its slot
                });
            }
            __syncthreads();
            if (S < 64) {
                for (uint32_t bb = wave * G; bb < nbins; bb += (COARSE_THREADS / 64) * G) {
                    const uint32_t bn = bb + gi;
                    if (gi < G && bn < nbins) {
                        const uint32_t n = s_cnt[bn], g0 = s_gcur[bn];
                        if (fj < (n < S ? n : S)) out[g0 + fj] = s_stage[bn * S + fj];
                        if (fj == 0 && n) { s_gcur[bn] = g0 + n; s_cnt[bn] = 0; }
                    }
                }
            } else {
                for (uint32_t bn = wave; bn < nbins; bn += COARSE_THREADS / 64) {
                    const uint32_t n = s_cnt[bn], g0 = s_gcur[bn], mm = n < S ? n : S;
                    for (uint32_t j = fj; j < mm; j += 64) out[g0 + j] = s_stage[bn * S + j];
                    if (lane == 0 && n) { s_gcur[bn] = g0 + n; s_cnt[bn] = 0; }
                }
            }
            __syncthreads();
        }
    }
}

// B1, 16-byte keys in PAIRS: a single 16-byte store to one of 8192 open sub-buckets costs a whole 32-byte HBM write atom (twice the bytes),
// two keys leaving together fill it. Same exchange-only protocol as above with a 16-byte parking slot per sub-bucket, exchanged by ONE
// LDS instruction, ds_wrxchg2_rtn_b64 (two adjacent 8-byte words swapped per lane, returned together). Verified to behave as an atomic 128-bit
// exchange on gfx950 (tools/xchg128_check: 65536 threads hammering 1..8192 slots, no torn pair, values conserved); a torn pair would also
// surface as a checksum mismatch in every k > 31 parity test. EMPTY = high word all ones (a key's high word has at most 62 bits).
// LDS: 8192 x (16 + 4) B = 160 KB, all of the CU.
typedef unsigned int v4u_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void lds_xchg128(unsigned long long* slot, uint64_t in_lo, uint64_t in_hi, uint64_t& out_lo, uint64_t& out_hi)
{
    const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)slot;
    v4u_t r;
    asm volatile("ds_wrxchg2_rtn_b64 %0, %1, %2, %3 offset1:1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r) : "v"(addr), "v"(in_lo), "v"(in_hi) : "memory");
    out_lo = (uint64_t)r.x | ((uint64_t)r.y << 32); out_hi = (uint64_t)r.z | ((uint64_t)r.w << 32);
}
__global__ __launch_bounds__(PAIR_THREADS) void k_expand_scatter_pair2(const PartDesc* __restrict__ parts, SegTable segs, uint32_t k,
                                                                        const uint64_t* __restrict__ b_start, u128* __restrict__ keys,
                                                                        const uint32_t* __restrict__ order /* workgroup -> partition of the batch (largest first), or nullptr */)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long s_pend[];       // [nsub][2] parked key (low word, high word) or EMPTY
    const PartDesc pd = parts[order ? order[blockIdx.x] : blockIdx.x];
    const uint32_t nsub = 1u << pd.sub_bits;
    uint32_t* s_cur = reinterpret_cast<uint32_t*>(s_pend + 2 * (size_t)nsub);        // [nsub] next free slot of the sub-bucket
    constexpr unsigned long long EMPTY = ~0ULL;
    for (uint32_t i = threadIdx.x; i < nsub; i += PAIR_THREADS) { s_pend[2 * i] = EMPTY; s_pend[2 * i + 1] = EMPTY; s_cur[i] = (uint32_t)(b_start[pd.sub_base + i] - pd.key_base); }
    __syncthreads();
    ulonglong2* out = reinterpret_cast<ulonglong2*>(keys + pd.key_base);              // one 16-byte key per element (x = low word, y = high word)
    for (uint32_t s = 0; s < segs.n_seg; s++) {
        const uint64_t r0 = segs.rec_off[(uint64_t)s * (segs.P + 1) + pd.part], r1 = segs.rec_off[(uint64_t)s * (segs.P + 1) + pd.part + 1];
        const ulonglong2* recs = reinterpret_cast<const ulonglong2*>(segs.rec[s]);     // 32-byte records: two elements each
        uint64_t r = r0 + threadIdx.x;
        ulonglong2 nx0 = make_ulonglong2(0, 0), nx1 = nx0;
        if (r < r1) { nx0 = recs[2 * r]; nx1 = recs[2 * r + 1]; }
        for (; r < r1; r += PAIR_THREADS) {
            const uint64_t R[4] = {nx0.x, nx0.y, nx1.x, nx1.y};
            if (r + PAIR_THREADS < r1) { nx0 = recs[2 * (r + PAIR_THREADS)]; nx1 = recs[2 * (r + PAIR_THREADS) + 1]; }   // next record in flight
            for_each_kmer32(R, k, [&](u128 c) {
                const uint32_t q = sub_index<2>(c, pd.shift);
                uint64_t h_lo = (uint64_t)c, h_hi = (uint64_t)(c >> 64);
                for (;;) {
                    uint64_t y_lo, y_hi;
                    lds_xchg128(&s_pend[2 * (size_t)q], EMPTY, EMPTY, y_lo, y_hi);
                    if (y_hi != EMPTY) {
                        const uint32_t p = atomicAdd(&s_cur[q], 2u);
                        out[p] = make_ulonglong2(y_lo, y_hi); out[p + 1] = make_ulonglong2(h_lo, h_hi);
                        break;
                    }
                    uint64_t z_lo, z_hi;
                    lds_xchg128(&s_pend[2 * (size_t)q], h_lo, h_hi, z_lo, z_hi);
                    if (z_hi == EMPTY) break;
                    h_lo = z_lo; h_hi = z_hi;
                }
            });
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nsub; i += PAIR_THREADS) { const unsigned long long lo = s_pend[2 * i], hi = s_pend[2 * i + 1]; if (hi != EMPTY) out[s_cur[i]] = make_ulonglong2(lo, hi); }
}

// B1, 32-byte version. With the expansion itself cheap the pair kernel is bound by its 16-byte stores: a sub-bucket's next pair arrives
// long after its line left L2, so every pair costs a whole 32-byte HBM write atom (PMC: 2 x the algorithmic bytes). Here a sub-bucket
// parks up to THREE keys (slots s0..s2) and the fourth arrival leaves with all of them as one aligned 32-byte quad. That needs 32 B of
// LDS per sub-bucket, so a workgroup owns 4096 sub-buckets: a partition split 8192 ways is expanded by TWO workgroups, each keeping the
// k-mers of its half of the key range (the expansion is a fraction of the kernel). Exchange-only protocol, keys conserved by every step:
// deposit = swap the held key into s0, s1, s2 in turn until EMPTY comes out (then it is parked); three keys came out instead -> all
// three slots were full: swap EMPTY into the three slots and leave with what came out + the held key. Fewer than 3 came out (two
// collectors raced): those keys go one by one to the BACK of the bucket (quads fill it from the front, singles from the back; the bucket
// size is exact, the two cursors meet).
constexpr int QUAD_THREADS = 1024, QUAD_SUB = 4096;
__global__ __launch_bounds__(QUAD_THREADS) void k_expand_scatter_quad(const PartDesc* __restrict__ parts, SegTable segs, uint32_t k,
                                                                       const uint64_t* __restrict__ b_start, const uint32_t* __restrict__ b_n,
                                                                       uint64_t* __restrict__ keys, uint32_t halves_log2)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long s_slot[];       // [3][QUAD_SUB] parked keys or EMPTY
    uint32_t* s_front = reinterpret_cast<uint32_t*>(s_slot + 3 * QUAD_SUB);           // [QUAD_SUB] next quad of the sub-bucket
    uint32_t* s_back = s_front + QUAD_SUB;                                            // [QUAD_SUB] one past the next single (counts down)
    const PartDesc pd = parts[blockIdx.x >> halves_log2];
    const uint32_t nsub = 1u << pd.sub_bits;
    const uint32_t half = blockIdx.x & ((1u << halves_log2) - 1), q0 = half * QUAD_SUB;
    if (q0 >= nsub) return;
    const uint32_t nq = min(nsub - q0, (uint32_t)QUAD_SUB);
    constexpr unsigned long long EMPTY = ~0ULL;
    for (uint32_t i = threadIdx.x; i < QUAD_SUB; i += QUAD_THREADS) {
        s_slot[i] = EMPTY; s_slot[QUAD_SUB + i] = EMPTY; s_slot[2 * QUAD_SUB + i] = EMPTY;
        if (i < nq) { const uint32_t st = (uint32_t)(b_start[pd.sub_base + q0 + i] - pd.key_base); s_front[i] = st; s_back[i] = st + b_n[pd.sub_base + q0 + i]; }
    }
    __syncthreads();
    uint64_t* out = keys + pd.key_base;
    const bool all_mine = nsub <= (uint32_t)QUAD_SUB;
    for (uint32_t s = 0; s < segs.n_seg; s++) {
        const uint64_t r0 = segs.rec_off[(uint64_t)s * (segs.P + 1) + pd.part], r1 = segs.rec_off[(uint64_t)s * (segs.P + 1) + pd.part + 1];
        const ulonglong2* recs = reinterpret_cast<const ulonglong2*>(segs.rec[s]);
        uint64_t r = r0 + threadIdx.x;
        ulonglong2 nx = r < r1 ? recs[r] : make_ulonglong2(0, 0);
        for (; r < r1; r += QUAD_THREADS) {
            const uint64_t R[2] = {nx.x, nx.y};
            if (r + QUAD_THREADS < r1) nx = recs[r + QUAD_THREADS];                  // next record in flight while this one is expanded
            for_each_kmer16(R, k, [&](uint64_t c) {
                const uint32_t q = (uint32_t)(c >> pd.shift) - q0;
                if (!all_mine && q >= (uint32_t)QUAD_SUB) return;                     // the other workgroup's half of the key range
                unsigned long long h = c, z;
                z = atomicExch(&s_slot[q], h); if (z == EMPTY) return; h = z;
                z = atomicExch(&s_slot[QUAD_SUB + q], h); if (z == EMPTY) return; h = z;
                z = atomicExch(&s_slot[2 * QUAD_SUB + q], h); if (z == EMPTY) return; h = z;
                const unsigned long long a = atomicExch(&s_slot[q], EMPTY), b = atomicExch(&s_slot[QUAD_SUB + q], EMPTY),
                                         d = atomicExch(&s_slot[2 * QUAD_SUB + q], EMPTY);
                if (a != EMPTY && b != EMPTY && d != EMPTY) {
                    const uint32_t p = atomicAdd(&s_front[q], 4u);
#ifdef GKC_EXP_NOSTORE
                    if (c == 0x123456789ULL)
#endif
                    {   ulonglong2* o = reinterpret_cast<ulonglong2*>(out + p);
                        o[0] = make_ulonglong2(a, b); o[1] = make_ulonglong2(d, h); }
                } else {
                    const uint32_t m = 1u + (a != EMPTY) + (b != EMPTY) + (d != EMPTY);
                    uint32_t p = atomicSub(&s_back[q], m) - m;
                    out[p++] = h;
                    if (a != EMPTY) out[p++] = a;
                    if (b != EMPTY) out[p++] = b;
                    if (d != EMPTY) out[p++] = d;
                }
            });
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nq; i += QUAD_THREADS) {
        uint32_t p = s_back[i];
        for (int j = 0; j < 3; j++) { const unsigned long long v = s_slot[j * QUAD_SUB + i]; if (v != EMPTY) out[--p] = v; }
    }
}


// B1, 64-BYTE LINES. What bounds a scattered store on this chip is the number of store requests, not their bytes (tools/scatter_bench,
// profiles/r02_scatter_store_calibration.txt: 8192 open cursors per workgroup, useful GB/s by bytes per store: 16 B 984, 32 B 1211,
// 64 B 2470, 128 B 3024; a 16-byte store occupies a 32-byte write at the fabric, WRITE_SIZE 2.07x): the pair kernel above runs exactly
// at the 16-byte rate (96 GB of keys in 98 ms). A full 64-byte line per store needs 8 staged keys per open bucket, and 160 KB of LDS
// hold that for 2048 buckets, not for 8192. So the scatter places keys at SUPER-bucket granularity (the top sub_bits - 2 bits: 2048
// super-buckets of 4 sub-buckets) and the level-1 sort splits every super-bucket 4 ways through LDS (k_super_sort): the sub-bucket sizes
// are known from the 13-bit histogram of k_expand_count, so nothing else changes downstream.
// Staging protocol, wait-free and exchange-only like the pair kernel: a key takes a ticket (one LDS add on the bucket's control word,
// ticket in its top bits) and is swapped into slot ticket % 8; EMPTY came out -> parked. Ticket 7 is the collector: after its own
// deposit it swaps EMPTY into all 8 slots (4 x ds_wrxchg2) and leaves with the line as ONE aligned 64-byte store at the bucket's
// front cursor. A depositor that has its ticket but has not swapped yet (another wave) leaves a hole: the collector then writes what
// it got as single keys at the BACK of the bucket (bucket sizes are exact, front and back meet), and the late key is picked up by a later
// round; a key that comes out of a deposit (a late one of an earlier round) is inserted again with a new ticket. Every step conserves keys.
constexpr int LINE_THREADS = 1024, LINE_SUPER_MAX = 2048, LINE_SLOT_WORDS = 8;      // 8 x 8 bytes staged per super-bucket
template <int KW> struct LineT { static constexpr uint32_t KEYS = 8 / KW, TK_SHIFT = (KW == 1) ? 29 : 30, TK_ONE = 1u << TK_SHIFT, BACK_MASK = TK_ONE - 1; };
__device__ __forceinline__ void lds_take_line(unsigned long long* slot0, uint64_t (&a)[8])       // swap EMPTY into 8 consecutive words, return what was there
{
    const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)slot0;
    const uint64_t e = ~0ULL;
    v4u_t r0, r1, r2, r3;
    asm volatile("ds_wrxchg2_rtn_b64 %0, %4, %5, %5 offset0:0 offset1:1\n\t"
                 "ds_wrxchg2_rtn_b64 %1, %4, %5, %5 offset0:2 offset1:3\n\t"
                 "ds_wrxchg2_rtn_b64 %2, %4, %5, %5 offset0:4 offset1:5\n\t"
                 "ds_wrxchg2_rtn_b64 %3, %4, %5, %5 offset0:6 offset1:7\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(addr), "v"(e) : "memory");
    a[0] = (uint64_t)r0.x | ((uint64_t)r0.y << 32); a[1] = (uint64_t)r0.z | ((uint64_t)r0.w << 32);
    a[2] = (uint64_t)r1.x | ((uint64_t)r1.y << 32); a[3] = (uint64_t)r1.z | ((uint64_t)r1.w << 32);
    a[4] = (uint64_t)r2.x | ((uint64_t)r2.y << 32); a[5] = (uint64_t)r2.z | ((uint64_t)r2.w << 32);
    a[6] = (uint64_t)r3.x | ((uint64_t)r3.y << 32); a[7] = (uint64_t)r3.z | ((uint64_t)r3.w << 32);
}
template <int KW, int RW>
__global__ __launch_bounds__(LINE_THREADS) void k_expand_scatter_line(const PartDesc* __restrict__ parts, SegTable segs, uint32_t k,
                                                                       const uint64_t* __restrict__ b_start, const uint32_t* __restrict__ b_n,
                                                                       typename KeyT<KW>::type* __restrict__ keys)
{
    typedef typename KeyT<KW>::type key_t;
    typedef LineT<KW> LT;
    extern __shared__ __attribute__((aligned(16))) unsigned long long s_stage[];      // [LINE_SUPER_MAX][8] words: 8 keys of 8 bytes / 4 keys of 16 bytes (low, high)
    uint32_t* s_tb = reinterpret_cast<uint32_t*>(s_stage + (size_t)LINE_SUPER_MAX * LINE_SLOT_WORDS);   // [ticket : 3 or 2 | keys written at the back : 29 or 30]
    uint32_t* s_front = s_tb + LINE_SUPER_MAX;                                        // next line of the super-bucket (slot index relative to key_base)
    const PartDesc pd = parts[blockIdx.x];
    const uint32_t nsup = 1u << (pd.sub_bits - 2), sshift = pd.shift + 2;
    constexpr unsigned long long EMPTY = ~0ULL;
    for (uint32_t i = threadIdx.x; i < nsup * LINE_SLOT_WORDS; i += LINE_THREADS) s_stage[i] = EMPTY;
    for (uint32_t i = threadIdx.x; i < nsup; i += LINE_THREADS) { s_tb[i] = 0; s_front[i] = (uint32_t)(b_start[pd.sub_base + 4 * i] - pd.key_base); }
    __syncthreads();
    uint64_t* out = reinterpret_cast<uint64_t*>(keys + pd.key_base);                  // in 8-byte words: key slot p starts at word p * KW
    auto bucket_end = [&](uint32_t sb) -> uint32_t { return (uint32_t)(b_start[pd.sub_base + 4 * sb + 3] - pd.key_base) + b_n[pd.sub_base + 4 * sb + 3]; };
    auto put_singles = [&](uint32_t sb, const uint64_t (&a)[8], uint32_t m) {         // m keys (the non-EMPTY ones of a) to the back of the super-bucket
        const uint32_t used = atomicAdd(&s_tb[sb], m) & LT::BACK_MASK;
        uint32_t p = bucket_end(sb) - used - m;
        if constexpr (KW == 1) { for (int j = 0; j < 8; j++) if (a[j] != EMPTY) out[p++] = a[j]; }
        else { for (int j = 0; j < 4; j++) if (a[2 * j + 1] != EMPTY) { *reinterpret_cast<ulonglong2*>(out + 2 * (size_t)p) = make_ulonglong2(a[2 * j], a[2 * j + 1]); p++; } }
    };
    auto insert = [&](key_t c) {
        const uint32_t sb = sub_index<KW>(c, sshift);
        uint64_t h_lo = (uint64_t)c, h_hi = 0;
        if constexpr (KW == 2) h_hi = (uint64_t)(c >> 64);
        for (;;) {
            const uint32_t tk = atomicAdd(&s_tb[sb], LT::TK_ONE) >> LT::TK_SHIFT;
            uint64_t z_lo, z_hi;
            if constexpr (KW == 1) { z_lo = atomicExch(&s_stage[(size_t)sb * 8 + tk], (unsigned long long)h_lo); z_hi = z_lo; }
            else lds_xchg128(&s_stage[(size_t)sb * 8 + 2 * tk], h_lo, h_hi, z_lo, z_hi);
            const bool came_out = z_hi != EMPTY;                                     // EMPTY = all ones in the (high) word: no key has it
            if (tk != LT::KEYS - 1) { if (!came_out) return; h_lo = z_lo; h_hi = z_hi; continue; }
            uint64_t a[8];
            lds_take_line(&s_stage[(size_t)sb * 8], a);
            uint32_t m = 0;
            if constexpr (KW == 1) { for (int j = 0; j < 8; j++) m += a[j] != EMPTY; }
            else { for (int j = 0; j < 4; j++) m += a[2 * j + 1] != EMPTY; }
            if (m == LT::KEYS) {
                const uint32_t p = atomicAdd(&s_front[sb], LT::KEYS);
#ifdef GKC_EXP_NOSTORE
                if (h_lo == 0x123456789ULL)
#endif
                {   ulonglong2* o = reinterpret_cast<ulonglong2*>(out + (size_t)p * KW);
                    o[0] = make_ulonglong2(a[0], a[1]); o[1] = make_ulonglong2(a[2], a[3]); o[2] = make_ulonglong2(a[4], a[5]); o[3] = make_ulonglong2(a[6], a[7]); }
            } else if (m) put_singles(sb, a, m);
            if (!came_out) return;
            h_lo = z_lo; h_hi = z_hi;                                                // a late key of an earlier round sat in the collector's slot: insert it again
        }
    };
    for (uint32_t s = 0; s < segs.n_seg; s++) {
        const uint64_t r0 = segs.rec_off[(uint64_t)s * (segs.P + 1) + pd.part], r1 = segs.rec_off[(uint64_t)s * (segs.P + 1) + pd.part + 1];
        const ulonglong2* recs = reinterpret_cast<const ulonglong2*>(segs.rec[s]);
        uint64_t r = r0 + threadIdx.x;
        if constexpr (RW == 2) {
            ulonglong2 nx = r < r1 ? recs[r] : make_ulonglong2(0, 0);
            for (; r < r1; r += LINE_THREADS) {
                const uint64_t R[2] = {nx.x, nx.y};
                if (r + LINE_THREADS < r1) nx = recs[r + LINE_THREADS];                // next record in flight while this one is expanded
                for_each_kmer16(R, k, insert);
            }
        } else {
            ulonglong2 nx0 = make_ulonglong2(0, 0), nx1 = nx0;
            if (r < r1) { nx0 = recs[2 * r]; nx1 = recs[2 * r + 1]; }
            for (; r < r1; r += LINE_THREADS) {
                const uint64_t R[4] = {nx0.x, nx0.y, nx1.x, nx1.y};
                if (r + LINE_THREADS < r1) { nx0 = recs[2 * (r + LINE_THREADS)]; nx1 = recs[2 * (r + LINE_THREADS) + 1]; }
                for_each_kmer32(R, k, insert);
            }
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nsup; i += LINE_THREADS) {                     // what is still parked leaves as single keys
        uint64_t a[8]; uint32_t m = 0;
        for (int j = 0; j < 8; j++) a[j] = s_stage[(size_t)i * 8 + j];
        if constexpr (KW == 1) { for (int j = 0; j < 8; j++) m += a[j] != EMPTY; }
        else { for (int j = 0; j < 4; j++) m += a[2 * j + 1] != EMPTY; }
        if (m) put_singles(i, a, m);
    }
}

// ------------------------------------------------------------------------------------------------ B2/B3 wave sort + RLE
struct SortOut;
__device__ __forceinline__ void put_count(const SortOut& O, uint64_t slot, uint32_t c);
// One WAVE per sub-bucket, keys in registers (KPL per lane, blocked index e = lane*KPL + r), bitonic network with
// all-ascending comparators: in-lane steps are register compare-exchanges, cross-lane steps are lane-xor shuffles.
// No LDS traffic, no barrier; the load is striped (coalesced) because a sort does not care about the initial order.
constexpr int SORT_THREADS = 256;
constexpr int HIST_LDS = 64;

struct SortOut {
    uint8_t* cnt8;            // [n_slots] abundance of the distinct key written at the same slot, saturated at 255 (0 = empty slot)
    uint32_t* cnt32;          // [n_slots] full abundance, written (and later read) only where cnt8 == 255: the flag/abundance
                              // plane costs 1 byte per slot of HBM traffic instead of 4
    unsigned long long* histo; uint32_t histo_max;
    uint32_t* over_list; uint32_t* over_count;      // buckets too large for the first wave tier -> k_wave_sort_big
    uint32_t* over2_list; uint32_t* over2_count;    // buckets too large for the workgroup tier -> HBM split level
    uint32_t* over3_list; uint32_t* over3_count;    // buckets too large for the double-size wave network -> k_wg_sort
    unsigned long long* n_sorted;                   // [0] buckets sorted here [1] keys sorted here
};

__device__ __forceinline__ void put_count(const SortOut& O, uint64_t slot, uint32_t c)
{
    O.cnt8[slot] = (uint8_t)(c < 255u ? c : 255u);
    if (c >= 255u) O.cnt32[slot] = c;
}

// lane-xor exchange with a COMPILE-TIME mask. Masks that stay inside a 16-lane row and map onto a DPP control (xor 1, 2, 3 =
// quad_perm; xor 7 = row_half_mirror; xor 15 = row_mirror; xor 8 = row_ror:8) are VALU moves with no LDS-crossbar round trip;
// xor 4 (two banked row shifts), xor 16 / 32 (gfx950 v_permlane16_swap / v_permlane32_swap + select) and 31 / 63 are available
// behind GKC_PERMLANE_SWAP (verified on the GPU, tools/dpp_check/dpp_check.hip) but measured slower than ds_bpermute here.
#ifndef GKC_PERMLANE_SWAP
#define GKC_PERMLANE_SWAP 0      // measured: with the swaps the network turns VALU-bound and gets slower; ds_bpermute overlaps on the LDS pipe
#endif
template <int M> __device__ __forceinline__ uint32_t lane_xor32(uint32_t v)
{
    if constexpr (M == 1)       return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
    else if constexpr (M == 2)  return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
    else if constexpr (M == 3)  return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x1B, 0xF, 0xF, true);    // quad_perm [3,2,1,0]
    else if constexpr (M == 7)  return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xF, 0xF, true);   // row_half_mirror
    else if constexpr (M == 15) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x140, 0xF, 0xF, true);   // row_mirror
    else if constexpr (M == 8)  return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x128, 0xF, 0xF, true);   // row_ror:8
    else if constexpr (M == 4 && GKC_PERMLANE_SWAP) {                                                      // two banked row shifts
        const uint32_t r = __builtin_amdgcn_update_dpp(v, v, 0x104, 0xF, 0x5, false);                        // row_shl:4 -> banks 0,2
        return __builtin_amdgcn_update_dpp(r, v, 0x114, 0xF, 0xA, false);                                    // row_shr:4 -> banks 1,3
    } else if constexpr (M == 16 && GKC_PERMLANE_SWAP) {                                                     // gfx950 v_permlane16_swap
        const auto p = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        return (lane_id() & 16) ? p[0] : p[1];
    } else if constexpr (M == 32 && GKC_PERMLANE_SWAP) {                                                     // gfx950 v_permlane32_swap
        const auto p = __builtin_amdgcn_permlane32_swap(v, v, false, false);
        return (lane_id() & 32) ? p[0] : p[1];
    } else if constexpr (M == 31 && GKC_PERMLANE_SWAP) return lane_xor32<16>(lane_xor32<15>(v));
    else if constexpr (M == 63 && GKC_PERMLANE_SWAP) return lane_xor32<32>(lane_xor32<31>(v));
    else                        return (uint32_t)__shfl_xor((int)v, M, 64);
}
template <int KW> struct Shfl;
template <> struct Shfl<1> {
    template <int M> static __device__ __forceinline__ uint64_t x(uint64_t v) { return ((uint64_t)lane_xor32<M>((uint32_t)(v >> 32)) << 32) | lane_xor32<M>((uint32_t)v); }
    static __device__ __forceinline__ uint64_t up(uint64_t v) { return (uint64_t)__shfl_up((unsigned long long)v, 1, 64); }
    static __device__ __forceinline__ uint64_t down(uint64_t v) { return (uint64_t)__shfl_down((unsigned long long)v, 1, 64); }
};
template <> struct Shfl<2> {
    template <int M> static __device__ __forceinline__ u128 x(u128 v) {
        const uint64_t lo = Shfl<1>::x<M>((uint64_t)v), hi = Shfl<1>::x<M>((uint64_t)(v >> 64));
        return ((u128)hi << 64) | lo; }
    static __device__ __forceinline__ u128 up(u128 v) {
        unsigned long long lo = __shfl_up((unsigned long long)v, 1, 64), hi = __shfl_up((unsigned long long)(v >> 64), 1, 64);
        return ((u128)hi << 64) | lo; }
    static __device__ __forceinline__ u128 down(u128 v) {
        unsigned long long lo = __shfl_down((unsigned long long)v, 1, 64), hi = __shfl_down((unsigned long long)(v >> 64), 1, 64);
        return ((u128)hi << 64) | lo; }
};

#ifndef GKC_CROSS_MINMAX
#define GKC_CROSS_MINMAX 1     // cross-lane steps of tagged keys as exec-masked v_min_f64 / v_max_f64 blocks (0: 64-bit compare + selects)
#endif
// In-lane compare-exchange. F (8-byte keys only): the keys of the bucket carry the exponent tag of a double in their 12 top bits (see
// TAG64 below), so they are positive normal doubles whose order is the integer order, and the exchange is the two native 64-bit
// instructions v_min_f64 / v_max_f64 instead of a 64-bit compare and four selects. Cross-lane steps compare the same bit patterns as integers.
constexpr uint64_t TAG64 = 0x4330000000000000ULL, TAG64_MANT = 0x000FFFFFFFFFFFFFULL;
template <int KW, bool F> __device__ __forceinline__ void ce_inlane(typename KeyT<KW>::type& a, typename KeyT<KW>::type& b)
{
    if constexpr (F && KW == 1) {
        const double x = __longlong_as_double((long long)a), y = __longlong_as_double((long long)b);
        double lo, hi;
        asm("v_min_f64 %0, %1, %2" : "=v"(lo) : "v"(x), "v"(y));
        asm("v_max_f64 %0, %1, %2" : "=v"(hi) : "v"(x), "v"(y));
        a = (uint64_t)__double_as_longlong(lo); b = (uint64_t)__double_as_longlong(hi);
    } else { const typename KeyT<KW>::type x = a, y = b; const bool sw = y < x; a = sw ? y : x; b = sw ? x : y; }
}

// bitonic network over N = 64*KPL keys (blocked index e = lane*KPL + r), all comparators ascending, as compile-time recursion so
// that every lane mask is a template constant
template <int KW, int KPL, int S, bool F = false> struct HalfClean {                    // e <-> e ^ S, then S/2, ..., 1
    static __device__ __forceinline__ void run(typename KeyT<KW>::type (&v)[KPL], const int lane) {
        typedef typename KeyT<KW>::type key_t;
        if constexpr (S >= 1) {
            if constexpr (S < KPL) {
#pragma unroll
                for (int r = 0; r < KPL; r++) if ((r & S) == 0) ce_inlane<KW, F>(v[r], v[r | S]);
            } else {
                constexpr int LS = S / KPL;
                const bool low = (lane & LS) == 0;
                if constexpr (F && KW == 1 && GKC_CROSS_MINMAX) {
                    // tagged keys are doubles in integer order: the lower lane of a pair keeps min, the upper max. Two exec-masked blocks of native
                    // 64-bit min / max per group of keys instead of a 64-bit compare (SGPR result, wait state) + xor + two selects per key
                    constexpr int C = KPL < 4 ? KPL : 4;
#pragma unroll
                    for (int r0 = 0; r0 < KPL; r0 += C) {
                        double y[C];
#pragma unroll
                        for (int u = 0; u < C; u++) y[u] = __longlong_as_double((long long)Shfl<KW>::template x<LS>(v[r0 + u]));
                        if (low) {
#pragma unroll
                            for (int u = 0; u < C; u++) { double d; asm volatile("v_min_f64 %0, %1, %2" : "=v"(d) : "v"(__longlong_as_double((long long)v[r0 + u])), "v"(y[u])); v[r0 + u] = (uint64_t)__double_as_longlong(d); }
                        } else {
#pragma unroll
                            for (int u = 0; u < C; u++) { double d; asm volatile("v_max_f64 %0, %1, %2" : "=v"(d) : "v"(__longlong_as_double((long long)v[r0 + u])), "v"(y[u])); v[r0 + u] = (uint64_t)__double_as_longlong(d); }
                        }
                    }
                } else {
#pragma unroll
                for (int r = 0; r < KPL; r++) { const key_t y = Shfl<KW>::template x<LS>(v[r]); const bool ylt = y < v[r]; v[r] = (ylt == low) ? y : v[r]; }
                }
            }
            HalfClean<KW, KPL, S / 2, F>::run(v, lane);
        }
    }
};
template <int KW, int KPL, int SIZE, bool F = false> struct BitonicMerge {              // sorted runs of SIZE/2 -> sorted runs of SIZE
    static __device__ __forceinline__ void run(typename KeyT<KW>::type (&v)[KPL], const int lane) {
        typedef typename KeyT<KW>::type key_t;
        if constexpr (SIZE >= 2) {
            BitonicMerge<KW, KPL, SIZE / 2, F>::run(v, lane);
            // mirror step: e <-> e ^ (SIZE-1)
            if constexpr (SIZE <= KPL) {
#pragma unroll
                for (int r = 0; r < KPL; r++) { const int pr = r ^ (SIZE - 1); if (pr > r) ce_inlane<KW, F>(v[r], v[pr]); }
            } else {
                constexpr int LMASK = SIZE / KPL - 1, TOP = (SIZE / KPL) >> 1;
                const bool low = (lane & TOP) == 0;
                if constexpr (F && KW == 1 && GKC_CROSS_MINMAX) {
                    // partner of (lane, r) is (lane ^ LMASK, KPL-1-r): registers r and KPL-1-r are exchanged together, so nothing is overwritten early
                    constexpr int H = KPL >= 2 ? KPL / 2 : 1, C = H < 2 ? H : 2;
#pragma unroll
                    for (int r0 = 0; r0 < H; r0 += C) {
                        double ya[C], yb[C];
#pragma unroll
                        for (int u = 0; u < C; u++) {
                            ya[u] = __longlong_as_double((long long)Shfl<KW>::template x<LMASK>(v[KPL - 1 - (r0 + u)]));
                            yb[u] = __longlong_as_double((long long)Shfl<KW>::template x<LMASK>(v[r0 + u]));
                        }
                        if (low) {
#pragma unroll
                            for (int u = 0; u < C; u++) {
                                double d; asm volatile("v_min_f64 %0, %1, %2" : "=v"(d) : "v"(__longlong_as_double((long long)v[r0 + u])), "v"(ya[u])); v[r0 + u] = (uint64_t)__double_as_longlong(d);
                                if (KPL - 1 - (r0 + u) != r0 + u) { double e; asm volatile("v_min_f64 %0, %1, %2" : "=v"(e) : "v"(__longlong_as_double((long long)v[KPL - 1 - (r0 + u)])), "v"(yb[u])); v[KPL - 1 - (r0 + u)] = (uint64_t)__double_as_longlong(e); }
                            }
                        } else {
#pragma unroll
                            for (int u = 0; u < C; u++) {
                                double d; asm volatile("v_max_f64 %0, %1, %2" : "=v"(d) : "v"(__longlong_as_double((long long)v[r0 + u])), "v"(ya[u])); v[r0 + u] = (uint64_t)__double_as_longlong(d);
                                if (KPL - 1 - (r0 + u) != r0 + u) { double e; asm volatile("v_max_f64 %0, %1, %2" : "=v"(e) : "v"(__longlong_as_double((long long)v[KPL - 1 - (r0 + u)])), "v"(yb[u])); v[KPL - 1 - (r0 + u)] = (uint64_t)__double_as_longlong(e); }
                            }
                        }
                    }
                } else {
                key_t w[KPL];
#pragma unroll
                for (int r = 0; r < KPL; r++) { const key_t y = Shfl<KW>::template x<LMASK>(v[KPL - 1 - r]); const bool ylt = y < v[r]; w[r] = (ylt == low) ? y : v[r]; }
#pragma unroll
                for (int r = 0; r < KPL; r++) v[r] = w[r];
                }
            }
            HalfClean<KW, KPL, SIZE / 4, F>::run(v, lane);
        }
    }
};
template <int KW, int KPL, bool F = false>
__device__ __forceinline__ void bitonic_wave(typename KeyT<KW>::type (&v)[KPL], const int lane) { BitonicMerge<KW, KPL, 64 * KPL, F>::run(v, lane); }

// sort + run-length count one bucket of n <= 64*KPL keys held by one wave; writes distinct keys / abundances at
// outk[start + j], O.cnt[start + j] (j-th distinct key) — ascending; slots start+nd .. start+n-1 keep abundance 0.
template <int KW, int KPL, bool F = false>
__device__ __forceinline__ void wave_sort_bucket(const typename KeyT<KW>::type* src /* first key of the bucket: LDS or global */,
                                                 typename KeyT<KW>::type* __restrict__ outk,
                                                 const uint64_t start, const uint32_t n, const SortOut& O, uint32_t* s_hc, const int lane)
{
    typedef typename KeyT<KW>::type key_t;
    key_t v[KPL];
    key_t top = 0;                                               // F: the 12 top bits every key of the bucket shares (replaced by the tag while sorting)
#pragma unroll
    for (int r = 0; r < KPL; r++) { const uint32_t i = r * 64 + lane; v[r] = i < n ? src[i] : KeyT<KW>::max(); }
    if constexpr (F && KW == 1) {
        top = src[0] & ~TAG64_MANT;
#pragma unroll
        for (int r = 0; r < KPL; r++) v[r] = (v[r] & TAG64_MANT) | TAG64;          // padding (all ones) becomes TAG64_PAD: above every key
    }
#ifndef GKC_EXP_NOSORT
    bitonic_wave<KW, KPL, F>(v, lane);
#endif
    // run-length count (B3). e = lane*KPL + r is the sorted rank
    const key_t prev_last = Shfl<KW>::up(v[KPL - 1]);
    const key_t next_first = Shfl<KW>::down(v[0]);
    // whole-lane bit masks (bit r = rank lane*KPL + r): one compare per key, the head / tail logic on the masks
    uint32_t neq = (lane == 0 || v[0] != prev_last) ? 1u : 0u;                    // key differs from the one before it (rank 0: always)
#pragma unroll
    for (int r = 1; r < KPL; r++) neq |= (v[r] != v[r - 1] ? 1u : 0u) << r;
    const uint32_t lane0 = (uint32_t)lane * KPL;
    const uint32_t have = n > lane0 ? (n - lane0 < (uint32_t)KPL ? n - lane0 : (uint32_t)KPL) : 0u;     // ranks of this lane below n
    const uint32_t inm = have >= 32u ? 0xFFFFFFFFu : ((1u << have) - 1u);
    const uint32_t nxt_differs = (v[KPL - 1] != next_first) ? 1u : 0u;
    const uint32_t lastm = (have && lane0 + have == n) ? (1u << (have - 1u)) : 0u;                    // rank n-1 closes its run
    const uint32_t headm = neq & inm;
    const uint32_t tailm = inm & ((neq >> 1) | (nxt_differs << (KPL - 1)) | lastm);
    const uint32_t nt = __popc(tailm);
    int lh = headm ? (int)(lane * KPL + 31 - __clz((int)headm)) : -1;
    uint32_t x = nt; int hx = lh;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d, 64); const int hy = __shfl_up(hx, d, 64); if (lane >= d) { x += y; hx = hy > hx ? hy : hx; } }
    uint32_t idx = x - nt;
    int cur = __shfl_up(hx, 1, 64); if (lane == 0) cur = -1;
#pragma unroll
    for (int r = 0; r < KPL; r++) {
        if ((tailm >> r) & 1) {
            // the run that ends here started at the last head at or before r in this lane, else at the carried-in head of an earlier lane
            const uint32_t hb_ = headm & ((2u << r) - 1u);
            const uint32_t c = hb_ ? (uint32_t)(r - (31 - __clz((int)hb_)) + 1) : (uint32_t)((int)(lane * KPL + r) - cur + 1);
#ifndef GKC_EXP_NORLESTORE
            if constexpr (F && KW == 1) outk[start + idx] = (v[r] & TAG64_MANT) | top; else outk[start + idx] = v[r];
            put_count(O, start + idx, c);
#else
            if (c == 0x7fffffffu) outk[start + idx] = v[r];
#endif
            idx++;
#ifndef GKC_EXP_NOHIST
            const uint32_t hb = c >= O.histo_max ? O.histo_max : c;              // Histogram::inc (Histogram.hpp:92)
            if (hb < HIST_LDS) atomicAdd(&s_hc[hb], 1u); else atomicAdd(&O.histo[hb], 1ULL);
#endif
        }
    }
}

// register capacity of the wave tiers: k_wave_sort up to 1024 (u64) / 512 (u128) keys, k_wave_sort_big twice that
template <int KW> struct WaveCap { static constexpr int KPL_MAX = (KW == 1) ? 8 : 4; static constexpr uint32_t CAP = 64 * KPL_MAX; };
template <int KW> struct WaveCapBig { static constexpr int KPL_MAX = (KW == 1) ? 16 : 8; static constexpr uint32_t CAP = 64 * KPL_MAX; };

template <int KW, int KPLMAX, bool F = false>
__device__ __forceinline__ void wave_sort_dispatch(const typename KeyT<KW>::type* src, typename KeyT<KW>::type* __restrict__ outk, const uint64_t start,
                                                   const uint32_t n, const SortOut& O, uint32_t* s_hc, const int lane)
{
    if (n <= 64) wave_sort_bucket<KW, 1, F>(src, outk, start, n, O, s_hc, lane);
    else if (n <= 128) wave_sort_bucket<KW, 2, F>(src, outk, start, n, O, s_hc, lane);
    else if (n <= 256 || KPLMAX == 4) wave_sort_bucket<KW, 4, F>(src, outk, start, n, O, s_hc, lane);
    else if (n <= 512 || KPLMAX == 8) wave_sort_bucket<KW, (KPLMAX >= 8 ? 8 : 4), F>(src, outk, start, n, O, s_hc, lane);
    else wave_sort_bucket<KW, KPLMAX, F>(src, outk, start, n, O, s_hc, lane);
}

// one WAVE per small bucket, straight from HBM (no LDS, no barrier)
// Appending to a device-wide list through ONE counter word saturates at ~9e7 appends/s (MI355X_MICROARCH.md, "dequeue"): the sort
// kernels therefore collect the buckets they pass on in an LDS list per workgroup and reserve list space once per flush.
constexpr int WGLIST_CAP = 128;
struct WgList { uint32_t n; uint32_t base; uint32_t item[WGLIST_CAP]; };
__device__ __forceinline__ void wglist_flush(WgList* L, uint32_t* count, uint32_t* list)     // all threads of the workgroup
{
    __syncthreads();
    const uint32_t n = min(L->n, (uint32_t)WGLIST_CAP);
    if (threadIdx.x == 0 && n) L->base = atomicAdd(count, n);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) list[L->base + i] = L->item[i];
    __syncthreads();
    if (threadIdx.x == 0) L->n = 0;
    __syncthreads();
}
// one lane: keep g for the next flush; a full LDS list falls back to the direct append (rare)
__device__ __forceinline__ void wglist_push(WgList* L, uint32_t g, uint32_t* count, uint32_t* list)
{
    const uint32_t i = atomicAdd(&L->n, 1u);
    if (i < (uint32_t)WGLIST_CAP) L->item[i] = g;
    else { const uint32_t slot = atomicAdd(count, 1u); list[slot] = g; }
}

#ifndef GKC_WS_WAVES
#define GKC_WS_WAVES 5      // waves per SIMD the register budget is cut for: 3 (151 VGPRs) 98 ms, 4: 87 ms, 5: 84 ms, 6: 85 ms per 1.2e10 keys
#endif
#ifndef GKC_T1_MID
#define GKC_T1_MID 0          // 1: first tier stops at 512 / 256 keys (8 / 4 per lane), a KPL-16 / 8 instance of k_wave_sort_big takes the next class
#endif
template <int KW> struct WaveCapT1 { static constexpr int KPL_MAX = GKC_T1_MID ? WaveCap<KW>::KPL_MAX : WaveCapBig<KW>::KPL_MAX; static constexpr uint32_t CAP = 64 * KPL_MAX; };
template <int KW, bool F>
__global__ __launch_bounds__(SORT_THREADS, GKC_WS_WAVES) void k_wave_sort(const typename KeyT<KW>::type* __restrict__ src, typename KeyT<KW>::type* __restrict__ outk,
                                                             const uint64_t* __restrict__ b_start, const uint32_t* __restrict__ b_n, uint32_t n_buckets, SortOut O,
                                                             const uint32_t* __restrict__ only_if /* nullptr, or: run only when this word is set */)
{
    __shared__ uint32_t s_hc[HIST_LDS];
    __shared__ WgList s_over;
    if (only_if && !*only_if) return;
    const int t = threadIdx.x, lane = t & 63;
    if (t < HIST_LDS) s_hc[t] = 0;
    if (t == 0) s_over.n = 0;
    __syncthreads();
    const uint32_t wave = (blockIdx.x * SORT_THREADS + t) >> 6, n_waves = (gridDim.x * SORT_THREADS) >> 6;
    uint32_t nb_done = 0; unsigned long long nk_done = 0;
    uint32_t n_next = wave < n_buckets ? b_n[wave] : 0; uint64_t start_next = wave < n_buckets ? b_start[wave] : 0;
    for (uint32_t g = wave; g < n_buckets; g += n_waves) {
        const uint32_t n = n_next; const uint64_t start = start_next;
        if (g + n_waves < n_buckets) { n_next = b_n[g + n_waves]; start_next = b_start[g + n_waves]; }     // next bucket's descriptor in flight during this sort
        if (n == 0) continue;
        if (n > WaveCapT1<KW>::CAP) { if (lane == 0) wglist_push(&s_over, g, O.over_count, O.over_list); continue; }
        nb_done++; nk_done += n;
        wave_sort_dispatch<KW, WaveCapT1<KW>::KPL_MAX, F>(src + start, outk, start, n, O, s_hc, lane);
    }
    wglist_flush(&s_over, O.over_count, O.over_list);
    if (t < HIST_LDS && s_hc[t]) atomicAdd(&O.histo[t], (unsigned long long)s_hc[t]);
    if (lane == 0 && nb_done) { atomicAdd(&O.n_sorted[0], (unsigned long long)nb_done); atomicAdd(&O.n_sorted[1], nk_done); }
}

// Level 1 after the coarse scatter: one WORKGROUP per bin. The bin's keys (one contiguous run in HBM, in no particular order) are loaded once, coalesced,
// and dropped into their sub-bucket's range inside LDS (cursor per sub-bucket seeded with the exact offsets k_expand_count computed: one LDS add per key,
// no counting pass); then the waves take the bin's sub-buckets one after the other and sort them straight out of LDS with the register network of
// k_wave_sort — distinct keys / abundances go to the head of the sub-bucket's own slot range in HBM, as everywhere else. A sub-bucket beyond the first
// tier's registers is written back grouped and handed to the next tier by its index; a bin that is one such sub-bucket is not even loaded.
constexpr int BIN_THREADS = 512;
constexpr uint32_t BIN_KEYS_MAX = BIN_WINDOW + BIN_ISO;                           // slots a bin can span
__global__ void k_bin_prefix(const uint32_t* __restrict__ bin_n, uint32_t nb, uint32_t* __restrict__ bin_pre /* [nb + 1] */)
{
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_carry;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nb; base += 1024) {
        const uint32_t v = base + t < nb ? bin_n[base + t] : 0u;
        uint32_t x = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
        if (lane == 63) s_w[wave] = x;
        __syncthreads();
        uint32_t pre = s_carry;
        for (int w = 0; w < wave; w++) pre += s_w[w];
        if (base + t < nb) bin_pre[base + t] = pre + x - v;
        __syncthreads();
        if (t == 1023) s_carry = pre + x;
        __syncthreads();
    }
    if (t == 0) bin_pre[nb] = s_carry;
}
template <int KW, bool F>
__global__ __launch_bounds__(BIN_THREADS, 4) void k_bin_sort(const PartDesc* __restrict__ parts, typename KeyT<KW>::type* __restrict__ keys,
                                                             const uint64_t* __restrict__ b_start, const uint32_t* __restrict__ b_n, BinTables bins, SortOut O, uint32_t dbg)
{
    typedef typename KeyT<KW>::type key_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    key_t* s_keys = reinterpret_cast<key_t*>(s_raw);                                // [BIN_KEYS_MAX]
    uint32_t* s_cur = reinterpret_cast<uint32_t*>(s_keys + BIN_KEYS_MAX);           // [BIN_SUBS_MAX] fill cursor of the sub-bucket inside s_keys
    uint32_t* s_off = s_cur + BIN_SUBS_MAX;                                         // [BIN_SUBS_MAX] first slot of the sub-bucket inside s_keys
    uint32_t* s_n = s_off + BIN_SUBS_MAX;                                           // [BIN_SUBS_MAX] its keys
    __shared__ uint32_t s_hc[HIST_LDS];
    __shared__ WgList s_over;
    __shared__ uint32_t s_next, s_total;
    if (*bins.bad) return;
    const int t = threadIdx.x, lane = t & 63;
    if (t < HIST_LDS) s_hc[t] = 0;
    if (t == 0) s_over.n = 0;
    const PartDesc pd = parts[blockIdx.y];                                          // grid: (workgroups per partition, partitions of the batch)
    const uint32_t nbins = bins.n[blockIdx.y];
    const uint32_t* first = bins.first + (uint64_t)blockIdx.y * (bins.nbmax + 1);
    const uint64_t* p_start = b_start + pd.sub_base;
    const uint32_t* p_n = b_n + pd.sub_base;
    uint32_t nb_done = 0; unsigned long long nk_done = 0;
    for (uint32_t bn = blockIdx.x; bn < nbins; bn += gridDim.x) {
        const uint32_t j0 = first[bn], j1 = first[bn + 1], ns = j1 - j0;
        const uint64_t start0 = p_start[j0];
        if (ns == 1) {
            const uint32_t n = p_n[j0];
            if (n > WaveCapT1<KW>::CAP) { if (t == 0) wglist_push(&s_over, (uint32_t)(pd.sub_base + j0), O.over_count, O.over_list); continue; }
        }
        __syncthreads();                                                            // LDS of the previous bin fully consumed
        if (t == 0) { s_next = 0; s_total = 0; }
        __syncthreads();
        {   uint32_t mine = 0;
            for (uint32_t i = t; i < ns; i += BIN_THREADS) {
                const uint32_t o = (uint32_t)(p_start[j0 + i] - start0), n = p_n[j0 + i];
                s_cur[i] = o; s_off[i] = o; s_n[i] = n; mine += n;
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) mine += __shfl_down(mine, d, 64);
            if (lane == 0 && mine) atomicAdd(&s_total, mine);
        }
        __syncthreads();
        const uint32_t n = s_total;
        const key_t* src = keys + start0;
        if (!(dbg & 2u))
        for (uint32_t i0 = 0; i0 < n; i0 += 4 * BIN_THREADS) {
            key_t v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { const uint32_t i = i0 + u * BIN_THREADS + t; if (i < n) v[u] = src[i]; }
#pragma unroll
            for (int u = 0; u < 4; u++) { const uint32_t i = i0 + u * BIN_THREADS + t;
                if (i < n) { const uint32_t q = sub_index<KW>(v[u], pd.shift) - j0; const uint32_t pos = atomicAdd(&s_cur[q], 1u); s_keys[pos] = v[u]; } }
        }
        __syncthreads();
        for (;;) {
            uint32_t i = 0;
            if (lane == 0) i = atomicAdd(&s_next, 1u);
            i = __builtin_amdgcn_readfirstlane(i);
            if (i >= ns) break;
            const uint32_t sn = s_n[i];
            if (sn == 0) continue;
            const uint32_t off = s_off[i];
            const uint64_t sstart = start0 + off;
            if (sn > WaveCapT1<KW>::CAP) {                                          // next tier: grouped copy back, index to the list
                for (uint32_t x = lane; x < sn; x += 64) keys[sstart + x] = s_keys[off + x];
                if (lane == 0) wglist_push(&s_over, (uint32_t)(pd.sub_base + j0 + i), O.over_count, O.over_list);
                continue;
            }
            nb_done++; nk_done += sn;
            if (!(dbg & 1u)) wave_sort_dispatch<KW, WaveCapT1<KW>::KPL_MAX, F>(s_keys + off, keys, sstart, sn, O, s_hc, lane);
        }
    }
    wglist_flush(&s_over, O.over_count, O.over_list);
    if (t < HIST_LDS && s_hc[t]) atomicAdd(&O.histo[t], (unsigned long long)s_hc[t]);
    if (lane == 0 && nb_done) { atomicAdd(&O.n_sorted[0], (unsigned long long)nb_done); atomicAdd(&O.n_sorted[1], nk_done); }
}

// COUNT FIRST, SORT THE DISTINCT KEYS (8-byte keys). With 30x coverage two thirds of the keys of a bucket are repeats of a k-mer that is already there: the
// sort network above moves all of them through ~45 compare-exchange stages only to collapse them afterwards. Here the wave first counts its bucket in a small
// LDS hash table (512 slots per wave: 64-bit CAS on the key, 32-bit add on the count; linear probing), then compacts the distinct entries into one 64-bit word
// each — [the key's bits below the bucket prefix | abundance] — and runs the register network over those only (a 128 / 256 wide network instead of 512 / 1024).
// The order of the packed words is the order of the keys, the abundance rides along for free, no run-length pass. A bucket whose distinct k-mers do not fit
// the table (low coverage, repeats-free data) is handed to the plain sort tier by index.
constexpr int WH_SLOTS = 512, WH_WAVES = SORT_THREADS / 64;
template <bool F_UNUSED>
__global__ __launch_bounds__(SORT_THREADS, GKC_WS_WAVES) void k_wave_hash_count(const uint64_t* __restrict__ src, uint64_t* __restrict__ outk,
                                                                                 const uint64_t* __restrict__ b_start, const uint32_t* __restrict__ b_n, const uint8_t* __restrict__ b_cons,
                                                                                 uint32_t n_buckets, uint32_t two_k, SortOut O, uint32_t* __restrict__ plain_count, uint32_t* __restrict__ plain_list)
{
    __shared__ unsigned long long s_tk[WH_WAVES][WH_SLOTS];                      // keys (or, after the counting, the packed distinct entries)
    __shared__ uint32_t s_tc[WH_WAVES][WH_SLOTS];                                // counts
    __shared__ uint32_t s_hc[HIST_LDS];
    __shared__ WgList s_over, s_plain;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    if (t < HIST_LDS) s_hc[t] = 0;
    if (t == 0) { s_over.n = 0; s_plain.n = 0; }
    unsigned long long* tk = s_tk[wv]; uint32_t* tc = s_tc[wv];
    constexpr unsigned long long EMPTY = ~0ULL;
#pragma unroll
    for (int j = 0; j < WH_SLOTS / 64; j++) { tk[j * 64 + lane] = EMPTY; tc[j * 64 + lane] = 0; }
    __syncthreads();
    const uint32_t wave = (blockIdx.x * SORT_THREADS + t) >> 6, n_waves = (gridDim.x * SORT_THREADS) >> 6;
    for (uint32_t g = wave; g < n_buckets; g += n_waves) {
        const uint32_t n = b_n[g];
        if (n == 0) continue;
        if (n > WaveCapT1<1>::CAP) { if (lane == 0) wglist_push(&s_over, g, O.over_count, O.over_list); continue; }
        const uint64_t start = b_start[g];
        const uint32_t low_bits = two_k - b_cons[g];                             // key bits below the bucket's shared prefix
        if (low_bits > 53) { if (lane == 0) wglist_push(&s_plain, g, plain_count, plain_list); continue; }     // no room for an 11-bit abundance beside the key
        const uint32_t cnt_bits = 64 - low_bits;
        const uint64_t low_mask = (1ULL << low_bits) - 1;
        // ---- count: every key goes into the table
        bool fail = false;
        uint64_t top = 0;
        for (uint32_t base = 0; base < n; base += 256) {                          // four keys per lane in flight: the LDS round trips of their CAS chains overlap
            uint64_t key[4]; uint32_t slot[4]; uint32_t pending = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t i = base + q * 64 + lane;
                key[q] = i < n ? src[start + i] : 0;
                if (i < n) { pending |= 1u << q; top = key[q] & ~low_mask; }
                slot[q] = (uint32_t)(mix64(key[q]) >> 40) & (WH_SLOTS - 1);
            }
            for (uint32_t probes = 0; pending; probes++) {
                unsigned long long old[4];
#pragma unroll
                for (int q = 0; q < 4; q++) if ((pending >> q) & 1) old[q] = atomicCAS(&tk[slot[q]], EMPTY, (unsigned long long)key[q]);
#pragma unroll
                for (int q = 0; q < 4; q++) if ((pending >> q) & 1) {
                    if (old[q] == EMPTY || old[q] == key[q]) { atomicAdd(&tc[slot[q]], 1u); pending &= ~(1u << q); }
                    else slot[q] = (slot[q] + 1) & (WH_SLOTS - 1);
                }
                if (probes >= 48) { fail = true; break; }
            }
        }
        __builtin_amdgcn_wave_barrier();
        top = __shfl(top, 0, 64);                                                // lane 0 always holds a key (n >= 1)
        // ---- collect the distinct entries (and leave the table empty for the next bucket)
        uint64_t e[WH_SLOTS / 64]; uint32_t nd_lane = 0;
#pragma unroll
        for (int j = 0; j < WH_SLOTS / 64; j++) {
            const unsigned long long kk = tk[j * 64 + lane]; const uint32_t cc = tc[j * 64 + lane];
            e[j] = kk == EMPTY ? EMPTY : (((kk & low_mask) << cnt_bits) | cc);
            nd_lane += kk != EMPTY;
        }
#pragma unroll
        for (int j = 0; j < WH_SLOTS / 64; j++) { tk[j * 64 + lane] = EMPTY; tc[j * 64 + lane] = 0; }
        if (__any(fail)) {                                                        // too many distinct k-mers for the table: the plain sort tier takes the bucket
            if (lane == 0) wglist_push(&s_plain, g, plain_count, plain_list);
            continue;
        }
        uint32_t x = nd_lane;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
        const uint32_t nd = __shfl(x, 63, 64);
        uint32_t o = x - nd_lane;
#pragma unroll
        for (int j = 0; j < WH_SLOTS / 64; j++) if (e[j] != EMPTY) tk[o++] = e[j];      // packed entries, compact, in the (now empty) key table
        __builtin_amdgcn_wave_barrier();
        // ---- sort the distinct entries: 64 / 128 / 256 wide network on the packed words
        auto emit = [&](uint64_t pv, uint32_t j) {
            const uint32_t c = (uint32_t)(pv & ((1ULL << cnt_bits) - 1));
            outk[start + j] = top | (pv >> cnt_bits);
            put_count(O, start + j, c);
            const uint32_t hb = c >= O.histo_max ? O.histo_max : c;
            if (hb < HIST_LDS) atomicAdd(&s_hc[hb], 1u); else atomicAdd(&O.histo[hb], 1ULL);
        };
        if (nd <= 64) {
            uint64_t v[1] = { (uint32_t)lane < nd ? tk[lane] : EMPTY };
            bitonic_wave<1, 1, false>(v, lane);
            if ((uint32_t)lane < nd) emit(v[0], (uint32_t)lane);
        } else if (nd <= 128) {
            uint64_t v[2];
#pragma unroll
            for (int r = 0; r < 2; r++) { const uint32_t i = r * 64 + lane; v[r] = i < nd ? tk[i] : EMPTY; }
            bitonic_wave<1, 2, false>(v, lane);
#pragma unroll
            for (int r = 0; r < 2; r++) { const uint32_t j = lane * 2 + r; if (j < nd) emit(v[r], j); }
        } else if (nd <= 256) {
            uint64_t v[4];
#pragma unroll
            for (int r = 0; r < 4; r++) { const uint32_t i = r * 64 + lane; v[r] = i < nd ? tk[i] : EMPTY; }
            bitonic_wave<1, 4, false>(v, lane);
#pragma unroll
            for (int r = 0; r < 4; r++) { const uint32_t j = lane * 4 + r; if (j < nd) emit(v[r], j); }
        } else {
            uint64_t v[8];
#pragma unroll
            for (int r = 0; r < 8; r++) { const uint32_t i = r * 64 + lane; v[r] = i < nd ? tk[i] : EMPTY; }
            bitonic_wave<1, 8, false>(v, lane);
#pragma unroll
            for (int r = 0; r < 8; r++) { const uint32_t j = lane * 8 + r; if (j < nd) emit(v[r], j); }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < WH_SLOTS / 64; j++) tk[j * 64 + lane] = EMPTY;       // the compacted entries go, the table is empty again
    }
    wglist_flush(&s_over, O.over_count, O.over_list);
    wglist_flush(&s_plain, plain_count, plain_list);
    if (t < HIST_LDS && s_hc[t]) atomicAdd(&O.histo[t], (unsigned long long)s_hc[t]);
}
// the buckets k_wave_hash_count handed back: plain sort network (first tier) on a list of bucket indices
template <int KW, bool F>
__global__ __launch_bounds__(SORT_THREADS, GKC_WS_WAVES) void k_wave_sort_list(const typename KeyT<KW>::type* __restrict__ src, typename KeyT<KW>::type* __restrict__ outk,
                                                                  const uint64_t* __restrict__ b_start, const uint32_t* __restrict__ b_n, const uint32_t* __restrict__ list, uint32_t n_list, SortOut O)
{
    __shared__ uint32_t s_hc[HIST_LDS];
    const int t = threadIdx.x, lane = t & 63;
    if (t < HIST_LDS) s_hc[t] = 0;
    __syncthreads();
    const uint32_t wave = (blockIdx.x * SORT_THREADS + t) >> 6, n_waves = (gridDim.x * SORT_THREADS) >> 6;
    for (uint32_t li = wave; li < n_list; li += n_waves) {
        const uint32_t g = list[li]; const uint32_t n = b_n[g]; const uint64_t start = b_start[g];
        wave_sort_dispatch<KW, WaveCapT1<KW>::KPL_MAX, F>(src + start, outk, start, n, O, s_hc, lane);
    }
    __syncthreads();
    if (t < HIST_LDS && s_hc[t]) atomicAdd(&O.histo[t], (unsigned long long)s_hc[t]);
}

// second tier: buckets up to twice the first tier's size (2048 / 1024 keys), one wave each with a double-size network; only
// ~10 % of the keys come here, so the lower occupancy of this kernel (64+ key registers) does not touch the first tier
template <int KW> struct WaveCapHuge { static constexpr int KPL = (KW == 1) ? 32 : 16; static constexpr uint32_t CAP = 64 * KPL; };
#ifndef GKC_WSB_WAVES
#define GKC_WSB_WAVES 3     // 2 (214 VGPRs): 16.3 ms, 3: 13.4 ms, 4: 13.2 ms
#endif
template <int KW, bool F, int KPL>
__global__ __launch_bounds__(SORT_THREADS, GKC_WSB_WAVES) void k_wave_sort_big(const typename KeyT<KW>::type* __restrict__ src, typename KeyT<KW>::type* __restrict__ outk,
                                                                 const uint64_t* __restrict__ b_start, const uint32_t* __restrict__ b_n,
                                                                 const uint32_t* __restrict__ list, uint32_t n_list, uint32_t n_min, uint32_t n_report /* buckets beyond go to the next list */, SortOut O)
{
    __shared__ uint32_t s_hc[HIST_LDS];
    __shared__ WgList s_over;
    const int t = threadIdx.x, lane = t & 63;
    if (t < HIST_LDS) s_hc[t] = 0;
    if (t == 0) s_over.n = 0;
    __syncthreads();
    const uint32_t wave = (blockIdx.x * SORT_THREADS + t) >> 6, n_waves = (gridDim.x * SORT_THREADS) >> 6;
    for (uint32_t li = wave; li < n_list; li += n_waves) {
        const uint32_t g = list[li];
        const uint32_t n = b_n[g];
        const uint64_t start = b_start[g];
        if (n_report && n > n_report) { if (lane == 0) wglist_push(&s_over, g, O.over3_count, O.over3_list); continue; }
        if (n <= n_min || n > 64u * KPL) continue;                 // another instance's class
        wave_sort_bucket<KW, KPL, F>(src + start, outk, start, n, O, s_hc, lane);
    }
    wglist_flush(&s_over, O.over3_count, O.over3_list);
    if (t < HIST_LDS && s_hc[t]) atomicAdd(&O.histo[t], (unsigned long long)s_hc[t]);
}


// Level 1 after the line scatter: one WORKGROUP of 4 waves per super-bucket (4 sub-buckets whose keys arrive mixed). The keys are loaded
// once (coalesced), split 4 ways through LDS on the two key bits below the super-bucket index — positions from wave ballots, the group
// bases are the sub-bucket sizes k_expand_count already knows, so no counting pass and no LDS atomics — and wave w sorts sub-bucket w
// straight out of LDS with the same register network as k_wave_sort; distinct keys / abundances go to the head of the sub-bucket's own
// slot range. A sub-bucket beyond the first tier's registers is written back in place and handed to the next tier by its index, a
// super-bucket beyond the LDS buffer goes to k_super_split_big: everything downstream still works on the 13-bit sub-buckets.
constexpr int SS_THREADS = 256;
template <int KW> struct SuperCap { static constexpr int RPT = (KW == 1) ? 12 : 6; static constexpr uint32_t CAP = RPT * SS_THREADS; };   // 3072 / 1536 keys: 24 KB of LDS
#ifndef GKC_SS_WAVES
#define GKC_SS_WAVES 4
#endif
template <int KW, bool F>
__global__ __launch_bounds__(SS_THREADS, GKC_SS_WAVES) void k_super_sort(const typename KeyT<KW>::type* __restrict__ src, typename KeyT<KW>::type* __restrict__ outk,
                                                                const uint64_t* __restrict__ b_start, const uint32_t* __restrict__ b_n, const uint8_t* __restrict__ b_cons,
                                                                uint32_t n_super, uint32_t two_k, SortOut O, uint32_t* __restrict__ big_count, uint32_t* __restrict__ big_list)
{
    typedef typename KeyT<KW>::type key_t;
    constexpr int RPT = SuperCap<KW>::RPT;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    key_t* s_keys = reinterpret_cast<key_t*>(s_raw);
    __shared__ uint32_t s_hc[HIST_LDS];
    __shared__ WgList s_over;
    __shared__ uint32_t s_wcnt[4][4];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (t < HIST_LDS) s_hc[t] = 0;
    if (t == 0) s_over.n = 0;
    __syncthreads();
    uint32_t nb_done = 0; unsigned long long nk_done = 0;
    for (uint32_t g = blockIdx.x; g < n_super; g += gridDim.x) {
        const uint4 nn = *reinterpret_cast<const uint4*>(b_n + 4 * (size_t)g);
        const uint32_t N = nn.x + nn.y + nn.z + nn.w;
        if (N == 0) continue;
        if (N > SuperCap<KW>::CAP) { if (t == 0) { const uint32_t slot = atomicAdd(big_count, 1u); big_list[slot] = g; } continue; }
        const uint64_t S = b_start[4 * (size_t)g];
        const uint32_t shift = two_k - b_cons[4 * (size_t)g];
        key_t v[RPT];
        uint32_t wc0 = 0, wc1 = 0, wc2 = 0;                       // keys of this wave per group (the fourth is the rest)
        uint32_t n_mine = 0;
#pragma unroll
        for (int r = 0; r < RPT; r++) {
            const uint32_t i = (uint32_t)r * SS_THREADS + t;
            const bool in = i < N;
            v[r] = in ? src[S + i] : (key_t)0;
            const uint32_t d = in ? (sub_index<KW>(v[r], shift) & 3u) : 4u;
            wc0 += (uint32_t)__popcll(__ballot(d == 0)); wc1 += (uint32_t)__popcll(__ballot(d == 1)); wc2 += (uint32_t)__popcll(__ballot(d == 2));
            n_mine += (uint32_t)__popcll(__ballot(in));
        }
        if (lane == 0) { s_wcnt[w][0] = wc0; s_wcnt[w][1] = wc1; s_wcnt[w][2] = wc2; s_wcnt[w][3] = n_mine - wc0 - wc1 - wc2; }
        __syncthreads();
        const uint32_t goff[4] = {0u, nn.x, nn.x + nn.y, nn.x + nn.y + nn.z};
        uint32_t wb0 = goff[0], wb1 = goff[1], wb2 = goff[2], wb3 = goff[3];           // where this wave's keys of each group start in the LDS buffer
        for (int ww = 0; ww < 4; ww++) if (ww < w) { wb0 += s_wcnt[ww][0]; wb1 += s_wcnt[ww][1]; wb2 += s_wcnt[ww][2]; wb3 += s_wcnt[ww][3]; }
        if (t < 4) {                                                                 // the 13-bit histogram and the keys that arrived must agree
            const uint32_t tot = s_wcnt[0][t] + s_wcnt[1][t] + s_wcnt[2][t] + s_wcnt[3][t];
            const uint32_t want = t == 0 ? nn.x : (t == 1 ? nn.y : (t == 2 ? nn.z : nn.w));
            if (tot != want) atomicAdd(&O.n_sorted[2], 1ULL);
        }
#pragma unroll
        for (int r = 0; r < RPT; r++) {
            const uint32_t i = (uint32_t)r * SS_THREADS + t;
            const uint32_t d = i < N ? (sub_index<KW>(v[r], shift) & 3u) : 4u;
            const uint64_t m0 = __ballot(d == 0), m1 = __ballot(d == 1), m2 = __ballot(d == 2), m3 = __ballot(d == 3);
            const uint64_t mine = d == 0 ? m0 : (d == 1 ? m1 : (d == 2 ? m2 : m3));
            const uint32_t base = d == 0 ? wb0 : (d == 1 ? wb1 : (d == 2 ? wb2 : wb3));
            const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(mine >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mine, 0u));
            if (d < 4) s_keys[base + below] = v[r];
            wb0 += (uint32_t)__popcll(m0); wb1 += (uint32_t)__popcll(m1); wb2 += (uint32_t)__popcll(m2); wb3 += (uint32_t)__popcll(m3);
        }
        __syncthreads();
        const uint32_t n = w == 0 ? nn.x : (w == 1 ? nn.y : (w == 2 ? nn.z : nn.w));
        const uint32_t go = w == 0 ? goff[0] : (w == 1 ? goff[1] : (w == 2 ? goff[2] : goff[3]));
        if (n) {
            const uint64_t start = S + go;                                             // == b_start[4g + w]: sub-buckets are packed inside the super-bucket
            if (n > WaveCapT1<KW>::CAP) {
                for (uint32_t i = lane; i < n; i += 64) outk[start + i] = s_keys[go + i];
                if (lane == 0) wglist_push(&s_over, 4 * g + (uint32_t)w, O.over_count, O.over_list);
            } else {
                nb_done++; nk_done += n;
                wave_sort_dispatch<KW, WaveCapT1<KW>::KPL_MAX, F>(s_keys + go, outk, start, n, O, s_hc, lane);
            }
        }
        __syncthreads();                                                               // the LDS buffer is free for the next super-bucket
    }
    wglist_flush(&s_over, O.over_count, O.over_list);
    if (t < HIST_LDS && s_hc[t]) atomicAdd(&O.histo[t], (unsigned long long)s_hc[t]);
    if (lane == 0 && nb_done) { atomicAdd(&O.n_sorted[0], (unsigned long long)nb_done); atomicAdd(&O.n_sorted[1], nk_done); }
}
// super-buckets beyond the LDS buffer (hot key prefixes): split 4 ways out of place (keys -> ping-pong buffer at the sub-buckets' own
// offsets), copied back by k_super_copy_back; their 4 sub-buckets join the next tier's list
template <int KW>
__global__ __launch_bounds__(SS_THREADS) void k_super_split_big(const typename KeyT<KW>::type* __restrict__ src, typename KeyT<KW>::type* __restrict__ dst,
                                                                 const uint64_t* __restrict__ b_start, const uint32_t* __restrict__ b_n, const uint8_t* __restrict__ b_cons,
                                                                 const uint32_t* __restrict__ big_list, uint32_t two_k, SortOut O)
{
    typedef typename KeyT<KW>::type key_t;
    __shared__ uint32_t s_cur[4];
    const uint32_t g = big_list[blockIdx.x];
    const uint4 nn = *reinterpret_cast<const uint4*>(b_n + 4 * (size_t)g);
    const uint32_t N = nn.x + nn.y + nn.z + nn.w;
    const uint64_t S = b_start[4 * (size_t)g];
    const uint32_t shift = two_k - b_cons[4 * (size_t)g];
    if (threadIdx.x == 0) { s_cur[0] = 0; s_cur[1] = nn.x; s_cur[2] = nn.x + nn.y; s_cur[3] = nn.x + nn.y + nn.z; }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    for (uint32_t i0 = 0; i0 < N; i0 += SS_THREADS) {
        const uint32_t i = i0 + threadIdx.x;
        const bool in = i < N;
        const key_t v = in ? src[S + i] : (key_t)0;
        const uint32_t d = in ? (sub_index<KW>(v, shift) & 3u) : 4u;
#pragma unroll
        for (uint32_t dd = 0; dd < 4; dd++) {
            const uint64_t m = __ballot(d == dd);
            if (!m) continue;
            uint32_t base = 0;
            if (lane == __ffsll((long long)m) - 1) base = atomicAdd(&s_cur[dd], (uint32_t)__popcll(m));
            base = __shfl(base, __ffsll((long long)m) - 1, 64);
            if (d == dd) dst[S + base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = v;
        }
    }
    if (threadIdx.x == 0) { const uint32_t slot = atomicAdd(O.over_count, 4u); for (uint32_t d = 0; d < 4; d++) O.over_list[slot + d] = 4 * g + d; }
}
template <int KW>
__global__ __launch_bounds__(SS_THREADS) void k_super_copy_back(typename KeyT<KW>::type* __restrict__ keys, const typename KeyT<KW>::type* __restrict__ from,
                                                                 const uint64_t* __restrict__ b_start, const uint32_t* __restrict__ b_n, const uint32_t* __restrict__ big_list)
{
    const uint32_t g = big_list[blockIdx.x];
    const uint4 nn = *reinterpret_cast<const uint4*>(b_n + 4 * (size_t)g);
    const uint32_t N = nn.x + nn.y + nn.z + nn.w;
    const uint64_t S = b_start[4 * (size_t)g];
    for (uint32_t i = threadIdx.x; i < N; i += SS_THREADS) keys[S + i] = from[S + i];
}

// Buckets beyond one wave's registers: a WORKGROUP of NW waves holds the bucket in registers (64*KPL keys per wave). Every wave
// runs the wave network on its part; the merges across waves are the same bitonic steps with the partner in another wave, exchanged
// through LDS at the SAME (register, lane) coordinate (mirror step: the reflected one) — consecutive lanes touch consecutive LDS
// words, no bank conflict, and only log2(NW)*(log2(NW)+1)/2 of the stages cross waves (3 for 4 waves, 6 for 8); the half-cleaners
// below 64*KPL stay inside the waves. Then one run-length count across the workgroup. No data-dependent LDS traffic at all
// (tools/lds_bench: a random 8-byte LDS access costs 5x a conflict-free one), unlike a split inside LDS.
template <int KW, int NW, int KPL, bool F>
__global__ __launch_bounds__(NW * 64, 2) void k_wg_sort(const typename KeyT<KW>::type* __restrict__ src, typename KeyT<KW>::type* __restrict__ outk,
                                                        const uint64_t* __restrict__ b_start, const uint32_t* __restrict__ b_n,
                                                        const uint32_t* __restrict__ list, uint32_t n_list, uint32_t n_min, uint32_t n_max_all /* 0, or: report buckets beyond every tier */, SortOut O)
{
    typedef typename KeyT<KW>::type key_t;
    constexpr uint32_t CAPW = 64 * KPL, CAP = NW * CAPW;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    key_t* s_x = reinterpret_cast<key_t*>(s_raw);                 // [NW][KPL][64] exchange buffer
    __shared__ key_t s_first[NW], s_last[NW];
    __shared__ uint32_t s_tails[NW]; __shared__ int s_head[NW];
    __shared__ uint32_t s_hc[HIST_LDS];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (t < HIST_LDS) s_hc[t] = 0;
    for (uint32_t li = blockIdx.x; li < n_list; li += gridDim.x) {
        const uint32_t g = list[li];
        const uint32_t n = b_n[g];
        if (n_max_all && n > n_max_all && t == 0) { const uint32_t slot = atomicAdd(O.over2_count, 1u); O.over2_list[slot] = g; }
        if (n <= n_min || n > CAP) continue;                      // another tier's bucket
        const uint64_t start = b_start[g];
        key_t v[KPL];
#pragma unroll
        for (int r = 0; r < KPL; r++) { const uint32_t i = w * CAPW + r * 64 + lane; v[r] = i < n ? src[start + i] : KeyT<KW>::max(); }
        key_t top = 0;
        if constexpr (F && KW == 1) {
            top = src[start] & ~TAG64_MANT;
#pragma unroll
            for (int r = 0; r < KPL; r++) v[r] = (v[r] & TAG64_MANT) | TAG64;
        }
        bitonic_wave<KW, KPL, F>(v, lane);
        key_t* mine = s_x + (size_t)w * CAPW;
#pragma unroll
        for (int SZ = 2; SZ <= NW; SZ <<= 1) {                    // merge sorted runs of SZ/2 waves into runs of SZ waves
            {   // mirror step: element E <-> E ^ (SZ*CAPW - 1): wave w ^ (SZ-1), register KPL-1-r, lane 63-lane
                __syncthreads();
#pragma unroll
                for (int r = 0; r < KPL; r++) mine[r * 64 + lane] = v[r];
                __syncthreads();
                const key_t* other = s_x + (size_t)(w ^ (SZ - 1)) * CAPW;
                const bool low = (w & (SZ >> 1)) == 0;
#pragma unroll
                for (int r = 0; r < KPL; r++) { const key_t y = other[(KPL - 1 - r) * 64 + (63 - lane)]; const bool ylt = y < v[r]; v[r] = (ylt == low) ? y : v[r]; }
            }
#pragma unroll
            for (int S = SZ >> 2; S >= 1; S >>= 1) {              // half-cleaners whose partner is another wave: w ^ S, same register and lane
                __syncthreads();
#pragma unroll
                for (int r = 0; r < KPL; r++) mine[r * 64 + lane] = v[r];
                __syncthreads();
                const key_t* other = s_x + (size_t)(w ^ S) * CAPW;
                const bool low = (w & S) == 0;
#pragma unroll
                for (int r = 0; r < KPL; r++) { const key_t y = other[r * 64 + lane]; const bool ylt = y < v[r]; v[r] = (ylt == low) ? y : v[r]; }
            }
            HalfClean<KW, KPL, CAPW / 2, F>::run(v, lane);        // the rest of the merge stays inside the wave
        }
        // run-length count across the workgroup; E = w*CAPW + lane*KPL + r is the sorted rank
        if (lane == 0) s_first[w] = v[0];
        if (lane == 63) s_last[w] = v[KPL - 1];
        __syncthreads();
        key_t prev_last = Shfl<KW>::up(v[KPL - 1]), next_first = Shfl<KW>::down(v[0]);
        if (lane == 0 && w > 0) prev_last = s_last[w - 1];
        if (lane == 63 && w < NW - 1) next_first = s_first[w + 1];
        const uint32_t E0 = w * CAPW + lane * KPL;
        uint32_t headm = 0, tailm = 0;
#pragma unroll
        for (int r = 0; r < KPL; r++) {
            const uint32_t e = E0 + r;
            const key_t pv = r ? v[r - 1] : prev_last;
            const key_t nx = (r < KPL - 1) ? v[r + 1] : next_first;
            const bool in = e < n;
            headm |= (uint32_t)(in && (e == 0 || v[r] != pv)) << r;
            tailm |= (uint32_t)(in && (e == n - 1 || v[r] != nx)) << r;
        }
        const uint32_t nt = __popc(tailm);
        int lh = headm ? (int)(E0 + 31 - __clz((int)headm)) : -1;
        uint32_t x = nt; int hx = lh;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d, 64); const int hy = __shfl_up(hx, d, 64); if (lane >= d) { x += y; hx = hy > hx ? hy : hx; } }
        if (lane == 63) { s_tails[w] = x; s_head[w] = hx; }
        __syncthreads();
        uint32_t idx = x - nt; int cur = __shfl_up(hx, 1, 64); if (lane == 0) cur = -1;
        for (int ww = 0; ww < w; ww++) { idx += s_tails[ww]; cur = s_head[ww] > cur ? s_head[ww] : cur; }
#pragma unroll
        for (int r = 0; r < KPL; r++) {
            const int e = (int)E0 + r;
            if ((headm >> r) & 1) cur = e;
            if ((tailm >> r) & 1) {
                const uint32_t c = (uint32_t)(e - cur + 1);
                if constexpr (F && KW == 1) outk[start + idx] = (v[r] & TAG64_MANT) | top; else outk[start + idx] = v[r];
                put_count(O, start + idx, c); idx++;
                const uint32_t hb = c >= O.histo_max ? O.histo_max : c;
                if (hb < HIST_LDS) atomicAdd(&s_hc[hb], 1u); else atomicAdd(&O.histo[hb], 1ULL);
            }
        }
    }
    __syncthreads();
    if (t < HIST_LDS && s_hc[t]) atomicAdd(&O.histo[t], (unsigned long long)s_hc[t]);
}

// ------------------------------------------------------------------------------------------------ tail: buckets beyond one wave, inside LDS
// Buckets larger than the first tier (k-mers that start with their minimizer pile up under one 20-bit prefix: a few hundred such clusters per
// partition, 11 % of the keys) used to go through three more tiers — a double-size wave network, a 4-wave merge through LDS, key -> key split levels in
// HBM with host round trips — at 2.7x / 3.5x / 27x the first tier's cost per key. Here ONE workgroup takes such a bucket (<= TailCap keys) into LDS and
// resolves it there: split the item on its next informative key bits (the bits every key of the item shares are skipped, so a cluster that was isolated by
// one split spreads over the next one), counting sort in place (keys wait in registers between the histogram and the scatter), repeat for the pieces that
// are still too large, then the waves sort the pieces with the register network straight out of LDS. No HBM traffic but the one load and the result, no
// host involvement. Larger buckets (poly-A, tandem repeats) still take the HBM split levels.
constexpr int TAIL_THREADS = 512, TAIL_RPT = 16, TAIL_MAX_ITEMS = 1024, TAIL_MAX_BIG = 128, TAIL_MAX_GROUPS = 256;
constexpr uint32_t TAIL_UNIFORM = 0xFFFFFFFFu;          // TailItem::cons of an item whose keys are all the same k-mer
template <int KW> struct TailCap { static constexpr uint32_t CAP = (KW == 1) ? (uint32_t)TAIL_THREADS * TAIL_RPT : (uint32_t)TAIL_THREADS * TAIL_RPT / 2; };   // 8192 / 4096 keys: 64 KB
struct TailItem { uint32_t off_n; uint32_t cons; };              // off:16 | n-1:16 (n <= 8192 -> fits), consumed key bits
template <int KW, bool F>
__global__ __launch_bounds__(TAIL_THREADS, 4) void k_lds_tail_sort(const typename KeyT<KW>::type* __restrict__ src, typename KeyT<KW>::type* __restrict__ outk,
                                                                     const uint64_t* __restrict__ b_start, const uint32_t* __restrict__ b_n, const uint8_t* __restrict__ b_cons,
                                                                     const uint32_t* __restrict__ list, uint32_t n_list, uint32_t two_k, SortOut O)
{
    typedef typename KeyT<KW>::type key_t;
    constexpr int RPT = (KW == 1) ? TAIL_RPT : TAIL_RPT / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    key_t* s_keys = reinterpret_cast<key_t*>(s_raw);                                 // [TailCap]
    __shared__ TailItem s_small[TAIL_MAX_ITEMS]; __shared__ TailItem s_big[TAIL_MAX_BIG];
    __shared__ uint32_t s_nsmall, s_nbig, s_fail;
    __shared__ uint32_t s_cnt[TAIL_MAX_GROUPS], s_cur[TAIL_MAX_GROUPS];
    __shared__ unsigned long long s_or[2];
    __shared__ uint32_t s_hc[HIST_LDS];
    __shared__ WgList s_over;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    constexpr int NW = TAIL_THREADS / 64;
    if (t < HIST_LDS) s_hc[t] = 0;
    if (t == 0) s_over.n = 0;
    __syncthreads();
    for (uint32_t li = blockIdx.x; li < n_list; li += gridDim.x) {
        const uint32_t g = list[li];
        const uint32_t N = b_n[g];
        if (N > TailCap<KW>::CAP) { if (t == 0) wglist_push(&s_over, g, O.over2_count, O.over2_list); continue; }      // HBM split levels
        const uint64_t S = b_start[g];
        for (uint32_t i = t; i < N; i += TAIL_THREADS) s_keys[i] = src[S + i];
        if (t == 0) { s_nsmall = 0; s_nbig = 1; s_fail = 0; s_big[0] = TailItem{ (0u << 16) | (N - 1), b_cons[g] }; }
        __syncthreads();
        // ---- split the items that are still too large, one at a time, all threads together
        for (uint32_t bi = 0; ; bi++) {
            __syncthreads();
            if (bi >= s_nbig || s_fail) break;
            const TailItem it = s_big[bi];
            const uint32_t off = it.off_n >> 16, n = (it.off_n & 0xFFFFu) + 1;
            key_t v[RPT];
            const key_t k0 = s_keys[off];
            key_t acc = 0;
#pragma unroll
            for (int r = 0; r < RPT; r++) { const uint32_t i = (uint32_t)r * TAIL_THREADS + t; v[r] = i < n ? s_keys[off + i] : k0; acc |= v[r] ^ k0; }
            if (t < 2) s_or[t] = 0;
            if (t < TAIL_MAX_GROUPS) s_cnt[t] = 0;
            __syncthreads();
            {   unsigned long long lo = (unsigned long long)acc, hi = (unsigned long long)((u128)acc >> 64);
#pragma unroll
                for (int dd = 32; dd >= 1; dd >>= 1) { lo |= __shfl_down(lo, dd, 64); hi |= __shfl_down(hi, dd, 64); }
                if (lane == 0) { atomicOr(&s_or[0], lo); if (KW == 2) atomicOr(&s_or[1], hi); }
            }
            __syncthreads();
            const unsigned long long olo = s_or[0], ohi = s_or[1];
            const uint32_t diff_bits = ohi ? 128 - __clzll((long long)ohi) : (olo ? 64 - __clzll((long long)olo) : 0);
            const uint32_t left0 = two_k - it.cons;
            const uint32_t left = diff_bits < left0 ? diff_bits : left0;                  // informative bits still unused
            if (left == 0) {                                                              // every key of the item is the same k-mer: one record, abundance n (emitted with the pieces)
                if (t == 0) { const uint32_t q = atomicAdd(&s_nsmall, 1u); if (q < (uint32_t)TAIL_MAX_ITEMS) s_small[q] = TailItem{ it.off_n, TAIL_UNIFORM }; else s_fail = 1; }
                continue;
            }
            uint32_t sb = 1; while (sb < 8 && sb < left && (n >> sb) > 384) sb++;
            const uint32_t shift = left - sb, G = 1u << sb, cons_child = it.cons + (left0 - left) + sb;
            uint32_t dg[RPT];
#pragma unroll
            for (int r = 0; r < RPT; r++) {
                const uint32_t i = (uint32_t)r * TAIL_THREADS + t;
                dg[r] = (uint32_t)(v[r] >> shift) & (G - 1);                             // (a generic shift: 0 <= shift < 2k)
                if (i < n) atomicAdd(&s_cnt[dg[r]], 1u);
            }
            __syncthreads();
            if (w == 0) {                                                                 // exclusive scan of G <= 256 counters by one wave
                uint32_t c4[4], sum = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) { c4[j] = (uint32_t)(lane * 4 + j) < G ? s_cnt[lane * 4 + j] : 0u; sum += c4[j]; }
                uint32_t x = sum;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
                uint32_t run = x - sum;
#pragma unroll
                for (int j = 0; j < 4; j++) if ((uint32_t)(lane * 4 + j) < G) {
                    s_cur[lane * 4 + j] = run;
                    const uint32_t nd = c4[j];
                    if (nd) {                                                             // the piece joins the small or the big list
                        const TailItem child{ ((off + run) << 16) | (nd - 1), cons_child };
                        if (nd <= WaveCapT1<KW>::CAP) { const uint32_t q = atomicAdd(&s_nsmall, 1u); if (q < (uint32_t)TAIL_MAX_ITEMS) s_small[q] = child; else s_fail = 1; }
                        else if (nd == n) s_fail = 1;                                     // cannot happen (left > 0 splits at least two ways); guards an endless loop
                        else { const uint32_t q = atomicAdd(&s_nbig, 1u); if (q < (uint32_t)TAIL_MAX_BIG) s_big[q] = child; else s_fail = 1; }
                    }
                    run += nd;
                }
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < RPT; r++) {
                const uint32_t i = (uint32_t)r * TAIL_THREADS + t;
                if (i < n) { const uint32_t p = atomicAdd(&s_cur[dg[r]], 1u); s_keys[off + p] = v[r]; }
            }
        }
        __syncthreads();
        if (s_fail) {                                                                     // lists overflowed (pathological key sets): the HBM split levels take the bucket as it was
            for (uint32_t i = t; i < N; i += TAIL_THREADS) outk[S + i] = s_keys[i];
            if (t == 0) wglist_push(&s_over, g, O.over2_count, O.over2_list);
            __syncthreads();
            continue;
        }
        // ---- the pieces: one wave each, register network straight out of LDS
        const uint32_t ns = s_nsmall;
        for (uint32_t i = w; i < ns; i += NW) {
            const TailItem it = s_small[i];
            const uint32_t off = it.off_n >> 16, n = (it.off_n & 0xFFFFu) + 1;
            if (it.cons == TAIL_UNIFORM) {
                if (lane == 0) {
                    outk[S + off] = s_keys[off]; put_count(O, S + off, n);
                    const uint32_t hb = n >= O.histo_max ? O.histo_max : n;
                    if (hb < HIST_LDS) atomicAdd(&s_hc[hb], 1u); else atomicAdd(&O.histo[hb], 1ULL);
                }
                continue;
            }
            wave_sort_dispatch<KW, WaveCapT1<KW>::KPL_MAX, F>(s_keys + off, outk, S + off, n, O, s_hc, lane);
        }
        __syncthreads();
    }
    wglist_flush(&s_over, O.over2_count, O.over2_list);
    if (t < HIST_LDS && s_hc[t]) atomicAdd(&O.histo[t], (unsigned long long)s_hc[t]);
}

// ------------------------------------------------------------------------------------------------ deeper levels
// A bucket too large for one wave (skewed key ranges: k-mers that START with their minimizer share their top 2m bits;
// repeats; too few partitions) is split again on its next key bits by one workgroup, keys -> keys (ping-pong buffers).
// When no bit is left every key of the bucket is the same k-mer: one record with abundance n.
struct SplitDesc { uint64_t start; uint32_t n; uint32_t left; uint32_t bits; uint32_t consumed; uint64_t child_base; };

template <int KW>
__global__ __launch_bounds__(EXPAND_THREADS) void k_split_count(const typename KeyT<KW>::type* __restrict__ src, const SplitDesc* __restrict__ descs,
                                                                 uint64_t* __restrict__ c_start, uint32_t* __restrict__ c_n, uint8_t* __restrict__ c_consumed,
                                                                 uint32_t* __restrict__ eff_shift)
{
    typedef typename KeyT<KW>::type key_t;
    __shared__ uint32_t s_hist[MAX_SUB];
    __shared__ uint32_t s_wsum[EXPAND_THREADS / 64];
    __shared__ unsigned long long s_or[2];
    const SplitDesc d = descs[blockIdx.x];
    const uint32_t nsub = 1u << d.bits, mask = nsub - 1;
    for (uint32_t i = threadIdx.x; i < nsub; i += EXPAND_THREADS) s_hist[i] = 0;
    if (threadIdx.x < 2) s_or[threadIdx.x] = 0;
    __syncthreads();
    // leading bits shared by every key of the bucket carry no information: skip them (e.g. k-mers that start with
    // their minimizer share 2m bits), so a skewed bucket resolves in one more level instead of several
    {
        const key_t k0 = src[d.start];
        key_t acc = 0;
        for (uint32_t i = threadIdx.x; i < d.n; i += EXPAND_THREADS) acc |= src[d.start + i] ^ k0;
        unsigned long long lo = (unsigned long long)acc, hi = (unsigned long long)((u128)acc >> 64);
#pragma unroll
        for (int dd = 32; dd >= 1; dd >>= 1) { lo |= __shfl_down(lo, dd, 64); hi |= __shfl_down(hi, dd, 64); }
        if ((threadIdx.x & 63) == 0) { atomicOr(&s_or[0], lo); if (KW == 2) atomicOr(&s_or[1], hi); }
    }
    __syncthreads();
    uint32_t diff_bits;                                    // number of low bits that may differ between keys
    {   const unsigned long long lo = s_or[0], hi = s_or[1];
        diff_bits = hi ? 128 - __clzll((long long)hi) : (lo ? 64 - __clzll((long long)lo) : 0); }
    const uint32_t left = diff_bits < d.left ? diff_bits : d.left;         // informative bits still unused
    const uint32_t bits = d.bits < left ? d.bits : left;
    const uint32_t shift = left - bits;
    const uint32_t cons_child = d.consumed + (d.left - left) + bits;      // == 2k when left == bits
    if (threadIdx.x == 0) eff_shift[blockIdx.x] = shift | (bits << 8);
    const uint32_t m2 = bits ? ((1u << bits) - 1) : 0;
    for (uint32_t i = threadIdx.x; i < d.n; i += EXPAND_THREADS) atomicAdd(&s_hist[(uint32_t)(src[d.start + i] >> shift) & m2], 1u);
    __syncthreads();
    (void)mask;
    const uint32_t per = (nsub + EXPAND_THREADS - 1) / EXPAND_THREADS;
    const uint32_t b = threadIdx.x * per;
    uint32_t loc = 0;
    for (uint32_t i = 0; i < per; i++) if (b + i < nsub) loc += s_hist[b + i];
    uint32_t x = loc; const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) { uint32_t y = __shfl_up(x, dd, 64); if (lane >= dd) x += y; }
    if (lane == 63) s_wsum[wave] = x;
    __syncthreads();
    uint32_t wpre = 0;
    for (int w = 0; w < wave; w++) wpre += s_wsum[w];
    uint32_t run = wpre + x - loc;
    for (uint32_t i = 0; i < per; i++) if (b + i < nsub) {
        c_start[d.child_base + b + i] = d.start + run; c_n[d.child_base + b + i] = s_hist[b + i];
        c_consumed[d.child_base + b + i] = (uint8_t)cons_child;
        run += s_hist[b + i];
    }
}
template <int KW>
__global__ __launch_bounds__(EXPAND_THREADS) void k_split_scatter(const typename KeyT<KW>::type* __restrict__ src, typename KeyT<KW>::type* __restrict__ dst,
                                                                   const SplitDesc* __restrict__ descs, const uint64_t* __restrict__ c_start,
                                                                   const uint32_t* __restrict__ eff_shift)
{
    __shared__ uint32_t s_cur[MAX_SUB];
    const SplitDesc d = descs[blockIdx.x];
    const uint32_t es = eff_shift[blockIdx.x];
    const uint32_t shift = es & 255, bits = es >> 8;
    const uint32_t nsub = 1u << d.bits, mask = bits ? ((1u << bits) - 1) : 0;
    for (uint32_t i = threadIdx.x; i < nsub; i += EXPAND_THREADS) s_cur[i] = (uint32_t)(c_start[d.child_base + i] - d.start);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < d.n; i += EXPAND_THREADS) {
        const typename KeyT<KW>::type key = src[d.start + i];
        const uint32_t slot = atomicAdd(&s_cur[(uint32_t)(key >> shift) & mask], 1u);
        dst[d.start + slot] = key;
    }
}
// buckets whose keys are all equal (no key bit left): one record
template <int KW>
__global__ void k_uniform_buckets(const typename KeyT<KW>::type* __restrict__ src, typename KeyT<KW>::type* __restrict__ outk, const SplitDesc* __restrict__ descs,
                                  uint32_t n, SortOut O)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const SplitDesc d = descs[i];
    outk[d.start] = src[d.start];
    const uint32_t c = d.n > 0x7FFFFFFFu ? 0x7FFFFFFFu : d.n;                       // CountNumber is int32
    put_count(O, d.start, c);
    atomicAdd(&O.histo[c >= O.histo_max ? O.histo_max : c], 1ULL);
}
__global__ void k_gather_buckets(const uint32_t* __restrict__ list, uint32_t n, const uint64_t* __restrict__ b_start, const uint32_t* __restrict__ b_n,
                                 const uint8_t* __restrict__ b_consumed, uint64_t* __restrict__ o_start, uint32_t* __restrict__ o_n, uint32_t* __restrict__ o_consumed)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t g = list[i];
    o_start[i] = b_start[g]; o_n[i] = b_n[g]; o_consumed[i] = b_consumed[g];
}

// ------------------------------------------------------------------------------------------------ B5 compaction by slot flags
// cnt[slot] != 0 marks a distinct k-mer (key in keys[slot]); partitions start on COMPACT_BLK-aligned slots, so the
// per-block prefix directly yields per-partition offsets.
constexpr int COMPACT_THREADS = 1024, COMPACT_ITEMS = 4, COMPACT_BLK = COMPACT_THREADS * COMPACT_ITEMS;
// one WAVE per compaction block (4096 flag bytes = 64 lanes x 4 x 16-byte loads), no LDS, no barrier
constexpr int BSUM_THREADS = 256;
__global__ __launch_bounds__(BSUM_THREADS) void k_flag_block_sums(const uint8_t* __restrict__ cnt8, const uint32_t* __restrict__ cnt32, uint64_t n_blocks,
                                                                   int32_t amin, int32_t amax, uint64_t* __restrict__ bs_distinct, uint64_t* __restrict__ bs_solid)
{
    static_assert(COMPACT_BLK == 64 * 4 * 16, "one wave covers a block with four 16-byte loads per lane");
    const uint64_t blk = (uint64_t)blockIdx.x * (BSUM_THREADS / 64) + (threadIdx.x >> 6);
    if (blk >= n_blocks) return;
    const int lane = threadIdx.x & 63;
    const bool all_solid = amin <= 1 && amax == 0x7fffffff;
    uint32_t d = 0, s = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint64_t base = blk * COMPACT_BLK + (uint64_t)(j * 64 + lane) * 16;
        const uint4 q = *reinterpret_cast<const uint4*>(cnt8 + base);
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const uint32_t b = (w[i >> 2] >> (8 * (i & 3))) & 255u;
            if (b) {
                d++;
                if (!all_solid) { const uint32_t c = count_at(cnt8, cnt32, base + i, b); s += ((int32_t)c >= amin && (int32_t)c <= amax); }
            }
        }
    }
    if (all_solid) s = d;
#pragma unroll
    for (int dd = 32; dd >= 1; dd >>= 1) { d += __shfl_down(d, dd, 64); s += __shfl_down(s, dd, 64); }
    if (lane == 0) { bs_distinct[blk] = d; bs_solid[blk] = s; }
}
// in-place exclusive scan of two u64 arrays of n entries (+ totals at [n]) in three launches: every workgroup scans its own
// chunk of 8192 entries and leaves the chunk totals, one workgroup scans the totals, a third pass adds them back
constexpr int SCAN2_ITEMS = 8, SCAN2_CHUNK = 1024 * SCAN2_ITEMS;
__device__ __forceinline__ void wg_scan2(uint64_t ta, uint64_t tb, uint64_t& ea, uint64_t& eb, uint64_t& tota, uint64_t& totb, uint64_t* s_a, uint64_t* s_b)
{   // exclusive prefix of (ta, tb) over the 1024 threads of the workgroup + workgroup totals
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    uint64_t xa = ta, xb = tb;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint64_t ya = __shfl_up((unsigned long long)xa, d, 64), yb = __shfl_up((unsigned long long)xb, d, 64); if (lane >= d) { xa += ya; xb += yb; } }
    if (lane == 63) { s_a[wave] = xa; s_b[wave] = xb; }
    __syncthreads();
    uint64_t pa = 0, pb = 0; tota = 0; totb = 0;
    for (int w = 0; w < 16; w++) { if (w < wave) { pa += s_a[w]; pb += s_b[w]; } tota += s_a[w]; totb += s_b[w]; }
    ea = pa + xa - ta; eb = pb + xb - tb;
}
__global__ __launch_bounds__(1024) void k_scan2_chunks(uint64_t* __restrict__ a, uint64_t* __restrict__ b, uint64_t n, uint64_t* __restrict__ ca, uint64_t* __restrict__ cb)
{
    __shared__ uint64_t s_a[16], s_b[16];
    const uint64_t i0 = (uint64_t)blockIdx.x * SCAN2_CHUNK + (uint64_t)threadIdx.x * SCAN2_ITEMS;
    uint64_t va[SCAN2_ITEMS], vb[SCAN2_ITEMS], ta = 0, tb = 0;
#pragma unroll
    for (int j = 0; j < SCAN2_ITEMS; j++) { va[j] = i0 + j < n ? a[i0 + j] : 0; vb[j] = i0 + j < n ? b[i0 + j] : 0; ta += va[j]; tb += vb[j]; }
    uint64_t ra, rb, tota, totb;
    wg_scan2(ta, tb, ra, rb, tota, totb, s_a, s_b);
#pragma unroll
    for (int j = 0; j < SCAN2_ITEMS; j++) if (i0 + j < n) { a[i0 + j] = ra; b[i0 + j] = rb; ra += va[j]; rb += vb[j]; }
    if (threadIdx.x == 0) { ca[blockIdx.x] = tota; cb[blockIdx.x] = totb; }
}
__global__ __launch_bounds__(1024) void k_scan2_totals(uint64_t* __restrict__ ca, uint64_t* __restrict__ cb, uint32_t n_chunks, uint64_t* __restrict__ a, uint64_t* __restrict__ b, uint64_t n)
{   // n_chunks <= 1024 * SCAN2_ITEMS (n < 2^26 entries): one round
    __shared__ uint64_t s_a[16], s_b[16];
    const uint32_t i0 = threadIdx.x * SCAN2_ITEMS;
    uint64_t va[SCAN2_ITEMS], vb[SCAN2_ITEMS], ta = 0, tb = 0;
#pragma unroll
    for (int j = 0; j < SCAN2_ITEMS; j++) { va[j] = i0 + j < n_chunks ? ca[i0 + j] : 0; vb[j] = i0 + j < n_chunks ? cb[i0 + j] : 0; ta += va[j]; tb += vb[j]; }
    uint64_t ra, rb, tota, totb;
    wg_scan2(ta, tb, ra, rb, tota, totb, s_a, s_b);
#pragma unroll
    for (int j = 0; j < SCAN2_ITEMS; j++) if (i0 + j < n_chunks) { ca[i0 + j] = ra; cb[i0 + j] = rb; ra += va[j]; rb += vb[j]; }
    if (threadIdx.x == 0) { a[n] = tota; b[n] = totb; }
}
__global__ __launch_bounds__(1024) void k_scan2_add(uint64_t* __restrict__ a, uint64_t* __restrict__ b, uint64_t n, const uint64_t* __restrict__ ca, const uint64_t* __restrict__ cb)
{
    const uint64_t oa = ca[blockIdx.x], ob = cb[blockIdx.x];
    const uint64_t i0 = (uint64_t)blockIdx.x * SCAN2_CHUNK + (uint64_t)threadIdx.x * SCAN2_ITEMS;
#pragma unroll
    for (int j = 0; j < SCAN2_ITEMS; j++) if (i0 + j < n) { a[i0 + j] += oa; b[i0 + j] += ob; }
}
// B5 dump: Count records {value, abundance} (Abundance.hpp:68-129), solid only, ascending
template <int KW>
__global__ __launch_bounds__(COMPACT_THREADS) void k_compact_flags(const typename KeyT<KW>::type* __restrict__ keys, const uint8_t* __restrict__ cnt8,
                                                                    const uint32_t* __restrict__ cnt32, uint64_t n_slots,
                                                                    const uint64_t* __restrict__ bp_solid, int32_t amin, int32_t amax, uint64_t* __restrict__ out)
{
    __shared__ uint32_t s_w[COMPACT_THREADS / 64];
    constexpr int OW = (KW == 1) ? 2 : 4;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint64_t base = (uint64_t)blockIdx.x * COMPACT_BLK + (uint64_t)t * COMPACT_ITEMS;
    uint32_t c[COMPACT_ITEMS]; uint32_t ok = 0, loc = 0;
    const uint32_t v4 = *reinterpret_cast<const uint32_t*>(cnt8 + base);                 // n_slots is a multiple of COMPACT_BLK
#pragma unroll
    for (int i = 0; i < COMPACT_ITEMS; i++) {
        const uint32_t b = (v4 >> (8 * i)) & 255u;
        c[i] = b ? count_at(cnt8, cnt32, base + i, b) : 0u;
        const bool o = c[i] != 0 && (int32_t)c[i] >= amin && (int32_t)c[i] <= amax;       // CountRange::includes (closed interval)
        ok |= (uint32_t)o << i; loc += o;
    }
    uint32_t x = loc;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
    if (lane == 63) s_w[wave] = x;
    __syncthreads();
    uint32_t pre = 0;
    for (int w = 0; w < wave; w++) pre += s_w[w];
    uint64_t o = bp_solid[blockIdx.x] + pre + x - loc;
#pragma unroll
    for (int i = 0; i < COMPACT_ITEMS; i++) if ((ok >> i) & 1) {
        const typename KeyT<KW>::type key = keys[base + i];
        uint64_t* dst = out + o * OW; o++;
        if (KW == 1) *reinterpret_cast<ulonglong2*>(dst) = make_ulonglong2((uint64_t)key, (uint64_t)c[i]);
        else {
            *reinterpret_cast<ulonglong2*>(dst) = make_ulonglong2((uint64_t)key, (uint64_t)((u128)key >> 64));
            *reinterpret_cast<ulonglong2*>(dst + 2) = make_ulonglong2((uint64_t)c[i], 0ULL);
        }
    }
}
__global__ void k_gather_u64(const uint64_t* __restrict__ a, const uint64_t* __restrict__ b, const uint64_t* __restrict__ idx, uint32_t n, uint64_t* __restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[2 * i] = a[idx[i]]; out[2 * i + 1] = b[idx[i]];
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
struct BatchBufs {
    DevBuf pd, keysA, keysB, cnt, cnt8, b_start[2], b_n[2], b_cons[2], over, over2, over3, bigs, misc, bs_d, bs_s, descs, effs, g_start, g_n, g_cons, pidx, ptot, bin_of, bin_first, bin_n, bin_pre, order;
    void release() { DevBuf* all[] = { &pd, &keysA, &keysB, &cnt, &cnt8, &b_start[0], &b_start[1], &b_n[0], &b_n[1], &b_cons[0], &b_cons[1], &over, &over2, &over3, &bigs, &misc, &bs_d, &bs_s,
                                        &descs, &effs, &g_start, &g_n, &g_cons, &pidx, &ptot, &bin_of, &bin_first, &bin_n, &bin_pre, &order };
                     for (DevBuf* d : all) d->release(); }
};

template <int KW, int RW>
static int count_batch(gkc_ctx* c, const std::vector<uint32_t>& batch_parts_in, const std::vector<uint64_t>& part_keys,
                       const SegTable& segs, std::vector<void*>& outputs)
{
    typedef typename KeyT<KW>::type key_t;
    const std::vector<uint32_t>& batch_parts = batch_parts_in;
    const uint32_t nb = (uint32_t)batch_parts.size();
    const uint32_t k = c->k;
    // --- host-built tables (sizes are known exactly from Stage A)
    std::vector<PartDesc> pd(nb);
    std::vector<uint64_t> pblk(nb + 1);
    uint64_t n_slots = 0, n_sub = 0;
    const uint32_t target = (KW == 1) ? SUB_TARGET : SUB_TARGET / 2;       // mean keys of a level-1 bucket (sorted inside LDS)
    const uint32_t max_bits1 = getenv("GKC_MAX_SUB_BITS") ? (uint32_t)atoi(getenv("GKC_MAX_SUB_BITS")) : (uint32_t)MAX_SUB_BITS;
    // GKC_SCATTER_LINE=1: keys leave Stage B's expansion as whole 64-byte lines into super-buckets of 4 sub-buckets, split again inside the level-1
    // sort (k_expand_scatter_line + k_super_sort). Bit-exact, but measured SLOWER than the default pair scatter + wave sort on 1e8 reads
    // (single lane: scatter 117 vs 95 ms, level-1 sort 157 vs 70 ms; profiles/r02_line_scatter_experiment.txt): an LDS atomic costs ~12 clk per
    // wave-instruction whatever the number of active lanes (profiles/r02_lds_bench.txt), so the collector's 9 extra LDS / store instructions per
    // divergent k-mer step outweigh the 2.5x cheaper stores. Kept as a measured experiment.
    static const bool line_env = getenv("GKC_SCATTER_LINE") ? atoi(getenv("GKC_SCATTER_LINE")) != 0 : false;
    const bool line = line_env && 2 * k >= 2 && max_bits1 >= 2 && getenv("GKC_SCATTER_NO_PAIR") == nullptr && getenv("GKC_SCATTER_QUAD") == nullptr;
    const uint32_t line_slots = line ? LineT<KW>::KEYS : 0u;
    for (uint32_t i = 0; i < nb; i++) {
        const uint64_t np = part_keys[batch_parts[i]];
        if (np >= (1ULL << 32)) GKC_FAIL(c, GKC_ERR_ARG, "partition %u holds %llu k-mers (>= 2^32): use more partitions", batch_parts[i], (unsigned long long)np);
        uint32_t bits = 0;
        while (bits < max_bits1 && bits < 2 * k && (np >> bits) > target) bits++;
        if (line && bits < 2) bits = 2;                                   // super-buckets are groups of 4 sub-buckets (2k >= 6)
        pd[i].part = batch_parts[i]; pd[i].sub_bits = bits; pd[i].shift = 2 * k - bits; pd[i].pad = 0;
        pd[i].key_base = n_slots; pd[i].sub_base = n_sub;
        pblk[i] = n_slots / COMPACT_BLK;
        const uint64_t pad_slots = line ? ((uint64_t)(line_slots - 1) << (bits - 2)) : (3ull << bits);   // super-buckets start on whole lines / sub-buckets on multiples of 4
        n_slots += (np + pad_slots + COMPACT_BLK - 1) / COMPACT_BLK * COMPACT_BLK;           // partitions start on compaction-block boundaries
        n_sub += (1ull << bits);
    }
    pblk[nb] = n_slots / COMPACT_BLK;
    if (n_sub >= (1ULL << 31)) GKC_FAIL(c, GKC_ERR_ARG, "too many sub-buckets in one batch");
    const uint64_t n_blocks = n_slots / COMPACT_BLK;
    BatchBufs B;
#define CB_TRY(expr) do { int rc__ = (expr); if (rc__ != GKC_OK) { B.release(); return rc__; } } while (0)
#define CB_HIP(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { B.release(); c->set_error(GKC_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); return GKC_ERR_HIP; } } while (0)
    CB_TRY(c->ensure(B.pd, nb * sizeof(PartDesc)));
    // the big working buffers are sized for the pass's batch budget, not for this batch: every batch then asks the allocator for exactly
    // the same blocks (a batch one partition larger or smaller would otherwise land in the next size class now and then)
    const uint64_t alloc_slots = std::max<uint64_t>(std::max<uint64_t>(n_slots, c->slots_hint), 4);
    CB_TRY(c->ensure(B.keysA, (size_t)alloc_slots * sizeof(key_t)));
    CB_TRY(c->ensure(B.cnt, (size_t)alloc_slots * 4)); CB_TRY(c->ensure(B.cnt8, (size_t)alloc_slots));
    CB_TRY(c->ensure(B.b_start[0], (size_t)n_sub * 8)); CB_TRY(c->ensure(B.b_n[0], (size_t)n_sub * 4)); CB_TRY(c->ensure(B.b_cons[0], (size_t)n_sub));
    CB_TRY(c->ensure(B.over, (size_t)(n_sub + 1) * 4)); CB_TRY(c->ensure(B.over2, (size_t)(n_sub + 1) * 4)); CB_TRY(c->ensure(B.over3, (size_t)(n_sub + 1) * 4));
    CB_TRY(c->ensure(B.misc, 64));
    CB_TRY(c->ensure(B.bs_d, (size_t)(n_blocks + 1) * 8)); CB_TRY(c->ensure(B.bs_s, (size_t)(n_blocks + 1) * 8));
    CB_TRY(c->ensure(B.pidx, (size_t)(nb + 1) * 8)); CB_TRY(c->ensure(B.ptot, (size_t)(nb + 1) * 16));
    // the expansion kernels run one workgroup per partition: workgroup i takes the i-th LARGEST partition, so that the launch does not end on one long
    // workgroup (partition sizes spread 2-3x around their mean). Only the assignment changes: the layout of the batch stays in partition order.
    std::vector<uint32_t> order(nb);
    for (uint32_t i = 0; i < nb; i++) order[i] = i;
    static const bool lpt = getenv("GKC_BATCH_LPT") ? atoi(getenv("GKC_BATCH_LPT")) != 0 : true;
    if (lpt) std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return part_keys[batch_parts[a]] > part_keys[batch_parts[b]]; });
    CB_TRY(c->ensure(B.order, (size_t)nb * 4));
    CB_HIP(hipMemcpyAsync(B.order.p, order.data(), (size_t)nb * 4, hipMemcpyHostToDevice, cur_stream(c)));
    CB_HIP(hipMemcpyAsync(B.pd.p, pd.data(), nb * sizeof(PartDesc), hipMemcpyHostToDevice, cur_stream(c)));
    CB_HIP(hipMemcpyAsync(B.pidx.p, pblk.data(), (size_t)(nb + 1) * 8, hipMemcpyHostToDevice, cur_stream(c)));
    CB_HIP(hipMemsetAsync(B.cnt8.p, 0, (size_t)std::max<uint64_t>(n_slots, 4), cur_stream(c)));
    CB_HIP(hipMemsetAsync(B.misc.p, 0, 64, cur_stream(c)));
    // GKC_BIN (8-byte keys): coarse scatter by bin + in-LDS split (k_expand_coarse, k_bin_sort) instead of the pair scatter + wave sort from HBM
    static const bool bin_env = getenv("GKC_BIN") ? atoi(getenv("GKC_BIN")) != 0 : false;
    const bool bin = bin_env && KW == 1 && RW == 2 && !line && getenv("GKC_SCATTER_NO_PAIR") == nullptr && getenv("GKC_SCATTER_QUAD") == nullptr &&
                     getenv("GKC_HASH_COUNT") == nullptr && k <= 31;
    BinTables BT{};
    if (bin) {
        static const uint32_t nbmax_env = getenv("GKC_BIN_NBMAX") ? (uint32_t)atoi(getenv("GKC_BIN_NBMAX")) : BIN_NBMAX;      // tests: force the fallback
        CB_TRY(c->ensure(B.bin_of, (size_t)n_sub * 2)); CB_TRY(c->ensure(B.bin_first, (size_t)nb * (BIN_NBMAX + 1) * 4));
        CB_TRY(c->ensure(B.bin_n, (size_t)nb * 4)); CB_TRY(c->ensure(B.bin_pre, (size_t)(nb + 1) * 4));
        BT.of_sub = (uint16_t*)B.bin_of.p; BT.first = (uint32_t*)B.bin_first.p; BT.n = (uint32_t*)B.bin_n.p; BT.bad = (uint32_t*)B.misc.p + 8;
        BT.nbmax = std::min<uint32_t>(nbmax_env, BIN_NBMAX);
    }

    {   ScopedTimer tm(c, "expand_count");
        hipLaunchKernelGGL((k_expand_count<KW, RW>), dim3(nb), dim3(EXPAND_THREADS), 0, cur_stream(c), (const PartDesc*)B.pd.p, segs, k,
                           (uint64_t*)B.b_start[0].p, (uint32_t*)B.b_n[0].p, (uint8_t*)B.b_cons[0].p, line_slots, BT, (const uint32_t*)B.order.p);
        CB_HIP(hipGetLastError());
    }
    if (bin && getenv("GKC_VERBOSE")) {
        std::vector<uint32_t> hb(nb); uint32_t bad = 0;
        CB_HIP(hipMemcpyAsync(hb.data(), B.bin_n.p, (size_t)nb * 4, hipMemcpyDeviceToHost, cur_stream(c)));
        CB_HIP(hipMemcpyAsync(&bad, BT.bad, 4, hipMemcpyDeviceToHost, cur_stream(c)));
        CB_HIP(hipStreamSynchronize(cur_stream(c)));
        uint64_t sum = 0, mx = 0, mxk = 0; for (uint32_t i = 0; i < nb; i++) { sum += hb[i]; if (hb[i] > mx) { mx = hb[i]; mxk = part_keys[batch_parts[i]]; } }
        fprintf(stderr, "[gkc] bins: %u partitions, %llu bins, max %llu (partition of %llu keys), fallback flag %u\n", nb, (unsigned long long)sum, (unsigned long long)mx, (unsigned long long)mxk, bad);
    }
    {   ScopedTimer tm(c, "expand_scatter");
        if constexpr (KW == 1 && RW == 2) if (bin) {
            // coarse scatter (whole runs per bin); the pair scatter behind it only runs when the bin tables did not fit (flag set by k_expand_count)
            const size_t lds = (size_t)COARSE_STAGE_KEYS * 8 + (size_t)BIN_NBMAX * 8 + (size_t)MAX_SUB * 2;
            static std::once_flag once; std::call_once(once, [&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_expand_coarse), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
            hipLaunchKernelGGL(k_expand_coarse, dim3(nb), dim3(COARSE_THREADS), lds, cur_stream(c), (const PartDesc*)B.pd.p, segs, k,
                               (const uint64_t*)B.b_start[0].p, BT, (uint64_t*)B.keysA.p, (const uint32_t*)B.order.p);
            const size_t lds2 = (size_t)MAX_SUB * 12;
            static std::once_flag once2; std::call_once(once2, [&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_expand_scatter_pair), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2); });
            hipLaunchKernelGGL(k_expand_scatter_pair, dim3(nb), dim3(PAIR_THREADS), lds2, cur_stream(c), (const PartDesc*)B.pd.p, segs, k,
                               (const uint64_t*)B.b_start[0].p, (uint64_t*)B.keysA.p, (const uint32_t*)BT.bad, (const uint32_t*)B.order.p);
        }
        if (bin) {} else
        if (line) {
            const size_t lds = (size_t)LINE_SUPER_MAX * (LINE_SLOT_WORDS * 8 + 8);          // 144 KB: 64 bytes staged + two control words per super-bucket
            static std::once_flag once; std::call_once(once, [&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_expand_scatter_line<KW, RW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
            hipLaunchKernelGGL((k_expand_scatter_line<KW, RW>), dim3(nb), dim3(LINE_THREADS), lds, cur_stream(c), (const PartDesc*)B.pd.p, segs, k,
                               (const uint64_t*)B.b_start[0].p, (const uint32_t*)B.b_n[0].p, (key_t*)B.keysA.p);
        } else if (KW == 1 && getenv("GKC_SCATTER_QUAD") != nullptr) {           // measured slower (double expansion + 3-slot protocol: 119 vs 96 ms), kept for experiments
            const size_t lds = (size_t)QUAD_SUB * 32;
            static std::once_flag once; std::call_once(once, [&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_expand_scatter_quad), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
            uint32_t max_bits = 0; for (uint32_t i = 0; i < nb; i++) max_bits = std::max(max_bits, pd[i].sub_bits);
            uint32_t hl2 = 0; while (((uint32_t)QUAD_SUB << hl2) < (1u << max_bits)) hl2++;
            hipLaunchKernelGGL(k_expand_scatter_quad, dim3(nb << hl2), dim3(QUAD_THREADS), lds, cur_stream(c), (const PartDesc*)B.pd.p, segs, k,
                               (const uint64_t*)B.b_start[0].p, (const uint32_t*)B.b_n[0].p, (uint64_t*)B.keysA.p, hl2);
        } else if (KW == 1 && getenv("GKC_SCATTER_NO_PAIR") == nullptr) {
            const size_t lds = (size_t)MAX_SUB * 12;
            static std::once_flag once; std::call_once(once, [&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_expand_scatter_pair), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
            hipLaunchKernelGGL(k_expand_scatter_pair, dim3(nb), dim3(PAIR_THREADS), lds, cur_stream(c), (const PartDesc*)B.pd.p, segs, k,
                               (const uint64_t*)B.b_start[0].p, (uint64_t*)B.keysA.p, (const uint32_t*)nullptr, (const uint32_t*)B.order.p);
        } else if (KW == 2 && k >= 32 && getenv("GKC_SCATTER_NO_PAIR") == nullptr) {
            const size_t lds = (size_t)MAX_SUB * 20;                           // 160 KB: the whole LDS of a CU
            static std::once_flag once; std::call_once(once, [&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_expand_scatter_pair2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
            hipLaunchKernelGGL(k_expand_scatter_pair2, dim3(nb), dim3(PAIR_THREADS), lds, cur_stream(c), (const PartDesc*)B.pd.p, segs, k,
                               (const uint64_t*)B.b_start[0].p, (u128*)B.keysA.p, (const uint32_t*)B.order.p);
        } else
        hipLaunchKernelGGL((k_expand_scatter<KW, RW>), dim3(nb), dim3(EXPAND_THREADS), 0, cur_stream(c), (const PartDesc*)B.pd.p, segs, k,
                           (const uint64_t*)B.b_start[0].p, (key_t*)B.keysA.p);
        CB_HIP(hipGetLastError());
    }
    SortOut O{};
    O.cnt8 = (uint8_t*)B.cnt8.p; O.cnt32 = (uint32_t*)B.cnt.p; O.histo = c->histo_of(c->pass); O.histo_max = c->histo_max;
    O.over_count = (uint32_t*)B.over.p; O.over_list = (uint32_t*)B.over.p + 1;
    O.over2_count = (uint32_t*)B.over2.p; O.over2_list = (uint32_t*)B.over2.p + 1;
    O.over3_count = (uint32_t*)B.over3.p; O.over3_list = (uint32_t*)B.over3.p + 1;
    O.n_sorted = (unsigned long long*)B.misc.p;

    // --- levels: sort what fits one wave, split the rest on the next key bits, repeat
    uint32_t min_bits1 = 64; for (uint32_t i = 0; i < nb; i++) min_bits1 = std::min(min_bits1, pd[i].sub_bits);
    // every bucket's keys share their top min_bits1 bits: when the rest fits a double's 52-bit mantissa the in-lane exchanges run as v_min/max_f64
    const bool tag = KW == 1 && 2 * k - min_bits1 <= 52 && getenv("GKC_NO_F64") == nullptr;
    constexpr bool FT = KW == 1;
    // GKC_HASH_COUNT=1: count-first first tier (8-byte keys): duplicates counted in a per-wave LDS hash table, only the distinct k-mers sorted (k_wave_hash_count).
    // Bit-exact; on the 30x synthetic reads it breaks even with the plain network (first tier 69 vs 69 ms single lane, deeper levels 16 vs 12 ms, 312 vs 304 ms
    // per step: profiles/r02_hash_count_experiment.txt) — the CAS chains cost what the smaller network saves. Kept as a measured experiment.
    static const bool hash_env = getenv("GKC_HASH_COUNT") ? atoi(getenv("GKC_HASH_COUNT")) != 0 : false;
    const bool hash_count = hash_env && KW == 1 && !line;
    int cur = 0;                                 // bucket arrays b_*[cur]
    uint64_t n_buckets = n_sub;
    key_t* src = (key_t*)B.keysA.p;
    for (int level = 1; n_buckets > 0; level++) {
        if (getenv("GKC_VERBOSE")) {             // diagnostic: bucket-size distribution of this level
            std::vector<uint32_t> hn(n_buckets);
            CB_HIP(hipMemcpyAsync(hn.data(), B.b_n[cur].p, (size_t)n_buckets * 4, hipMemcpyDeviceToHost, cur_stream(c)));
            CB_HIP(hipStreamSynchronize(cur_stream(c)));
            const uint32_t edges[] = {0, 64, 128, 256, 512, 1024, 2048, 4096, 0xffffffffu};
            uint64_t nb_[9] = {0}, nk_[9] = {0};
            for (uint32_t v : hn) { int e = 0; while (v > edges[e]) e++; nb_[e]++; nk_[e] += v; }
            fprintf(stderr, "[gkc] level %d sizes:", level);
            for (int e = 0; e < 9; e++) fprintf(stderr, " <=%u: %llu b / %llu k;", edges[e], (unsigned long long)nb_[e], (unsigned long long)nk_[e]);
            fprintf(stderr, "\n");
        }
        CB_HIP(hipMemsetAsync(B.over.p, 0, 4, cur_stream(c)));
        CB_HIP(hipMemsetAsync(B.over2.p, 0, 4, cur_stream(c)));
        CB_HIP(hipMemsetAsync(B.over3.p, 0, 4, cur_stream(c)));
        if (level == 1 && line) {
            // level 1 after the line scatter: a workgroup per super-bucket splits it through LDS and sorts its 4 sub-buckets (k_super_sort)
            ScopedTimer tm(c, "bucket_sort");
            const uint32_t n_super = (uint32_t)(n_buckets / 4);
            CB_TRY(c->ensure(B.bigs, ((size_t)n_super + 1) * 4));
            CB_HIP(hipMemsetAsync(B.bigs.p, 0, 4, cur_stream(c)));
            const size_t lds = (size_t)SuperCap<KW>::CAP * sizeof(key_t);
            const unsigned grid = (unsigned)std::min<uint64_t>(n_super, 256 * 6);
            if (tag) hipLaunchKernelGGL((k_super_sort<KW, FT>), dim3(grid), dim3(SS_THREADS), lds, cur_stream(c), (const key_t*)src, (key_t*)B.keysA.p,
                               (const uint64_t*)B.b_start[cur].p, (const uint32_t*)B.b_n[cur].p, (const uint8_t*)B.b_cons[cur].p, n_super, 2 * k, O, (uint32_t*)B.bigs.p, (uint32_t*)B.bigs.p + 1);
            else hipLaunchKernelGGL((k_super_sort<KW, false>), dim3(grid), dim3(SS_THREADS), lds, cur_stream(c), (const key_t*)src, (key_t*)B.keysA.p,
                               (const uint64_t*)B.b_start[cur].p, (const uint32_t*)B.b_n[cur].p, (const uint8_t*)B.b_cons[cur].p, n_super, 2 * k, O, (uint32_t*)B.bigs.p, (uint32_t*)B.bigs.p + 1);
            CB_HIP(hipGetLastError());
            uint32_t n_big = 0;
            CB_HIP(hipMemcpyAsync(&n_big, B.bigs.p, 4, hipMemcpyDeviceToHost, cur_stream(c)));
            CB_HIP(hipStreamSynchronize(cur_stream(c)));
            if (n_big) {                                                              // super-buckets beyond the LDS buffer: 4-way split out of place, copied back
                if (!B.keysB.p) CB_TRY(c->ensure(B.keysB, (size_t)alloc_slots * sizeof(key_t)));
                hipLaunchKernelGGL((k_super_split_big<KW>), dim3(n_big), dim3(SS_THREADS), 0, cur_stream(c), (const key_t*)src, (key_t*)B.keysB.p, (const uint64_t*)B.b_start[cur].p,
                                   (const uint32_t*)B.b_n[cur].p, (const uint8_t*)B.b_cons[cur].p, (const uint32_t*)B.bigs.p + 1, 2 * k, O);
                hipLaunchKernelGGL((k_super_copy_back<KW>), dim3(n_big), dim3(SS_THREADS), 0, cur_stream(c), (key_t*)B.keysA.p, (const key_t*)B.keysB.p, (const uint64_t*)B.b_start[cur].p,
                                   (const uint32_t*)B.b_n[cur].p, (const uint32_t*)B.bigs.p + 1);
                CB_HIP(hipGetLastError());
            }
        } else
        if (KW == 1 && hash_count) {
            // first tier, count-first variant: duplicates are counted in a per-wave LDS hash table, only the distinct k-mers are sorted (k_wave_hash_count);
            // what does not fit its table comes back in a list for the plain network
            ScopedTimer tm(c, level == 1 ? "bucket_sort" : "bucket_sort_deep");
            CB_TRY(c->ensure(B.bigs, ((size_t)n_buckets + 1) * 4));
            CB_HIP(hipMemsetAsync(B.bigs.p, 0, 4, cur_stream(c)));
            const unsigned grid = (unsigned)std::min<uint64_t>((n_buckets + 3) / 4, 256 * 32);
            if constexpr (KW == 1) {
                hipLaunchKernelGGL((k_wave_hash_count<false>), dim3(grid), dim3(SORT_THREADS), 0, cur_stream(c), (const uint64_t*)src, (uint64_t*)B.keysA.p, (const uint64_t*)B.b_start[cur].p,
                                   (const uint32_t*)B.b_n[cur].p, (const uint8_t*)B.b_cons[cur].p, (uint32_t)n_buckets, 2 * k, O, (uint32_t*)B.bigs.p, (uint32_t*)B.bigs.p + 1);
            }
            CB_HIP(hipGetLastError());
            uint32_t n_plain = 0;
            CB_HIP(hipMemcpyAsync(&n_plain, B.bigs.p, 4, hipMemcpyDeviceToHost, cur_stream(c)));
            CB_HIP(hipStreamSynchronize(cur_stream(c)));
            if (n_plain) {
                const unsigned g2 = (unsigned)std::min<uint64_t>((n_plain + 3) / 4, 256 * 32);
                if (tag) hipLaunchKernelGGL((k_wave_sort_list<KW, FT>), dim3(g2), dim3(SORT_THREADS), 0, cur_stream(c), (const key_t*)src, (key_t*)B.keysA.p, (const uint64_t*)B.b_start[cur].p,
                                   (const uint32_t*)B.b_n[cur].p, (const uint32_t*)B.bigs.p + 1, n_plain, O);
                else hipLaunchKernelGGL((k_wave_sort_list<KW, false>), dim3(g2), dim3(SORT_THREADS), 0, cur_stream(c), (const key_t*)src, (key_t*)B.keysA.p, (const uint64_t*)B.b_start[cur].p,
                                   (const uint32_t*)B.b_n[cur].p, (const uint32_t*)B.bigs.p + 1, n_plain, O);
                CB_HIP(hipGetLastError());
            }
        } else
        {   ScopedTimer tm(c, level == 1 ? "bucket_sort" : "bucket_sort_deep");
            const unsigned grid = (unsigned)std::min<uint64_t>((n_buckets + 3) / 4, 256 * 32);
            const uint32_t* only_if = nullptr;
            if (level == 1 && bin) {
                // level 1 after the coarse scatter: a workgroup per bin splits it inside LDS and sorts its sub-buckets (k_bin_sort); the plain wave sort
                // below then only runs when the batch fell back to the pair scatter (more bins in a partition than the coarse scatter has counters for)
                const size_t lds = (size_t)BIN_KEYS_MAX * sizeof(key_t) + (size_t)BIN_SUBS_MAX * 12;
                static std::once_flag once;
                std::call_once(once, [&] {
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_bin_sort<KW, FT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_bin_sort<KW, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                });
                static const unsigned bwg = getenv("GKC_BIN_GRID") ? (unsigned)atoi(getenv("GKC_BIN_GRID")) : 2048u;       // workgroups of the launch, about
                static const uint32_t dbg = getenv("GKC_BIN_DBG") ? (uint32_t)atoi(getenv("GKC_BIN_DBG")) : 0u;
                const unsigned gx = std::min(64u, std::max(2u, (bwg + nb - 1) / nb));
                if (tag) hipLaunchKernelGGL((k_bin_sort<KW, FT>), dim3(gx, nb), dim3(BIN_THREADS), lds, cur_stream(c), (const PartDesc*)B.pd.p, (key_t*)B.keysA.p,
                                   (const uint64_t*)B.b_start[cur].p, (const uint32_t*)B.b_n[cur].p, BT, O, dbg);
                else hipLaunchKernelGGL((k_bin_sort<KW, false>), dim3(gx, nb), dim3(BIN_THREADS), lds, cur_stream(c), (const PartDesc*)B.pd.p, (key_t*)B.keysA.p,
                                   (const uint64_t*)B.b_start[cur].p, (const uint32_t*)B.b_n[cur].p, BT, O, dbg);
                only_if = BT.bad;
            }
            if (tag) hipLaunchKernelGGL((k_wave_sort<KW, FT>), dim3(grid), dim3(SORT_THREADS), 0, cur_stream(c), (const key_t*)src, (key_t*)B.keysA.p,
                               (const uint64_t*)B.b_start[cur].p, (const uint32_t*)B.b_n[cur].p, (uint32_t)n_buckets, O, only_if);
            else hipLaunchKernelGGL((k_wave_sort<KW, false>), dim3(grid), dim3(SORT_THREADS), 0, cur_stream(c), (const key_t*)src, (key_t*)B.keysA.p,
                               (const uint64_t*)B.b_start[cur].p, (const uint32_t*)B.b_n[cur].p, (uint32_t)n_buckets, O, only_if);
            CB_HIP(hipGetLastError());
        }
        uint32_t n_mid = 0; unsigned long long split_bad = 0;
        CB_HIP(hipMemcpyAsync(&n_mid, B.over.p, 4, hipMemcpyDeviceToHost, cur_stream(c)));
        CB_HIP(hipMemcpyAsync(&split_bad, (unsigned long long*)B.misc.p + 2, 8, hipMemcpyDeviceToHost, cur_stream(c)));
        CB_HIP(hipStreamSynchronize(cur_stream(c)));
        if (split_bad) { B.release(); GKC_FAIL(c, GKC_ERR_HIP, "internal error: %llu sub-buckets received another number of keys than the expansion counted", split_bad); }
        if (!n_mid) break;
        // GKC_TAIL_LDS=1: every bucket beyond the first tier is resolved by one workgroup inside LDS (k_lds_tail_sort) instead of the double-size wave network,
        // the 4-wave merge and (up to 8192 keys) the HBM split levels. Bit-exact, measured SLOWER (1e8 reads, k=31: 348 vs 302 ms per step, the tail kernel
        // ~60 ms single lane for 11 % of the keys against 23 + 8 ms of the tiers it replaces; k=63: 650 vs 469 ms; profiles/r02_tail_lds_experiment.txt):
        // one bucket per workgroup serialises ~10 barriers and LDS atomics on a handful of counters per split. Kept as a measured experiment.
        static const bool tail_lds = getenv("GKC_TAIL_LDS") ? atoi(getenv("GKC_TAIL_LDS")) != 0 : false;
        if (tail_lds) {
            // every bucket beyond the first tier that fits the LDS buffer is resolved by one workgroup inside LDS (k_lds_tail_sort); larger ones join over2
            ScopedTimer tm(c, "bucket_sort_tail");
            const size_t lds = (size_t)TailCap<KW>::CAP * sizeof(key_t);
            static std::once_flag once;
            std::call_once(once, [&] {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_lds_tail_sort<KW, FT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_lds_tail_sort<KW, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            });
            const unsigned grid = (unsigned)std::min<uint64_t>(n_mid, 256 * 2);
            if (tag) hipLaunchKernelGGL((k_lds_tail_sort<KW, FT>), dim3(grid), dim3(TAIL_THREADS), lds, cur_stream(c), (const key_t*)src, (key_t*)B.keysA.p, (const uint64_t*)B.b_start[cur].p,
                               (const uint32_t*)B.b_n[cur].p, (const uint8_t*)B.b_cons[cur].p, (const uint32_t*)O.over_list, n_mid, 2 * k, O);
            else hipLaunchKernelGGL((k_lds_tail_sort<KW, false>), dim3(grid), dim3(TAIL_THREADS), lds, cur_stream(c), (const key_t*)src, (key_t*)B.keysA.p, (const uint64_t*)B.b_start[cur].p,
                               (const uint32_t*)B.b_n[cur].p, (const uint8_t*)B.b_cons[cur].p, (const uint32_t*)O.over_list, n_mid, 2 * k, O);
            CB_HIP(hipGetLastError());
        }
        uint32_t n_mid2 = 0;
        if (!tail_lds) {
        {   ScopedTimer tm(c, "bucket_sort_big");                 // up to 2x the first tier: double-size wave network
            const unsigned grid = (unsigned)std::min<uint64_t>((n_mid + 3) / 4, 256 * 16);
            constexpr int KB = WaveCapHuge<KW>::KPL;
            if (GKC_T1_MID) {
                if (tag) hipLaunchKernelGGL((k_wave_sort_big<KW, FT, KB / 2>), dim3(grid), dim3(SORT_THREADS), 0, cur_stream(c), (const key_t*)src, (key_t*)B.keysA.p,
                               (const uint64_t*)B.b_start[cur].p, (const uint32_t*)B.b_n[cur].p, (const uint32_t*)O.over_list, n_mid, 0u, 0u, O);
                else hipLaunchKernelGGL((k_wave_sort_big<KW, false, KB / 2>), dim3(grid), dim3(SORT_THREADS), 0, cur_stream(c), (const key_t*)src, (key_t*)B.keysA.p,
                               (const uint64_t*)B.b_start[cur].p, (const uint32_t*)B.b_n[cur].p, (const uint32_t*)O.over_list, n_mid, 0u, 0u, O);
            }
            const uint32_t big_min = GKC_T1_MID ? 64u * (KB / 2) : 0u;
            if (tag) hipLaunchKernelGGL((k_wave_sort_big<KW, FT, KB>), dim3(grid), dim3(SORT_THREADS), 0, cur_stream(c), (const key_t*)src, (key_t*)B.keysA.p,
                               (const uint64_t*)B.b_start[cur].p, (const uint32_t*)B.b_n[cur].p, (const uint32_t*)O.over_list, n_mid, big_min, 64u * KB, O);
            else hipLaunchKernelGGL((k_wave_sort_big<KW, false, KB>), dim3(grid), dim3(SORT_THREADS), 0, cur_stream(c), (const key_t*)src, (key_t*)B.keysA.p,
                               (const uint64_t*)B.b_start[cur].p, (const uint32_t*)B.b_n[cur].p, (const uint32_t*)O.over_list, n_mid, big_min, 64u * KB, O);
            CB_HIP(hipGetLastError());
            CB_HIP(hipMemcpyAsync(&n_mid2, B.over3.p, 4, hipMemcpyDeviceToHost, cur_stream(c)));
            CB_HIP(hipStreamSynchronize(cur_stream(c)));
        }
        if (!n_mid2) break;
        {
            ScopedTimer tm(c, "bucket_sort_wg");                  // beyond one wave: workgroups of 4 / 8 waves, merges across waves through LDS
            constexpr int K1 = WaveCapHuge<KW>::KPL / 2;
            constexpr uint32_t C0 = WaveCapHuge<KW>::CAP, C1 = 4 * 64 * K1, C2 = 8 * 64 * K1;
            static std::once_flag once;
            std::call_once(once, [&] {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_wg_sort<KW, 4, K1, FT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(C1 * sizeof(key_t)));
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_wg_sort<KW, 4, K1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(C1 * sizeof(key_t)));
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_wg_sort<KW, 8, K1, FT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(C2 * sizeof(key_t)));
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_wg_sort<KW, 8, K1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(C2 * sizeof(key_t)));
            });
            // measured (ms per 1.2e10 keys, wg tier + HBM split levels + their sorts): up to 4096 keys here: 40, up to 8192: 44, none: 47
            const uint32_t wg_max = getenv("GKC_WG_MAX") ? (uint32_t)atoi(getenv("GKC_WG_MAX")) : C1;     // buckets beyond go to the HBM split
            const unsigned grid = (unsigned)std::min<uint64_t>(n_mid2, 256 * 4);
            if (tag) hipLaunchKernelGGL((k_wg_sort<KW, 4, K1, FT>), dim3(grid), dim3(256), C1 * sizeof(key_t), cur_stream(c), (const key_t*)src, (key_t*)B.keysA.p,
                               (const uint64_t*)B.b_start[cur].p, (const uint32_t*)B.b_n[cur].p, (const uint32_t*)O.over3_list, n_mid2, wg_max >= C1 ? C0 : 0xffffffffu, std::max(wg_max, C0), O);
            else hipLaunchKernelGGL((k_wg_sort<KW, 4, K1, false>), dim3(grid), dim3(256), C1 * sizeof(key_t), cur_stream(c), (const key_t*)src, (key_t*)B.keysA.p,
                               (const uint64_t*)B.b_start[cur].p, (const uint32_t*)B.b_n[cur].p, (const uint32_t*)O.over3_list, n_mid2, wg_max >= C1 ? C0 : 0xffffffffu, std::max(wg_max, C0), O);
            if (wg_max > C1) {
            if (tag) hipLaunchKernelGGL((k_wg_sort<KW, 8, K1, FT>), dim3(grid), dim3(512), C2 * sizeof(key_t), cur_stream(c), (const key_t*)src, (key_t*)B.keysA.p,
                               (const uint64_t*)B.b_start[cur].p, (const uint32_t*)B.b_n[cur].p, (const uint32_t*)O.over3_list, n_mid2, C1, 0u, O);
            else hipLaunchKernelGGL((k_wg_sort<KW, 8, K1, false>), dim3(grid), dim3(512), C2 * sizeof(key_t), cur_stream(c), (const key_t*)src, (key_t*)B.keysA.p,
                               (const uint64_t*)B.b_start[cur].p, (const uint32_t*)B.b_n[cur].p, (const uint32_t*)O.over3_list, n_mid2, C1, 0u, O);
            }
            CB_HIP(hipGetLastError());
        }
        }   // !tail_lds
        uint32_t n_over = 0;
        CB_HIP(hipMemcpyAsync(&n_over, B.over2.p, 4, hipMemcpyDeviceToHost, cur_stream(c)));
        CB_HIP(hipStreamSynchronize(cur_stream(c)));
        if (!n_over) break;
        ScopedTimer tm(c, "split_levels");
        // fetch (start, n, consumed bits) of the oversize buckets
        CB_TRY(c->ensure(B.g_start, (size_t)n_over * 8)); CB_TRY(c->ensure(B.g_n, (size_t)n_over * 4)); CB_TRY(c->ensure(B.g_cons, (size_t)n_over * 4));
        hipLaunchKernelGGL(k_gather_buckets, dim3((n_over + 255) / 256), dim3(256), 0, cur_stream(c), (const uint32_t*)O.over2_list, n_over,
                           (const uint64_t*)B.b_start[cur].p, (const uint32_t*)B.b_n[cur].p, (const uint8_t*)B.b_cons[cur].p,
                           (uint64_t*)B.g_start.p, (uint32_t*)B.g_n.p, (uint32_t*)B.g_cons.p);
        std::vector<uint64_t> h_start(n_over); std::vector<uint32_t> h_n(n_over), h_cons(n_over);
        CB_HIP(hipMemcpyAsync(h_start.data(), B.g_start.p, (size_t)n_over * 8, hipMemcpyDeviceToHost, cur_stream(c)));
        CB_HIP(hipMemcpyAsync(h_n.data(), B.g_n.p, (size_t)n_over * 4, hipMemcpyDeviceToHost, cur_stream(c)));
        CB_HIP(hipMemcpyAsync(h_cons.data(), B.g_cons.p, (size_t)n_over * 4, hipMemcpyDeviceToHost, cur_stream(c)));
        CB_HIP(hipStreamSynchronize(cur_stream(c)));
        std::vector<SplitDesc> split, uni;
        uint64_t n_child = 0;
        for (uint32_t i = 0; i < n_over; i++) {
            SplitDesc d{}; d.start = h_start[i]; d.n = h_n[i]; d.consumed = h_cons[i];
            const uint32_t left = 2 * k - d.consumed;
            if (left == 0) { uni.push_back(d); continue; }
            // deeper levels see clustered keys (that is why the bucket was oversize): split 8x finer than the mean asks for
            uint32_t bits = 1;
            while (bits < (uint32_t)MAX_SUB_BITS && bits < left && (d.n >> bits) > target) bits++;
            static const int extra_env = getenv("GKC_SPLIT_EXTRA") ? atoi(getenv("GKC_SPLIT_EXTRA")) : 2;
            bits = std::min<uint32_t>(std::min<uint32_t>(bits + (uint32_t)extra_env, (uint32_t)MAX_SUB_BITS), left);
            d.bits = bits; d.left = left; d.child_base = n_child; n_child += (1ull << bits);
            split.push_back(d);
        }
        { std::lock_guard<std::mutex> lk(c->mu); c->stats_now().oversize_buckets += n_over; }
        if (getenv("GKC_VERBOSE")) {
            uint64_t kk = 0, mx = 0; for (uint32_t i = 0; i < n_over; i++) { kk += h_n[i]; mx = std::max<uint64_t>(mx, h_n[i]); }
            fprintf(stderr, "[gkc] level %d: %llu buckets sorted from, %u oversize (%llu keys, max %llu), %zu uniform, %llu children\n", level,
                    (unsigned long long)n_buckets, n_over, (unsigned long long)kk, (unsigned long long)mx, uni.size(), (unsigned long long)n_child);
        }
        if (n_child >= (1ULL << 31)) { B.release(); GKC_FAIL(c, GKC_ERR_ARG, "too many sub-buckets while splitting: use more partitions"); }
        key_t* dst = src;
        if (!uni.empty()) {
            CB_TRY(c->ensure(B.descs, uni.size() * sizeof(SplitDesc)));
            CB_HIP(hipMemcpyAsync(B.descs.p, uni.data(), uni.size() * sizeof(SplitDesc), hipMemcpyHostToDevice, cur_stream(c)));
            hipLaunchKernelGGL((k_uniform_buckets<KW>), dim3((unsigned)((uni.size() + 255) / 256)), dim3(256), 0, cur_stream(c), (const key_t*)src, (key_t*)B.keysA.p,
                               (const SplitDesc*)B.descs.p, (uint32_t)uni.size(), O);
            CB_HIP(hipGetLastError());
            CB_HIP(hipStreamSynchronize(cur_stream(c)));
        }
        const int nxt = cur ^ 1;
        if (!split.empty()) {
            if (!B.keysB.p) CB_TRY(c->ensure(B.keysB, (size_t)alloc_slots * sizeof(key_t)));
            dst = (src == (key_t*)B.keysA.p) ? (key_t*)B.keysB.p : (key_t*)B.keysA.p;
            CB_TRY(c->ensure(B.descs, split.size() * sizeof(SplitDesc)));
            CB_TRY(c->ensure(B.effs, split.size() * 4));
            CB_TRY(c->ensure(B.b_start[nxt], (size_t)n_child * 8)); CB_TRY(c->ensure(B.b_n[nxt], (size_t)n_child * 4)); CB_TRY(c->ensure(B.b_cons[nxt], (size_t)n_child));
            if (B.over.bytes < (size_t)(n_child + 1) * 4) { CB_TRY(c->ensure(B.over, (size_t)(n_child + 1) * 4)); O.over_count = (uint32_t*)B.over.p; O.over_list = (uint32_t*)B.over.p + 1; }
            if (B.over2.bytes < (size_t)(n_child + 1) * 4) { CB_TRY(c->ensure(B.over2, (size_t)(n_child + 1) * 4)); O.over2_count = (uint32_t*)B.over2.p; O.over2_list = (uint32_t*)B.over2.p + 1; }
            if (B.over3.bytes < (size_t)(n_child + 1) * 4) { CB_TRY(c->ensure(B.over3, (size_t)(n_child + 1) * 4)); O.over3_count = (uint32_t*)B.over3.p; O.over3_list = (uint32_t*)B.over3.p + 1; }
            CB_HIP(hipMemcpyAsync(B.descs.p, split.data(), split.size() * sizeof(SplitDesc), hipMemcpyHostToDevice, cur_stream(c)));
            hipLaunchKernelGGL((k_split_count<KW>), dim3((unsigned)split.size()), dim3(EXPAND_THREADS), 0, cur_stream(c), (const key_t*)src, (const SplitDesc*)B.descs.p,
                               (uint64_t*)B.b_start[nxt].p, (uint32_t*)B.b_n[nxt].p, (uint8_t*)B.b_cons[nxt].p, (uint32_t*)B.effs.p);
            hipLaunchKernelGGL((k_split_scatter<KW>), dim3((unsigned)split.size()), dim3(EXPAND_THREADS), 0, cur_stream(c), (const key_t*)src, dst, (const SplitDesc*)B.descs.p,
                               (const uint64_t*)B.b_start[nxt].p, (const uint32_t*)B.effs.p);
            CB_HIP(hipGetLastError());
            CB_HIP(hipStreamSynchronize(cur_stream(c)));     // descs / host vectors are reused next level
        }
        cur = nxt; n_buckets = n_child; src = dst;
    }

    // --- compaction: per-block (distinct, solid) sums -> prefix -> Count records
    uint64_t total_solid = 0;
    std::vector<uint64_t> ptot((size_t)(nb + 1) * 2);
    {   ScopedTimer tm(c, "compact");
        if (n_blocks) {
            hipLaunchKernelGGL(k_flag_block_sums, dim3((unsigned)((n_blocks + BSUM_THREADS / 64 - 1) / (BSUM_THREADS / 64))), dim3(BSUM_THREADS), 0, cur_stream(c), (const uint8_t*)B.cnt8.p, (const uint32_t*)B.cnt.p, n_blocks, c->amin, c->amax,
                               (uint64_t*)B.bs_d.p, (uint64_t*)B.bs_s.p);
        }
        {   const uint32_t n_chunks = (uint32_t)((n_blocks + SCAN2_CHUNK - 1) / SCAN2_CHUNK);
            if (n_chunks > (uint32_t)SCAN2_CHUNK) { B.release(); GKC_FAIL(c, GKC_ERR_ARG, "batch too large for the block-sum scan"); }
            CB_TRY(c->ensure(B.g_start, (size_t)std::max<uint32_t>(n_chunks, 1) * 16));          // chunk totals (scratch buffer, free at this point)
            uint64_t* ca = (uint64_t*)B.g_start.p; uint64_t* cb = ca + std::max<uint32_t>(n_chunks, 1);
            if (n_chunks) hipLaunchKernelGGL(k_scan2_chunks, dim3(n_chunks), dim3(1024), 0, cur_stream(c), (uint64_t*)B.bs_d.p, (uint64_t*)B.bs_s.p, n_blocks, ca, cb);
            hipLaunchKernelGGL(k_scan2_totals, dim3(1), dim3(1024), 0, cur_stream(c), ca, cb, n_chunks, (uint64_t*)B.bs_d.p, (uint64_t*)B.bs_s.p, n_blocks);
            if (n_chunks) hipLaunchKernelGGL(k_scan2_add, dim3(n_chunks), dim3(1024), 0, cur_stream(c), (uint64_t*)B.bs_d.p, (uint64_t*)B.bs_s.p, n_blocks, (const uint64_t*)ca, (const uint64_t*)cb);
        }
        hipLaunchKernelGGL(k_gather_u64, dim3((nb + 1 + 255) / 256), dim3(256), 0, cur_stream(c), (const uint64_t*)B.bs_d.p, (const uint64_t*)B.bs_s.p,
                           (const uint64_t*)B.pidx.p, nb + 1, (uint64_t*)B.ptot.p);
        CB_HIP(hipGetLastError());
        CB_HIP(hipMemcpyAsync(ptot.data(), B.ptot.p, (size_t)(nb + 1) * 16, hipMemcpyDeviceToHost, cur_stream(c)));
        CB_HIP(hipStreamSynchronize(cur_stream(c)));
        total_solid = ptot[2 * nb + 1];
        constexpr int OW = (KW == 1) ? 2 : 4;
        void* out = c->dalloc((size_t)std::max<uint64_t>(total_solid, 1) * OW * 8);
        if (!out) { B.release(); return GKC_ERR_NOMEM; }
        { std::lock_guard<std::mutex> lk(c->mu); outputs.push_back(out); }
        if (n_blocks) {
            hipLaunchKernelGGL((k_compact_flags<KW>), dim3((unsigned)n_blocks), dim3(COMPACT_THREADS), 0, cur_stream(c), (const key_t*)B.keysA.p, (const uint8_t*)B.cnt8.p, (const uint32_t*)B.cnt.p, n_slots,
                               (const uint64_t*)B.bs_s.p, c->amin, c->amax, (uint64_t*)out);
            CB_HIP(hipGetLastError());
        }
        CB_HIP(hipStreamSynchronize(cur_stream(c)));
        // streamed results: the batch's records go to the host sink on the copy stream while the lanes count the next batches
        const uint8_t* h_base = nullptr; hipEvent_t landed = nullptr;
        if (c->sink && total_solid) {
            const uint64_t bytes = total_solid * OW * 8;
            std::lock_guard<std::mutex> lk(c->mu);
            if (c->sink_used + bytes > c->sink_cap) c->sink_overflow = true;          // the records stay on the device (gkc_partition_counts still serves them)
            else {
                h_base = (const uint8_t*)c->sink + c->sink_used; c->sink_used += bytes;
                if (hipEventCreateWithFlags(&landed, hipEventDisableTiming) == hipSuccess &&
                    hipMemcpyAsync((void*)h_base, out, bytes, hipMemcpyDeviceToHost, c->copy_stream) == hipSuccess &&
                    hipEventRecord(landed, c->copy_stream) == hipSuccess) c->landed_events.push_back(landed);
                else { (void)hipGetLastError(); if (landed) (void)hipEventDestroy(landed); landed = nullptr; h_base = nullptr; c->sink_overflow = true; }
            }
        }
        {   std::lock_guard<std::mutex> lk(c->mu);
            for (uint32_t i = 0; i < nb; i++) {
                Dataset& D = c->datasets[(size_t)c->pass * c->nb_partitions + batch_parts[i]];
                const uint64_t s0 = ptot[2 * i + 1], s1 = ptot[2 * (i + 1) + 1];
                D.d_counts = (const uint8_t*)out + s0 * OW * 8;
                D.h_counts = h_base ? h_base + s0 * OW * 8 : nullptr; D.landed = landed;
                D.n_solid = s1 - s0; D.n_distinct = ptot[2 * (i + 1)] - ptot[2 * i]; D.n_kmers = part_keys[batch_parts[i]]; D.done = true;
                c->stats_now().kmers_nb_distinct += D.n_distinct; c->stats_now().kmers_nb_solid += D.n_solid;
            }
        }
        c->cv_done.notify_all();
    }
    B.release();
#undef CB_TRY
#undef CB_HIP
    return GKC_OK;
}

int gkc_count_pass(gkc_ctx* c)
{
    const uint32_t Pn = c->nb_partitions;
    const uint32_t n_seg = (uint32_t)c->segments.size();
    c->drain_pending();                                     // multi-GPU: the records other ranks sent must have arrived
    {   // a pass counted again (a retry after GKC_ERR_NOMEM, or gkc_finish_pass called twice) starts from a clean slate: what the
        // batches of the failed attempt added to the histogram, to the counters and to the result list must not be counted twice
        auto it = c->pass_outputs.find(c->pass);
        if (it != c->pass_outputs.end()) { for (void* p : it->second) c->dfree(p); it->second.clear(); }
        for (uint32_t p = 0; p < Pn; p++) c->datasets[(size_t)c->pass * Pn + p] = Dataset();
        gkc_stats& S = c->stats_now(); S.kmers_nb_distinct = 0; S.kmers_nb_solid = 0; S.oversize_buckets = 0;
        GKC_HIP(c, hipMemsetAsync(c->histo_of(c->pass), 0, ((size_t)c->histo_max + 1) * 8, c->stream));
    }
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

    // Batches of consecutive partitions, each bounded by a key budget that stays the SAME through the pass: equal batches ask the caching
    // allocator for the same block sizes again and again, so after the first batches no hipMalloc / hipFree happens at all (hipMalloc
    // costs ~22 ms per GB here; re-deriving the budget from the shrinking free memory made every batch a new size and a k=63 pass, where
    // the results take half the HBM, spent seconds in the allocator). Per key slot a batch needs the key twice (a split level uses
    // the ping-pong buffer) and 5 B of abundance planes — its working set, returned to the pool afterwards — and leaves one Count
    // record per SOLID distinct key resident, so the peak is at the END of the pass: all results + the last working sets.
    //  * the budget is what the working sets may take beside the results the whole pass will leave:
    //    (0.95 free - keys * rec * 1.05 d) / (lanes * work), d = solid records per key; capped, and cut into equal shares;
    //  * d comes from a small probe batch the first time the memory may bind (see below) and is kept by the context;
    //  * safety net: before every batch the commitments (finished results, the other lanes' running batches) are checked, and if the
    //    batch does not fit the extra lanes retire and the main lane halves its batches (slow: blocks change size; never seen when
    //    the plan holds).
    const size_t key_bytes = c->key_words == 1 ? 8 : 16, rec_bytes = c->key_words == 1 ? 16 : 32;
    const size_t work_per_key = 2 * key_bytes + 5;
    // Two LANES: Stage B's kernels are bound by different things (expand: store atoms and LDS, sorts: VALU, compaction: HBM), and a single
    // in-order stream leaves most of the chip waiting on whichever bound the current kernel has. Two host threads therefore take batches
    // from one queue, each on its own stream (thread-local stream override, cur_stream()): measured 264 -> 229 ms for the same work.
    const uint64_t total_keys = [&] { uint64_t t = 0; for (uint64_t v : part_keys) t += v; return t; }();
    const uint64_t max_part = [&] { uint64_t t = 0; for (uint64_t v : part_keys) t = std::max(t, v); return t; }();
    int lanes = getenv("GKC_STAGEB_LANES") ? atoi(getenv("GKC_STAGEB_LANES")) : 2;
    if (lanes < 1) lanes = 1;
    if (lanes > 4) lanes = 4;
    if (total_keys < 50000000ULL || c->key_budget) lanes = 1;               // small inputs (and the tests' tiny forced budgets): one lane
    bool tight = false;                                                       // memory is running out: the extra lanes retire, one lane finishes the pass
    const double avail0 = [&] {                                              // memory this pass may use: free now + blocks parked in the caching allocator
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = (size_t)8 << 30;
        return (double)(free_b + c->pool.cached_bytes);
    }();
    static const size_t cap_env = getenv("GKC_BATCH_KEYS") ? (size_t)atoll(getenv("GKC_BATCH_KEYS")) : 0;
    // Few, large batches: every batch ends with the drain of ~12 kernels (the expand kernels run one 6 ms workgroup per partition) and
    // eight host round trips; 8 -> 4 batches per 1.2e10 keys: 320 -> 304 ms. Equal shares, a whole number of batches per lane.
    // With a host sink (streamed results) the batches are three times smaller: the first records start over PCIe sooner and the copy that is left
    // when the last batch has been counted is shorter (1e8 reads, abundance-min 2: 431 -> ms per step with everything landed; profiles/r02_*)
    const size_t cap_default = c->key_words == 1 ? (size_t)3200000000ULL : (size_t)1600000000ULL;
    const size_t cap = cap_env ? cap_env : (c->sink ? cap_default / 3 : cap_default);   // the same with one lane or two
    uint64_t done_keys = 0, done_solid = 0;                                  // this pass: finished batches (keys, resident records) (guarded by plan_mu)
    // budget for a given solid-per-key ratio d. Deterministic in (free memory rounded to GB, total keys, d rounded up to 0.05): every
    // pass of a context plans the same sizes, so from the second pass on all blocks are parked already.
    const double avail_q = std::floor(avail0 / 1e9) * 1e9;
    int plan_lanes = (c->key_budget || total_keys < 50000000ULL) ? 1 : 2;                                  // one lane gets the same batches as two would (same blocks whichever way a pass runs) ...
    auto plan_budget = [&](double d) -> size_t {
        const double work = (double)plan_lanes * (double)work_per_key;
        const double dq = std::min(1.0, std::ceil(1.05 * d / 0.05) * 0.05);
        const double mem = (0.95 * avail_q - (double)total_keys * (double)rec_bytes * dq) / work;
        const uint64_t bmem = mem > (double)((size_t)1 << 20) ? (uint64_t)mem : ((uint64_t)1 << 20);
        const uint64_t b = std::min<uint64_t>(bmem, cap);
        const uint64_t per_round = b * (uint64_t)plan_lanes;
        const uint64_t rounds = std::max<uint64_t>((total_keys + per_round - 1) / per_round, 1);
        const uint64_t share = total_keys / (rounds * (uint64_t)plan_lanes);
        return (size_t)std::min<uint64_t>(b + max_part, share + share / 64 + max_part);           // a little over the share: no small last batch
    };
    size_t fixed_budget = 0;                                                 // set below, after the probe
    bool probe_pending = false; const size_t probe_keys = (size_t)std::max<uint64_t>(total_keys / 256, 16000000ULL);
    double inflight[4] = { 0, 0, 0, 0 };                                     // bytes each lane's running batch may still claim (working set + its results)
    size_t last_b[4] = { 0, 0, 0, 0 };
    auto per_key_now = [&]() -> double {
        const double d_est = done_keys ? std::min(1.0, 1.05 * (double)done_solid / (double)done_keys) : (c->d_hint > 0 ? std::min(1.0, 1.05 * c->d_hint) : 1.0);
        return (double)work_per_key + (double)rec_bytes * d_est;
    };
    auto budget_now = [&](int lane) -> size_t {                              // keys of the next batch of one lane (called under plan_mu)
        if (c->key_budget) return c->key_budget;
        double committed = 0; for (int l = 0; l < 4; l++) if (l != lane) committed += inflight[l];
        const double left = 0.98 * avail0 - (double)done_solid * (double)rec_bytes - committed;      // (the plan keeps 0.95: headroom between plan and net)
        const size_t fits = left > 0 ? (size_t)(left / per_key_now()) : 0;
        size_t b = fixed_budget;
        if (b > fits) {
            if (getenv("GKC_POOL_DEBUG")) fprintf(stderr, "[gkc plan] lane %d: budget %.3e does not fit (%.3e): done_solid %.3e committed %.1f GB\n", lane, (double)b, (double)fits, (double)done_solid, committed / 1e9);
            if (lanes > 1) tight = true;                                     // first the extra lanes retire ...
            if (lane != 0) return b;
            while (b > fits && b > ((size_t)1 << 20)) b /= 2;                // ... then the main lane's batches shrink
            c->slots_hint = 0;                                               // (exact buffer sizes from here on)
        }
        if (b != last_b[lane]) {                                             // another batch size: the parked blocks have the wrong sizes, and
            if (last_b[lane]) c->pool.trim();                                // reusing larger ones would keep exactly the memory that ran out
            last_b[lane] = b;
        }
        return b;
    };
    std::vector<void*>& outputs = c->pass_outputs[c->pass];
    std::mutex plan_mu; uint32_t next_p = 0; int first_rc = GKC_OK;
    auto carve = [&](std::vector<uint32_t>& batch, int lane) -> bool {   // next batch of consecutive partitions; false when nothing is left
        std::lock_guard<std::mutex> lk(plan_mu);
        batch.clear();
        if (first_rc != GKC_OK) return false;
        inflight[lane] = 0;
        size_t budget = budget_now(lane);
        if (tight && lane != 0) return false;
        if (probe_pending) { probe_pending = false; budget = probe_keys; }      // the pass's first batch is the small probe batch, every pass (same batches, same blocks)
        uint64_t acc = 0;
        while (next_p < Pn) {
            const uint32_t p = next_p;
            if (part_keys[p] == 0) {                           // nothing to count (e.g. a partition another rank owns): an empty, finished dataset
                Dataset& D = c->datasets[(size_t)c->pass * Pn + p];
                D.d_counts = nullptr; D.n_solid = 0; D.n_distinct = 0; D.n_kmers = 0; D.done = true;
                next_p++; continue;
            }
            if (!batch.empty() && acc + part_keys[p] > budget) break;
            batch.push_back(p); acc += part_keys[p]; next_p++;
        }
        // the expand kernels run one workgroup per partition, one workgroup per CU: a batch of 545 partitions takes three rounds on 256 CUs
        // with the last one 13 % full — whole multiples of the CU count leave no such tail
        static const size_t align = getenv("GKC_BATCH_ALIGN") ? (size_t)atoll(getenv("GKC_BATCH_ALIGN")) : 0;
        if (align && next_p < Pn && batch.size() > align)
            for (const size_t keep = batch.size() / align * align; batch.size() > keep; batch.pop_back()) { next_p = batch.back(); acc -= part_keys[batch.back()]; }
        inflight[lane] = (double)acc * per_key_now();
        return !batch.empty();
    };
    auto lane_main = [&](hipStream_t st, int lane) {
        (void)hipSetDevice(c->device);
        gkc_tl_stream = st;
        std::vector<uint32_t> batch;
        while (carve(batch, lane)) {
            const int r = (c->key_words == 1) ? count_batch<1, 2>(c, batch, part_keys, segs, outputs) : count_batch<2, 4>(c, batch, part_keys, segs, outputs);
            std::lock_guard<std::mutex> lk(plan_mu);
            inflight[lane] = 0;
            if (r != GKC_OK) { if (first_rc == GKC_OK) first_rc = r; break; }
            for (uint32_t p : batch) { const Dataset& D = c->datasets[(size_t)c->pass * Pn + p]; done_keys += D.n_kmers; done_solid += D.n_solid; }
        }
        (void)hipStreamSynchronize(st);
        gkc_tl_stream = nullptr;
    };
    (void)hipStreamSynchronize(c->stream);                                   // Stage A and the table uploads are complete before the lanes start
    // d not known yet and the memory may bind: count a small PROBE batch first (the first partitions holding ~0.4 % of the keys) and take
    // its ratio. The context keeps that first estimate (until the configuration changes), so every later pass plans the same sizes.
    // In later passes the same small batch is simply the first one in the queue (it runs beside the other lane's first batch).
    probe_pending = !c->key_budget && plan_budget(1.0) < plan_budget(1e-9);
    if (probe_pending && c->d_hint <= 0) {
        std::vector<uint32_t> batch;
        fixed_budget = probe_keys; c->slots_hint = 0;
        if (carve(batch, 0)) {
            const int r = (c->key_words == 1) ? count_batch<1, 2>(c, batch, part_keys, segs, outputs) : count_batch<2, 4>(c, batch, part_keys, segs, outputs);
            inflight[0] = 0; last_b[0] = 0;
            if (r != GKC_OK) { d_recptr.release(); d_recoff.release(); return r; }
            for (uint32_t p : batch) { const Dataset& D = c->datasets[(size_t)c->pass * Pn + p]; done_keys += D.n_kmers; done_solid += D.n_solid; }
            if (done_keys) c->d_hint = std::max(1e-6, (double)done_solid / (double)done_keys);
        }
    }
    fixed_budget = c->d_hint > 0 ? plan_budget(c->d_hint) : plan_budget(1.0);   // without a ratio the memory does not bind even at d = 1
    if (getenv("GKC_POOL_DEBUG")) fprintf(stderr, "[gkc plan] avail %.1f GB, keys %.3e, d_hint %.4f, lanes %d, budget %.3e\n", avail0 / 1e9, (double)total_keys, c->d_hint, lanes, (double)fixed_budget);
    if (!c->key_budget && fixed_budget < 250000000ULL) {                       // ... unless there is little room: one lane, larger batches
        lanes = 1; plan_lanes = 1; fixed_budget = c->d_hint > 0 ? plan_budget(c->d_hint) : plan_budget(1.0);
    }
    {   uint64_t nonempty = 0; for (uint64_t v : part_keys) nonempty += v != 0;
        const uint64_t avg_part = std::max<uint64_t>(total_keys / std::max<uint64_t>(nonempty, 1), 1);
        const uint64_t hint = (uint64_t)fixed_budget + ((uint64_t)fixed_budget / avg_part + 2) * ((3ull << MAX_SUB_BITS) + COMPACT_BLK);
        c->slots_hint = c->key_budget ? 0 : (hint + COMPACT_BLK - 1) / COMPACT_BLK * COMPACT_BLK;
    }
    for (int l = 1; l < lanes; l++)
        if (!c->lane_streams[l - 1] && hipStreamCreateWithFlags(&c->lane_streams[l - 1], hipStreamNonBlocking) != hipSuccess) { c->lane_streams[l - 1] = nullptr; lanes = l; break; }
    {
        std::vector<std::thread> extra;
        for (int l = 1; l < lanes; l++) extra.emplace_back(lane_main, c->lane_streams[l - 1], l);
        lane_main(c->stream, 0);
        for (auto& t : extra) t.join();
    }
    rc = first_rc;
    c->cv_done.notify_all();
    if (getenv("GKC_POOL_DEBUG")) fprintf(stderr, "[gkc pool] mallocs %llu failed %llu trims %llu, %.1f ms in hipMalloc, cached %.2f GB\n", (unsigned long long)c->pool.n_malloc,
                                          (unsigned long long)c->pool.n_fail, (unsigned long long)c->pool.n_trim, c->pool.malloc_ms, (double)c->pool.cached_bytes / 1e9);
    d_recptr.release(); d_recoff.release();
    return rc;
}

// explicit result checksum entry (used by the C-ABI)
int gkc_result_checksum_impl(gkc_ctx* c, uint64_t* checksum, uint64_t* sum_abundance)
{
    DevBuf d; GKC_TRY(c->ensure(d, 16));
    GKC_HIP(c, hipMemsetAsync(d.p, 0, 16, cur_stream(c)));
    for (const Dataset& D : c->datasets) {
        if (!D.done || !D.n_solid) continue;
        if (c->key_words == 1) hipLaunchKernelGGL((k_result_checksum<1>), dim3(1024), dim3(256), 0, cur_stream(c), (const uint64_t*)D.d_counts, D.n_solid, (unsigned long long*)d.p);
        else                   hipLaunchKernelGGL((k_result_checksum<2>), dim3(1024), dim3(256), 0, cur_stream(c), (const uint64_t*)D.d_counts, D.n_solid, (unsigned long long*)d.p);
    }
    uint64_t h[2];
    hipError_t e = hipMemcpyAsync(h, d.p, 16, hipMemcpyDeviceToHost, cur_stream(c));
    if (e == hipSuccess) e = hipStreamSynchronize(cur_stream(c));
    d.release();
    if (e != hipSuccess) GKC_FAIL(c, GKC_ERR_HIP, "result checksum failed: %s", hipGetErrorString(e));
    *checksum = h[0]; *sum_abundance = h[1];
    return GKC_OK;
}
