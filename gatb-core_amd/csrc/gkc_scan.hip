// gkc_scan.hip — Stage A on gfx950: reads -> super-k-mer records bucketed by minimizer partition.
//
// Replaces (reference, under /root/reference/gatb-core/src/gatb/):
//   A1 Data::ConvertASCII                      tools/misc/api/Data.hpp:185
//   A2 validity window of ModelAbstract::iterate kmer/impl/Model.hpp:749-759
//   A3 ModelMinimizer LUT / next / rescan       kmer/impl/Model.hpp:1032-1064, 1107-1139, 1254-1287
//   A4 Sequence2SuperKmer + KmerFunctor         kmer/impl/Sequence2SuperKmer.hpp:81-159
//   A5 FillPartitions::processSuperkmer         kmer/impl/SortingCountAlgorithm.cpp:1081-1151
//   A6 SuperKmer::save + SuperKmerBinFiles      kmer/impl/Model.hpp:1386-1471, tools/storage/impl/Storage.cpp:360-580
//
// MI355X design (not a translation of the per-thread rolling loop of the reference):
//   * one 512-thread workgroup per tile of 8192 k-mer start positions; ASCII is read once with coalesced 16-byte loads
//     and turned into three bit-planes in LDS (big-endian 2-bit, little-endian 2-bit, invalid mask);
//   * the minimizer order key of every m-mer is computed position-parallel (lexicographic/KMC2 order: pure ALU, the
//     reverse complement comes from the little-endian plane for free; frequency order: one 4-byte gather from an
//     L2-resident table) and written to LDS; the window minimum of k-m+1 keys uses a shared-core + prefix/suffix-min
//     scheme, ~3 LDS reads per k-mer instead of k-m+1;
//   * super-k-mer boundaries come from one workgroup-wide max-scan (wave64 shuffles + 4 partials in LDS); records are
//     emitted at run ENDS so a single forward scan suffices, including the length cap;
//   * the k-mer integer itself is never formed here: a record is a 16/32-byte aligned copy of the 2-bit plane.
//   The pass runs twice per batch (count, then emit at exact offsets): no bucket can overflow whatever the skew.
#include "gkc_common.hpp"
#include "gkc_device.hpp"
#include <algorithm>

struct ScanParams {
    const uint8_t* bases; uint64_t n_bases;
    const uint32_t* rsbits;
    uint32_t k, m, nb_mm, maxs, maxs_magic, mmask, mask_ma1;   // maxs_magic = ceil(2^32 / maxs): d / maxs == umulhi(d, magic) for d < 2^13
    int freq_mode;
    const uint32_t* mkey_lut; const uint32_t* key2val; uint32_t default_key;
    const uint16_t* repart; uint32_t nb_passes, pass;
    unsigned long long* cnt_rec; unsigned long long* cnt_kmers; unsigned long long* cursor;
    uint64_t* arena;
    int identity_part;            // sampling mode: the 'partition' of a super-k-mer is its minimizer value (4^m bins)
    uint64_t n_tiles; uint32_t n_parts;
    uint32_t* wg_cnt;                    // LDSPART count: [grid][P] records of this workgroup
    const unsigned long long* wg_base;   // LDSPART emit : [grid][P] first arena slot of this workgroup in the partition
    const unsigned long long* rec_off;   // LDSPART emit : [P] first record of the partition in the arena
    uint32_t* desc; uint32_t desc_cap_wg; uint2* desc_tile; uint32_t* desc_overflow;   // record descriptors written by the count pass
    uint8_t* desc_fine; uint32_t fine_shift;     // two-level scan, <= 16 partitions per group: `repart` is the FINE table, the group is partition >> fine_shift, and the
                                                 // partition's low bits travel beside the descriptor into the record's 4 spare bits (k_refine_* read them back)
    unsigned long long* gstats;   // [0] valid k-mers [1] invalid k-mers [2] records emitted/counted
};

// ------------------------------------------------------------------------------------------------
// read-start bitmask: bit g set iff some read starts at base g, or g == n_bases (offsets[n_reads])
// ------------------------------------------------------------------------------------------------
// The offsets come from the caller: an offset beyond n_bases, a decreasing pair, offsets[0] != 0 or offsets[n_reads] != n_bases sets *bad
// (the push then fails with GKC_ERR_ARG) and never touches memory outside the mask.
// also the read-length statistics of BankStats::update (BankKmers.hpp:176-186) when len_stats != nullptr: [0] max of ~length (= ~shortest), [1] longest, [2] sum of squares.
// Grid-stride: a thread keeps its own minimum / maximum / sum, a wave reduces them once at the end — three atomics per WAVE of the launch, not per 64 reads
// (78 000 workgroups hitting three addresses took 21 ms per push).
__global__ void k_mark_read_starts(const uint64_t* __restrict__ offsets, uint64_t n_entries, uint64_t n_bases, uint32_t* __restrict__ bits, uint32_t* __restrict__ bad,
                                   unsigned long long* __restrict__ len_stats)
{
    unsigned long long inv = 0, mx = 0, sq = 0; bool any = false;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_entries; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t g = offsets[i], nxt = i + 1 < n_entries ? offsets[i + 1] : 0;
        const bool wrong = g > n_bases || (i + 1 < n_entries && nxt < g) || (i == 0 && g != 0) || (i + 1 == n_entries && g != n_bases);
        if (wrong && bad) *bad = 1u;
        if (g <= n_bases) atomicOr(&bits[g >> 5], 1u << (g & 31));
        if (len_stats && i + 1 < n_entries && !wrong) {
            const unsigned long long len = nxt - g;
            inv = ~len > inv ? ~len : inv; mx = len > mx ? len : mx; sq += len * len; any = true;
        }
    }
    if (len_stats) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const unsigned long long a = __shfl_xor(inv, d, 64), b = __shfl_xor(mx, d, 64);
            inv = a > inv ? a : inv; mx = b > mx ? b : mx; sq += __shfl_xor(sq, d, 64);
        }
        const bool wave_any = __any(any) != 0;
        if ((threadIdx.x & 63) == 0 && wave_any) { atomicMax(&len_stats[0], inv); atomicMax(&len_stats[1], mx); atomicAdd(&len_stats[2], sq); }
    }
}

// LDS index of per-position arrays: thread t owns positions 16t..16t+15; one pad word per 16 entries makes the lane
// stride 17 dwords, so the 32 lanes of an LDS access group hit 32 different banks (stride 16 would be a 16-way conflict)
#define MKI(p) ((p) + ((p) >> 4))
constexpr int BE_PAD = 12;   // zero words after the tile planes so record extraction may read past the halo

// builds the 16/32-byte record of the super-k-mer that starts at tile-local position `start` (nbk k-mers) from the
// big-endian 2-bit plane in LDS and stores it at arena slot `slot`
template <int RW>
__device__ __forceinline__ void store_record(const uint32_t* s_be, int start, uint32_t nbk, uint32_t k, uint64_t* arena, unsigned long long slot, const uint64_t spare = 0 /* the 4 bits below the nucleotides */)
{
    const int w0 = start >> 4, sh = 2 * (start & 15);
    uint64_t A[RW + 1];
#pragma unroll
    for (int i = 0; i <= RW; i++) A[i] = ((uint64_t)s_be[w0 + 2 * i] << 32) | s_be[w0 + 2 * i + 1];
    uint64_t B[RW];
#pragma unroll
    for (int i = 0; i < RW; i++) B[i] = sh ? ((A[i] << sh) | (A[i + 1] >> (64 - sh))) : A[i];
    uint64_t R[RW];
    R[0] = ((uint64_t)nbk << 56) | (B[0] >> 8);
#pragma unroll
    for (int i = 1; i < RW; i++) R[i] = (B[i - 1] << 56) | (B[i] >> 8);
    // zero everything after the k+nbK-1 nucleotides of this super-k-mer
    const int n = (int)k + (int)nbk - 1;
    if (n < 28) R[0] &= ~((1ULL << (56 - 2 * n)) - 1);
#pragma unroll
    for (int i = 1; i < RW; i++) {
        const int have = n - 28 - 32 * (i - 1);       // nucleotides of word i that belong to the record
        if (have <= 0) R[i] = 0;
        else if (have < 32) R[i] &= ~((1ULL << (64 - 2 * have)) - 1);
    }
    R[RW - 1] |= spare;
    uint64_t* dst = arena + slot * RW;
#pragma unroll
    for (int i = 0; i < RW; i += 2) store16(dst + i, R[i], R[i + 1]);
}

// 16 ASCII bases (4 dwords) -> little-endian 2-bit word + invalid mask (A1), SWAR
__device__ __forceinline__ void encode16(const uint32_t (&dw)[4], uint32_t& le, uint32_t& bad)
{
    le = 0; bad = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t wv = dw[q];
        uint32_t c4 = (wv >> 1) & 0x03030303u;
        c4 = (c4 | (c4 >> 6)) & 0x000F000Fu;
        c4 = (c4 | (c4 >> 12)) & 0xFFu;                                     // c0 | c1<<2 | c2<<4 | c3<<6
        le |= c4 << (8 * q);
        const uint32_t u = wv & 0xDFDFDFDFu;
        auto nz = [](uint32_t v) { return (((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v) & 0x80808080u; };   // 0x80 where the byte is non-zero
        uint32_t b4 = nz(u ^ 0x41414141u) & nz(u ^ 0x43434343u) & nz(u ^ 0x47474747u) & nz(u ^ 0x54545454u);
        b4 >>= 7;
        b4 = (b4 | (b4 >> 7) | (b4 >> 14) | (b4 >> 21)) & 0xFu;
        bad |= b4 << (4 * q);
    }
}
__device__ __forceinline__ uint32_t rev2bit(uint32_t x)                         // reverse the 16 two-bit groups
{
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
    return __builtin_bswap32(x);
}
__device__ __forceinline__ void load16(const uint8_t* bases, uint64_t g0, uint64_t n_bases, uint32_t (&dw)[4])
{
    if (g0 + 16 <= n_bases) {
        const uint4 v = *reinterpret_cast<const uint4*>(bases + g0);
        dw[0] = v.x; dw[1] = v.y; dw[2] = v.z; dw[3] = v.w;
    } else {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t x = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) { const uint64_t g = g0 + 4 * q + b; const uint32_t c = (g < n_bases) ? bases[g] : 0u; x |= c << (8 * b); }
            dw[q] = x;
        }
    }
}

// LDSPART: persistent workgroups with a static tile assignment (tile = blockIdx.x, += gridDim.x, identical in the count
// and the emit launch). Per-partition counters / record cursors live in LDS (64-bit LDS atomics): the count launch
// leaves a [workgroup][partition] matrix, a tiny prefix kernel turns it into private record ranges, and the emit launch
// needs no global atomic at all. !LDSPART (nb_partitions too large for LDS): global atomics per record.
template <bool EMIT, int RW, bool LDSPART, bool FINE = false /* count pass of the two-level scan with the fine bits kept beside the descriptors */>
__global__ __launch_bounds__(SCAN_THREADS, 4) void k_scan_tile(ScanParams P)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_part[];       // per-partition record counter of this workgroup (4 B)
    __shared__ uint32_t s_be[SCAN_WORDS + BE_PAD];
    __shared__ uint32_t s_le[SCAN_WORDS + BE_PAD];
    __shared__ uint16_t s_bad[SCAN_WORDS + 8];
    __shared__ uint32_t s_rs[SCAN_TILE / 32 + 8];
    __shared__ uint32_t s_mk[MKI(SCAN_TILE + 16 * SCAN_HALO_WORDS + 16)];   // padded: see MKI
    __shared__ uint16_t s_end[MKI(SCAN_TILE)];                             // per run end: (nbK-1) << 4 | j
    __shared__ uint32_t s_lastmz[SCAN_THREADS], s_firstmz[SCAN_THREADS + 1];
    __shared__ uint8_t  s_lastvalid[SCAN_THREADS], s_firstvalid[SCAN_THREADS + 1];
    __shared__ int      s_wavecarry[SCAN_THREADS / 64];
    __shared__ unsigned long long s_stat[3];
    __shared__ uint32_t s_toff, s_tcnt;

    const int t = threadIdx.x;
    const uint32_t k = P.k, m = P.m;

    if (t < 3) s_stat[t] = 0;
    if (t == 0) { s_toff = 0; s_tcnt = 0; }
    if (t < BE_PAD) { s_be[SCAN_WORDS + t] = 0; s_le[SCAN_WORDS + t] = 0; }
    if (t < 8) s_bad[SCAN_WORDS + t] = 0;
    if (LDSPART) {
        for (uint32_t p = t; p < P.n_parts; p += SCAN_THREADS)
            s_part[p] = 0u;
    }
    uint32_t nv_acc = 0, ni_acc = 0, n_rec = 0;

  // the tile's bases (16 per thread + the halo words) and read-start bits are fetched one tile AHEAD into registers: with two workgroups
  // per CU and ~7 barriers per tile nothing else hides the global-load latency (measured: a third of the kernel)
  static_assert(SCAN_WORDS <= 2 * SCAN_THREADS && SCAN_TILE / 32 + 8 <= SCAN_THREADS, "one or two words and one read-start word per thread");
  uint32_t pfA[4] = {0, 0, 0, 0}, pfB[4] = {0, 0, 0, 0}, pfR = 0;
  auto prefetch = [&](uint64_t tile_) {
      const uint64_t t0_ = tile_ * SCAN_TILE; const int tt = threadIdx.x;
      load16(P.bases, t0_ + 16ull * tt, P.n_bases, pfA);
      if (tt + SCAN_THREADS < SCAN_WORDS) load16(P.bases, t0_ + 16ull * (tt + SCAN_THREADS), P.n_bases, pfB);
      if (tt < SCAN_TILE / 32 + 8) pfR = P.rsbits[t0_ / 32 + tt];
  };
  if (blockIdx.x < P.n_tiles) prefetch(blockIdx.x);
  for (uint64_t tile = blockIdx.x; tile < P.n_tiles; tile += gridDim.x) {
    const uint64_t t0 = tile * SCAN_TILE;
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));                                // keep per-lane address math inside the loop: hoisting it
                                                               // (LICM) costs ~150 VGPRs and halves the occupancy
    __syncthreads();                                           // LDS of the previous tile fully consumed
    if (!EMIT && LDSPART && P.desc && t == 0 && tile != blockIdx.x) {      // close the previous tile's descriptor range (its reservations are all in:
        P.desc_tile[tile - gridDim.x] = make_uint2(s_toff, s_tcnt);        // the barrier above; nobody touches the two words again before step 5)
        s_toff += s_tcnt; s_tcnt = 0;
    }

    // ---- step 0: ASCII -> bit planes (A1) ----
    {
        uint32_t le, bad;
        encode16(pfA, le, bad);
        s_be[t] = rev2bit(le); s_le[t] = le; s_bad[t] = (uint16_t)bad;
        if (t + SCAN_THREADS < SCAN_WORDS) { encode16(pfB, le, bad); s_be[t + SCAN_THREADS] = rev2bit(le); s_le[t + SCAN_THREADS] = le; s_bad[t + SCAN_THREADS] = (uint16_t)bad; }
        if (t < SCAN_TILE / 32 + 8) s_rs[t] = pfR;
    }
    if (tile + gridDim.x < P.n_tiles) prefetch(tile + gridDim.x);     // in flight during the rest of this tile
    __syncthreads();

    // ---- step 1: order key of the m-mer starting at every position (A3: LUT semantics) ----
    for (int w = t; w < SCAN_WORDS; w += SCAN_THREADS) {
        const uint32_t xh = s_be[w], xl = s_be[w + 1], yl = s_le[w], yh = s_le[w + 1];
        const uint32_t fsh = 32u - 2u * m, rcx = 0xAAAAAAAAu & P.mmask;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            // forward m-mer = the top 2m bits of the big-endian window shifted left by j nucleotides; reverse complement = the low 2m bits of the
            // little-endian window shifted right by j: one funnel shift (v_alignbit) each instead of 64-bit shifts
            const uint32_t f32 = j ? __builtin_amdgcn_alignbit(xh, xl, 32 - 2 * j) : xh;
            uint32_t fw = f32 >> fsh;
            uint32_t key;
            if (P.freq_mode) {
                key = P.mkey_lut[fw];                         // order key of canonical(fw) under (freq_order[c], c)
            } else {
                uint32_t rc = ((j ? __builtin_amdgcn_alignbit(yh, yl, 2 * j) : yl) & P.mmask) ^ rcx;
                uint32_t c = fw < rc ? fw : rc;               // canonical m-mer
                uint32_t a = ~(c | (c >> 2));
                a = (a >> 1) & a & P.mask_ma1;                // "AA" anywhere but as prefix (KMC2 rule)
                key = a ? P.mmask : c;
            }
            s_mk[17 * w + j] = key;
        }
    }
    __syncthreads();

    // ---- step 2: minimizer = window minimum of nb_mm keys (always the true minimum: Model.hpp:1107-1139 keeps it by
    //      rescanning whenever the tracked one leaves the window) ----
    const int p0 = 16 * t;
    const uint32_t Wn = P.nb_mm;
    uint32_t mz[16];
    if (Wn >= 16) {
        // MKI(16 t + i) = 17 t + i + (i >> 4): the part that depends on i is wave-uniform (scalar), one vector add per LDS address
        const uint32_t* mk_t = s_mk + 17 * t;
        uint32_t core = P.default_key;                        // the default minimizer 4^m-1 takes part (Model.hpp:1260)
        for (uint32_t i = 15; i < Wn; i++) { uint32_t v = mk_t[i + (i >> 4)]; core = v < core ? v : core; }
        uint32_t suf = 0xFFFFFFFFu;
        uint32_t sufL[16];
        sufL[15] = suf;
#pragma unroll
        for (int j = 14; j >= 0; j--) { uint32_t v = mk_t[j]; suf = v < suf ? v : suf; sufL[j] = suf; }
        uint32_t pre = 0xFFFFFFFFu;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            uint32_t r = sufL[j] < core ? sufL[j] : core;
            mz[j] = pre < r ? pre : r;
            const uint32_t i = Wn + (uint32_t)j;
            uint32_t v = mk_t[i + (i >> 4)]; pre = v < pre ? v : pre;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            uint32_t best = P.default_key;
            for (uint32_t i = 0; i < Wn; i++) { uint32_t v = s_mk[MKI(p0 + j + i)]; best = v < best ? v : best; }
            mz[j] = best;
        }
    }

    // ---- step 3: which positions hold a k-mer, and which of those are valid (A2) ----
    // window-OR of the invalid / read-start bit planes over k (k-1) positions for all 16 positions of the thread at once:
    // doubling (windows 1,2,4,...,32) on a 128-bit value, then the binary decomposition of the window length.
    uint32_t existsmask, validmask;
    {
        const int q = t >> 1, off = (t & 1) * 16;
        uint64_t rlo = (uint64_t)s_rs[q] | ((uint64_t)s_rs[q + 1] << 32);
        uint64_t rhi = (uint64_t)s_rs[q + 2] | ((uint64_t)s_rs[q + 3] << 32);
        if (off) { rlo = (rlo >> 16) | (rhi << 48); rhi >>= 16; }
        rlo = (rlo >> 1) | (rhi << 63); rhi >>= 1;            // read starts matter inside (g, g+k-1]: shift by one, window k-1
        uint64_t blo = (uint64_t)s_bad[t] | ((uint64_t)s_bad[t + 1] << 16) | ((uint64_t)s_bad[t + 2] << 32) | ((uint64_t)s_bad[t + 3] << 48);
        uint64_t bhi = (uint64_t)s_bad[t + 4] | ((uint64_t)s_bad[t + 5] << 16) | ((uint64_t)s_bad[t + 6] << 32) | ((uint64_t)s_bad[t + 7] << 48);
        auto window_or16 = [](uint64_t lo, uint64_t hi, uint32_t len) -> uint32_t {
            // returns bits j=0..15 : OR of input bits j .. j+len-1   (len in [0,63])
            uint32_t acc = 0, ofs = 0;
            uint64_t alo = lo, ahi = hi;                      // window 1
#pragma unroll
            for (uint32_t p = 1; p <= 32; p <<= 1) {
                if (len & p) { acc |= (uint32_t)(ofs ? ((alo >> ofs) | (ahi << (64 - ofs))) : alo); ofs += p; }
                const uint64_t nlo = alo | ((alo >> p) | (ahi << (64 - p))), nhi = ahi | (ahi >> p);   // window 2p
                alo = nlo; ahi = nhi;
            }
            return acc & 0xFFFFu;
        };
        // read starts (one per read) and invalid letters (rare) are SPARSE: instead of the doubling network, walk the set bits of the
        // window and OR in the range of positions each one reaches — 0..2 iterations per wave on ordinary reads, any density stays exact
        auto window_or16_sparse = [](uint64_t lo, uint64_t hi, uint32_t len) -> uint32_t {
            // bits j=0..15 : OR of input bits j .. j+len-1   (len in [0,63]): input bit b reaches j in [b-len+1, b]
            if (len == 0) return 0u;
            uint32_t acc = 0;
            const uint32_t top = 15u + len;                                   // first input bit that no longer matters (<= 78)
            uint64_t w = top >= 64 ? lo : (lo & ((1ULL << top) - 1));
            while (w) {
                const uint32_t b = (uint32_t)__builtin_ctzll(w); w &= w - 1;
                const uint32_t j1 = b < 15u ? b : 15u, j0 = b + 1u > len ? b + 1u - len : 0u;
                acc |= ((2u << j1) - 1u) & ~((1u << j0) - 1u);
            }
            if (top > 64) {
                uint64_t v = hi & ((1ULL << (top - 64)) - 1);
                while (v) {
                    const uint32_t b = 64u + (uint32_t)__builtin_ctzll(v); v &= v - 1;
                    const uint32_t j0 = b + 1u - len;                         // b >= 64 > 15: reaches j in [b-len+1, 15]
                    if (j0 <= 15u) acc |= 0xFFFFu & ~((1u << j0) - 1u);
                }
            }
            return acc & 0xFFFFu;
        };
        (void)window_or16;
        const uint32_t rsany = window_or16_sparse(rlo, rhi, k - 1);
        const uint32_t badany = window_or16_sparse(blo, bhi, k);
        // k-mers may not run past the end of the batch
        const long long lim = (long long)P.n_bases - (long long)k - (long long)(t0 + p0);      // last j that still fits
        const uint32_t fits = lim >= 15 ? 0xFFFFu : (lim < 0 ? 0u : ((2u << (uint32_t)lim) - 1u));
        existsmask = ~rsany & fits;
        validmask = existsmask & ~badany;
    }

    // ---- step 4: natural super-k-mer starts and the workgroup-wide "last start" max-scan (A4) ----
    s_lastmz[t] = mz[15]; s_lastvalid[t] = (validmask >> 15) & 1;
    s_firstmz[t] = mz[0]; s_firstvalid[t] = validmask & 1;
    if (t == 0) { s_firstvalid[SCAN_THREADS] = 0; s_firstmz[SCAN_THREADS] = 0; }
    __syncthreads();
    uint32_t nsmask = 0;
    {
        const bool pv = t > 0 ? (s_lastvalid[t - 1] != 0) : false;  // tile start forces a new super-k-mer
        const uint32_t pm = t > 0 ? s_lastmz[t - 1] : 0;
        // a valid position starts a run when the position before is not valid or holds another minimizer: whole-thread bit masks
        uint32_t neq = mz[0] != pm ? 1u : 0u;
#pragma unroll
        for (int j = 1; j < 16; j++) neq |= (mz[j] != mz[j - 1] ? 1u : 0u) << j;
        const uint32_t pvmask = (validmask << 1) | (pv ? 1u : 0u);
        nsmask = validmask & (~pvmask | neq) & 0xFFFFu;
    }
    int carry;   // position (tile-local) of the last natural start before p0, or -1
    {
        int v = nsmask ? (p0 + 31 - __clz((int)nsmask)) : -1;
        int lane = t & 63, wave = t >> 6;
        int x = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { int y = __shfl_up(x, d, 64); if (lane >= d) x = y > x ? y : x; }
        if (lane == 63) s_wavecarry[wave] = x;
        __syncthreads();
        int wc = -1;
        for (int w2 = 0; w2 < wave; w2++) { int y = s_wavecarry[w2]; wc = y > wc ? y : wc; }
        int ex = __shfl_up(x, 1, 64);
        if (lane == 0) ex = -1;
        carry = ex > wc ? ex : wc;
    }

    // ---- step 5: emit one record per run end (A4 cap, A5 pass filter + partition, A6 bucket write) ----
    // run ends are first compacted per thread into LDS (own 16 slots), so the divergent emission loop runs
    // max-over-lanes(#ends) times instead of 16
    {
        int ls = carry, n_end = 0;
        const bool nxt_valid = s_firstvalid[t + 1] != 0;       // t==255: sentinel (tile end)
        const uint32_t nxt_mz = s_firstmz[t + 1];
        const int maxs = (int)P.maxs;
        // The cap (a run cut every maxs k-mers) only ever bites on runs of a REPEATED minimizer value (a minimizer stays at most nb_mm <= maxs
        // positions in the window otherwise): a thread whose positions cannot reach rank maxs-1 of their run takes run ends straight from
        // the bit masks (no division, no per-position arithmetic); the test is wave-uniform so both loops stay divergence-free.
        const uint32_t goes_on = validmask & ~nsmask;                      // positions that continue the run of the position before
        const int jcap = carry + maxs - 1 - p0;                           // where the run entering this thread would reach rank maxs-1
        const bool may_cap = maxs < 18 || ((goes_on & 1u) && (jcap <= 0 || (jcap <= 15 && (goes_on & ((2u << jcap) - 1u)) == ((2u << jcap) - 1u))));
        const bool fast = __ballot(may_cap) == 0ull;
        uint32_t ends = 0;
        if (fast) {
            // run ends as a bit mask; the 16 minimizers go to the thread's own slots with static indices (no per-position branch), the
            // emission loop below walks the set bits
            const uint32_t b15 = (!nxt_valid || (nxt_mz != mz[15])) ? 1u : 0u;
            ends = validmask & ((((~(validmask >> 1)) | (nsmask >> 1)) & 0x7FFFu) | (b15 << 15));
#pragma unroll
            for (int j = 0; j < 16; j++) s_mk[MKI(p0) + j] = mz[j];
            n_end = __popc(ends);
        } else
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const bool v = (validmask >> j) & 1;
            const int p = p0 + j;
            if ((nsmask >> j) & 1) ls = p;
            bool boundary;
            if (j < 15) boundary = !((validmask >> (j + 1)) & 1) || ((nsmask >> (j + 1)) & 1);
            else boundary = !nxt_valid || (nxt_mz != mz[15]);
            // run length so far and its position inside the cap: exact division by the launch constant `maxs` through a
            // multiply-high (a hardware integer division here would be ~1/3 of the kernel's instructions)
            const uint32_t d0 = (uint32_t)(p - ls);                           // < tile size whenever v is set
            const uint32_t q0 = __umulhi(d0, P.maxs_magic);
            const uint32_t r0 = d0 - q0 * (uint32_t)maxs;                    // (p - ls) % maxs
            const bool is_end = v && (boundary || r0 + 1 == (uint32_t)maxs);
            if (is_end) {
                const int start = ls + (int)(q0 * (uint32_t)maxs);
                s_mk[MKI(p0) + n_end] = mz[j];                                    // own slots; s_mk is dead after step 2
                s_end[MKI(p0) + n_end] = (uint16_t)(((uint32_t)(p - start) << 4) | (uint32_t)j);
                n_end++;
            }
        }
        // descriptor stream of the count pass (LDSPART): (partition, nbK, start) per record, so that the emit pass does
        // not have to recompute minimizers. Slots are reserved per thread; filtered-out records leave a ~0 hole.
        uint32_t dbase = 0; bool dstore = false;
        if (!EMIT && LDSPART && P.desc) {
            // one LDS add per WAVE (512 returning adds on one word serialise): wave prefix of the counts, lane 0 reserves the wave's total
            const int lane = t & 63;
            uint32_t x = (uint32_t)n_end;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
            const uint32_t wtot = __shfl(x, 63, 64);
            uint32_t wbase = 0;
            if (lane == 0 && wtot) wbase = atomicAdd(&s_tcnt, wtot);
            wbase = __shfl(wbase, 0, 64);
            if (n_end) {
                dbase = s_toff + wbase + x - (uint32_t)n_end;
                dstore = dbase + (uint32_t)n_end <= P.desc_cap_wg;
                if (!dstore) *P.desc_overflow = 1u;
            }
        }
        auto next_end = [&](int e, uint32_t& nbk, uint32_t& key, int& start) {          // the e-th run end of the thread (called in order)
            if (fast) {
                const uint32_t j = (uint32_t)__builtin_ctz(ends); ends &= ends - 1u;
                const uint32_t below = nsmask & ((2u << j) - 1u);                       // run starts at or before j inside the thread
                start = below ? p0 + 31 - __clz((int)below) : carry;
                nbk = (uint32_t)(p0 + (int)j - start) + 1u;
                key = s_mk[MKI(p0) + j];
            } else {
                const uint32_t info = s_end[MKI(p0) + e];
                nbk = (info >> 4) + 1;
                start = p0 + (int)(info & 15u) - (int)(info >> 4);
                key = s_mk[MKI(p0) + e];
            }
        };
        if (!EMIT && LDSPART) {
            // count pass: the minimizer -> partition gathers (L2) of up to 4 run ends are issued back to back, then consumed: a thread has 1..3 ends
            // on ordinary reads, and a divergent one-end-per-iteration loop would expose one gather latency per end
            uint32_t* dst = P.desc + (uint64_t)blockIdx.x * P.desc_cap_wg + dbase;
#pragma unroll 1
            for (int e0 = 0; e0 < n_end; e0 += 4) {
                uint32_t dpart[4], dlow[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    dlow[u] = 0xFFFFFFFFu; dpart[u] = 0;
                    if (e0 + u < n_end) {
                        uint32_t nbk, key; int start;
                        next_end(e0 + u, nbk, key, start);
                        const uint32_t value = P.freq_mode ? P.key2val[key] : key;
                        if (!(P.nb_passes > 1 && (value % P.nb_passes) != P.pass)) {     // SortingCountAlgorithm.cpp:1083
                            dpart[u] = P.identity_part ? value : P.repart[value];
                            dlow[u] = ((nbk - 1) << DESC_START_BITS) | (uint32_t)start;
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; u++) if (e0 + u < n_end) {
                    if (dlow[u] == 0xFFFFFFFFu) { if (dstore) dst[e0 + u] = 0xFFFFFFFFu; continue; }      // filtered out: a hole in the descriptor stream
                    n_rec++;
                    const uint32_t grp = FINE ? dpart[u] >> P.fine_shift : dpart[u];
                    atomicAdd(&s_part[grp], 1u);
                    if (dstore) {
                        dst[e0 + u] = (grp << (DESC_START_BITS + DESC_NBK_BITS)) | dlow[u];
                        if constexpr (FINE) P.desc_fine[(uint64_t)blockIdx.x * P.desc_cap_wg + dbase + e0 + u] = (uint8_t)(dpart[u] & ((1u << P.fine_shift) - 1u));
                    }
                }
            }
        } else
#pragma unroll 1
        for (int e = 0; e < n_end; e++) {
            uint32_t nbk, key; int start;
            next_end(e, nbk, key, start);
            const uint32_t value = P.freq_mode ? P.key2val[key] : key;
            if (P.nb_passes > 1 && (value % P.nb_passes) != P.pass) continue;          // SortingCountAlgorithm.cpp:1083
            const uint32_t full = P.identity_part ? value : P.repart[value];
            const uint32_t part = full >> P.fine_shift;
            n_rec++;
            if (!EMIT) {
                { atomicAdd(&P.cnt_rec[part], 1ULL); atomicAdd(&P.cnt_kmers[part], (unsigned long long)nbk); }
            } else {
                const unsigned long long slot = LDSPART ? (P.wg_base[(uint64_t)blockIdx.x * P.n_parts + part] + atomicAdd(&s_part[part], 1u))
                                                        : atomicAdd(&P.cursor[part], 1ULL);
                store_record<RW>(s_be, start, nbk, k, P.arena, slot, (uint64_t)(full & ((1u << P.fine_shift) - 1u)));
            }
        }
    }

    nv_acc += __popc(validmask); ni_acc += __popc(existsmask & ~validmask);
  }   // tile loop
    if (!EMIT && LDSPART && P.desc && blockIdx.x < P.n_tiles) { // close the last tile's descriptor range
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint64_t last = blockIdx.x + (P.n_tiles - 1 - blockIdx.x) / gridDim.x * (uint64_t)gridDim.x;
            P.desc_tile[last] = make_uint2(s_toff, s_tcnt);
        }
    }

    // ---- statistics (Sequence2SuperKmer.hpp:103,108) ----
    {
        uint32_t nv = nv_acc, ni = ni_acc;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { nv += __shfl_down(nv, d, 64); ni += __shfl_down(ni, d, 64); n_rec += __shfl_down(n_rec, d, 64); }
        if ((t & 63) == 0) { atomicAdd(&s_stat[0], (unsigned long long)nv); atomicAdd(&s_stat[1], (unsigned long long)ni); atomicAdd(&s_stat[2], (unsigned long long)n_rec); }
        __syncthreads();
        if (t < 3 && s_stat[t]) atomicAdd(&P.gstats[t], s_stat[t]);
    }
    if (LDSPART && !EMIT) {
        for (uint32_t p = t; p < P.n_parts; p += SCAN_THREADS) P.wg_cnt[(uint64_t)blockIdx.x * P.n_parts + p] = s_part[p];
    }
}

// Emit pass driven by the count pass's descriptors: only the big-endian 2-bit plane is rebuilt (no m-mer keys, no window
// minimum, no scans); one thread per record builds and stores it at its exact slot (LDS counter + private base).
template <int RW>
__global__ __launch_bounds__(SCAN_THREADS) void k_emit_desc(ScanParams P)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long s_cur[];   // absolute arena cursor per partition (this kernel's
                                                                                 // static LDS is tiny, so 8 B per partition fit; a
                                                                                 // per-record gather of the base from HBM/L2 would
                                                                                 // triple the kernel's read traffic)
    __shared__ uint32_t s_be[SCAN_WORDS + BE_PAD];
    if (threadIdx.x < BE_PAD) s_be[SCAN_WORDS + threadIdx.x] = 0;
    uint32_t* s_km = reinterpret_cast<uint32_t*>(s_cur + P.n_parts);              // k-mers per partition of this workgroup
    for (uint32_t p = threadIdx.x; p < P.n_parts; p += SCAN_THREADS) { s_cur[p] = P.wg_base[(uint64_t)blockIdx.x * P.n_parts + p]; s_km[p] = 0u; }
    const uint32_t* desc = P.desc + (uint64_t)blockIdx.x * P.desc_cap_wg;
    const uint8_t* dfine = P.fine_shift ? P.desc_fine + (uint64_t)blockIdx.x * P.desc_cap_wg : nullptr;
    for (uint64_t tile = blockIdx.x; tile < P.n_tiles; tile += gridDim.x) {
        int t = threadIdx.x;
        asm volatile("" : "+v"(t));
        const uint64_t t0 = tile * SCAN_TILE;
        __syncthreads();
        for (int w = t; w < SCAN_WORDS; w += SCAN_THREADS) {
            uint32_t dw[4], le, bad;
            load16(P.bases, t0 + 16ull * w, P.n_bases, dw);
            encode16(dw, le, bad);
            s_be[w] = rev2bit(le);
        }
        __syncthreads();
        const uint2 td = P.desc_tile[tile];
        for (uint32_t i = t; i < td.y; i += SCAN_THREADS) {
            const uint32_t d = desc[td.x + i];
            const uint64_t spare = dfine ? (uint64_t)dfine[td.x + i] : 0ULL;          // (both loads in flight together)
            if (d == 0xFFFFFFFFu) continue;
            const uint32_t part = d >> (DESC_START_BITS + DESC_NBK_BITS), nbk = ((d >> DESC_START_BITS) & ((1u << DESC_NBK_BITS) - 1)) + 1u; const int start = (int)(d & (uint32_t)(SCAN_TILE - 1));
            const unsigned long long slot = atomicAdd(&s_cur[part], 1ULL);
            atomicAdd(&s_km[part], nbk);
            store_record<RW>(s_be, start, nbk, P.k, P.arena, slot, spare);
        }
    }
    __syncthreads();
    for (uint32_t p = threadIdx.x; p < P.n_parts; p += SCAN_THREADS) if (s_km[p]) atomicAdd(&P.cnt_kmers[p], (unsigned long long)s_km[p]);
}

// per-partition prefix over workgroups: base[w][p] = records of workgroups < w (partition-relative); totals per partition
__global__ void k_wg_prefix(const uint32_t* __restrict__ wg_cnt, uint32_t n_wg, uint32_t n_parts,
                            unsigned long long* __restrict__ wg_base, unsigned long long* __restrict__ tot_rec)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_parts) return;
    unsigned long long run = 0;
    for (uint32_t w = 0; w < n_wg; w++) {
        const uint32_t v = wg_cnt[(uint64_t)w * n_parts + p];
        wg_base[(uint64_t)w * n_parts + p] = run;
        run += v;
    }
    tot_rec[p] = run;
}
// base[w][p] += rec_off[p]: absolute arena slots
__global__ void k_wg_base_abs(unsigned long long* __restrict__ wg_base, const unsigned long long* __restrict__ rec_off, uint32_t n_parts, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) wg_base[i] += rec_off[i % n_parts];
}
// k-mers per partition = sum of nbK over its records (the count pass only counts records, 4 B of LDS per partition)
template <int RW>
__global__ void k_partition_kmers(const uint64_t* __restrict__ arena, const unsigned long long* __restrict__ rec_off, const unsigned long long* __restrict__ tot_rec,
                                  unsigned long long* __restrict__ out)
{
    __shared__ unsigned long long s[4];
    const uint32_t p = blockIdx.x;
    const uint64_t r0 = rec_off[p], n = tot_rec[p];
    unsigned long long v = 0;
    for (uint64_t r = threadIdx.x; r < n; r += blockDim.x) v += arena[(r0 + r) * RW] >> 56;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d, 64);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) out[p] = s[0] + s[1] + s[2] + s[3];
}

static int launch_scan(gkc_ctx* c, const ScanParams& P, bool emit, bool ldspart, unsigned grid_n, size_t dyn_lds)
{
    dim3 grid(grid_n), block(SCAN_THREADS);
#define GKC_LAUNCH(E, R, L) hipLaunchKernelGGL((k_scan_tile<E, R, L>), grid, block, dyn_lds, c->stream, P)
    if (c->record_bytes == 16) {
        if (ldspart) { if (emit) GKC_LAUNCH(true, 2, true); else if (P.fine_shift) hipLaunchKernelGGL((k_scan_tile<false, 2, true, true>), grid, block, dyn_lds, c->stream, P); else GKC_LAUNCH(false, 2, true); }
        else         { if (emit) GKC_LAUNCH(true, 2, false); else GKC_LAUNCH(false, 2, false); }
    } else {
        if (ldspart) { if (emit) GKC_LAUNCH(true, 4, true); else if (P.fine_shift) hipLaunchKernelGGL((k_scan_tile<false, 4, true, true>), grid, block, dyn_lds, c->stream, P); else GKC_LAUNCH(false, 4, true); }
        else         { if (emit) GKC_LAUNCH(true, 4, false); else GKC_LAUNCH(false, 4, false); }
    }
#undef GKC_LAUNCH
    GKC_HIP(c, hipGetLastError());
    return GKC_OK;
}

// ------------------------------------------------------------------------------------------------ two-level Stage A: refine
// partition of a record = repart[minimizer of its first k-mer] (every k-mer of a super-k-mer has the same minimizer): the window minimum
// of step 2 of k_scan_tile recomputed for ONE window, from the record's own bit string
template <int RW>
__device__ __forceinline__ uint32_t record_partition(const uint64_t (&R)[RW], const ScanParams& P)
{
    // the first k-mer of the record and its reverse complement, once; m-mer j (j = 0 .. k-m) is then a window of each: forward at nucleotide j from the
    // left, reverse complement at nucleotide j from the right (a revcomp per m-mer made this kernel VALU-bound: 22 of them per record)
    uint32_t best = P.default_key;
    const uint32_t k = P.k, m = P.m;
    const uint64_t S0 = (R[0] << 8) | (R[1] >> 56);
    if (RW == 2 || k <= 32) {
        const uint64_t F = k == 32 ? S0 : (S0 >> (64 - 2 * k));
        const uint64_t RC = revcomp64(F, k);
        for (uint32_t j = 0; j < P.nb_mm; j++) {
            const uint32_t fw = (uint32_t)(F >> (2 * (k - m - j))) & P.mmask;
            uint32_t key;
            if (P.freq_mode) key = P.mkey_lut[fw];
            else {
                const uint32_t rc = (uint32_t)(RC >> (2 * j)) & P.mmask;
                const uint32_t cn = fw < rc ? fw : rc;
                uint32_t a = ~(cn | (cn >> 2));
                a = (a >> 1) & a & P.mask_ma1;
                key = a ? P.mmask : cn;
            }
            best = key < best ? key : best;
        }
    } else {
        const uint64_t S1 = (R[1] << 8) | (R[RW > 2 ? 2 : 1] >> 56);
        const u128 F = ((((u128)S0) << 64) | S1) >> (128 - 2 * k);
        const u128 RC = revcomp128(F, k);
        for (uint32_t j = 0; j < P.nb_mm; j++) {
            const uint32_t fw = (uint32_t)(F >> (2 * (k - m - j))) & P.mmask;
            uint32_t key;
            if (P.freq_mode) key = P.mkey_lut[fw];
            else {
                const uint32_t rc = (uint32_t)(RC >> (2 * j)) & P.mmask;
                const uint32_t cn = fw < rc ? fw : rc;
                uint32_t a = ~(cn | (cn >> 2));
                a = (a >> 1) & a & P.mask_ma1;
                key = a ? P.mmask : cn;
            }
            best = key < best ? key : best;
        }
    }
    const uint32_t value = P.freq_mode ? P.key2val[best] : best;
    return P.repart[value];
}
constexpr int REFINE_THREADS = 256, REFINE_MAX_FINE = 64;
// lanes of the wave that hold the same value b (< 2^bits) as this lane, among the lanes where `active` holds (all lanes of the wave must call this)
__device__ __forceinline__ unsigned long long wave_peers(uint32_t b, uint32_t bits, bool active)
{
    unsigned long long peers = __ballot(active);
    for (uint32_t i = 0; i < bits; i++) { const unsigned long long bal = __ballot((b >> i) & 1u); peers &= ((b >> i) & 1u) ? bal : ~bal; }
    return peers;
}
// one workgroup per coarse group: records -> fine partition; counts per fine partition + the fine id of every record
template <int RW>
__global__ __launch_bounds__(REFINE_THREADS) void k_refine_count(ScanParams P, const uint64_t* __restrict__ arena, const unsigned long long* __restrict__ coarse_off,
                                                                  uint32_t shift, uint8_t* __restrict__ fine_id, unsigned long long* __restrict__ cnt_rec,
                                                                  unsigned long long* __restrict__ cnt_km, uint32_t* __restrict__ bad)
{
    __shared__ uint32_t s_rec[REFINE_MAX_FINE]; __shared__ unsigned long long s_km[REFINE_MAX_FINE];
    const uint32_t g = blockIdx.x, nf = 1u << shift;
    if (threadIdx.x < nf) { s_rec[threadIdx.x] = 0; s_km[threadIdx.x] = 0; }
    __syncthreads();
    const unsigned long long r0 = coarse_off[g], r1 = coarse_off[g + 1];
    const int lane = threadIdx.x & 63;
    ulonglong2 nx[RW / 2];                                              // the next record of the thread is in flight while this one is worked on
    if (r0 + threadIdx.x < r1) { const ulonglong2* src = reinterpret_cast<const ulonglong2*>(arena + (r0 + threadIdx.x) * RW);
#pragma unroll
        for (int i = 0; i < RW / 2; i++) nx[i] = src[i]; }
    for (unsigned long long rb = r0; rb < r1; rb += REFINE_THREADS) {     // (uniform trip count: the ballots see all lanes)
        const unsigned long long r = rb + threadIdx.x;
        const bool in = r < r1;
        uint32_t b = 0, nk = 0;
        uint64_t R[RW];
#pragma unroll
        for (int i = 0; i < RW; i += 2) { R[i] = nx[i / 2].x; R[i + 1] = nx[i / 2].y; }
        if (r + REFINE_THREADS < r1) { const ulonglong2* src = reinterpret_cast<const ulonglong2*>(arena + (r + REFINE_THREADS) * RW);
#pragma unroll
            for (int i = 0; i < RW / 2; i++) nx[i] = src[i]; }
        if (in) {
            if (fine_id == nullptr) b = (uint32_t)R[RW - 1] & 15u;      // the scan left the partition's low bits in the record's spare bits
            else { b = record_partition<RW>(R, P) - (g << shift); }
            nk = (uint32_t)(R[0] >> 56);
            if (b >= nf) { *bad = 1u; b = 0; nk = 0; }                   // cannot happen: the coarse group of a record is its partition >> shift
            if (fine_id) fine_id[r] = (uint8_t)b;
        }
        // one LDS add per fine partition and wave, not per record: all lanes hitting the same 8 counters serialise
        const unsigned long long peers = wave_peers(b, shift, in);
        uint32_t km = 0;                                                // sum of nk (< 64) over the peers, bit by bit
#pragma unroll
        for (int i = 0; i < 6; i++) km += (uint32_t)__popcll(__ballot((nk >> i) & 1u) & peers) << i;
        if (in && (peers & ((1ULL << lane) - 1ULL)) == 0) { atomicAdd(&s_rec[b], (uint32_t)__popcll(peers)); atomicAdd(&s_km[b], (unsigned long long)km); }
    }
    __syncthreads();
    if (threadIdx.x < nf && ((g << shift) + threadIdx.x) < P.n_parts) { cnt_rec[(g << shift) + threadIdx.x] = s_rec[threadIdx.x]; cnt_km[(g << shift) + threadIdx.x] = s_km[threadIdx.x]; }
}
template <int RW>
__global__ __launch_bounds__(REFINE_THREADS) void k_refine_scatter(const uint64_t* __restrict__ arena, const unsigned long long* __restrict__ coarse_off, uint32_t shift,
                                                                    const uint8_t* __restrict__ fine_id, const unsigned long long* __restrict__ fine_off,
                                                                    uint32_t n_parts, uint64_t* __restrict__ out)
{
    __shared__ unsigned long long s_cur[REFINE_MAX_FINE];
    const uint32_t g = blockIdx.x, nf = 1u << shift;
    if (threadIdx.x < nf) s_cur[threadIdx.x] = ((g << shift) + threadIdx.x) < n_parts ? fine_off[(g << shift) + threadIdx.x] : 0;
    __syncthreads();
    const unsigned long long r0 = coarse_off[g], r1 = coarse_off[g + 1];
    const int lane = threadIdx.x & 63;
    ulonglong2 nx[RW / 2]; uint32_t nb_ = 0;                            // the next record (and its fine id) of the thread is in flight while this one is placed
    if (r0 + threadIdx.x < r1) { const ulonglong2* src0 = reinterpret_cast<const ulonglong2*>(arena + (r0 + threadIdx.x) * RW); if (fine_id) nb_ = fine_id[r0 + threadIdx.x];
#pragma unroll
        for (int i = 0; i < RW / 2; i++) nx[i] = src0[i]; }
    for (unsigned long long rb = r0; rb < r1; rb += REFINE_THREADS) {     // (uniform trip count: the ballots see all lanes)
        const unsigned long long r = rb + threadIdx.x;
        const bool in = r < r1;
        ulonglong2 cur[RW / 2];
#pragma unroll
        for (int i = 0; i < RW / 2; i++) cur[i] = nx[i];
        if (fine_id == nullptr) { nb_ = (uint32_t)cur[RW / 2 - 1].y & 15u; cur[RW / 2 - 1].y &= ~15ull; }      // the spare bits go back to zero (Stage B keeps weights there)
        const uint32_t b = in ? nb_ : 0u;
        if (r + REFINE_THREADS < r1) { const ulonglong2* src1 = reinterpret_cast<const ulonglong2*>(arena + (r + REFINE_THREADS) * RW); if (fine_id) nb_ = fine_id[r + REFINE_THREADS];
#pragma unroll
            for (int i = 0; i < RW / 2; i++) nx[i] = src1[i]; }
        // one returning LDS add per fine partition and wave (the lowest peer reserves for all, the others take their rank)
        const unsigned long long peers = wave_peers(b, shift, in);
        const int leader = peers ? __ffsll((long long)peers) - 1 : 0;
        unsigned long long base = 0;
        if (in && lane == leader) base = atomicAdd(&s_cur[b], (unsigned long long)__popcll(peers));
        base = __shfl(base, leader, 64);
        if (in) {
            const unsigned long long slot = base + (unsigned long long)__popcll(peers & ((1ULL << lane) - 1ULL));
            ulonglong2* dst = reinterpret_cast<ulonglong2*>(out + slot * RW);
#pragma unroll
            for (int i = 0; i < RW / 2; i++) dst[i] = cur[i];
        }
    }
}

constexpr uint32_t SCAN_LDS_PARTS_MAX = 16384;     // 64 KB of LDS counters at most
constexpr size_t SCAN_STATIC_LDS = 64 * 1024;      // static LDS of k_scan_tile (upper bound used for residency)

int gkc_scan_push(gkc_ctx* c, const char* d_bases, const uint64_t* d_offsets, uint64_t n_reads, uint64_t n_bases)
{
    // above SCAN_COARSE_MAX partitions the scan buckets into groups of 2^coarse_shift partitions (Pn groups) and the groups are split at the end
    const uint32_t Pfine = c->nb_partitions, cshift = c->coarse_shift;
    const uint32_t Pn = cshift ? ((Pfine - 1) >> cshift) + 1 : Pfine;
    if (cshift && (1u << cshift) > (uint32_t)REFINE_MAX_FINE) GKC_FAIL(c, GKC_ERR_ARG, "too many partitions for the two-level scan");
    if (n_bases >= (1ULL << 40)) GKC_FAIL(c, GKC_ERR_ARG, "a single push is limited to 2^40 bases");
    const uint64_t n_tiles = (n_bases + SCAN_TILE - 1) / SCAN_TILE;
    if (n_tiles >= (1ULL << 31)) GKC_FAIL(c, GKC_ERR_ARG, "too many tiles in one push; split the batch");

    Segment seg; seg.rec_off.assign(Pn + 1, 0); seg.nkmers.assign(Pn, 0); seg.owned = true;
    if (n_tiles == 0) { seg.rec_off.assign(Pfine + 1, 0); seg.nkmers.assign(Pfine, 0); c->segments.push_back(seg); c->stats_now().nb_sequences += n_reads; return GKC_OK; }

    // read-start bitmask (+ slack so every tile can read its halo words)
    const size_t rs_words = (size_t)(n_tiles * SCAN_TILE / 32 + 64);
    GKC_TRY(c->ensure(c->d_rsbits, rs_words * 4));
    GKC_HIP(c, hipMemsetAsync(c->d_rsbits.p, 0, rs_words * 4, c->stream));
    {
        uint64_t n_entries = n_reads + 1;
        dim3 g((unsigned)std::min<uint64_t>((n_entries + 255) / 256, 4096)), b(256);
        // the validity flag lives in the last word of the (zeroed) mask allocation's slack and is read back with the counters below
        hipLaunchKernelGGL(k_mark_read_starts, g, b, 0, c->stream, d_offsets, n_entries, n_bases, (uint32_t*)c->d_rsbits.p, (uint32_t*)c->d_rsbits.p + rs_words - 1,
                           (unsigned long long*)((uint32_t*)c->d_rsbits.p + rs_words - 8));      // (... and the read-length statistics in the six words before it)
        GKC_HIP(c, hipGetLastError());
    }
    // geometry: persistent workgroups when the partition cursors fit in LDS
    const bool ldspart = Pn <= SCAN_LDS_PARTS_MAX && !gkc_tun().scan_global_atomics;
    const size_t dyn_lds = ldspart ? (size_t)Pn * 4 : 0;
    unsigned grid_n;
    if (ldspart) {
        int cus = 256; hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
        unsigned per_cu = (unsigned)std::min<size_t>(8, std::max<size_t>(1, (160 * 1024) / (SCAN_STATIC_LDS + dyn_lds)));
        // multi-GPU: the exchange of the previous push runs on the communicator's stream while this scan runs; the scan's workgroups are persistent (they
        // hold their CU until the last tile), so 16 CUs are left to the RCCL send / receive kernels instead of making them wait for the scan to drain
        if (c->comm_world > 1 && cus > 64) cus -= 16;
        grid_n = (unsigned)std::min<uint64_t>(n_tiles, (uint64_t)cus * per_cu);
    } else grid_n = (unsigned)std::min<uint64_t>(n_tiles, 1u << 20);

    // counters: [0..P) records, [P..2P) k-mers, [2P..3P) cursors / rec_off, [3P..3P+4) stats
    const size_t n_cnt = (size_t)3 * Pn + 4;
    GKC_TRY(c->ensure(c->d_scan_counters, n_cnt * 8));
    GKC_HIP(c, hipMemsetAsync(c->d_scan_counters.p, 0, n_cnt * 8, c->stream));
    unsigned long long* cnt = (unsigned long long*)c->d_scan_counters.p;
    if (ldspart) GKC_TRY(c->ensure(c->d_scan_matrix, (size_t)grid_n * Pn * 12));      // [G][P] u64 bases, then [G][P] u32 counts

    ScanParams P{};
    P.bases = (const uint8_t*)d_bases; P.n_bases = n_bases; P.rsbits = (const uint32_t*)c->d_rsbits.p;
    P.k = c->k; P.m = c->m; P.nb_mm = c->k - c->m + 1; P.maxs = c->maxs; P.maxs_magic = (uint32_t)(((1ULL << 32) + c->maxs - 1) / c->maxs);
    P.mmask = (uint32_t)((1ULL << (2 * c->m)) - 1);
    P.mask_ma1 = (uint32_t)(0x5555555555555555ULL & ((1ULL << ((c->m - 2) * 2)) - 1));
    P.freq_mode = c->minimizer_type == GKC_MINIMIZER_FREQ;
    P.mkey_lut = (const uint32_t*)c->d_mkey_lut.p; P.key2val = (const uint32_t*)c->d_key2val.p; P.default_key = c->default_key;
    // two-level scan with <= 16 partitions per group: the scan looks the FINE partition up and leaves its low bits in the record (4 spare bits below the nucleotides),
    // so the refine level does not recompute minimizers; more partitions per group: the group table, and k_refine_count recomputes
    const bool stash = cshift > 0 && cshift <= 4 && !gkc_tun().refine_recompute;
    P.fine_shift = stash ? cshift : 0u; P.desc_fine = nullptr;
    P.repart = (cshift && !stash) ? (const uint16_t*)c->d_repart_coarse.p : (const uint16_t*)c->d_repart.p; P.nb_passes = c->nb_passes; P.pass = c->pass;
    P.cnt_rec = cnt; P.cnt_kmers = cnt + Pn; P.cursor = cnt + 2 * (size_t)Pn; P.gstats = cnt + 3 * (size_t)Pn;
    P.arena = nullptr;
    P.n_tiles = n_tiles; P.n_parts = Pn;
    P.wg_base = ldspart ? (unsigned long long*)c->d_scan_matrix.p : nullptr;
    P.wg_cnt = ldspart ? (uint32_t*)((unsigned long long*)c->d_scan_matrix.p + (size_t)grid_n * Pn) : nullptr;
    P.rec_off = cnt + 2 * (size_t)Pn;
    // descriptor stream (count pass -> emit pass): DESC_PER_TILE u32 per tile on average, per-workgroup regions
    const bool use_desc = ldspart && Pn <= DESC_PARTS_MAX && !gkc_tun().scan_no_desc;
    if (use_desc) {
        const uint64_t tiles_per_wg = (n_tiles + grid_n - 1) / grid_n;
        const uint64_t cap = tiles_per_wg * (SCAN_TILE * 5 / 32);
        if (cap < (1ULL << 32)) {
            GKC_TRY(c->ensure(c->d_desc, (size_t)grid_n * cap * (stash ? 5 : 4)));
            if (stash) P.desc_fine = (uint8_t*)c->d_desc.p + (size_t)grid_n * cap * 4;
            GKC_TRY(c->ensure(c->d_desc_tile, (size_t)n_tiles * 8));
            P.desc = (uint32_t*)c->d_desc.p; P.desc_cap_wg = (uint32_t)cap; P.desc_tile = (uint2*)c->d_desc_tile.p;
            P.desc_overflow = (uint32_t*)(cnt + 3 * (size_t)Pn + 3);
        }
    }

    {   ScopedTimer tm(c, "scan_count");
        GKC_TRY(launch_scan(c, P, false, ldspart, grid_n, dyn_lds));
        if (ldspart) {
            hipLaunchKernelGGL(k_wg_prefix, dim3((Pn + 255) / 256), dim3(256), 0, c->stream, (const uint32_t*)P.wg_cnt, grid_n, Pn,
                               (unsigned long long*)P.wg_base, cnt);
            GKC_HIP(c, hipGetLastError());
        }
    }
    std::vector<unsigned long long> h(n_cnt);
    uint32_t bad_offsets = 0;
    GKC_HIP(c, hipMemcpyAsync(h.data(), cnt, n_cnt * 8, hipMemcpyDeviceToHost, c->stream));
    GKC_HIP(c, hipMemcpyAsync(&bad_offsets, (uint32_t*)c->d_rsbits.p + rs_words - 1, 4, hipMemcpyDeviceToHost, c->stream));
    unsigned long long len_stats[3] = {0, 0, 0};
    GKC_HIP(c, hipMemcpyAsync(len_stats, (uint32_t*)c->d_rsbits.p + rs_words - 8, sizeof(len_stats), hipMemcpyDeviceToHost, c->stream));
    GKC_HIP(c, hipStreamSynchronize(c->stream));
    if (bad_offsets) GKC_FAIL(c, GKC_ERR_ARG, "read offsets are not a CSR table of the bases (need offsets[0] == 0, non-decreasing, offsets[n_reads] == n_bases = %llu)", (unsigned long long)n_bases);
    uint64_t total = 0;
    for (uint32_t p = 0; p < Pn; p++) { seg.rec_off[p] = total; total += h[p]; seg.nkmers[p] = h[Pn + p]; }
    seg.rec_off[Pn] = total;
    {   gkc_stats& S = c->stats_now();
        S.kmers_nb_valid += h[3 * (size_t)Pn + 0]; S.kmers_nb_invalid += h[3 * (size_t)Pn + 1];
        S.nb_sequences += n_reads; S.nb_bases += n_bases;
        if (len_stats[1]) {                                             // BankStats::update (BankKmers.hpp:176-186)
            const uint64_t mn = ~len_stats[0];
            S.seq_len_min = S.seq_len_min == 0 || mn < S.seq_len_min ? mn : S.seq_len_min;      // (0 = no read yet; an empty read does not count, like an empty line in the reference's reader)
            S.seq_len_max = std::max<uint64_t>(S.seq_len_max, len_stats[1]); S.seq_len_sq_sum += len_stats[2];
        }
        S.nb_superkmers += total; S.superkmer_bytes += total * c->record_bytes;
    }

    void* arena = nullptr;
    if (total) {
        arena = c->dalloc((size_t)total * c->record_bytes);
        if (!arena) return GKC_ERR_NOMEM;
        c->owned_arenas.push_back(arena);
        // cursors (global-atomic mode) / partition offsets (LDS mode) start at the exact partition offsets
        GKC_HIP(c, hipMemcpyAsync(cnt + 2 * (size_t)Pn, seg.rec_off.data(), (size_t)Pn * 8, hipMemcpyHostToDevice, c->stream));
        GKC_HIP(c, hipMemsetAsync(cnt + 3 * (size_t)Pn, 0, 4 * 8, c->stream));
        P.arena = (uint64_t*)arena;
        {   ScopedTimer tm(c, "scan_emit");
            if (ldspart) {
                const uint64_t nn = (uint64_t)grid_n * Pn;
                hipLaunchKernelGGL(k_wg_base_abs, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, c->stream, (unsigned long long*)P.wg_base,
                                   (const unsigned long long*)(cnt + 2 * (size_t)Pn), Pn, nn);
            }
            const bool light = P.desc && h[3 * (size_t)Pn + 3] == 0;
            if (light) {                                          // descriptors complete: light emit kernel (also sums k-mers per partition)
                if (c->record_bytes == 16) hipLaunchKernelGGL((k_emit_desc<2>), dim3(grid_n), dim3(SCAN_THREADS), (size_t)Pn * 12, c->stream, P);
                else                       hipLaunchKernelGGL((k_emit_desc<4>), dim3(grid_n), dim3(SCAN_THREADS), (size_t)Pn * 12, c->stream, P);
                GKC_HIP(c, hipGetLastError());
            } else
            GKC_TRY(launch_scan(c, P, true, ldspart, grid_n, dyn_lds));
            if (ldspart) {          // k-mers per partition (Stage B sizes its key buffers from them)
                if (light) { /* accumulated by k_emit_desc into cnt[P..2P) */ }
                else if (c->record_bytes == 16) hipLaunchKernelGGL((k_partition_kmers<2>), dim3(Pn), dim3(256), 0, c->stream, (const uint64_t*)arena,
                                                              (const unsigned long long*)(cnt + 2 * (size_t)Pn), (const unsigned long long*)cnt, cnt + Pn);
                else                       hipLaunchKernelGGL((k_partition_kmers<4>), dim3(Pn), dim3(256), 0, c->stream, (const uint64_t*)arena,
                                                              (const unsigned long long*)(cnt + 2 * (size_t)Pn), (const unsigned long long*)cnt, cnt + Pn);
                GKC_HIP(c, hipGetLastError());
                std::vector<unsigned long long> hk(Pn);
                GKC_HIP(c, hipMemcpyAsync(hk.data(), cnt + Pn, (size_t)Pn * 8, hipMemcpyDeviceToHost, c->stream));
                GKC_HIP(c, hipStreamSynchronize(c->stream));
                for (uint32_t p = 0; p < Pn; p++) seg.nkmers[p] = hk[p];
            }
        }
    }
    c->d_desc.release(); c->d_desc_tile.release();             // back to the pool: Stage B / the refine level may reuse the space
    if (cshift) {
        // second level: every group -> its 2^cshift partitions (minimizer of the record recomputed from its first k-mer)
        ScopedTimer tm(c, "scan_refine");
        Segment fine; fine.rec_off.assign(Pfine + 1, 0); fine.nkmers.assign(Pfine, 0); fine.owned = true;
        if (total) {
            DevBuf d_fid, d_cnt, d_coff;
            struct Guard { DevBuf *a, *b, *cc; ~Guard() { a->release(); b->release(); cc->release(); } } guard{&d_fid, &d_cnt, &d_coff};
            if (!stash) GKC_TRY(c->ensure(d_fid, (size_t)total));
            GKC_TRY(c->ensure(d_cnt, ((size_t)2 * Pfine + 2) * 8)); GKC_TRY(c->ensure(d_coff, ((size_t)Pn + 1) * 8));
            GKC_HIP(c, hipMemsetAsync(d_cnt.p, 0, ((size_t)2 * Pfine + 2) * 8, c->stream));
            GKC_HIP(c, hipMemcpyAsync(d_coff.p, seg.rec_off.data(), ((size_t)Pn + 1) * 8, hipMemcpyHostToDevice, c->stream));
            ScanParams Q = P; Q.repart = (const uint16_t*)c->d_repart.p; Q.n_parts = Pfine;
            unsigned long long* fc = (unsigned long long*)d_cnt.p;
            if (c->record_bytes == 16) hipLaunchKernelGGL((k_refine_count<2>), dim3(Pn), dim3(REFINE_THREADS), 0, c->stream, Q, (const uint64_t*)arena, (const unsigned long long*)d_coff.p,
                                                          cshift, stash ? (uint8_t*)nullptr : (uint8_t*)d_fid.p, fc, fc + Pfine, (uint32_t*)(fc + 2 * (size_t)Pfine));
            else                       hipLaunchKernelGGL((k_refine_count<4>), dim3(Pn), dim3(REFINE_THREADS), 0, c->stream, Q, (const uint64_t*)arena, (const unsigned long long*)d_coff.p,
                                                          cshift, stash ? (uint8_t*)nullptr : (uint8_t*)d_fid.p, fc, fc + Pfine, (uint32_t*)(fc + 2 * (size_t)Pfine));
            GKC_HIP(c, hipGetLastError());
            std::vector<unsigned long long> hc((size_t)2 * Pfine + 1);
            GKC_HIP(c, hipMemcpyAsync(hc.data(), fc, hc.size() * 8, hipMemcpyDeviceToHost, c->stream));
            GKC_HIP(c, hipStreamSynchronize(c->stream));
            if (hc[2 * (size_t)Pfine]) GKC_FAIL(c, GKC_ERR_HIP, "internal error: a record left its partition group in the refine level");
            uint64_t run = 0;
            for (uint32_t p = 0; p < Pfine; p++) { fine.rec_off[p] = run; run += hc[p]; fine.nkmers[p] = hc[Pfine + p]; }
            fine.rec_off[Pfine] = run;
            if (run != total) GKC_FAIL(c, GKC_ERR_HIP, "internal error: refine level lost records");
            void* arena2 = c->dalloc((size_t)total * c->record_bytes);
            if (!arena2) return GKC_ERR_NOMEM;
            GKC_HIP(c, hipMemcpyAsync(fc, fine.rec_off.data(), (size_t)Pfine * 8, hipMemcpyHostToDevice, c->stream));
            if (c->record_bytes == 16) hipLaunchKernelGGL((k_refine_scatter<2>), dim3(Pn), dim3(REFINE_THREADS), 0, c->stream, (const uint64_t*)arena, (const unsigned long long*)d_coff.p, cshift,
                                                          stash ? (const uint8_t*)nullptr : (const uint8_t*)d_fid.p, (const unsigned long long*)fc, Pfine, (uint64_t*)arena2);
            else                       hipLaunchKernelGGL((k_refine_scatter<4>), dim3(Pn), dim3(REFINE_THREADS), 0, c->stream, (const uint64_t*)arena, (const unsigned long long*)d_coff.p, cshift,
                                                          stash ? (const uint8_t*)nullptr : (const uint8_t*)d_fid.p, (const unsigned long long*)fc, Pfine, (uint64_t*)arena2);
            hipError_t e = hipGetLastError();
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            if (e != hipSuccess) { c->dfree(arena2); GKC_FAIL(c, GKC_ERR_HIP, "refine scatter failed: %s", hipGetErrorString(e)); }
            for (auto& a : c->owned_arenas) if (a == arena) a = arena2;
            c->dfree(arena); arena = arena2;
        }
        seg = std::move(fine);
    }
    seg.d_records = arena;
    c->segments.push_back(std::move(seg));
    return GKC_OK;
}

// ------------------------------------------------------------------------------------------------
// RepartitorAlgorithm support (kmer/impl/RepartitionAlgorithm.cpp:395-475 SampleRepart): per-minimizer super-k-mer and
// k-mer counts of a sample of reads, computed by the same scan kernel with "partition = minimizer value".
// ------------------------------------------------------------------------------------------------
int gkc_scan_sample(gkc_ctx* c, const char* d_bases, const uint64_t* d_offsets, uint64_t n_reads, uint64_t n_bases,
                    uint64_t* h_superkmers, uint64_t* h_kmers)
{
    const uint64_t nm = 1ULL << (2 * c->m);
    const uint64_t n_tiles = (n_bases + SCAN_TILE - 1) / SCAN_TILE;
    if (n_tiles == 0) return GKC_OK;
    if (n_tiles >= (1ULL << 31)) GKC_FAIL(c, GKC_ERR_ARG, "sample too large");
    const size_t rs_words = (size_t)(n_tiles * SCAN_TILE / 32 + 64);
    GKC_TRY(c->ensure(c->d_rsbits, rs_words * 4));
    GKC_HIP(c, hipMemsetAsync(c->d_rsbits.p, 0, rs_words * 4, c->stream));
    hipLaunchKernelGGL(k_mark_read_starts, dim3((unsigned)((n_reads + 1 + 255) / 256)), dim3(256), 0, c->stream, d_offsets, n_reads + 1, n_bases, (uint32_t*)c->d_rsbits.p, (uint32_t*)nullptr, (unsigned long long*)nullptr);
    DevBuf cnt; GKC_TRY(c->ensure(cnt, (size_t)(2 * nm + 4) * 8));
    hipError_t e = hipMemsetAsync(cnt.p, 0, (size_t)(2 * nm + 4) * 8, c->stream);
    ScanParams P{};
    P.bases = (const uint8_t*)d_bases; P.n_bases = n_bases; P.rsbits = (const uint32_t*)c->d_rsbits.p;
    P.k = c->k; P.m = c->m; P.nb_mm = c->k - c->m + 1; P.maxs = c->maxs; P.maxs_magic = (uint32_t)(((1ULL << 32) + c->maxs - 1) / c->maxs);
    P.mmask = (uint32_t)(nm - 1);
    P.mask_ma1 = (uint32_t)(0x5555555555555555ULL & ((1ULL << ((c->m - 2) * 2)) - 1));
    P.freq_mode = c->minimizer_type == GKC_MINIMIZER_FREQ;
    P.mkey_lut = (const uint32_t*)c->d_mkey_lut.p; P.key2val = (const uint32_t*)c->d_key2val.p; P.default_key = c->default_key;
    P.repart = (const uint16_t*)c->d_repart.p; P.nb_passes = 1; P.pass = 0;
    P.cnt_rec = (unsigned long long*)cnt.p; P.cnt_kmers = P.cnt_rec + nm; P.cursor = nullptr; P.gstats = P.cnt_rec + 2 * nm;
    P.identity_part = 1; P.n_tiles = n_tiles; P.n_parts = 0;
    int rc = (e == hipSuccess) ? launch_scan(c, P, false, false, (unsigned)std::min<uint64_t>(n_tiles, 1u << 20), 0) : GKC_ERR_HIP;
    if (rc == GKC_OK) {
        std::vector<uint64_t> a(nm), b(nm);
        e = hipMemcpyAsync(a.data(), cnt.p, nm * 8, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(b.data(), (uint64_t*)cnt.p + nm, nm * 8, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e == hipSuccess) for (uint64_t i = 0; i < nm; i++) { h_superkmers[i] += a[i]; h_kmers[i] += b[i]; }
    }
    cnt.release();
    if (rc != GKC_OK) return rc;
    if (e != hipSuccess) GKC_FAIL(c, GKC_ERR_HIP, "minimizer sampling failed: %s", hipGetErrorString(e));
    return GKC_OK;
}

// ------------------------------------------------------------------------------------------------
// EXACT Repartitor sample (RepartitionAlgorithm.cpp:135-215 SampleRepart over Sequence2SuperKmer.hpp:81-159): one thread per read walks the read the
// way the reference does — k-mer by k-mer, the super-k-mer closed when the minimizer VALUE changes, at an invalid k-mer, at the length cap or at the
// end of the read — and records per minimizer the number of super-k-mers, of k-mers and of kx-mers (runs of successive k-mers on one strand, cut after
// _kx = 4 extensions: :186-203), which is what Repartitor::computeDistrib balances on (PartiInfo.cpp:48-106). No tiles here: the tile scan of Stage A
// may split a super-k-mer at a tile border, harmless for counting, not for these statistics. The sample is a few 10^4..10^6 reads: speed is irrelevant.
// mode 0: per read, the number of super-k-mers of the pass (the reference stops its sample when their running total exceeds a threshold, :205-212);
// mode 1: the statistics of reads [0, n_reads).
// ------------------------------------------------------------------------------------------------
__global__ void k_sample_exact(ScanParams P, const uint64_t* __restrict__ offsets, uint64_t n_reads, int mode, uint32_t* __restrict__ per_read,
                               unsigned long long* __restrict__ nsk, unsigned long long* __restrict__ nk, unsigned long long* __restrict__ nkx)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const uint64_t b = offsets[r], e = offsets[r + 1];
    const uint32_t k = P.k, m = P.m;
    uint32_t n_here = 0;
    if (e - b >= k) {
        const u128 kmask = KeyT<2>::mask(k);
        const uint32_t DEFAULT = 0xFFFFFFFFu;                    // SuperKmer::DEFAULT_MINIMIZER (Model.hpp:1349): "no minimizer yet"
        u128 fw = 0, rv = 0; uint32_t good = 0;                  // rolling k-mer (Model.hpp:637-657); good = valid nucleotides in a row
        uint32_t sk_min = DEFAULT, sk_size = 0, kx_size = 0, n_kx = 0; bool prev_which = false;
        auto close_sk = [&]() {                                  // Sequence2SuperKmer::processSuperkmer -> SampleRepart::processSuperkmer
            if (sk_size > 0 && sk_min != DEFAULT && (sk_min % P.nb_passes) == P.pass) {
                n_here++;
                if (mode == 1) { atomicAdd(&nsk[sk_min], 1ULL); atomicAdd(&nk[sk_min], (unsigned long long)sk_size); atomicAdd(&nkx[sk_min], (unsigned long long)(n_kx + 1)); }
            }
            sk_size = 0; n_kx = 0; kx_size = 0;
        };
        for (uint64_t g = b; g < e; g++) {
            const uint32_t ch = P.bases[g];
            const bool ok = nt_valid(ch) != 0;
            const uint32_t code = nt_code(ch);
            fw = ((fw << 2) | (u128)code) & kmask;
            rv = (rv >> 2) | ((u128)(code ^ 2u) << (2 * (k - 1)));
            good = ok ? good + 1 : 0;
            if (g + 1 < b + k) continue;                         // the first k-mer ends at base b + k - 1
            if (good < k) { close_sk(); sk_min = DEFAULT; continue; }       // invalid k-mer: close, restart "from new" (Sequence2SuperKmer.hpp:94-106)
            // minimizer of this k-mer: minimum order key over its k-m+1 m-mers, the default minimizer taking part (Model.hpp:1254-1287)
            uint32_t best = P.default_key;
            const uint64_t k0 = g + 1 - k;                       // first base of the k-mer
            uint32_t mf = 0;
            for (uint32_t j = 0; j < k; j++) {
                mf = ((mf << 2) | nt_code(P.bases[k0 + j])) & P.mmask;
                if (j + 1 < m) continue;
                uint32_t key;
                if (P.freq_mode) key = P.mkey_lut[mf];
                else {
                    const uint32_t rc = (uint32_t)revcomp64(mf, m);
                    const uint32_t cn = mf < rc ? mf : rc;
                    uint32_t a = ~(cn | (cn >> 2));
                    a = (a >> 1) & a & P.mask_ma1;
                    key = a ? P.mmask : cn;
                }
                best = key < best ? key : best;
            }
            const uint32_t h = P.freq_mode ? P.key2val[best] : best;
            const bool which = fw < rv;                          // strand of the canonical k-mer (Model.hpp:294)
            if (sk_min == DEFAULT) sk_min = h;
            if (h != sk_min || sk_size >= P.maxs) close_sk();
            sk_min = h;
            if (sk_size > 0) {                                   // kx-mer bookkeeping of SampleRepart::processSuperkmer (:186-203), done on the fly
                if (which != prev_which || kx_size >= 4) { n_kx++; kx_size = 0; } else kx_size++;
            }
            prev_which = which;
            sk_size++;
        }
        close_sk();                                              // "output last superK"
    }
    if (mode == 0) per_read[r] = n_here;
}

// MmersFrequency (RepartitionAlgorithm.cpp:88-120): occurrences of every canonical m-mer at VALID m-mer positions
__global__ void k_count_mmers(const uint8_t* __restrict__ bases, const uint64_t* __restrict__ offsets, uint64_t n_reads, uint32_t m,
                              unsigned int* __restrict__ counts)
{
    const uint32_t mask = (uint32_t)((1ULL << (2 * m)) - 1);
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_reads; r += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t b = offsets[r], e = offsets[r + 1];
        uint32_t fw = 0, rv = 0, good = 0;
        for (uint64_t g = b; g < e; g++) {
            const uint32_t ch = bases[g];
            if (nt_valid(ch)) { const uint32_t code = nt_code(ch); fw = ((fw << 2) | code) & mask; rv = (rv >> 2) | ((code ^ 2u) << (2 * (m - 1))); good++; }
            else { good = 0; fw = 0; rv = 0; }
            if (good >= m) atomicAdd(&counts[fw < rv ? fw : rv], 1u);
        }
    }
}
int gkc_scan_count_mmers(gkc_ctx* c, uint32_t m, const char* d_bases, const uint64_t* d_offsets, uint64_t n_reads, uint32_t* h_counts)
{
    const uint64_t nm = 1ULL << (2 * m);
    DevBuf cnt; GKC_TRY(c->ensure(cnt, nm * 4));
    hipError_t e = hipMemcpyAsync(cnt.p, h_counts, nm * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_count_mmers, dim3((unsigned)std::min<uint64_t>((n_reads + 255) / 256 + 1, 8192)), dim3(256), 0, c->stream,
                           (const uint8_t*)d_bases, d_offsets, n_reads, m, (unsigned int*)cnt.p);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(h_counts, cnt.p, nm * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    cnt.release();
    if (e != hipSuccess) GKC_FAIL(c, GKC_ERR_HIP, "m-mer counting failed: %s", hipGetErrorString(e));
    return GKC_OK;
}

// ------------------------------------------------------------------------------------------------
// parity surface: partition -> reference wire format [u8 nbK][payload] (Model.hpp:1386-1471)
// (host-side re-encoding of the device records; not on the timed path)
// ------------------------------------------------------------------------------------------------
int gkc_export_superkmers(gkc_ctx* c, uint32_t part, uint8_t* out, uint64_t cap, uint64_t* nb, uint64_t* nsk, uint64_t* nk)
{
    const uint32_t k = c->k; const int RW = c->record_bytes / 8;
    uint64_t bytes = 0, n_sk = 0, n_k = 0;
    for (const Segment& s : c->segments) {
        const uint64_t n = s.rec_off[part + 1] - s.rec_off[part];
        if (!n) continue;
        std::vector<uint64_t> recs((size_t)n * RW);
        GKC_HIP(c, hipMemcpy(recs.data(), (const uint8_t*)s.d_records + s.rec_off[part] * c->record_bytes, (size_t)n * c->record_bytes, hipMemcpyDeviceToHost));
        for (uint64_t r = 0; r < n; r++) {
            const uint64_t* R = &recs[(size_t)r * RW];
            const uint32_t nbk = (uint32_t)(R[0] >> 56);
            const uint32_t nn = k + nbk - 1;
            auto nt = [&](uint32_t i) -> uint32_t {
                if (i < 28) return (uint32_t)(R[0] >> (54 - 2 * i)) & 3u;
                uint32_t j = i - 28; return (uint32_t)(R[1 + (j >> 5)] >> (62 - 2 * (j & 31))) & 3u;
            };
            const uint64_t need = 1 + (nn + 3) / 4;
            if (bytes + need > cap) GKC_FAIL(c, GKC_ERR_CAPACITY, "super-k-mer export buffer too small");
            uint8_t* o = out + bytes;
            o[0] = (uint8_t)nbk;
            // first k-mer little-endian by bytes: byte b holds nucleotides k-4b-4 .. k-4b-1, last one in the low bits
            uint32_t ob = 1; int rem = (int)k; int pos = (int)k;
            while (rem >= 4) { uint8_t v = 0; for (int q = 0; q < 4; q++) v |= (uint8_t)(nt((uint32_t)(pos - 1 - q)) << (2 * q)); o[ob++] = v; pos -= 4; rem -= 4; }
            uint8_t v = 0; for (int q = 0; q < rem; q++) v |= (uint8_t)(nt((uint32_t)(rem - 1 - q)) << (2 * q));
            int uid = rem; uint32_t idx = k;
            for (;;) {
                while (uid < 4 && idx < nn) { v |= (uint8_t)(nt(idx) << (2 * uid)); uid++; idx++; }
                if (uid > 0) o[ob++] = v;
                if (idx >= nn) break;
                v = 0; uid = 0;
            }
            bytes += ob; n_sk++; n_k += nbk;
        }
    }
    *nb = bytes; *nsk = n_sk; *nk = n_k;
    return GKC_OK;
}

// exact sample statistics of the first reads of a bank, stopped like SampleRepart (see k_sample_exact)
int gkc_scan_sample_exact(gkc_ctx* c, const char* bases, const uint64_t* offsets, uint64_t n_reads, uint64_t max_superkmers,
                          uint64_t* h_nsk, uint64_t* h_nk, uint64_t* h_nkx, uint64_t* reads_used)
{
    const uint64_t nm = 1ULL << (2 * c->m);
    DevBuf d_stats, d_per, db, dof;
    struct Guard { std::vector<DevBuf*> v; ~Guard() { for (DevBuf* b : v) b->release(); } } guard; guard.v = { &d_stats, &d_per, &db, &dof };
    GKC_TRY(c->ensure(d_stats, (size_t)3 * nm * 8));
    GKC_HIP(c, hipMemsetAsync(d_stats.p, 0, (size_t)3 * nm * 8, c->stream));
    ScanParams P{};
    P.k = c->k; P.m = c->m; P.maxs = c->maxs; P.mmask = (uint32_t)(nm - 1);
    P.mask_ma1 = (uint32_t)(0x5555555555555555ULL & ((1ULL << ((c->m - 2) * 2)) - 1));
    P.freq_mode = c->minimizer_type == GKC_MINIMIZER_FREQ;
    P.mkey_lut = (const uint32_t*)c->d_mkey_lut.p; P.key2val = (const uint32_t*)c->d_key2val.p; P.default_key = c->default_key;
    P.nb_passes = c->nb_passes; P.pass = 0;
    unsigned long long* st = (unsigned long long*)d_stats.p;
    const uint64_t CHUNK = 1u << 18;
    uint64_t seen = 0, used = 0; bool stop = false;
    for (uint64_t r0 = 0; r0 < n_reads && !stop; r0 += CHUNK) {
        const uint64_t n = std::min<uint64_t>(CHUNK, n_reads - r0), nb = offsets[r0 + n] - offsets[r0];
        GKC_TRY(c->ensure(db, (size_t)nb + 64)); GKC_TRY(c->ensure(dof, (size_t)(n + 1) * 8)); GKC_TRY(c->ensure(d_per, (size_t)n * 4));
        std::vector<uint64_t> rel(offsets + r0, offsets + r0 + n + 1);
        for (uint64_t& v : rel) v -= offsets[r0];
        if (nb) GKC_HIP(c, hipMemcpyAsync(db.p, bases + offsets[r0], (size_t)nb, hipMemcpyHostToDevice, c->stream));
        GKC_HIP(c, hipMemcpyAsync(dof.p, rel.data(), (size_t)(n + 1) * 8, hipMemcpyHostToDevice, c->stream));
        P.bases = (const uint8_t*)db.p; P.n_bases = nb;
        const unsigned grid = (unsigned)((n + 127) / 128);
        hipLaunchKernelGGL(k_sample_exact, dim3(grid), dim3(128), 0, c->stream, P, (const uint64_t*)dof.p, n, 0, (uint32_t*)d_per.p, st, st + nm, st + 2 * nm);
        GKC_HIP(c, hipGetLastError());
        std::vector<uint32_t> per(n);
        GKC_HIP(c, hipMemcpyAsync(per.data(), d_per.p, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
        GKC_HIP(c, hipStreamSynchronize(c->stream));
        // the reference checks its cancel flag between sequences: the read in which the running total first EXCEEDS the threshold is still counted
        uint64_t take = n;
        for (uint64_t i = 0; i < n; i++) { seen += per[i]; if (seen > max_superkmers) { take = i + 1; stop = true; break; } }
        hipLaunchKernelGGL(k_sample_exact, dim3((unsigned)((take + 127) / 128)), dim3(128), 0, c->stream, P, (const uint64_t*)dof.p, take, 1, (uint32_t*)nullptr, st, st + nm, st + 2 * nm);
        GKC_HIP(c, hipGetLastError());
        GKC_HIP(c, hipStreamSynchronize(c->stream));
        used += take;
    }
    std::vector<uint64_t> h((size_t)3 * nm);
    GKC_HIP(c, hipMemcpy(h.data(), d_stats.p, h.size() * 8, hipMemcpyDeviceToHost));
    for (uint64_t i = 0; i < nm; i++) { if (h_nsk) h_nsk[i] += h[i]; if (h_nk) h_nk[i] += h[nm + i]; if (h_nkx) h_nkx[i] += h[2 * nm + i]; }
    if (reads_used) *reads_used = used;
    return GKC_OK;
}
