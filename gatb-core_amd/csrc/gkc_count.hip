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
    uint32_t pad;          // 0, or (entries of a sliced batch, see count_batch) log2(slices of the partition) << 16 | slice: the entry expands that share of the partition's records
    uint64_t key_base;     // first key of the partition in the batch key buffer
    uint64_t sub_base;     // first sub-bucket of the partition in the batch sub-bucket tables
    uint64_t aux;          // entries of a sliced batch: first word of the entry's row in the per-(slice, sub-bucket) tables
};
// SEVERAL WORKGROUPS PER PARTITION (round 4). The expansion kernels run one workgroup per work-list ENTRY. Normally an entry is a partition. A batch with partitions far
// beyond the size the batches are planned for (a caller that configures 256 partitions for 1e8 reads — the reference's own Configuration at -max-memory 200000 —
// gets 5e7 k-mers per partition: 256 workgroups of very different sizes for 256 CUs, 2 waves per SIMD) is SLICED instead: a partition is 2^s entries, entry w
// expanding the w-th share of its records (of every segment). The counting pass leaves every slice's per-sub-bucket counts c[w][b] in a table; the LAST slice of a
// partition to finish adds them up, lays the partition's sub-buckets out as always and gives every slice its own place inside every sub-bucket:
//     [slice 0's pairs][slice 1's pairs] ... [slice W-1's pairs][the odd keys of the slices, one each at most, in slice order]
// — the pair scatter writes aligned 16-byte pairs and is left with at most one parked key per sub-bucket at the end: a slice's c & ~1 keys fill its pair range
// (every range starts on an even slot: sub-buckets start on multiples of 4), its odd key goes to its slot in the tail. No hole, no global cursor, no extra
// expansion work; the sort tiers see the same contiguous sub-buckets as ever.
struct SliceTables { uint32_t* start; /* in: c[w][b]; out: first slot of the slice's pair range, relative to the partition's first key */ uint32_t* tail; /* slot of the slice's odd key */ uint32_t* done; /* [entries] slices of the partition that have counted (at the partition's first entry) */ };
// records [lo, hi) of a segment's partition range [r0, r1) that slice w of 2^sl_bits takes
__device__ __forceinline__ void slice_range(uint64_t r0, uint64_t r1, uint32_t pad, uint64_t& lo, uint64_t& hi)
{
    const uint32_t sl_bits = pad >> 16, w = pad & 0xFFFFu;
    const uint64_t n = r1 - r0;
    lo = r0 + ((n * w) >> sl_bits); hi = r0 + ((n * (w + 1)) >> sl_bits);
}

struct SegTable {          // device copy of the segment list
    const uint8_t* const* rec;      // [n_seg] arena pointers
    const uint64_t* rec_off;        // [n_seg][P+1]
    uint32_t n_seg, P;
    const uint64_t* rec_end;        // nullptr, or (one segment only) [P]: one past the last record of the partition when its range is not full (deduplicated copy)
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
#ifndef GKC_KMER32_WORDS
#define GKC_KMER32_WORDS 1
#endif
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
#if GKC_KMER32_WORDS
    // Round 5: the forward k-mer from a copy of the string moved DOWN by `down` bits once per record (D): k-mer i is the 128-bit window of D at bit 2i with its top `down` bits
    // (the nucleotides before i) masked off. The loop counter is the same in every lane, so with the 16 steps of a word unrolled the window is four v_alignbit at constant shifts
    // of five words picked once per 16 k-mers, instead of six variable 64-bit shifts and three word selects per k-mer (k = 63, two lanes: Stage B 277.4 -> 273.5 ms, A/B in one
    // call; the same loop NOT unrolled measured slower than the 64-bit form).
    uint64_t D0, D1, D2, D3;
    if (down == 64) { D0 = 0; D1 = S0; D2 = S1; D3 = S2; }
    else { const uint32_t up = 64 - down; D0 = S0 >> down; D1 = (S1 >> down) | (S0 << up); D2 = (S2 >> down) | (S1 << up); D3 = (S3 >> down) | (S2 << up); }
    const uint32_t Dw[8] = { (uint32_t)(D0 >> 32), (uint32_t)D0, (uint32_t)(D1 >> 32), (uint32_t)D1, (uint32_t)(D2 >> 32), (uint32_t)D2, (uint32_t)(D3 >> 32), (uint32_t)D3 };
    const uint64_t mask_hi = down == 64 ? 0ull : (~0ull >> down);
    const uint32_t mh0 = (uint32_t)(mask_hi >> 32), mh1 = (uint32_t)mask_hi;
    for (uint32_t a = 0; a < 4; a++) {
        const uint32_t w0 = a == 0 ? Dw[0] : (a == 1 ? Dw[1] : (a == 2 ? Dw[2] : Dw[3])), w1 = a == 0 ? Dw[1] : (a == 1 ? Dw[2] : (a == 2 ? Dw[3] : Dw[4]));
        const uint32_t w2 = a == 0 ? Dw[2] : (a == 1 ? Dw[3] : (a == 2 ? Dw[4] : Dw[5])), w3 = a == 0 ? Dw[3] : (a == 1 ? Dw[4] : (a == 2 ? Dw[5] : Dw[6]));
        const uint32_t w4 = a == 0 ? Dw[4] : (a == 1 ? Dw[5] : (a == 2 ? Dw[6] : Dw[7]));
#pragma unroll
        for (int q = 0; q < 16; q++) {
            if (16 * a + q >= nbk) return;
            uint32_t v0 = w0, v1 = w1, v2 = w2, v3 = w3;
            if (q) { v0 = __builtin_amdgcn_alignbit(w0, w1, 32 - 2 * q); v1 = __builtin_amdgcn_alignbit(w1, w2, 32 - 2 * q); v2 = __builtin_amdgcn_alignbit(w2, w3, 32 - 2 * q); v3 = __builtin_amdgcn_alignbit(w3, w4, 32 - 2 * q); }
            fh = ((uint64_t)(v0 & mh0) << 32) | (v1 & mh1); fl = ((uint64_t)v2 << 32) | v3;
            if (16 * a + q) {
                const uint64_t c = (uint64_t)((v3 & 3u) ^ 2u);
                rl = (rl >> 2) | (rh << 62); rh >>= 2;
                if (sh >= 64) rh |= c << (sh - 64); else rl |= c << sh;
            }
            const bool fwd = fh < rh || (fh == rh && fl < rl);
            f(fwd ? (((u128)fh << 64) | fl) : (((u128)rh << 64) | rl));
        }
    }
#else
    for (uint32_t i = 0; i < nbk; i++) {
        const bool fwd = fh < rh || (fh == rh && fl < rl);
        f(fwd ? (((u128)fh << 64) | fl) : (((u128)rh << 64) | rl));
        window(i + 1);
        const uint64_t c = (uint64_t)(((uint32_t)fl & 3u) ^ 2u);
        rl = (rl >> 2) | (rh << 62); rh >>= 2;
        if (sh >= 64) rh |= c << (sh - 64); else rl |= c << sh;
    }
#endif
}
// Counting pass for 16-byte keys: only the sub-bucket of every k-mer is needed = the top `bits` (<= 13) bits of min(forward, reverse
// complement). Those are decided by the TOP 64 bits of the two (the first 32 nt of the k-mer / the reverse complement of its last 32):
// when the top words tie, both give the same sub-bucket. So the walk keeps two 64-bit words only: one funnel shift for the forward
// top word, a rolling reverse complement of the last 32 nt. calls f(sub-bucket) for every k-mer of the record (32 <= k <= 63).
#ifndef GKC_SUB32_WORDS
#define GKC_SUB32_WORDS 1
#endif
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
#if GKC_SUB32_WORDS
    // The same walk on 32-bit words (round 5): the sub-bucket is at most 14 bits, so the top THIRTY-TWO bits of the two strands decide it just as well (a tie there leaves the
    // top 14 bits equal either way), and the loop counter is the same in every lane: with the record's string as eight 32-bit words W[] and a copy T[] of it moved up by 2k bits
    // once per record (nucleotide i + k of the string = nucleotide i of T), k-mer i reads fixed words at fixed shifts once the 16 steps of a word are unrolled —
    // one v_alignbit for the forward top word, one v_bfe for the nucleotide that enters, a shift-or for the rolling reverse complement, v_min, a shift:
    // ~8 VALU per k-mer against ~30 of the 64-bit form below (five 64-bit shifts, two 4-way word selects, a 64-bit compare).
    const uint32_t W[9] = { (uint32_t)(S0 >> 32), (uint32_t)S0, (uint32_t)(S1 >> 32), (uint32_t)S1, (uint32_t)(S2 >> 32), (uint32_t)S2, (uint32_t)(S3 >> 32), (uint32_t)S3, 0u };
    const uint32_t kw = (2 * k) >> 5, kb = (2 * k) & 31;                      // k in [32, 63]: kw = 2 or 3 (wave-uniform)
    uint32_t T[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t hi = kw == 2 ? W[j + 2] : W[j + 3], lo = kw == 2 ? W[j + 3] : W[j + 4];
        T[j] = kb ? __builtin_amdgcn_alignbit(hi, lo, 32 - kb) : hi;
    }
    uint32_t r32 = (uint32_t)(rtop >> 32);
    const uint32_t sh32 = 32 - bits;
    for (uint32_t a = 0; a < 4; a++) {                                        // (a: wave-uniform; the words of this round picked once)
        const uint32_t w0 = a == 0 ? W[0] : (a == 1 ? W[1] : (a == 2 ? W[2] : W[3])), w1 = a == 0 ? W[1] : (a == 1 ? W[2] : (a == 2 ? W[3] : W[4]));
        const uint32_t ta = a == 0 ? T[0] : (a == 1 ? T[1] : (a == 2 ? T[2] : T[3]));
#pragma unroll
        for (int q = 0; q < 16; q++) {
            if (16 * a + q >= nbk) return;
            const uint32_t f32 = q ? __builtin_amdgcn_alignbit(w0, w1, 32 - 2 * q) : w0;
            const uint32_t m32 = f32 < r32 ? f32 : r32;
            f(bits ? m32 >> sh32 : 0u);
            r32 = (r32 >> 2) | ((((ta >> (30 - 2 * q)) & 3u) ^ 2u) << 30);    // nucleotide i + k enters the next k-mer at the right
        }
    }
#else
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
#endif
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


// ------------------------------------------------------------------------------------------------ same-address relief (round 5)
// A low-complexity read (poly-A, (AC)n) or a repeat family at hundreds of copies sends 10^5 .. 10^7 keys of ONE k-mer through Stage B: every lane of a wave then
// asks for the same LDS counter, the same parking slot, the same global cursor, and same-address atomics are served one lane at a time (1e8 reads with 1 % such
// reads: k_expand_scatter_pair 1.08 s, k_giant_scatter 0.55 s, k_deep_split 0.29 s per step instead of 42 / 0.2 / 9 ms — profiles/r05_skewed_input.txt). When at least
// SAME_MIN lanes of the wave want what its first active lane wants, ONE lane asks for all of them (the others take their place by their rank among those lanes);
// the rest of the wave, and every wave of ordinary input (6 scalar instructions to find out), goes on as before.
constexpr int SAME_MIN = 8;
__device__ __forceinline__ uint32_t rank_in(uint64_t mask) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u)); }
// Do many lanes of the wave ask for one thing? (>= SAME_MIN lanes agree with the first or with the last active lane.) The expansion kernels ask once per RECORD, with the
// sub-bucket of its first k-mer — all active lanes are in the first step of their record then — and run the per-k-mer relief only for the records of a wave that says yes:
// ordinary input pays the check once per ~11 k-mers (the per-k-mer check alone cost k_expand_count 8.3 -> 11.7 ms).
__device__ __forceinline__ bool wave_same_hint(uint32_t idx)
{
    const uint32_t lead = (uint32_t)__builtin_amdgcn_readfirstlane((int)idx);
    if (__popcll(__ballot(idx == lead)) >= SAME_MIN) return true;
    const uint64_t act = __ballot(true);
    const uint32_t tail = (uint32_t)__builtin_amdgcn_readlane((int)idx, 63 - (int)__builtin_clzll(act));
    return __popcll(__ballot(idx == tail)) >= SAME_MIN;
}
// ctr[idx] += 1 for every active lane
__device__ __forceinline__ void wave_add1(uint32_t* ctr, uint32_t idx)
{
    const uint32_t lead = (uint32_t)__builtin_amdgcn_readfirstlane((int)idx);
    const uint64_t same = __ballot(idx == lead);
    if (__popcll(same) >= SAME_MIN && idx == lead) { if (rank_in(same) == 0) atomicAdd(&ctr[lead], (uint32_t)__popcll(same)); }
    else atomicAdd(&ctr[idx], 1u);
}
// slot = ctr[idx]++ for every active lane
__device__ __forceinline__ uint32_t wave_take1(uint32_t* ctr, uint32_t idx)
{
    const uint32_t lead = (uint32_t)__builtin_amdgcn_readfirstlane((int)idx);
    const uint64_t same = __ballot(idx == lead);
    if (__popcll(same) >= SAME_MIN && idx == lead) {
        const uint32_t r = rank_in(same);
        uint32_t base = 0;
        if (r == 0) base = atomicAdd(&ctr[lead], (uint32_t)__popcll(same));
        base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)__builtin_ctzll(same));
        return base + r;
    }
    return atomicAdd(&ctr[idx], 1u);
}

constexpr int EXPAND_THREADS = 512;
// A key on its way through the sort is the canonical k-mer shifted left by wb WEIGHT BITS with (multiplicity - 1) of its super-k-mer record below it: identical records
// of a partition may be merged before the expansion (k_dedupe_*), their k-mers then count `weight` times. wb is chosen per batch (weight_bits_of): 2 .. 4 — the
// records have 4 spare bits below their nucleotides — as many as the sort can carry: 8-byte keys sort with the f64-tagged network while what is left of a key below
// its sub-bucket index fits the 61 bits that network orders (KTAG64: 2k + wb - sub_bits <= 61). A key of 2k + wb bits may be up to WEIGHT_DROP_MAX = 2
// bits longer than the 64 / 128 it is stored in (k = 31, k = 63 with wb = 3, 4): the TOP bits fall off in the shift, and nothing is lost — every key of a level-1
// sub-bucket shares them (they are the top bits of the sub-bucket's index), k_expand_count notes them beside the sub-bucket (b_consumed, bits 4-5) and the two kernels
// that write the Count records (k_gather_counts, k_root_write) put them back. The order inside a sub-bucket does not depend on bits all its keys share. The stored
// key must never be all ones (the scatter's EMPTY, the sort's padding). With one bit dropped that would take a k-mer [C|G] G..G — neither is canonical (their reverse
// complements C..C[C|G] are smaller); with two, A G..G is canonical and would be all ones at the largest weight — so with two bits dropped a record stands for at
// most 2^wb - 1 copies (weight_cap_of): the weight field is never all ones, whatever the k-mer.
// Measured (1e8 reads, k = 30, no bit dropped: profiles/r04_weight_bits_experiment.txt): weights up to 4 / 8 / 16 merge 1.65x / 1.85x / 1.98x, step 230 / 215 / 211 ms.
constexpr int WEIGHT_BITS_MIN = 2, WEIGHT_BITS_MAX = 4, WEIGHT_DROP_MAX = 2;
constexpr uint32_t CONS_BITS_MASK = 0x0Fu, CONS_GIANT = 0x80u; constexpr int CONS_DROP_SHIFT = 4;      // b_consumed: sub-bucket bits | dropped top bits << 4 | giant flag


// ------------------------------------------------------------------------------------------------ B1 expand_count
// Sub-buckets beyond the first sort tier are handed to the later tiers by LIST, and the lists are made right here, where the exact sizes are first known:
// the tier kernels then run back to back with no host round trip in between (the counts stay on the device; the kernels read them there).
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
// one lane: keep g for the next flush; a full LDS list falls back to the direct append (rare). Appending to a device-wide list through ONE counter
// word saturates at ~9e7 appends/s (MI355X_MICROARCH.md, "dequeue"): hence one reservation per workgroup and list
__device__ __forceinline__ void wglist_push(WgList* L, uint32_t g, uint32_t* count, uint32_t* list)
{
    const uint32_t i = atomicAdd(&L->n, 1u);
    if (i < (uint32_t)WGLIST_CAP) L->item[i] = g;
    else { const uint32_t slot = atomicAdd(count, 1u); list[slot] = g; }
}
constexpr uint32_t GIANT_MIN = 32768, GIANT_MAX = 64, GIANT_WGS = 64, GIANT_CHUNK = 8192;      // see k_giant_or
struct TierLists {
    uint32_t* big_list;   uint32_t* big_count;      // cap1 < n <= cap2: double-size wave network (k_wave_sort_big)
    uint32_t* wg_list;    uint32_t* wg_count;       // cap2 < n <= cap3: workgroup tier (k_wg_sort)
    uint32_t* split_list; uint32_t* split_count;    // n > cap3: split again on the next informative key bits (k_deep_split)
    uint32_t* giant_list; uint32_t* giant_count;    // of those, the first GIANT_MAX beyond GIANT_MIN keys (also flagged in b_consumed, bit 7): first split by many workgroups
    uint32_t cap1, cap2, cap3;
};

template <int KW, int RW>
__global__ __launch_bounds__(EXPAND_THREADS) void k_expand_count(const PartDesc* __restrict__ parts, SegTable segs, uint32_t k,
                                                                  uint64_t* __restrict__ b_start, uint32_t* __restrict__ b_n, uint8_t* __restrict__ b_consumed,
                                                                  TierLists T,
                                                                  const uint32_t* __restrict__ order /* i-th partition to take (largest first), or nullptr */,
                                                                  uint32_t nb, uint32_t* __restrict__ ticket /* partitions are handed out to the workgroups of the launch */,
                                                                  uint32_t drop /* top bits of a k-mer that do not fit the stored key (<= sub_bits of every partition) */,
                                                                  SliceTables ST /* sliced batch: the tables (start != nullptr) */)
{
    typedef typename KeyT<KW>::type key_t;
    __shared__ uint32_t s_hist[MAX_SUB];
    __shared__ uint32_t s_last;
    __shared__ uint32_t s_wsum[EXPAND_THREADS / 64];
    __shared__ WgList s_big, s_wg, s_split;
    __shared__ uint32_t s_item;
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) s_item = atomicAdd(ticket, 1u);
    __syncthreads();
    if (s_item >= nb) break;
    const uint32_t bi = order ? order[s_item] : s_item;
    const PartDesc pd = parts[bi];
    const uint32_t nsub = 1u << pd.sub_bits;
    for (uint32_t i = threadIdx.x; i < nsub; i += EXPAND_THREADS) s_hist[i] = 0;
    if (threadIdx.x == 0) { s_big.n = 0; s_wg.n = 0; s_split.n = 0; }
    __syncthreads();
    for (uint32_t s = 0; s < segs.n_seg; s++) {
        uint64_t r0 = segs.rec_off[(uint64_t)s * (segs.P + 1) + pd.part], r1 = segs.rec_end ? segs.rec_end[pd.part] : segs.rec_off[(uint64_t)s * (segs.P + 1) + pd.part + 1];
        if (pd.pad) slice_range(r0, r1, pd.pad, r0, r1);
        const uint8_t* base = segs.rec[s];
        for (uint64_t r = r0 + threadIdx.x; r < r1; r += EXPAND_THREADS) {
            uint64_t R[RW]; load_rec<RW>(base, r, R);
            if constexpr (KW == 2 && RW == 4) {
                bool first = true, armed = false;                                 // (same-address relief armed per record: wave_same_hint)
                auto add = [&](uint32_t sb) { if (first) { first = false; armed = wave_same_hint(sb); } if (armed) wave_add1(s_hist, sb); else atomicAdd(&s_hist[sb], 1u); };
                if (k >= 32) for_each_sub32(R, k, pd.sub_bits, [&](uint32_t sb) { add(sb); });
                else for_each_kmer_fast<KW, RW>(R, k, [&](key_t c) { add(sub_index<KW>(c, pd.shift)); });
            } else {
                bool first = true, armed = false;
                for_each_kmer_fast<KW, RW>(R, k, [&](key_t c) { const uint32_t sb = sub_index<KW>(c, pd.shift); if (first) { first = false; armed = wave_same_hint(sb); } if (armed) wave_add1(s_hist, sb); else atomicAdd(&s_hist[sb], 1u); });
            }
        }
    }
    __syncthreads();
    const uint32_t n_slices = 1u << (pd.pad >> 16), slice = pd.pad & 0xFFFFu;
    const uint64_t row0 = pd.aux - (uint64_t)slice * nsub;                  // the partition's rows of the slice tables: one per slice, consecutive
    if (ST.start) {
        // sliced batch: the slice's counts go to its row; the last slice of the partition to get here lays the partition out
        for (uint32_t i = threadIdx.x; i < nsub; i += EXPAND_THREADS) ST.start[pd.aux + i] = s_hist[i];
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) s_last = atomicAdd(&ST.done[bi - slice], 1u) == n_slices - 1u ? 1u : 0u;
        __syncthreads();
        if (!s_last) continue;
        __threadfence();
        if (n_slices > 1) {
            for (uint32_t i = threadIdx.x; i < nsub; i += EXPAND_THREADS) {
                uint32_t n = 0;
                for (uint32_t w = 0; w < n_slices; w++) n += ST.start[row0 + (uint64_t)w * nsub + i];
                s_hist[i] = n;
            }
            __syncthreads();
        }
    }
    // exclusive scan of the counters -> absolute key offsets of the sub-buckets; every sub-bucket starts on a multiple of 4 slots (pair scatter)
    const uint32_t per = (nsub + EXPAND_THREADS - 1) / EXPAND_THREADS;      // <= 16
    const uint32_t b = threadIdx.x * per;
    uint32_t loc = 0;
    for (uint32_t i = 0; i < per; i++) if (b + i < nsub) loc += (s_hist[b + i] + 3u) & ~3u;
    uint32_t x = loc;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { uint32_t y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
    if (lane == 63) s_wsum[wave] = x;
    __syncthreads();
    uint32_t wpre = 0;
    for (int w = 0; w < wave; w++) wpre += s_wsum[w];
    uint32_t run = wpre + x - loc;
    for (uint32_t i = 0; i < per; i++) if (b + i < nsub) {
        const uint32_t j = b + i, n = s_hist[j];
        const uint32_t g = (uint32_t)(pd.sub_base + j);
        const uint32_t cons = pd.sub_bits | ((drop ? j >> (pd.sub_bits - drop) : 0u) << CONS_DROP_SHIFT);      // the sub-bucket's keys all start with these `drop` bits
        b_start[g] = pd.key_base + run; b_n[g] = n; b_consumed[g] = (uint8_t)cons;
        if (ST.start) {                                                      // every slice's place inside the sub-bucket: pair ranges in slice order, then the odd keys
            uint32_t pos = run, oddm = 0;
            for (uint32_t w = 0; w < n_slices; w++) {
                const uint64_t at = row0 + (uint64_t)w * nsub + j;
                const uint32_t cw = ST.start[at];
                ST.start[at] = pos; pos += cw & ~1u; oddm |= (cw & 1u) << w;
            }
            for (uint32_t w = 0; w < n_slices; w++) ST.tail[row0 + (uint64_t)w * nsub + j] = pos + __popc(oddm & ((1u << w) - 1u));      // (read only by a slice whose count is odd)
        }
        if (n > T.cap1) {
            if (n <= T.cap2) wglist_push(&s_big, g, T.big_count, T.big_list);
            else if (n <= T.cap3) wglist_push(&s_wg, g, T.wg_count, T.wg_list);
            else {
                wglist_push(&s_split, g, T.split_count, T.split_list);
                if (n > GIANT_MIN) { const uint32_t y = atomicAdd(T.giant_count, 1u); if (y < GIANT_MAX) { T.giant_list[y] = g; b_consumed[g] = (uint8_t)(cons | CONS_GIANT); } }
            }
        }
        run += (n + 3u) & ~3u;
    }
    wglist_flush(&s_big, T.big_count, T.big_list);
    wglist_flush(&s_wg, T.wg_count, T.wg_list);
    wglist_flush(&s_split, T.split_count, T.split_list);
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
                                                                       const uint32_t* __restrict__ order /* i-th partition to take (largest first), or nullptr */,
                                                                       uint32_t nb, uint32_t* __restrict__ ticket /* partitions are handed out to the workgroups of the launch */, uint32_t wb /* weight bits */,
                                                                       SliceTables ST /* sliced batch (start != nullptr): the entry is a slice of a partition's records with its own place in every sub-bucket */)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long s_pend[];       // [nsub] parked key or EMPTY
    __shared__ uint32_t s_item;
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) s_item = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint32_t item = s_item;
    if (item >= nb) break;
    const PartDesc pd = parts[order ? order[item] : item];
    const uint32_t nsub = 1u << pd.sub_bits;
    uint32_t* s_cur = reinterpret_cast<uint32_t*>(s_pend + nsub);                     // [nsub] next free slot of the sub-bucket
    constexpr unsigned long long EMPTY = ~0ULL;
    for (uint32_t i = threadIdx.x; i < nsub; i += PAIR_THREADS) { s_pend[i] = EMPTY; s_cur[i] = ST.start ? ST.start[pd.aux + i] : (uint32_t)(b_start[pd.sub_base + i] - pd.key_base); }
    __syncthreads();
    uint64_t* out = keys + pd.key_base;
    for (uint32_t s = 0; s < segs.n_seg; s++) {
        uint64_t r0 = segs.rec_off[(uint64_t)s * (segs.P + 1) + pd.part], r1 = segs.rec_end ? segs.rec_end[pd.part] : segs.rec_off[(uint64_t)s * (segs.P + 1) + pd.part + 1];
        if (pd.pad) slice_range(r0, r1, pd.pad, r0, r1);
        const ulonglong2* recs = reinterpret_cast<const ulonglong2*>(segs.rec[s]);
        unsigned long long* const s_comb0 = s_pend + (((size_t)nsub * 12 + 15) / 16) * 2;                   // behind the cursors, 16-byte aligned
        unsigned long long* const s_comb = s_comb0 + (size_t)(threadIdx.x >> 6) * 64;                         // this wave's 64 words
        bool rec_first = true, armed = false;                                    // same-address relief armed per record (wave_same_hint)
        auto emit = [&](uint64_t c, unsigned long long wq) {
            const uint32_t q = (uint32_t)(c >> pd.shift);
            unsigned long long h = (c << wb) | wq;                               // (bits beyond the 64th fall off: see above) never all ones
            if (rec_first) { rec_first = false; armed = wave_same_hint(q); }
            // same-address relief (see wave_add1): >= SAME_MIN lanes of the wave with keys for ONE sub-bucket (one k-mer at 10^6 copies) pair up among themselves —
            // keys into the wave's LDS words by rank, one reservation for all the pairs, lane r < pairs stores words 2r, 2r + 1; an odd last lane takes the usual way
            if (armed) {
                const uint32_t lead = (uint32_t)__builtin_amdgcn_readfirstlane((int)q);
                const uint64_t same = __ballot(q == lead);
                const uint32_t ns = (uint32_t)__popcll(same);
                if (ns >= (uint32_t)SAME_MIN) {
                    bool done = false;
                    if (q == lead) {
                        const uint32_t r = rank_in(same), np = ns >> 1;
                        s_comb[r] = h;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        uint32_t base = 0;
                        if (r == 0) base = atomicAdd(&s_cur[lead], 2u * np);
                        base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)__builtin_ctzll(same));
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        if (r < np) store16(out + base + 2 * r, s_comb[2 * r], s_comb[2 * r + 1]);
                        done = !((ns & 1u) && r == ns - 1u);
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");          // (the words are reused by the wave's next group)
                    }
                    if (done) return;
                }
            }
            for (;;) {
                const unsigned long long y = atomicExch(&s_pend[q], EMPTY);
                if (y != EMPTY) {
                    const uint32_t p = atomicAdd(&s_cur[q], 2u);
                    store16(out + p, y, h);
                    break;
                }
                const unsigned long long z = atomicExch(&s_pend[q], h);
                if (z == EMPTY) break;
                h = z;
            }
        };
        uint64_t r = r0 + threadIdx.x;
        ulonglong2 nx = r < r1 ? recs[r] : make_ulonglong2(0, 0);
        for (; r < r1; r += PAIR_THREADS) {
            const uint64_t R[2] = {nx.x, nx.y};
            if (r + PAIR_THREADS < r1) nx = recs[r + PAIR_THREADS];                  // next record in flight while this one is expanded
            const unsigned long long wq = R[1] & ((1ull << wb) - 1ull);                              // the record's weight - 1 (below the nucleotides; 0 unless the records were deduplicated)
            rec_first = true;
            for_each_kmer16(R, k, [&](uint64_t c) { emit(c, wq); });
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nsub; i += PAIR_THREADS) { const unsigned long long v = s_pend[i]; if (v != EMPTY) out[ST.start ? ST.tail[pd.aux + i] : s_cur[i]] = v; }      // (a slice's odd key: its slot in the sub-bucket's tail)
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
                                                                        const uint32_t* __restrict__ order /* i-th partition to take (largest first), or nullptr */,
                                                                        uint32_t nb, uint32_t* __restrict__ ticket /* partitions are handed out to the workgroups of the launch */, uint32_t wb /* weight bits */,
                                                                        SliceTables ST /* sliced batch: see k_expand_scatter_pair */)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long s_pend[];       // [nsub][2] parked key (low word, high word) or EMPTY
  for (;;) {
    // the ticket travels through the first word of the parking table (dead between two partitions): the table + cursors take the CU's whole 160 KB of LDS
    __syncthreads();
    if (threadIdx.x == 0) *reinterpret_cast<volatile uint32_t*>(s_pend) = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint32_t item = *reinterpret_cast<volatile uint32_t*>(s_pend);
    __syncthreads();
    if (item >= nb) break;
    const PartDesc pd = parts[order ? order[item] : item];
    const uint32_t nsub = 1u << pd.sub_bits;
    uint32_t* s_cur = reinterpret_cast<uint32_t*>(s_pend + 2 * (size_t)nsub);        // [nsub] next free slot of the sub-bucket
    constexpr unsigned long long EMPTY = ~0ULL;
    for (uint32_t i = threadIdx.x; i < nsub; i += PAIR_THREADS) { s_pend[2 * i] = EMPTY; s_pend[2 * i + 1] = EMPTY; s_cur[i] = ST.start ? ST.start[pd.aux + i] : (uint32_t)(b_start[pd.sub_base + i] - pd.key_base); }
    __syncthreads();
    ulonglong2* out = reinterpret_cast<ulonglong2*>(keys + pd.key_base);              // one 16-byte key per element (x = low word, y = high word)
    for (uint32_t s = 0; s < segs.n_seg; s++) {
        uint64_t r0 = segs.rec_off[(uint64_t)s * (segs.P + 1) + pd.part], r1 = segs.rec_end ? segs.rec_end[pd.part] : segs.rec_off[(uint64_t)s * (segs.P + 1) + pd.part + 1];
        if (pd.pad) slice_range(r0, r1, pd.pad, r0, r1);
        const ulonglong2* recs = reinterpret_cast<const ulonglong2*>(segs.rec[s]);     // 32-byte records: two elements each
        uint64_t r = r0 + threadIdx.x;
        ulonglong2 nx0 = make_ulonglong2(0, 0), nx1 = nx0;
        if (r < r1) { nx0 = recs[2 * r]; nx1 = recs[2 * r + 1]; }
        for (; r < r1; r += PAIR_THREADS) {
            const uint64_t R[4] = {nx0.x, nx0.y, nx1.x, nx1.y};
            if (r + PAIR_THREADS < r1) { nx0 = recs[2 * (r + PAIR_THREADS)]; nx1 = recs[2 * (r + PAIR_THREADS) + 1]; }   // next record in flight
            const uint64_t wq = R[3] & ((1ull << wb) - 1ull);                                                   // the record's weight - 1
            bool rec_first = true, armed = false;                                                                // same-address relief armed per record (wave_same_hint)
            for_each_kmer32(R, k, [&](u128 c) {
                const uint32_t q = sub_index<2>(c, pd.shift);
                const u128 st = (c << wb) | (u128)wq;                                    // (bits beyond the 128th fall off) never all ones
                uint64_t h_lo = (uint64_t)st, h_hi = (uint64_t)(st >> 64);
                if (rec_first) { rec_first = false; armed = wave_same_hint(q); }
                if (armed) {   // same-address relief, as in k_expand_scatter_pair — without LDS words (the parking table of 8192 sub-buckets takes the CU's whole LDS, and the partitions
                    // that need this are the ones with 8192): a lane's partner is the NEXT lane of the group, its key comes over by ds_bpermute
                    const uint32_t lead = (uint32_t)__builtin_amdgcn_readfirstlane((int)q);
                    const uint64_t same = __ballot(q == lead);
                    const uint32_t ns = (uint32_t)__popcll(same);
                    if (ns >= (uint32_t)SAME_MIN) {
                        bool done = false;
                        if (q == lead) {
                            const int lane = threadIdx.x & 63;
                            const uint32_t r = rank_in(same), np = ns >> 1;
                            const uint64_t above = same & ~((2ull << lane) - 1ull);
                            const int pl = above ? (int)__builtin_ctzll(above) : lane;
                            const uint64_t p_lo = (uint64_t)__shfl((unsigned long long)h_lo, pl, 64), p_hi = (uint64_t)__shfl((unsigned long long)h_hi, pl, 64);
                            uint32_t base = 0;
                            if (r == 0) base = atomicAdd(&s_cur[lead], 2u * np);
                            base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)__builtin_ctzll(same));
                            if (!(r & 1u) && r + 1u < ns) { out[base + r] = make_ulonglong2(h_lo, h_hi); out[base + r + 1] = make_ulonglong2(p_lo, p_hi); }
                            done = !((ns & 1u) && r == ns - 1u);
                        }
                        if (done) return;
                    }
                }
                for (;;) {
                    uint64_t y_lo, y_hi;
                    lds_xchg128(&s_pend[2 * (size_t)q], EMPTY, EMPTY, y_lo, y_hi);
                    if (y_hi != EMPTY || y_lo != EMPTY) {
                        const uint32_t p = atomicAdd(&s_cur[q], 2u);
                        out[p] = make_ulonglong2(y_lo, y_hi); out[p + 1] = make_ulonglong2(h_lo, h_hi);
                        break;
                    }
                    uint64_t z_lo, z_hi;
                    lds_xchg128(&s_pend[2 * (size_t)q], h_lo, h_hi, z_lo, z_hi);
                    if (z_hi == EMPTY && z_lo == EMPTY) break;
                    h_lo = z_lo; h_hi = z_hi;
                }
            });
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nsub; i += PAIR_THREADS) { const unsigned long long lo = s_pend[2 * i], hi = s_pend[2 * i + 1]; if (hi != EMPTY || lo != EMPTY) out[ST.start ? ST.tail[pd.aux + i] : s_cur[i]] = make_ulonglong2(lo, hi); }
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
    uint8_t* cnt8;            // [n_slots] abundance of the distinct key written at the same slot, saturated at 255
    uint32_t* cnt32;          // [n_slots] full abundance, written (and later read) only where cnt8 == 255: the abundance
                              // plane costs 1 byte per distinct key of HBM traffic instead of 4
    unsigned long long* histo; uint32_t histo_max;
    uint32_t* nd;             // [n_sub] distinct k-mers of the level-1 sub-bucket (they sit at the head of its slot range; a split sub-bucket: anywhere in it)
    uint32_t* ns;             // [n_sub] of those, the ones inside the solidity window (== nd, the same array, when the window is open)
    int32_t amin, amax; uint32_t all_solid;
    uint32_t wb;              // weight bits below the k-mer in every key of the batch
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
// In-lane compare-exchange. F (8-byte keys only): the keys of the bucket carry a tag in their top bits that makes them positive normal doubles, whose order is the
// order of their bit patterns, and the exchange is the two native 64-bit instructions v_min_f64 / v_max_f64 (they return one of their operands, bit for bit)
// instead of a 64-bit compare and four selects. Cross-lane steps compare the same bit patterns as integers.
//   TAG64 / TAG64_MANT (the record deduplication's hash words): exponent 0x433 above 52 bits.
//   KTAG64 / KTAG64_MANT (the key sorts): only the top THREE bits are the tag — sign 0, exponent bits 10 and 9 = 1, 0 — and the 9 exponent bits below them belong
//   to the key with the 52 mantissa bits: exponents 0x400 .. 0x5FF, never 0 (denormal), never 0x7FF (infinity / NaN), so 61 key bits sort this way. (Rounds 2-3
//   kept the whole exponent fixed: 52 bits, which made every partition take 12-13 sub-bucket bits whatever its size.) The three bits the tag replaces are bits all
//   keys of a bucket share: 2k + weight bits - sub-bucket bits <= 61, checked per launch, integer path otherwise.
constexpr uint64_t TAG64 = 0x4330000000000000ULL, TAG64_MANT = 0x000FFFFFFFFFFFFFULL;
constexpr uint64_t KTAG64 = 0x4000000000000000ULL, KTAG64_MANT = 0x1FFFFFFFFFFFFFFFULL; constexpr int KTAG64_BITS = 61;
constexpr int KTAG128_BITS = 125;            // 16-byte keys: the same tag on the TOP word (three shared top bits), all 64 bits of the low word are key bits
template <int KW> struct KTagBits { static constexpr int value = KW == 1 ? KTAG64_BITS : KTAG128_BITS; };
// 16-byte keys (round 4): the order of two keys written out on their 64-bit halves. The compiler's own `y < x` on unsigned __int128 costs ~23 VALU instructions
// per compare-exchange on gfx950 (the i1 results are materialised through v_cndmask / v_and, 5 s_nop per exchange); spelled out it is 3 compares + 2 SALU ops,
// and the exchange 8 selects (11 VALU) — or, with the TOP word tagged like an 8-byte key (KTag<2>: its three top bits, which all keys of a bucket share, replaced
// by the double tag), v_min_f64 / v_max_f64 on the top words and 4 selects on the low ones (9 VALU).
template <int KW> __device__ __forceinline__ bool key_lt(typename KeyT<KW>::type a, typename KeyT<KW>::type b)
{
    if constexpr (KW == 1) return a < b;
    else { const uint64_t ah = (uint64_t)(a >> 64), al = (uint64_t)a, bh = (uint64_t)(b >> 64), bl = (uint64_t)b; return (ah < bh) | ((ah == bh) & (al < bl)); }
}
// the k-mers of two keys differ (the low WMASK bits are the weight)
template <int KW> __device__ __forceinline__ bool key_differs(typename KeyT<KW>::type a, typename KeyT<KW>::type b, uint32_t WMASK)
{
    if constexpr (KW == 1) return (a ^ b) > (uint64_t)WMASK;
    else return (((uint64_t)(a >> 64) ^ (uint64_t)(b >> 64)) | (((uint64_t)a ^ (uint64_t)b) & ~(uint64_t)WMASK)) != 0;
}
// tag of the f64-ordered form of a key: the whole 8-byte key, or the top word of a 16-byte one
template <int KW> struct KTag;
template <> struct KTag<1> { static __device__ __forceinline__ uint64_t mant() { return KTAG64_MANT; }  static __device__ __forceinline__ uint64_t tag() { return KTAG64; } };
template <> struct KTag<2> { static __device__ __forceinline__ u128 mant() { return ((u128)KTAG64_MANT << 64) | (u128)~0ULL; }  static __device__ __forceinline__ u128 tag() { return (u128)KTAG64 << 64; } };
template <int KW, bool F> __device__ __forceinline__ void ce_inlane(typename KeyT<KW>::type& a, typename KeyT<KW>::type& b)
{
    if constexpr (F && KW == 1) {
        const double x = __longlong_as_double((long long)a), y = __longlong_as_double((long long)b);
        double lo, hi;
        asm("v_min_f64 %0, %1, %2" : "=v"(lo) : "v"(x), "v"(y));
        asm("v_max_f64 %0, %1, %2" : "=v"(hi) : "v"(x), "v"(y));
        a = (uint64_t)__double_as_longlong(lo); b = (uint64_t)__double_as_longlong(hi);
    } else if constexpr (F && KW == 2) {
        const uint64_t ah = (uint64_t)(a >> 64), al = (uint64_t)a, bh = (uint64_t)(b >> 64), bl = (uint64_t)b;
        const bool sw = (bh < ah) | ((bh == ah) & (bl < al));
        const double x = __longlong_as_double((long long)ah), y = __longlong_as_double((long long)bh);
        double mn, mx;
        asm("v_min_f64 %0, %1, %2" : "=v"(mn) : "v"(x), "v"(y));
        asm("v_max_f64 %0, %1, %2" : "=v"(mx) : "v"(x), "v"(y));
        const uint64_t l0 = sw ? bl : al, l1 = sw ? al : bl;
        a = ((u128)(uint64_t)__double_as_longlong(mn) << 64) | l0; b = ((u128)(uint64_t)__double_as_longlong(mx) << 64) | l1;
    } else { const typename KeyT<KW>::type x = a, y = b; const bool sw = key_lt<KW>(y, x); a = sw ? y : x; b = sw ? x : y; }
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
                for (int r = 0; r < KPL; r++) { const key_t y = Shfl<KW>::template x<LS>(v[r]); const bool ylt = key_lt<KW>(y, v[r]); v[r] = (ylt == low) ? y : v[r]; }
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
                for (int r = 0; r < KPL; r++) { const key_t y = Shfl<KW>::template x<LMASK>(v[KPL - 1 - r]); const bool ylt = key_lt<KW>(y, v[r]); w[r] = (ylt == low) ? y : v[r]; }
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
// nd_out / ns_out (wave-uniform): distinct k-mers of the bucket / those inside the solidity window.
template <int KW, int KPL, bool F = false>
__device__ __forceinline__ void wave_sort_bucket(const typename KeyT<KW>::type* src /* first key of the bucket */,
                                                 typename KeyT<KW>::type* __restrict__ outk,
                                                 const uint64_t start, const uint32_t n, const SortOut& O, uint32_t* s_hc, const int lane,
                                                 uint32_t& nd_out, uint32_t& ns_out)
{
    typedef typename KeyT<KW>::type key_t;
    key_t v[KPL];
    key_t top = 0;                                               // F: the 3 top bits every key of the bucket shares (replaced by the tag while sorting)
#pragma unroll
    for (int r = 0; r < KPL; r++) { const uint32_t i = r * 64 + lane; v[r] = i < n ? src[i] : KeyT<KW>::max(); }
    if constexpr (F) {
        top = src[0] & ~KTag<KW>::mant();
#pragma unroll
        for (int r = 0; r < KPL; r++) v[r] = (v[r] & KTag<KW>::mant()) | KTag<KW>::tag();        // padding (all ones) becomes the largest tagged value: not below any key
    }
    bitonic_wave<KW, KPL, F>(v, lane);
    // run-length count (B3), weighted: e = lane*KPL + r is the sorted rank; a key is the k-mer above O.wb bits of (multiplicity - 1): equal k-mers are
    // adjacent whatever their weights, the abundance of a run is the sum of its weights
    const key_t prev_last = Shfl<KW>::up(v[KPL - 1]);
    const key_t next_first = Shfl<KW>::down(v[0]);
    const uint32_t WB = O.wb, WMASK = (1u << WB) - 1u;
    auto differs = [WMASK](key_t a, key_t b) -> uint32_t { return key_differs<KW>(a, b, WMASK) ? 1u : 0u; };
    // whole-lane bit masks (bit r = rank lane*KPL + r): one compare per key, the tail logic on the masks
    uint32_t neq = lane == 0 ? 1u : differs(v[0], prev_last);                      // k-mer differs from the one before it (rank 0: always)
#pragma unroll
    for (int r = 1; r < KPL; r++) neq |= differs(v[r], v[r - 1]) << r;
    const uint32_t lane0 = (uint32_t)lane * KPL;
    const uint32_t have = n > lane0 ? (n - lane0 < (uint32_t)KPL ? n - lane0 : (uint32_t)KPL) : 0u;     // ranks of this lane below n
    const uint32_t inm = have >= 32u ? 0xFFFFFFFFu : ((1u << have) - 1u);
    const uint32_t nxt_differs = differs(v[KPL - 1], next_first);
    const uint32_t lastm = (have && lane0 + have == n) ? (1u << (have - 1u)) : 0u;                    // rank n-1 closes its run
    const uint32_t tailm = inm & ((neq >> 1) | (nxt_differs << (KPL - 1)) | lastm);
    const uint32_t nt = __popc(tailm);
    // weights of the lane's ranks: total, and the part up to its last run end
    uint32_t wtot = 0, wlast = 0;
    {   const uint32_t lastbit = tailm ? (uint32_t)(31 - __clz((int)tailm)) : 0u;
#pragma unroll
        for (int r = 0; r < KPL; r++) {
            const uint32_t w = ((inm >> r) & 1u) ? ((uint32_t)v[r] & WMASK) + 1u : 0u;
            wtot += w;
            if ((uint32_t)r <= lastbit) wlast += w;
        }
    }
    uint32_t x = nt, t = wtot;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d, 64), u = __shfl_up(t, d, 64); if (lane >= d) { x += y; t += u; } }
    uint32_t idx = x - nt;
    nd_out = (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
    const uint32_t wbase = t - wtot;                                              // weight of all ranks before this lane
    // weight up to the last run end before this lane (0: none): the weights are monotone along the ranks, so a max-scan carries it
    uint32_t lt = tailm ? wbase + wlast : 0u;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(lt, d, 64); if (lane >= d) lt = y > lt ? y : lt; }
    uint32_t prevw = __shfl_up(lt, 1, 64); if (lane == 0) prevw = 0;
    uint32_t nsol = 0, run = wbase;
#pragma unroll
    for (int r = 0; r < KPL; r++) {
        run += ((inm >> r) & 1u) ? ((uint32_t)v[r] & WMASK) + 1u : 0u;
        if ((tailm >> r) & 1) {
            const uint32_t c = run - prevw;                                       // the run that ends here: every weight since the previous run end
            prevw = run;
            if constexpr (F) outk[start + idx] = ((v[r] & KTag<KW>::mant()) | top) >> WB; else outk[start + idx] = v[r] >> WB;
            put_count(O, start + idx, c);
            idx++;
            nsol += ((int32_t)c >= O.amin && (int32_t)c <= O.amax) ? 1u : 0u;      // CountRange::includes (closed interval)
            const uint32_t hb = c >= O.histo_max ? O.histo_max : c;              // Histogram::inc (Histogram.hpp:92)
            if (hb < HIST_LDS) atomicAdd(&s_hc[hb], 1u); else atomicAdd(&O.histo[hb], 1ULL);
        }
    }
    if (O.all_solid) ns_out = nd_out;
    else {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) nsol += __shfl_xor(nsol, d, 64);
        ns_out = nsol;
    }
}

// register capacity of the wave tiers: k_wave_sort up to 1024 (u64) / 512 (u128) keys, k_wave_sort_big twice that
template <int KW> struct WaveCap { static constexpr int KPL_MAX = (KW == 1) ? 8 : 4; static constexpr uint32_t CAP = 64 * KPL_MAX; };
template <int KW> struct WaveCapBig { static constexpr int KPL_MAX = (KW == 1) ? 16 : 8; static constexpr uint32_t CAP = 64 * KPL_MAX; };

template <int KW, int KPLMAX, bool F = false>
__device__ __forceinline__ void wave_sort_dispatch(const typename KeyT<KW>::type* src, typename KeyT<KW>::type* __restrict__ outk, const uint64_t start,
                                                   const uint32_t n, const SortOut& O, uint32_t* s_hc, const int lane, uint32_t& nd, uint32_t& ns)
{
    if (n <= 64) wave_sort_bucket<KW, 1, F>(src, outk, start, n, O, s_hc, lane, nd, ns);
    else if (n <= 128) wave_sort_bucket<KW, 2, F>(src, outk, start, n, O, s_hc, lane, nd, ns);
    else if (n <= 256 || KPLMAX == 4) wave_sort_bucket<KW, 4, F>(src, outk, start, n, O, s_hc, lane, nd, ns);
    else if (n <= 512 || KPLMAX == 8) wave_sort_bucket<KW, (KPLMAX >= 8 ? 8 : 4), F>(src, outk, start, n, O, s_hc, lane, nd, ns);
    else wave_sort_bucket<KW, KPLMAX, F>(src, outk, start, n, O, s_hc, lane, nd, ns);
}

// one WAVE per small bucket, straight from HBM (no LDS, no barrier)
#ifndef GKC_WS_WAVES
#define GKC_WS_WAVES 5      // waves per SIMD the register budget is cut for: 3 (151 VGPRs) 98 ms, 4: 87 ms, 5: 84 ms, 6: 85 ms per 1.2e10 keys
#endif
// register capacity of the first tier: one wave holds 16 (u64) / 8 (u128) keys per lane
template <int KW> struct WaveCapT1 { static constexpr int KPL_MAX = WaveCapBig<KW>::KPL_MAX; static constexpr uint32_t CAP = 64 * KPL_MAX; };
__device__ __forceinline__ void put_nd(const SortOut& O, uint32_t g, uint32_t nd, uint32_t ns, int lane)
{
    if (lane == 0) { O.nd[g] = nd; if (!O.all_solid) O.ns[g] = ns; }
}
template <int KW, bool F>
__global__ __launch_bounds__(SORT_THREADS, GKC_WS_WAVES) void k_wave_sort(const typename KeyT<KW>::type* __restrict__ src, typename KeyT<KW>::type* __restrict__ outk,
                                                             const uint64_t* __restrict__ b_start, const uint32_t* __restrict__ b_n, uint32_t n_buckets, SortOut O)
{
    __shared__ uint32_t s_hc[HIST_LDS];
    const int t = threadIdx.x, lane = t & 63;
    if (t < HIST_LDS) s_hc[t] = 0;
    __syncthreads();
    const uint32_t wave = (blockIdx.x * SORT_THREADS + t) >> 6, n_waves = (gridDim.x * SORT_THREADS) >> 6;
    uint32_t n_next = wave < n_buckets ? b_n[wave] : 0; uint64_t start_next = wave < n_buckets ? b_start[wave] : 0;
    for (uint32_t g = wave; g < n_buckets; g += n_waves) {
        const uint32_t n = n_next; const uint64_t start = start_next;
        if (g + n_waves < n_buckets) { n_next = b_n[g + n_waves]; start_next = b_start[g + n_waves]; }     // next bucket's descriptor in flight during this sort
        if (n == 0 || n > WaveCapT1<KW>::CAP) continue;            // empty, or a later tier's bucket (listed by k_expand_count)
        uint32_t nd, ns;
        wave_sort_dispatch<KW, WaveCapT1<KW>::KPL_MAX, F>(src + start, outk, start, n, O, s_hc, lane, nd, ns);
        put_nd(O, g, nd, ns, lane);
    }
    __syncthreads();
    if (t < HIST_LDS && s_hc[t]) atomicAdd(&O.histo[t], (unsigned long long)s_hc[t]);
}

// second tier: buckets up to twice the first tier's size (2048 / 1024 keys), one wave each with a double-size network; only
// ~5 % of the keys come here, so the lower occupancy of this kernel (64+ key registers) does not touch the first tier
template <int KW> struct WaveCapHuge { static constexpr int KPL = (KW == 1) ? 32 : 16; static constexpr uint32_t CAP = 64 * KPL; };
#ifndef GKC_WSB_WAVES
#define GKC_WSB_WAVES 3     // 2 (214 VGPRs): 16.3 ms, 3: 13.4 ms, 4: 13.2 ms
#endif
template <int KW, bool F, int KPL>
__global__ __launch_bounds__(SORT_THREADS, GKC_WSB_WAVES) void k_wave_sort_big(const typename KeyT<KW>::type* __restrict__ src, typename KeyT<KW>::type* __restrict__ outk,
                                                                 const uint64_t* __restrict__ b_start, const uint32_t* __restrict__ b_n,
                                                                 const uint32_t* __restrict__ list, const uint32_t* __restrict__ n_list_p /* on the device: written by k_expand_count */, SortOut O)
{
    __shared__ uint32_t s_hc[HIST_LDS];
    const int t = threadIdx.x, lane = t & 63;
    if (t < HIST_LDS) s_hc[t] = 0;
    __syncthreads();
    const uint32_t n_list = *n_list_p;
    const uint32_t wave = (blockIdx.x * SORT_THREADS + t) >> 6, n_waves = (gridDim.x * SORT_THREADS) >> 6;
    for (uint32_t li = wave; li < n_list; li += n_waves) {
        const uint32_t g = list[li];
        const uint32_t n = b_n[g];
        const uint64_t start = b_start[g];
        uint32_t nd, ns;
        wave_sort_bucket<KW, KPL, F>(src + start, outk, start, n, O, s_hc, lane, nd, ns);
        put_nd(O, g, nd, ns, lane);
    }
    __syncthreads();
    if (t < HIST_LDS && s_hc[t]) atomicAdd(&O.histo[t], (unsigned long long)s_hc[t]);
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
                                                        const uint32_t* __restrict__ list, const uint32_t* __restrict__ n_list_p /* on the device: written by k_expand_count */, SortOut O)
{
    typedef typename KeyT<KW>::type key_t;
    constexpr uint32_t CAPW = 64 * KPL, CAP = NW * CAPW;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    key_t* s_x = reinterpret_cast<key_t*>(s_raw);                 // [NW][KPL][64] exchange buffer
    __shared__ key_t s_first[NW], s_last[NW];
    __shared__ uint32_t s_tails[NW], s_wsum[NW], s_lt[NW];
    __shared__ uint32_t s_hc[HIST_LDS];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    __shared__ uint32_t s_sol[NW];
    if (t < HIST_LDS) s_hc[t] = 0;
    const uint32_t n_list = *n_list_p;
    for (uint32_t li = blockIdx.x; li < n_list; li += gridDim.x) {
        const uint32_t g = list[li];
        const uint32_t n = b_n[g];
        if (n > CAP) continue;                                    // cannot happen (k_expand_count lists n <= cap3 = CAP here)
        const uint64_t start = b_start[g];
        key_t v[KPL];
#pragma unroll
        for (int r = 0; r < KPL; r++) { const uint32_t i = w * CAPW + r * 64 + lane; v[r] = i < n ? src[start + i] : KeyT<KW>::max(); }
        key_t top = 0;
        if constexpr (F) {
            top = src[start] & ~KTag<KW>::mant();
#pragma unroll
            for (int r = 0; r < KPL; r++) v[r] = (v[r] & KTag<KW>::mant()) | KTag<KW>::tag();
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
                for (int r = 0; r < KPL; r++) { const key_t y = other[(KPL - 1 - r) * 64 + (63 - lane)]; const bool ylt = key_lt<KW>(y, v[r]); v[r] = (ylt == low) ? y : v[r]; }
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
                for (int r = 0; r < KPL; r++) { const key_t y = other[r * 64 + lane]; const bool ylt = key_lt<KW>(y, v[r]); v[r] = (ylt == low) ? y : v[r]; }
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
        const uint32_t WB = O.wb, WMASK = (1u << WB) - 1u;
        auto differs = [WMASK](key_t a, key_t b) -> bool { return key_differs<KW>(a, b, WMASK); };     // the k-mers above the weight bits differ
        uint32_t tailm = 0, inm = 0;
#pragma unroll
        for (int r = 0; r < KPL; r++) {
            const uint32_t e = E0 + r;
            const key_t nx = (r < KPL - 1) ? v[r + 1] : next_first;
            const bool in = e < n;
            inm |= (uint32_t)in << r;
            tailm |= (uint32_t)(in && (e == n - 1 || differs(v[r], nx))) << r;
        }
        (void)prev_last;
        const uint32_t nt = __popc(tailm);
        uint32_t wtot = 0, wlast = 0;                              // weights of the lane's ranks: total, and the part up to its last run end
        {   const uint32_t lastbit = tailm ? (uint32_t)(31 - __clz((int)tailm)) : 0u;
#pragma unroll
            for (int r = 0; r < KPL; r++) {
                const uint32_t wgt = ((inm >> r) & 1u) ? ((uint32_t)v[r] & WMASK) + 1u : 0u;
                wtot += wgt;
                if ((uint32_t)r <= lastbit) wlast += wgt;
            }
        }
        uint32_t x = nt, tw = wtot;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d, 64), u = __shfl_up(tw, d, 64); if (lane >= d) { x += y; tw += u; } }
        if (lane == 63) { s_tails[w] = x; s_wsum[w] = tw; }
        __syncthreads();
        uint32_t idx = x - nt, wbase = tw - wtot;
        for (int ww = 0; ww < w; ww++) { idx += s_tails[ww]; wbase += s_wsum[ww]; }
        uint32_t lt = tailm ? wbase + wlast : 0u;                  // weight up to the lane's last run end; carried by a max-scan (monotone along the ranks)
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(lt, d, 64); if (lane >= d) lt = y > lt ? y : lt; }
        if (lane == 63) s_lt[w] = lt;
        __syncthreads();
        uint32_t prevw = __shfl_up(lt, 1, 64); if (lane == 0) prevw = 0;
        for (int ww = 0; ww < w; ww++) prevw = s_lt[ww] > prevw ? s_lt[ww] : prevw;
        uint32_t nsol = 0, run = wbase;
#pragma unroll
        for (int r = 0; r < KPL; r++) {
            run += ((inm >> r) & 1u) ? ((uint32_t)v[r] & WMASK) + 1u : 0u;
            if ((tailm >> r) & 1) {
                const uint32_t c = run - prevw;
                prevw = run;
                if constexpr (F) outk[start + idx] = ((v[r] & KTag<KW>::mant()) | top) >> WB; else outk[start + idx] = v[r] >> WB;
                put_count(O, start + idx, c); idx++;
                nsol += ((int32_t)c >= O.amin && (int32_t)c <= O.amax) ? 1u : 0u;
                const uint32_t hb = c >= O.histo_max ? O.histo_max : c;
                if (hb < HIST_LDS) atomicAdd(&s_hc[hb], 1u); else atomicAdd(&O.histo[hb], 1ULL);
            }
        }
        // distinct / solid k-mers of the bucket: the tails of all waves
        if (!O.all_solid) {
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) nsol += __shfl_xor(nsol, d, 64);
            if (lane == 0) s_sol[w] = nsol;
            __syncthreads();
        }
        if (t == 0) {
            uint32_t nd = 0, ns = 0;
            for (int ww = 0; ww < NW; ww++) { nd += s_tails[ww]; if (!O.all_solid) ns += s_sol[ww]; }
            O.nd[g] = nd; if (!O.all_solid) O.ns[g] = ns;
        }
    }
    __syncthreads();
    if (t < HIST_LDS && s_hc[t]) atomicAdd(&O.histo[t], (unsigned long long)s_hc[t]);
}

// ------------------------------------------------------------------------------------------------ deeper levels
// A sub-bucket too large for the sort tiers (skewed key ranges: k-mers that START with their minimizer share their top 2m bits; repeats; too few
// partitions) is split again on its next INFORMATIVE key bits — the leading bits every key of it shares are skipped — by one workgroup, keys -> keys
// into the other key buffer. Pieces that fit a wave are appended to ONE list that a single launch sorts at the end (all pieces of all levels, chip-wide:
// a root of 10^6 keys has thousands of pieces), pieces that are still too large go to the next level's queue. When no bit is left all keys are one
// k-mer: one record with abundance n. Everything is decided on the device: the levels are launched back to back, a level whose queue is empty returns
// at once (round 2 fetched the lists to the host between the levels: ~1 ms of latency per level and batch for 1.3 % of the keys). Results land in the
// primary key buffer at the piece's own slots; the root's nd / ns counters collect them.
struct SplitPlan { uint32_t left, bits, shift; };
__device__ __forceinline__ SplitPlan split_plan(unsigned long long or_lo, unsigned long long or_hi, uint32_t consumed, uint32_t key_bits /* 2k + weight bits */, uint32_t max_bits, uint32_t wb)
{
    const uint32_t diff_bits = or_hi ? 128 - __clzll((long long)or_hi) : (or_lo ? 64 - __clzll((long long)or_lo) : 0);   // number of low bits that may differ between keys
    const uint32_t have = key_bits - consumed;
    const uint32_t span = diff_bits < have ? diff_bits : have;              // low bits still unused and not shared by all keys; the lowest wb are not k-mer bits
    SplitPlan p; p.left = span > wb ? span - wb : 0u; p.bits = 0; p.shift = 0;   // informative k-mer bits (0: all keys are one k-mer)
    if (p.left == 0) return p;
    // always as many bits as the tables hold: the pieces that come out small are listed in runs (see k_deep_split), and a cluster under a longer shared prefix spreads
    p.bits = min(max_bits, p.left);
    p.shift = span - p.bits;                                                // >= wb: equal k-mers stay together whatever their weights
    return p;
}
struct DeepItem { uint64_t start; uint32_t n; uint32_t root; uint32_t consumed; uint32_t buf; };      // buf: 0 = keys are in the primary buffer, 1 = in the ping-pong buffer
struct SortItem { uint64_t start; uint32_t n_buf; uint32_t root; };                                     // n_buf: n | buf << 31
constexpr int DEEP_THREADS = 256, DEEP_MLP = 8, DEEP_WINDOWS = 1024;
// Two launches per level share the items by size: items up to DEEP_SMALL_N keys are cut on DEEP_SMALL_BITS bits (16 KB of LDS: 8 workgroups per CU, the items are
// latency-bound: ~10 barriers and ~6 dependent memory round trips each), larger ones on MAX_SUB_BITS bits (72 KB: 2 per CU) so that a cluster of 10^4 keys under a
// 7-bit longer prefix still comes out in pieces a wave can sort instead of coming back at the next level.
constexpr uint32_t DEEP_SMALL_BITS = 10, DEEP_SMALL_N = 8192;
template <int KW>
__global__ __launch_bounds__(DEEP_THREADS) void k_deep_split(typename KeyT<KW>::type* keysA, typename KeyT<KW>::type* keysB,
                                                              const uint32_t* __restrict__ root_list /* level 1: the split list of k_expand_count; else nullptr */,
                                                              const uint64_t* __restrict__ b_start, const uint32_t* __restrict__ b_n, const uint8_t* __restrict__ b_cons,
                                                              const DeepItem* __restrict__ q_in, const uint32_t* __restrict__ n_in_p, uint32_t* __restrict__ ticket,
                                                              DeepItem* __restrict__ q_out, uint32_t* __restrict__ n_out_p,
                                                              SortItem* __restrict__ sort_list, uint32_t* __restrict__ n_sort_p,
                                                              uint32_t two_k, uint32_t max_bits /* <= log2 of the LDS tables */, uint32_t n_lo, uint32_t n_hi /* this launch: items with n_lo < n <= n_hi */,
                                                              uint32_t cap1, SortOut O)
{
    typedef typename KeyT<KW>::type key_t;
    extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];
    uint32_t* s_cnt = s_dyn;                                // [1 << max_bits] keys of every piece
    uint32_t* s_cur = s_dyn + (1u << max_bits);             // [1 << max_bits] first slot of the piece, then its scatter cursor (relative to the item)
    __shared__ uint32_t s_gs[DEEP_WINDOWS], s_ge[DEEP_WINDOWS];    // runs of small pieces by window of their first slot: first slot, one past the last
    __shared__ uint32_t s_wsum[DEEP_THREADS / 64];
    __shared__ unsigned long long s_or[3];                  // OR of (key ^ first key), low / high word; sum of the weights
    __shared__ uint32_t s_item, s_base_sort, s_base_q;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t n_items = *n_in_p;
    for (;;) {
        __syncthreads();                                      // LDS of the previous item fully consumed
        if (t == 0) s_item = atomicAdd(ticket, 1u);           // items differ by orders of magnitude: dynamic hand-out
        if (t < 3) s_or[t] = 0;
        __syncthreads();
        const uint32_t it = s_item;
        if (it >= n_items) break;
        DeepItem d;
        if (root_list) {
            const uint32_t g = root_list[it]; const uint32_t cb = b_cons[g];
            if (cb & CONS_GIANT) continue;                    // a giant: split by many workgroups (k_giant_*)
            d.start = b_start[g]; d.n = b_n[g]; d.root = g; d.consumed = cb & CONS_BITS_MASK; d.buf = 0;
        } else d = q_in[it];
        if (d.n <= n_lo || d.n > n_hi) continue;              // the other launch's item
        const key_t* src = (d.buf ? keysB : keysA) + d.start;
        key_t* dst = (d.buf ? keysA : keysB) + d.start;
        // the root's abundance plane is read slot by slot by the gather (its records are not at the head of its range): clear it first
        if (root_list) {
            uint8_t* z = O.cnt8 + d.start;                    // d.start is a multiple of 4
            for (uint32_t i = t; i < d.n / 4; i += DEEP_THREADS) reinterpret_cast<uint32_t*>(z)[i] = 0u;
            if (t < (d.n & 3u)) z[(d.n & ~3u) + t] = 0;
        }
        // leading bits shared by every key of the item carry no information: skip them
        {
            const key_t k0 = src[0];
            key_t acc = 0;
            unsigned long long wsum = 0;                           // the item's total weight (its abundance if it turns out to be one k-mer)
            uint32_t i = t;
            for (; i + (DEEP_MLP - 1) * DEEP_THREADS < d.n; i += DEEP_MLP * DEEP_THREADS) {       // a root of 10^6 keys is walked by ONE workgroup: keep DEEP_MLP loads in flight
                key_t v[DEEP_MLP];
#pragma unroll
                for (int u = 0; u < DEEP_MLP; u++) v[u] = src[i + u * DEEP_THREADS];
#pragma unroll
                for (int u = 0; u < DEEP_MLP; u++) { acc |= v[u] ^ k0; wsum += ((uint32_t)v[u] & ((1u << O.wb) - 1u)) + 1u; }
            }
            for (; i < d.n; i += DEEP_THREADS) { const key_t v1 = src[i]; acc |= v1 ^ k0; wsum += ((uint32_t)v1 & ((1u << O.wb) - 1u)) + 1u; }
            unsigned long long lo = (unsigned long long)acc, hi = (unsigned long long)((u128)acc >> 64);
#pragma unroll
            for (int dd = 32; dd >= 1; dd >>= 1) { lo |= __shfl_down(lo, dd, 64); hi |= __shfl_down(hi, dd, 64); wsum += __shfl_down(wsum, dd, 64); }
            if (lane == 0) { if (lo) atomicOr(&s_or[0], lo); if (KW == 2 && hi) atomicOr(&s_or[1], hi); atomicAdd(&s_or[2], wsum); }
        }
        __syncthreads();
        const SplitPlan P = split_plan(s_or[0], s_or[1], d.consumed, two_k, max_bits, O.wb);
        const uint32_t left = P.left;
        if (left == 0) {                                       // one k-mer, abundance n (CountNumber is int32)
            if (t == 0) {
                keysA[d.start] = src[0] >> O.wb;
                const uint32_t c = s_or[2] > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)s_or[2];
                put_count(O, d.start, c);
                atomicAdd(&O.histo[c >= O.histo_max ? O.histo_max : c], 1ULL);
                atomicAdd(&O.nd[d.root], 1u);
                if (!O.all_solid && (int32_t)c >= O.amin && (int32_t)c <= O.amax) atomicAdd(&O.ns[d.root], 1u);
            }
            continue;
        }
        const uint32_t bits = P.bits, shift = P.shift, nsub = 1u << bits, mask = nsub - 1u;
        const uint32_t cons_child = two_k - shift;
        for (uint32_t i = t; i < nsub; i += DEEP_THREADS) s_cnt[i] = 0;
        __syncthreads();
        {   uint32_t i = t;
            for (; i + (DEEP_MLP - 1) * DEEP_THREADS < d.n; i += DEEP_MLP * DEEP_THREADS) {
                key_t v[DEEP_MLP];
#pragma unroll
                for (int u = 0; u < DEEP_MLP; u++) v[u] = src[i + u * DEEP_THREADS];
#pragma unroll
                for (int u = 0; u < DEEP_MLP; u++) wave_add1(s_cnt, (uint32_t)(v[u] >> shift) & mask);
            }
            for (; i < d.n; i += DEEP_THREADS) wave_add1(s_cnt, (uint32_t)(src[i] >> shift) & mask);
        }
        __syncthreads();
        // exclusive scan of the piece sizes -> first slot of every piece
        const uint32_t per = (nsub + DEEP_THREADS - 1) / DEEP_THREADS, b = t * per;
        uint32_t loc = 0;
        for (uint32_t i = 0; i < per; i++) if (b + i < nsub) loc += s_cnt[b + i];
        uint32_t x = loc;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) { const uint32_t xx = __shfl_up(x, dd, 64); if (lane >= dd) x += xx; }
        if (lane == 63) s_wsum[wave] = x;
        __syncthreads();
        uint32_t run = x - loc;
        for (int w = 0; w < wave; w++) run += s_wsum[w];
        for (uint32_t i = 0; i < per; i++) if (b + i < nsub) { s_cur[b + i] = run; run += s_cnt[b + i]; }
        // What is listed for the sort is not the single piece: the item is always cut on as many bits as the tables hold (a cluster under a longer shared
        // prefix then still spreads over several pieces instead of coming back whole at the next level), and pieces of at most M = cap1 / 2 keys whose first
        // slot lies in the same window of M slots are listed as ONE run of < cap1 keys (consecutive pieces are consecutive key ranges and consecutive in
        // memory) — the sparse background of a clustered item becomes a few full sorts instead of hundreds of tiny ones.
        const uint32_t M = cap1 / 2, nwin = d.n / M + 1;
        const bool merge = nwin <= (uint32_t)DEEP_WINDOWS;
        if (merge) for (uint32_t i = t; i < nwin; i += DEEP_THREADS) { s_gs[i] = 0xFFFFFFFFu; s_ge[i] = 0u; }
        __syncthreads();
        uint32_t cls = 0;                                      // items this thread lists: sort list (low half) / next level (high half)
        for (uint32_t i = 0; i < per; i++) if (b + i < nsub) {
            const uint32_t v = s_cnt[b + i];
            if (!v) continue;
            if (v > cap1) cls += 0x10000u;
            else if (v > M) cls += 1u;
            else if (merge) { const uint32_t e = s_cur[b + i], w = e / M; atomicMin(&s_gs[w], e); atomicMax(&s_ge[w], e + v); }
        }
        __syncthreads();
        if (merge) for (uint32_t w = t; w < nwin; w += DEEP_THREADS) if (s_ge[w]) cls += 1u;
        uint32_t y = cls;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) { const uint32_t yy = __shfl_up(y, dd, 64); if (lane >= dd) y += yy; }
        if (lane == 63) s_wsum[wave] = y;
        __syncthreads();
        uint32_t ypre = y - cls, ytot = 0;
        for (int w = 0; w < DEEP_THREADS / 64; w++) { if (w < wave) ypre += s_wsum[w]; ytot += s_wsum[w]; }
        if (t == 0) {                                          // one reservation per item and list
            s_base_sort = (ytot & 0xFFFFu) ? atomicAdd(n_sort_p, ytot & 0xFFFFu) : 0u;
            s_base_q = (ytot >> 16) ? atomicAdd(n_out_p, ytot >> 16) : 0u;
        }
        __syncthreads();
        {   uint32_t is = s_base_sort + (ypre & 0xFFFFu), iq = s_base_q + (ypre >> 16);
            const uint32_t buf_bit = (d.buf ^ 1u) << 31;
            for (uint32_t i = 0; i < per; i++) if (b + i < nsub) {
                const uint32_t v = s_cnt[b + i];
                if (!v) continue;
                const uint64_t st = d.start + s_cur[b + i];
                if (v > cap1) { DeepItem c; c.start = st; c.n = v; c.root = d.root; c.consumed = cons_child; c.buf = d.buf ^ 1u; q_out[iq++] = c; }
                else if (v > M) { SortItem si; si.start = st; si.n_buf = v | buf_bit; si.root = d.root; sort_list[is++] = si; }
            }
            if (merge) for (uint32_t w = t; w < nwin; w += DEEP_THREADS) if (s_ge[w]) {
                SortItem si; si.start = d.start + s_gs[w]; si.n_buf = (s_ge[w] - s_gs[w]) | buf_bit; si.root = d.root; sort_list[is++] = si;
            }
        }
        // an item of more windows than the tables hold (beyond DEEP_WINDOWS * M keys: a root the giant path did not take, or a huge piece of one): the same
        // runs, DEEP_WINDOWS windows at a time, each round with its own reservation — every item lists <= 2 n / M + 1 entries whatever its size (the list's
        // capacity is sized for that: sort_cap in count_batch)
        if (!merge) for (uint32_t w0 = 0; w0 < nwin; w0 += DEEP_WINDOWS) {
            const uint32_t nw = min((uint32_t)DEEP_WINDOWS, nwin - w0);
            __syncthreads();
            for (uint32_t i = t; i < nw; i += DEEP_THREADS) { s_gs[i] = 0xFFFFFFFFu; s_ge[i] = 0u; }
            __syncthreads();
            for (uint32_t i = 0; i < per; i++) if (b + i < nsub) {
                const uint32_t v = s_cnt[b + i];
                if (!v || v > M) continue;
                const uint32_t e = s_cur[b + i], w = e / M;
                if (w >= w0 && w - w0 < nw) { atomicMin(&s_gs[w - w0], e); atomicMax(&s_ge[w - w0], e + v); }
            }
            __syncthreads();
            uint32_t c2 = 0;
            for (uint32_t w = t; w < nw; w += DEEP_THREADS) if (s_ge[w]) c2++;
            uint32_t y2 = c2;
#pragma unroll
            for (int dd = 1; dd < 64; dd <<= 1) { const uint32_t yy = __shfl_up(y2, dd, 64); if (lane >= dd) y2 += yy; }
            if (lane == 63) s_wsum[wave] = y2;
            __syncthreads();
            uint32_t pre2 = y2 - c2, tot2 = 0;
            for (int w = 0; w < DEEP_THREADS / 64; w++) { if (w < wave) pre2 += s_wsum[w]; tot2 += s_wsum[w]; }
            if (t == 0) s_base_sort = tot2 ? atomicAdd(n_sort_p, tot2) : 0u;
            __syncthreads();
            uint32_t is = s_base_sort + pre2;
            const uint32_t buf_bit = (d.buf ^ 1u) << 31;
            for (uint32_t w = t; w < nw; w += DEEP_THREADS) if (s_ge[w]) {
                SortItem si; si.start = d.start + s_gs[w]; si.n_buf = (s_ge[w] - s_gs[w]) | buf_bit; si.root = d.root; sort_list[is++] = si;
            }
        }
        __syncthreads();                                      // the first slots in s_cur have been read: they now become the scatter's cursors
        {   uint32_t i = t;
            for (; i + (DEEP_MLP - 1) * DEEP_THREADS < d.n; i += DEEP_MLP * DEEP_THREADS) {
                key_t v[DEEP_MLP];
#pragma unroll
                for (int u = 0; u < DEEP_MLP; u++) v[u] = src[i + u * DEEP_THREADS];
#pragma unroll
                for (int u = 0; u < DEEP_MLP; u++) { const uint32_t slot = wave_take1(s_cur, (uint32_t)(v[u] >> shift) & mask); dst[slot] = v[u]; }
            }
            for (; i < d.n; i += DEEP_THREADS) { const key_t key = src[i]; const uint32_t slot = wave_take1(s_cur, (uint32_t)(key >> shift) & mask); dst[slot] = key; }
        }
    }
}
// ---- giants: a root beyond GIANT_MIN keys (the k-mers that START with the partition's hottest minimizer: 10^5 .. 10^7 keys under one 20-bit prefix) would keep
// ONE workgroup busy for milliseconds while the chip idles. Its first split is therefore done by GIANT_WGS workgroups together, in chunks of GIANT_CHUNK keys:
// OR of the keys (informative bits) -> histogram of the chunk in LDS, added to the giant's global histogram -> one workgroup scans it into piece offsets and lists
// the pieces -> every key takes its slot from the piece's global cursor. The pieces then go the way of all pieces (sort list / next level's queue).
constexpr int GIANT_THREADS = 1024, GIANT_MLP = GIANT_CHUNK / GIANT_THREADS;
struct GiantTables { unsigned long long* gor; uint32_t* ghist; uint32_t* gcur; const uint32_t* list; const uint32_t* count; };   // [MAX][4] (OR low, OR high, total weight), [MAX][MAX_SUB], [MAX][MAX_SUB]
template <int KW>
__global__ __launch_bounds__(GIANT_THREADS) void k_giant_or(const typename KeyT<KW>::type* __restrict__ keysA, GiantTables G, const uint64_t* __restrict__ b_start, const uint32_t* __restrict__ b_n,
                                                             uint8_t* __restrict__ cnt8, uint32_t wb)
{
    typedef typename KeyT<KW>::type key_t;
    const uint32_t y = blockIdx.y;
    if (y >= min(*G.count, GIANT_MAX)) return;
    const uint32_t g = G.list[y];
    const uint64_t start = b_start[g]; const uint32_t n = b_n[g];
    const key_t* src = keysA + start;
    const key_t k0 = src[0];
    key_t acc = 0;
    unsigned long long wsum = 0;
    for (uint32_t c0 = blockIdx.x * GIANT_CHUNK; c0 < n; c0 += gridDim.x * GIANT_CHUNK) {
        key_t v[GIANT_MLP];
#pragma unroll
        for (int u = 0; u < GIANT_MLP; u++) { const uint32_t i = c0 + u * GIANT_THREADS + threadIdx.x; v[u] = i < n ? src[i] : k0; }
#pragma unroll
        for (int u = 0; u < GIANT_MLP; u++) { acc |= v[u] ^ k0; if (c0 + u * GIANT_THREADS + threadIdx.x < n) wsum += ((uint32_t)v[u] & ((1u << wb) - 1u)) + 1u; }
        // the root's abundance plane is read slot by slot by the gather: clear it (start is a multiple of 4)
        for (uint32_t i = threadIdx.x; i < GIANT_CHUNK / 4; i += GIANT_THREADS) if (c0 + 4 * i < n) {
            if (c0 + 4 * i + 4 <= n) reinterpret_cast<uint32_t*>(cnt8 + start + c0)[i] = 0u;
            else for (uint32_t q = c0 + 4 * i; q < n; q++) cnt8[start + q] = 0;
        }
    }
    unsigned long long lo = (unsigned long long)acc, hi = (unsigned long long)((u128)acc >> 64);
#pragma unroll
    for (int dd = 32; dd >= 1; dd >>= 1) { lo |= __shfl_down(lo, dd, 64); hi |= __shfl_down(hi, dd, 64); wsum += __shfl_down(wsum, dd, 64); }
    if ((threadIdx.x & 63) == 0) { if (lo) atomicOr(&G.gor[4 * y], lo); if (KW == 2 && hi) atomicOr(&G.gor[4 * y + 1], hi); if (wsum) atomicAdd(&G.gor[4 * y + 2], wsum); }
}
template <int KW>
__global__ __launch_bounds__(GIANT_THREADS) void k_giant_hist(const typename KeyT<KW>::type* __restrict__ keysA, GiantTables G, const uint64_t* __restrict__ b_start, const uint32_t* __restrict__ b_n,
                                                               const uint8_t* __restrict__ b_cons, uint32_t two_k, uint32_t max_bits, uint32_t wb)
{
    typedef typename KeyT<KW>::type key_t;
    __shared__ uint32_t s_cnt[MAX_SUB];
    const uint32_t y = blockIdx.y;
    if (y >= min(*G.count, GIANT_MAX)) return;
    const uint32_t g = G.list[y];
    const uint64_t start = b_start[g]; const uint32_t n = b_n[g];
    if ((uint64_t)blockIdx.x * GIANT_CHUNK >= n) return;
    const SplitPlan P = split_plan(G.gor[4 * y], G.gor[4 * y + 1], b_cons[g] & CONS_BITS_MASK, two_k, min(max_bits, (uint32_t)MAX_SUB_BITS), wb);
    if (P.left == 0) return;
    const uint32_t nsub = 1u << P.bits, mask = nsub - 1u;
    for (uint32_t i = threadIdx.x; i < nsub; i += GIANT_THREADS) s_cnt[i] = 0;
    __syncthreads();
    const key_t* src = keysA + start;
    for (uint32_t c0 = blockIdx.x * GIANT_CHUNK; c0 < n; c0 += gridDim.x * GIANT_CHUNK) {
        key_t v[GIANT_MLP];
#pragma unroll
        for (int u = 0; u < GIANT_MLP; u++) { const uint32_t i = c0 + u * GIANT_THREADS + threadIdx.x; v[u] = i < n ? src[i] : (key_t)0; }
#pragma unroll
        for (int u = 0; u < GIANT_MLP; u++) if (c0 + u * GIANT_THREADS + threadIdx.x < n) wave_add1(s_cnt, (uint32_t)(v[u] >> P.shift) & mask);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nsub; i += GIANT_THREADS) { const uint32_t h = s_cnt[i]; if (h) atomicAdd(&G.ghist[(size_t)y * MAX_SUB + i], h); }
}
// one workgroup per giant: histogram -> piece offsets (the scatter's cursors) and the piece lists
template <int KW>
__global__ __launch_bounds__(GIANT_THREADS) void k_giant_plan(typename KeyT<KW>::type* __restrict__ keysA, GiantTables G, const uint64_t* __restrict__ b_start, const uint32_t* __restrict__ b_n,
                                                               const uint8_t* __restrict__ b_cons, DeepItem* __restrict__ q_out, uint32_t* __restrict__ n_out_p,
                                                               SortItem* __restrict__ sort_list, uint32_t* __restrict__ n_sort_p,
                                                               uint32_t two_k, uint32_t max_bits, uint32_t cap1, SortOut O)
{
    __shared__ uint32_t s_wsum[GIANT_THREADS / 64];
    __shared__ uint32_t s_base_sort, s_base_q;
    const uint32_t y = blockIdx.x;
    if (y >= min(*G.count, GIANT_MAX)) return;
    const uint32_t g = G.list[y];
    const uint64_t start = b_start[g];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const SplitPlan P = split_plan(G.gor[4 * y], G.gor[4 * y + 1], b_cons[g] & CONS_BITS_MASK, two_k, min(max_bits, (uint32_t)MAX_SUB_BITS), O.wb);
    if (P.left == 0) {                                         // one k-mer, abundance = the total weight (CountNumber is int32)
        if (t == 0) {
            const unsigned long long wsum = G.gor[4 * y + 2];
            const uint32_t c = wsum > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)wsum;
            keysA[start] = keysA[start] >> O.wb;                // the k-mer without its weight bits, where the gather expects it
            put_count(O, start, c);
            atomicAdd(&O.histo[c >= O.histo_max ? O.histo_max : c], 1ULL);
            atomicAdd(&O.nd[g], 1u);
            if (!O.all_solid && (int32_t)c >= O.amin && (int32_t)c <= O.amax) atomicAdd(&O.ns[g], 1u);
        }
        return;
    }
    const uint32_t nsub = 1u << P.bits, cons_child = two_k - P.shift;
    const uint32_t* hist = G.ghist + (size_t)y * MAX_SUB; uint32_t* cur = G.gcur + (size_t)y * MAX_SUB;
    const uint32_t per = (nsub + GIANT_THREADS - 1) / GIANT_THREADS, b = t * per;
    uint32_t loc = 0;
    for (uint32_t i = 0; i < per; i++) if (b + i < nsub) loc += hist[b + i];
    uint32_t x = loc;
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) { const uint32_t xx = __shfl_up(x, dd, 64); if (lane >= dd) x += xx; }
    if (lane == 63) s_wsum[wave] = x;
    __syncthreads();
    uint32_t run0 = x - loc;
    for (int w = 0; w < wave; w++) run0 += s_wsum[w];
    // the thread's consecutive pieces are listed in runs of <= cap1 keys (consecutive pieces are consecutive key ranges and consecutive in memory); a piece beyond
    // cap1 goes to the next level. walk(emit): the same walk counts (emit = false) and writes (emit = true)
    uint32_t is = 0, iq = 0;
    auto walk = [&](bool emit) -> uint32_t {
        uint32_t cls = 0, run = run0, rs = 0, rn = 0;
        auto flush = [&]() { if (rn) { if (emit) { SortItem si; si.start = start + rs; si.n_buf = rn | (1u << 31); si.root = g; sort_list[is++] = si; } cls += 1u; rn = 0; } };   // (the pieces are in the ping-pong buffer)
        for (uint32_t i = 0; i < per; i++) if (b + i < nsub) {
            const uint32_t sn = hist[b + i];
            if (emit) cur[b + i] = run;
            if (sn > cap1) {
                flush();
                if (emit) { DeepItem c; c.start = start + run; c.n = sn; c.root = g; c.consumed = cons_child; c.buf = 1u; q_out[iq++] = c; }
                cls += 0x10000u;
            } else if (sn) {
                if (rn + sn > cap1) flush();
                if (!rn) rs = run;
                rn += sn;
            }
            run += sn;
        }
        flush();
        return cls;
    };
    const uint32_t cls = walk(false);
    uint32_t yv = cls;
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) { const uint32_t yy = __shfl_up(yv, dd, 64); if (lane >= dd) yv += yy; }
    __syncthreads();
    if (lane == 63) s_wsum[wave] = yv;
    __syncthreads();
    uint32_t ypre = yv - cls, ytot = 0;
    for (int w = 0; w < GIANT_THREADS / 64; w++) { if (w < wave) ypre += s_wsum[w]; ytot += s_wsum[w]; }
    if (t == 0) {
        s_base_sort = (ytot & 0xFFFFu) ? atomicAdd(n_sort_p, ytot & 0xFFFFu) : 0u;
        s_base_q = (ytot >> 16) ? atomicAdd(n_out_p, ytot >> 16) : 0u;
    }
    __syncthreads();
    is = s_base_sort + (ypre & 0xFFFFu); iq = s_base_q + (ypre >> 16);
    (void)walk(true);
}
template <int KW>
__global__ __launch_bounds__(GIANT_THREADS) void k_giant_scatter(const typename KeyT<KW>::type* __restrict__ keysA, typename KeyT<KW>::type* __restrict__ keysB, GiantTables G,
                                                                  const uint64_t* __restrict__ b_start, const uint32_t* __restrict__ b_n, const uint8_t* __restrict__ b_cons,
                                                                  uint32_t two_k, uint32_t max_bits, uint32_t wb)
{
    typedef typename KeyT<KW>::type key_t;
    const uint32_t y = blockIdx.y;
    if (y >= min(*G.count, GIANT_MAX)) return;
    const uint32_t g = G.list[y];
    const uint64_t start = b_start[g]; const uint32_t n = b_n[g];
    if ((uint64_t)blockIdx.x * GIANT_CHUNK >= n) return;
    const SplitPlan P = split_plan(G.gor[4 * y], G.gor[4 * y + 1], b_cons[g] & CONS_BITS_MASK, two_k, min(max_bits, (uint32_t)MAX_SUB_BITS), wb);
    if (P.left == 0) return;
    const uint32_t mask = (1u << P.bits) - 1u;
    const key_t* src = keysA + start; key_t* dst = keysB + start;
    uint32_t* cur = G.gcur + (size_t)y * MAX_SUB;
    for (uint32_t c0 = blockIdx.x * GIANT_CHUNK; c0 < n; c0 += gridDim.x * GIANT_CHUNK) {
        key_t v[GIANT_MLP];
#pragma unroll
        for (int u = 0; u < GIANT_MLP; u++) { const uint32_t i = c0 + u * GIANT_THREADS + threadIdx.x; v[u] = i < n ? src[i] : (key_t)0; }
#pragma unroll
        for (int u = 0; u < GIANT_MLP; u++) if (c0 + u * GIANT_THREADS + threadIdx.x < n) { const uint32_t slot = wave_take1(cur, (uint32_t)(v[u] >> P.shift) & mask); dst[slot] = v[u]; }
    }
}

// the pieces the split levels listed: one wave each, like k_wave_sort; results at the piece's own slots of the primary buffer
template <int KW, bool F>
__global__ __launch_bounds__(SORT_THREADS, GKC_WS_WAVES) void k_sort_items(typename KeyT<KW>::type* keysA, const typename KeyT<KW>::type* keysB,
                                                                            const SortItem* __restrict__ list, const uint32_t* __restrict__ n_list_p, uint32_t first, SortOut O)
{
    __shared__ uint32_t s_hc[HIST_LDS];
    const int t = threadIdx.x, lane = t & 63;
    if (t < HIST_LDS) s_hc[t] = 0;
    __syncthreads();
    const uint32_t n_list = *n_list_p;
    const uint32_t wave = (blockIdx.x * SORT_THREADS + t) >> 6, n_waves = (gridDim.x * SORT_THREADS) >> 6;
    for (uint32_t li = first + wave; li < n_list; li += n_waves) {
        const SortItem si = list[li];
        const uint32_t n = si.n_buf & 0x7FFFFFFFu;
        const typename KeyT<KW>::type* src = ((si.n_buf >> 31) ? keysB : (const typename KeyT<KW>::type*)keysA) + si.start;
        uint32_t nd, ns;
        wave_sort_dispatch<KW, WaveCapT1<KW>::KPL_MAX, F>(src, keysA, si.start, n, O, s_hc, lane, nd, ns);
        if (lane == 0) { atomicAdd(&O.nd[si.root], nd); if (!O.all_solid && ns) atomicAdd(&O.ns[si.root], ns); }
    }
    __syncthreads();
    if (t < HIST_LDS && s_hc[t]) atomicAdd(&O.histo[t], (unsigned long long)s_hc[t]);
}

// ------------------------------------------------------------------------------------------------ B5 dump: per-bucket counts -> prefix -> Count records
// The sort tiers leave every sub-bucket's distinct k-mers at the head of its own slot range (ascending, abundance in the byte plane at the same slot) and
// its number of distinct / solid k-mers in nd[] / ns[]. Sub-bucket order is key order, so an exclusive prefix over ns[] is the position of every
// sub-bucket's first record in the partition-major, ascending result array: one scan of 4 bytes per SUB-BUCKET instead of two passes over one byte per
// SLOT (round 2: block sums + plane scan, 24 GB per 1.2e10 keys, plus the memset of the plane), then one gather.
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
// (nd, ns) u32[n] -> exclusive prefixes (off_d, off_s) u64[n + 1] in three launches: every workgroup scans its own chunk of 8192 entries and leaves the chunk
// totals, one workgroup scans the totals, a third pass adds them back
__global__ __launch_bounds__(1024) void k_scan2_chunks(const uint32_t* __restrict__ nd, const uint32_t* __restrict__ ns, uint64_t n, uint64_t* __restrict__ a, uint64_t* __restrict__ b,
                                                       uint64_t* __restrict__ ca, uint64_t* __restrict__ cb)
{
    __shared__ uint64_t s_a[16], s_b[16];
    const uint64_t i0 = (uint64_t)blockIdx.x * SCAN2_CHUNK + (uint64_t)threadIdx.x * SCAN2_ITEMS;
    uint64_t va[SCAN2_ITEMS], vb[SCAN2_ITEMS], ta = 0, tb = 0;
#pragma unroll
    for (int j = 0; j < SCAN2_ITEMS; j++) { va[j] = i0 + j < n ? nd[i0 + j] : 0; vb[j] = i0 + j < n ? ns[i0 + j] : 0; ta += va[j]; tb += vb[j]; }
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
// Count records {value, abundance} (Abundance.hpp:68-129), solid only, ascending: one WAVE per group of 64 consecutive sub-buckets. The head slots of the
// group's sub-buckets (a sub-bucket that was split: all of its slots, the records sit at the pieces' heads) are walked as ONE flat sequence — lane <-> flat
// index, the owning sub-bucket found by a 6-step search over the group's span prefix in LDS — so every lane has independent loads in flight whatever the
// sub-bucket sizes; the solid entries leave, in flat order, from the group's first output position on (consecutive groups write consecutive output).
constexpr int GATHER_THREADS = 256, GATHER_UNROLL = 4;
template <int KW>
__global__ __launch_bounds__(GATHER_THREADS) void k_gather_counts(const typename KeyT<KW>::type* __restrict__ keys, const uint8_t* __restrict__ cnt8, const uint32_t* __restrict__ cnt32,
                                                                   const uint64_t* __restrict__ b_start, const uint32_t* __restrict__ b_n, const uint32_t* __restrict__ nd,
                                                                   const uint64_t* __restrict__ off_s, uint32_t n_buckets, uint32_t split_min /* sub-buckets beyond were split */,
                                                                   int32_t amin, int32_t amax, uint32_t all_solid, uint64_t* __restrict__ out,
                                                                   const uint8_t* __restrict__ b_cons, uint32_t top_shift /* where a sub-bucket's dropped top bits go back: 2k - their number */)
{
    constexpr int OW = (KW == 1) ? 2 : 4;
    typedef typename KeyT<KW>::type key_t;
    constexpr uint64_t START_MASK = (1ULL << 62) - 1ULL;          // (slot numbers are far below 2^62: the dropped bit rides on top of the sub-bucket's first slot in LDS)
    __shared__ uint32_t s_pre[GATHER_THREADS / 64][64], s_skip[GATHER_THREADS / 64][64];
    __shared__ uint64_t s_start[GATHER_THREADS / 64][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t wave = (blockIdx.x * GATHER_THREADS + threadIdx.x) >> 6, n_waves = (gridDim.x * GATHER_THREADS) >> 6;
    const uint64_t lt_mask = (1ULL << lane) - 1ULL;
    uint32_t* pre = s_pre[wv]; uint32_t* skp = s_skip[wv]; uint64_t* sta = s_start[wv];
    for (uint64_t g0 = (uint64_t)wave * 64; g0 < n_buckets; g0 += (uint64_t)n_waves * 64) {
        const uint64_t g = g0 + lane;
        const bool in = g < n_buckets;
        const uint32_t my_nd = in ? nd[g] : 0u;
        const uint32_t my_n = in ? b_n[g] : 0u;
        const bool root = my_n > split_min;                     // its records are written by k_gather_roots; here only its output range is stepped over
        const uint32_t span = root ? 0u : my_nd;
        const uint32_t rs = (root && my_nd) ? (uint32_t)(off_s[g + 1] - off_s[g]) : 0u;
        uint32_t x = span, z = rs;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d, 64), w = __shfl_up(z, d, 64); if (lane >= d) { x += y; z += w; } }
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
        if (total == 0) continue;
        pre[lane] = x - span;
        skp[lane] = z - rs;
        sta[lane] = in ? (b_start[g] | ((uint64_t)((b_cons[g] >> CONS_DROP_SHIFT) & 3u) << 62)) : 0ull;
        uint64_t o = off_s[g0];                                 // first output record of the group
        // (the wave's own LDS writes are visible to its later reads: same wave, in order)
        for (uint32_t s0 = 0; s0 < total; s0 += 64 * GATHER_UNROLL) {
            uint64_t slot[GATHER_UNROLL]; uint32_t b8[GATHER_UNROLL], sk[GATHER_UNROLL], tp[GATHER_UNROLL];
            typename KeyT<KW>::type kk[GATHER_UNROLL];                                  // the keys are fetched with the abundance bytes, not behind them: all loads of a step in flight at once
#pragma unroll
            for (int u = 0; u < GATHER_UNROLL; u++) {
                const uint32_t s = s0 + u * 64 + lane;
                b8[u] = 0; slot[u] = 0; sk[u] = 0; kk[u] = 0; tp[u] = 0;
                if (s < total) {
                    uint32_t lo = 0;
#pragma unroll
                    for (int st = 32; st >= 1; st >>= 1) if (pre[lo + st] <= s) lo += st;     // largest b with pre[b] <= s (lo + st <= 63)
                    const uint64_t st0 = sta[lo];
                    slot[u] = (st0 & START_MASK) + (s - pre[lo]); tp[u] = (uint32_t)(st0 >> 62);
                    sk[u] = skp[lo];
                    b8[u] = cnt8[slot[u]];
                    kk[u] = keys[slot[u]];
                }
            }
#pragma unroll
            for (int u = 0; u < GATHER_UNROLL; u++) {
                if (s0 + u * 64 >= total) break;
                uint32_t c = b8[u];
                if (c == 255u) c = cnt32[slot[u]];
                const bool ok = c != 0 && (all_solid || ((int32_t)c >= amin && (int32_t)c <= amax));      // CountRange::includes (closed interval)
                const unsigned long long bal = __ballot(ok);
                if (ok) {
                    const key_t key = kk[u] | ((key_t)tp[u] << top_shift);
                    uint64_t* dst = out + (o + sk[u] + __popcll(bal & lt_mask)) * OW;
                    if (KW == 1) store16(dst, (uint64_t)key, (uint64_t)c);
                    else {
                        store16(dst, (uint64_t)key, (uint64_t)((u128)key >> 64));
                        store16(dst + 2, (uint64_t)c, 0ULL);
                    }
                }
                o += __popcll(bal);
            }
        }
    }
}
// The sub-buckets that were split ("roots": 4097 .. 10^7 keys): their records sit at the heads of their pieces, anywhere in the slot range. They are walked in
// chunks of ROOT_CHUNK slots, all chunks of all roots in parallel: chunk table (k_root_chunks) -> solid slots per chunk (k_root_count) -> prefix (k_root_scan) ->
// records at root offset + chunk prefix (k_root_write).
constexpr int ROOT_THREADS = 1024, ROOT_CHUNK = 4096;
struct RootTables { uint32_t* base; /* [n_roots + 1] first chunk of the root */ uint32_t* root_of; /* [chunks] */ uint32_t* cnt; /* [chunks + 1] solid slots, then their exclusive prefix */ uint32_t* n_chunks; };
__device__ __forceinline__ uint32_t wg_excl_scan_1024(uint32_t v, uint32_t* s_w, uint32_t& total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
    __syncthreads();
    if (lane == 63) s_w[wave] = x;
    __syncthreads();
    uint32_t pre = x - v; total = 0;
    for (int w = 0; w < ROOT_THREADS / 64; w++) { if (w < wave) pre += s_w[w]; total += s_w[w]; }
    return pre;
}
__global__ __launch_bounds__(ROOT_THREADS) void k_root_chunks(const uint32_t* __restrict__ root_list, const uint32_t* __restrict__ n_roots_p, const uint32_t* __restrict__ b_n, RootTables R)
{
    __shared__ uint32_t s_w[ROOT_THREADS / 64];
    const uint32_t n_roots = *n_roots_p;
    uint32_t carry = 0;
    for (uint32_t i0 = 0; i0 < n_roots; i0 += ROOT_THREADS) {
        const uint32_t i = i0 + threadIdx.x;
        const uint32_t nch = i < n_roots ? (b_n[root_list[i]] + ROOT_CHUNK - 1) / ROOT_CHUNK : 0u;
        uint32_t tot; const uint32_t pre = wg_excl_scan_1024(nch, s_w, tot);
        if (i < n_roots) {
            R.base[i] = carry + pre;
            for (uint32_t c = 0; c < nch; c++) R.root_of[carry + pre + c] = i;          // chunk -> root (1-2 chunks per root; a giant: a few hundred stores, nobody waits for them)
        }
        carry += tot;
    }
    if (threadIdx.x == 0) { R.base[n_roots] = carry; *R.n_chunks = carry; }
}
__global__ __launch_bounds__(256) void k_root_count(const uint8_t* __restrict__ cnt8, const uint32_t* __restrict__ cnt32, const uint64_t* __restrict__ b_start, const uint32_t* __restrict__ b_n,
                                                    const uint32_t* __restrict__ root_list, RootTables R, int32_t amin, int32_t amax, uint32_t all_solid)
{
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * 256 + threadIdx.x) >> 6, n_waves = (gridDim.x * 256) >> 6;
    const uint32_t n_chunks = *R.n_chunks;
    for (uint32_t c = wave; c < n_chunks; c += n_waves) {
        const uint32_t i = R.root_of[c], g = root_list[i];
        const uint64_t start = b_start[g]; const uint32_t n = b_n[g], s0 = (c - R.base[i]) * ROOT_CHUNK;
        uint32_t cnt = 0;
        // 16 slots per lane and step (the range starts on a multiple of 4 slots: 4-byte loads)
        for (uint32_t j = s0 + 4 * lane; j < s0 + ROOT_CHUNK && j < n; j += 256) {
            uint32_t w;
            if (j + 4 <= n) w = *reinterpret_cast<const uint32_t*>(cnt8 + start + j);
            else { w = 0; for (uint32_t q = 0; j + q < n; q++) w |= (uint32_t)cnt8[start + j + q] << (8 * q); }
            if (!w) continue;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t b8 = (w >> (8 * q)) & 255u;
                if (b8) { const uint32_t cc = b8 == 255u ? cnt32[start + j + q] : b8; cnt += (all_solid || ((int32_t)cc >= amin && (int32_t)cc <= amax)) ? 1u : 0u; }
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d, 64);
        if (lane == 0) R.cnt[c] = cnt;
    }
}
__global__ __launch_bounds__(ROOT_THREADS) void k_root_scan(RootTables R)
{
    __shared__ uint32_t s_w[ROOT_THREADS / 64];
    const uint32_t n_chunks = *R.n_chunks;
    uint32_t carry = 0;
    for (uint32_t i0 = 0; i0 < n_chunks; i0 += ROOT_THREADS) {
        const uint32_t i = i0 + threadIdx.x;
        const uint32_t v = i < n_chunks ? R.cnt[i] : 0u;
        uint32_t tot; const uint32_t pre = wg_excl_scan_1024(v, s_w, tot);
        if (i < n_chunks) R.cnt[i] = carry + pre;
        carry += tot;
    }
    if (threadIdx.x == 0) R.cnt[n_chunks] = carry;
}
template <int KW>
__global__ __launch_bounds__(256) void k_root_write(const typename KeyT<KW>::type* __restrict__ keys, const uint8_t* __restrict__ cnt8, const uint32_t* __restrict__ cnt32,
                                                    const uint64_t* __restrict__ b_start, const uint32_t* __restrict__ b_n, const uint32_t* __restrict__ root_list, RootTables R,
                                                    const uint64_t* __restrict__ off_s, int32_t amin, int32_t amax, uint32_t all_solid, uint64_t* __restrict__ out,
                                                    const uint8_t* __restrict__ b_cons, uint32_t top_shift /* as in k_gather_counts */)
{
    constexpr int OW = (KW == 1) ? 2 : 4;
    typedef typename KeyT<KW>::type key_t;
    const int lane = threadIdx.x & 63;
    const uint64_t lt_mask = (1ULL << lane) - 1ULL;
    const uint32_t wave = (blockIdx.x * 256 + threadIdx.x) >> 6, n_waves = (gridDim.x * 256) >> 6;
    const uint32_t n_chunks = *R.n_chunks;
    for (uint32_t c = wave; c < n_chunks; c += n_waves) {
        const uint32_t i = R.root_of[c], g = root_list[i];
        const key_t top = (key_t)((b_cons[g] >> CONS_DROP_SHIFT) & 3u) << top_shift;
        const uint64_t start = b_start[g]; const uint32_t n = b_n[g], s0 = (c - R.base[i]) * ROOT_CHUNK;
        uint64_t o = off_s[g] + (R.cnt[c] - R.cnt[R.base[i]]);
        for (uint32_t j0 = s0; j0 < s0 + ROOT_CHUNK && j0 < n; j0 += 64) {       // (uniform trip count: the ballots see all lanes)
            const uint32_t j = j0 + lane;
            uint32_t cc = 0;
            if (j < n) { const uint32_t b8 = cnt8[start + j]; if (b8) cc = b8 == 255u ? cnt32[start + j] : b8; }
            const bool ok = cc != 0 && (all_solid || ((int32_t)cc >= amin && (int32_t)cc <= amax));
            const unsigned long long bal = __ballot(ok);
            if (ok) {
                const key_t key = keys[start + j] | top;
                uint64_t* dst = out + (o + __popcll(bal & lt_mask)) * OW;
                if (KW == 1) store16(dst, (uint64_t)key, (uint64_t)cc);
                else {
                    store16(dst, (uint64_t)key, (uint64_t)((u128)key >> 64));
                    store16(dst + 2, (uint64_t)cc, 0ULL);
                }
            }
            o += __popcll(bal);
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

// ------------------------------------------------------------------------------------------------ super-k-mer deduplication (k <= 31)
// With sequencing coverage c a genomic super-k-mer is read ~c times, and every copy that lies whole and error-free inside its read is the SAME record (same
// minimizer, same extent) up to the strand. Expanding, scattering and sorting its k-mers once with a weight instead of c times is less work for every later
// kernel of Stage B (30x synthetic reads, 1 % substitutions: 1.66x fewer keys with weights up to 4; the copies cut by a read end or changed by an error stay
// single). The reference has no such step (it counts k-mer by k-mer: PartitionsCommand.cpp:944-1128); the result is the same multiset of k-mers.
//   k_dedupe_bin   one workgroup per partition (persistent, ticket): every record in its strand-canonical form (the smaller of the nucleotide string and its
//                  reverse complement: the same canonical k-mers either way) is dropped into one of <= 4096 bins of the partition by a hash of its content
//                  (LDS histogram -> scan -> LDS cursors), in a scratch arena laid out like the partition's records;
//   k_dedupe_sort  one wave per bin (~64 records): 128-bit register sort, run lengths, and the bin is rewritten in place as one record per run and per
//                  2^wb copies (wb: the batch's weight bits) — (copies - 1) in the record's spare bits below its nucleotides — followed by empty records (nbK = 0: the
//                  expansion kernels skip them). A bin beyond the wave's registers is left as it is (weights 1).
// bins of 100..200 records (measured, 1e8 reads, one lane, k_dedupe_bin + k_dedupe_sort per step: mean <= 24: 46.6 ms, 48: 45.8, 96: 45.1, 160: 41.1, 200: 40.3; with 512-record
// bins and a double-size network, mean <= 320: 44.4, 440: 41.9): fewer bins are fewer open 32-byte sectors per workgroup in the binning scatter — 1024 instead of 4096,
// which two workgroups per CU keep inside the XCD's L2 — at the price of a longer sorting network per record
constexpr int DD_THREADS = 1024, DD_BINS_MAX = 4096, DD_BIN_TARGET = 200;
constexpr int DD_KPL_MAX = 4;
// records per lane / per bin that are deduplicated (16-byte records; half for 32-byte ones), slot bits inside the sorted words, waves of a sorting workgroup (64 KB of LDS windows)
template <int RW> struct DDCap {
    static constexpr int KPL_MAX = RW == 2 ? DD_KPL_MAX : DD_KPL_MAX / 2; static constexpr int SLOTS = 64 * KPL_MAX;
    static constexpr int SLOT_BITS = SLOTS <= 128 ? 7 : SLOTS <= 256 ? 8 : 9;
    static constexpr int WAVES = 65536 / (SLOTS * RW * 8) > 16 ? 16 : 65536 / (SLOTS * RW * 8);
};
struct DedupeTables { uint32_t* bin_start; /* [nb][DD_BINS_MAX + 1] first record of the bin, relative to the partition's first record */ uint32_t* bin_log2; /* [nb] */ };
template <int RW> struct DRec { uint64_t w[RW]; };
template <int RW> __device__ __forceinline__ DRec<RW> dd_load(const uint64_t* p) { DRec<RW> r;
#pragma unroll
    for (int i = 0; i < RW; i += 2) { const ulonglong2 q = *reinterpret_cast<const ulonglong2*>(p + i); r.w[i] = q.x; r.w[i + 1] = q.y; } return r; }
template <int RW> __device__ __forceinline__ void dd_store(uint64_t* p, const DRec<RW>& r) {
#pragma unroll
    for (int i = 0; i < RW; i += 2) *reinterpret_cast<ulonglong2*>(p + i) = make_ulonglong2(r.w[i], r.w[i + 1]); }
template <int RW> __device__ __forceinline__ bool dd_equal(const DRec<RW>& a, const DRec<RW>& b) { bool e = true;
#pragma unroll
    for (int i = 0; i < RW; i++) e = e && a.w[i] == b.w[i]; return e; }

// strand-canonical form: the smaller of the record's nucleotide string and its reverse complement (same nbK)
__device__ __forceinline__ void dd_canonical(DRec<2>& R, uint32_t k)
{
    const uint32_t nbk = (uint32_t)(R.w[0] >> 56);
    if (nbk == 0) return;
    const uint32_t L = k + nbk - 1;                                   // nucleotides of the record (<= 58)
    const uint64_t s_hi = (R.w[0] << 8) | (R.w[1] >> 56), s_lo = R.w[1] << 8;     // the string, left-aligned in 128 bits
    // reverse complement of all 64 positions, then the L real ones moved back to the left (the complemented padding falls off)
    const u128 rc = (((u128)revcomp64(s_lo, 32)) << 64) | revcomp64(s_hi, 32);
    const u128 rv = rc << (128 - 2 * L);
    const u128 fw = (((u128)s_hi) << 64) | s_lo;
    if (rv < fw) {
        const uint64_t h = (uint64_t)(rv >> 64), l = (uint64_t)rv;
        R.w[0] = ((uint64_t)nbk << 56) | (h >> 8); R.w[1] = (h << 56) | (l >> 8);
    }
}
__device__ __forceinline__ void dd_canonical(DRec<4>& R, uint32_t k)
{
    const uint32_t nbk = (uint32_t)(R.w[0] >> 56);
    if (nbk == 0) return;
    const uint32_t L = k + nbk - 1;                                   // nucleotides of the record (<= 122)
    const uint64_t S0 = (R.w[0] << 8) | (R.w[1] >> 56), S1 = (R.w[1] << 8) | (R.w[2] >> 56), S2 = (R.w[2] << 8) | (R.w[3] >> 56), S3 = R.w[3] << 8;
    const u128 fh = (((u128)S0) << 64) | S1, fl = (((u128)S2) << 64) | S3;                     // the string, left-aligned in 256 bits
    const u128 rh = (((u128)revcomp64(S3, 32)) << 64) | revcomp64(S2, 32), rl = (((u128)revcomp64(S1, 32)) << 64) | revcomp64(S0, 32);   // all 128 positions reversed
    const uint32_t sh = 256 - 2 * L;                                  // in [12, 192]
    u128 vh, vl;
    if (sh >= 128) { vh = sh == 128 ? rl : (rl << (sh - 128)); vl = 0; }
    else { vh = (rh << sh) | (rl >> (128 - sh)); vl = rl << sh; }
    if (vh < fh || (vh == fh && vl < fl)) {
        const uint64_t a0 = (uint64_t)(vh >> 64), a1 = (uint64_t)vh, a2 = (uint64_t)(vl >> 64), a3 = (uint64_t)vl;
        R.w[0] = ((uint64_t)nbk << 56) | (a0 >> 8); R.w[1] = (a0 << 56) | (a1 >> 8); R.w[2] = (a1 << 56) | (a2 >> 8); R.w[3] = (a2 << 56) | (a3 >> 8);
    }
}
template <int RW> __device__ __forceinline__ uint64_t dd_hash64(const DRec<RW>& R)
{
    uint64_t h = R.w[0];
#pragma unroll
    for (int i = 1; i < RW; i++) h = (h ^ (R.w[i] * 0x9E3779B97F4A7C15ULL)) * 0xBF58476D1CE4E5B9ULL;
    return h ^ (h >> 29);
}

template <int RW>
__global__ __launch_bounds__(DD_THREADS) void k_dedupe_bin(const PartDesc* __restrict__ parts, SegTable segs, uint32_t k, const uint64_t* __restrict__ rec_base /* [nb + 1] */,
                                                            uint64_t* __restrict__ arena, DedupeTables D, uint32_t nb, uint32_t* __restrict__ ticket)
{
    __shared__ uint32_t s_cnt[DD_BINS_MAX];
    __shared__ uint32_t s_wsum[DD_THREADS / 64];
    __shared__ uint32_t s_item;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (;;) {
        __syncthreads();
        if (t == 0) s_item = atomicAdd(ticket, 1u);
        __syncthreads();
        const uint32_t bi = s_item;
        if (bi >= nb) break;
        const PartDesc pd = parts[bi];
        const uint64_t base = rec_base[bi]; const uint32_t nrec = (uint32_t)(rec_base[bi + 1] - base);
        uint32_t lg = 0; while (lg < 12 && (nrec >> lg) > (uint32_t)DD_BIN_TARGET) lg++;
        const uint32_t nbin = 1u << lg, hsh = 64 - lg;
        for (uint32_t i = t; i < nbin; i += DD_THREADS) s_cnt[i] = 0;
        __syncthreads();
        for (uint32_t s = 0; s < segs.n_seg; s++) {
            const uint64_t r0 = segs.rec_off[(uint64_t)s * (segs.P + 1) + pd.part], r1 = segs.rec_off[(uint64_t)s * (segs.P + 1) + pd.part + 1];
            const uint64_t* recs = reinterpret_cast<const uint64_t*>(segs.rec[s]);
            for (uint64_t r = r0 + t; r < r1; r += DD_THREADS) {
                DRec<RW> R = dd_load<RW>(recs + r * RW);
                dd_canonical(R, k);
                wave_add1(s_cnt, lg ? (uint32_t)(dd_hash64<RW>(R) >> hsh) : 0u);
            }
        }
        __syncthreads();
        // exclusive scan -> first record of every bin; the cursors replace the counts
        const uint32_t per = (nbin + DD_THREADS - 1) / DD_THREADS, b = t * per;
        uint32_t loc = 0;
        for (uint32_t i = 0; i < per; i++) if (b + i < nbin) loc += s_cnt[b + i];
        uint32_t x = loc;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
        if (lane == 63) s_wsum[wave] = x;
        __syncthreads();
        uint32_t run = x - loc;
        for (int w = 0; w < wave; w++) run += s_wsum[w];
        uint32_t* bs = D.bin_start + (size_t)bi * (DD_BINS_MAX + 1);
        for (uint32_t i = 0; i < per; i++) if (b + i < nbin) { const uint32_t c = s_cnt[b + i]; s_cnt[b + i] = run; bs[b + i] = run; run += c; }
        if (t == 0) { bs[nbin] = nrec; D.bin_log2[bi] = lg; }
        __syncthreads();
        uint64_t* out = arena + base * RW;
        for (uint32_t s = 0; s < segs.n_seg; s++) {
            const uint64_t r0 = segs.rec_off[(uint64_t)s * (segs.P + 1) + pd.part], r1 = segs.rec_off[(uint64_t)s * (segs.P + 1) + pd.part + 1];
            const uint64_t* recs = reinterpret_cast<const uint64_t*>(segs.rec[s]);
            for (uint64_t r = r0 + t; r < r1; r += DD_THREADS) {
                DRec<RW> R = dd_load<RW>(recs + r * RW);
                dd_canonical(R, k);
                const uint32_t slot = wave_take1(s_cnt, lg ? (uint32_t)(dd_hash64<RW>(R) >> hsh) : 0u);
                dd_store<RW>(out + (uint64_t)slot * RW, R);
            }
        }
    }
}

// One bin: the records are brought into hash order by sorting 64-bit words [hash : 43][slot : 8] in the f64-tagged register network of k_wave_sort (equal records have
// equal hashes and end up adjacent; two different records under one 43-bit hash merely stay unmerged), then fetched in that order through the wave's LDS window.
template <int RW, int KPL>
__device__ __forceinline__ uint32_t dd_sort_bin(const DRec<RW> (&in)[DDCap<RW>::KPL_MAX] /* record r * 64 + lane of the bin */, const uint32_t n, const int lane,
                                                uint64_t* __restrict__ s_win /* [SLOTS][RW] of this wave */,
                                                DRec<RW> (&rec)[DDCap<RW>::KPL_MAX], uint32_t (&cnt)[DDCap<RW>::KPL_MAX], unsigned long long& in_keys, const uint32_t WCAP /* copies one record stands for at most */)
{
    constexpr int KM = DDCap<RW>::KPL_MAX, SB = DDCap<RW>::SLOT_BITS;
    uint64_t key[KPL];
#pragma unroll
    for (int r = 0; r < KPL; r++) {
        const uint32_t i = r * 64 + lane;
        if (i < n) { dd_store<RW>(s_win + (size_t)i * RW, in[r]); key[r] = TAG64 | ((dd_hash64<RW>(in[r]) >> (13 + SB)) << SB) | (uint64_t)i; }   // 51 - SB hash bits, SB slot bits
        else key[r] = TAG64 | TAG64_MANT;
    }
    bitonic_wave<1, KPL, true>(key, lane);
    // the records in sorted order (rank e = lane * KPL + r), run ends by full comparison with the next record
#pragma unroll
    for (int r = 0; r < KM; r++) { cnt[r] = 0;
#pragma unroll
        for (int i = 0; i < RW; i++) rec[r].w[i] = 0; }
#pragma unroll
    for (int r = 0; r < KPL; r++) { const uint32_t e = (uint32_t)lane * KPL + r; if (e < n) rec[r] = dd_load<RW>(s_win + (size_t)((uint32_t)key[r] & ((1u << SB) - 1u)) * RW); }
    DRec<RW> next_first;
#pragma unroll
    for (int i = 0; i < RW; i++) next_first.w[i] = (uint64_t)__shfl_down((unsigned long long)rec[0].w[i], 1, 64);
    const uint32_t lane0 = (uint32_t)lane * KPL;
    uint32_t tailm = 0;
#pragma unroll
    for (int r = 0; r < KPL; r++) {
        const uint32_t e = lane0 + r;
        const bool same = r < KPL - 1 ? dd_equal<RW>(rec[r], rec[r + 1 < KM ? r + 1 : r]) : dd_equal<RW>(rec[r], next_first);
        tailm |= (uint32_t)(e < n && (e == n - 1 || !same)) << r;
    }
    int lt = tailm ? (int)(lane0 + 31 - __clz((int)tailm)) : -1;          // rank of the lane's last run end; carried by a max-scan
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(lt, d, 64); if (lane >= d) lt = y > lt ? y : lt; }
    int prev = __shfl_up(lt, 1, 64); if (lane == 0) prev = -1;
    uint32_t nout = 0;
#pragma unroll
    for (int r = 0; r < KPL; r++) if ((tailm >> r) & 1) {
        const int e = (int)lane0 + r; cnt[r] = (uint32_t)(e - prev); prev = e; nout += (cnt[r] + WCAP - 1) / WCAP;
        in_keys += (unsigned long long)cnt[r] * (uint32_t)(rec[r].w[0] >> 56);
    }
    return nout;
}
// one workgroup per partition (persistent, ticket): its waves take the bins in order; what a bin is rewritten as goes right behind the output of the bin before it
// (a chain through LDS: the wave waits for its predecessor's end, never for more), so the partition's deduplicated records end up contiguous at the front of its
// range — the expansion kernels then walk 0.6x the records instead of stepping over holes. In place: everything left of a bin's output has been read already.
template <int RW>
__global__ __launch_bounds__(DDCap<RW>::WAVES * 64) void k_dedupe_sort(uint64_t* __restrict__ arena, const uint64_t* __restrict__ rec_base, DedupeTables D, const PartDesc* __restrict__ parts,
                                                             uint64_t* __restrict__ rec_end /* [P] */, uint32_t nb, uint32_t* __restrict__ ticket,
                                                             unsigned long long* __restrict__ totals /* [0] k-mers in [1] k-mers out */, uint32_t WCAP /* copies one record may stand for */)
{
    constexpr int KM = DDCap<RW>::KPL_MAX, SLOTS = DDCap<RW>::SLOTS, DDS_WAVES = DDCap<RW>::WAVES;
    __shared__ __attribute__((aligned(16))) uint64_t s_win[DDS_WAVES][SLOTS * RW];      // 64 KB
    __shared__ volatile uint32_t s_next, s_pos;             // bin whose output may be placed now; where
    __shared__ uint32_t s_item;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    unsigned long long ik = 0, ok = 0;
    for (;;) {
        __syncthreads();
        if (t == 0) { s_item = atomicAdd(ticket, 1u); s_next = 0; s_pos = 0; }
        __syncthreads();
        const uint32_t bi = s_item;
        if (bi >= nb) break;
        const uint32_t nbin = 1u << D.bin_log2[bi];
        const uint32_t* bs = D.bin_start + (size_t)bi * (DD_BINS_MAX + 1);
        uint64_t* part = arena + rec_base[bi] * RW;
        // the next bin of the wave is in flight (registers) while this one is sorted
        DRec<RW> nx[KM]; uint32_t nx_s0 = 0, nx_n = 0;
        auto fetch = [&](uint32_t bin_) {
            nx_s0 = 0; nx_n = 0;
            if (bin_ < nbin) {
                nx_s0 = bs[bin_]; nx_n = bs[bin_ + 1] - nx_s0;
                if (nx_n >= 2 && nx_n <= (uint32_t)SLOTS) {
#pragma unroll
                    for (int r = 0; r < KM; r++) { const uint32_t i = r * 64 + lane; if (i < nx_n) nx[r] = dd_load<RW>(part + (size_t)(nx_s0 + i) * RW); }
                }
            }
        };
        fetch(wave);
        for (uint32_t bin = wave; bin < nbin; bin += DDS_WAVES) {
            const uint32_t s0 = nx_s0, n = nx_n;
            DRec<RW> in[KM];
#pragma unroll
            for (int r = 0; r < KM; r++) in[r] = nx[r];
            fetch(bin + DDS_WAVES);
            // A bin beyond the wave's registers — one record copied 10^5 .. 10^7 times: the super-k-mers of poly-A / (AC)n reads, of a repeat family at hundreds of copies (every
            // copy hashes into this bin) — is deduplicated CHUNK by chunk of SLOTS records (round 5): each chunk sorted and merged like a bin of its own by the code below, its
            // output placed behind the output of the chunk before (always left of the chunk's own records: in place). Up to round 4 such a bin was moved as it was: 1e8 reads
            // with 1 % low-complexity reads then sent 2e7 keys of ONE k-mer through one parking slot of the scatter and one cursor of the giant split
            // (k_expand_scatter_pair 1.08 s, k_giant_scatter 0.55 s, k_deep_split 0.29 s per step: profiles/r05_skewed_input.txt). An ordinary bin is its own single chunk.
            const uint32_t nch = n > (uint32_t)SLOTS ? (n + SLOTS - 1) / SLOTS : 1u;
            uint32_t pos = 0;
            for (uint32_t ch = 0; ch < nch; ch++) {
                uint32_t cn = n;
                if (nch > 1) {
                    cn = min((uint32_t)SLOTS, n - ch * SLOTS);
#pragma unroll
                    for (int r = 0; r < KM; r++) { const uint32_t i = r * 64 + lane; if (i < cn) in[r] = dd_load<RW>(part + (size_t)(s0 + ch * SLOTS + i) * RW); }
                }
                DRec<RW> rec[KM]; uint32_t cnt[KM];
#pragma unroll
                for (int r = 0; r < KM; r++) cnt[r] = 0;
                uint32_t nout = 0; int kpl = 0;                                   // kpl 0: one record (or none), moved as it is
                if (cn >= 2 && cn <= 64) { nout = dd_sort_bin<RW, 1>(in, cn, lane, s_win[wave], rec, cnt, ik, WCAP); kpl = 1; }
                else if (cn > 64 && cn <= 128) { nout = dd_sort_bin<RW, 2>(in, cn, lane, s_win[wave], rec, cnt, ik, WCAP); kpl = 2; }
                else if (KM >= 4 && cn > 128 && cn <= 256) { nout = dd_sort_bin<RW, (KM >= 4 ? 4 : KM)>(in, cn, lane, s_win[wave], rec, cnt, ik, WCAP); kpl = 4; }
                else if (KM >= 8 && cn > 256 && cn <= 512) { nout = dd_sort_bin<RW, (KM >= 8 ? 8 : KM)>(in, cn, lane, s_win[wave], rec, cnt, ik, WCAP); kpl = 8; }
                uint32_t x = nout;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
                const uint32_t total = kpl ? (uint32_t)__builtin_amdgcn_readlane((int)x, 63) : cn;
                if (ch == 0) {
                    // my turn? (the bins of a partition are handed to the waves in order: the wave of bin - 1 is another wave of this workgroup)
                    while (s_next != bin) __builtin_amdgcn_s_sleep(1);
                    pos = s_pos;
                }
                const uint32_t pos0 = pos;
                if (!kpl && cn) {
                    // one record: moved left BEFORE the next bin may place its output (which may reach into this bin's old range)
                    DRec<RW> q;
                    if (lane < cn) q = dd_load<RW>(part + (size_t)(s0 + ch * SLOTS + lane) * RW);
                    if (lane < cn) dd_store<RW>(part + (size_t)(pos0 + lane) * RW, q);
                }
                if (ch + 1 == nch) {                                              // the last chunk is in registers: the next bin may go
                    __threadfence_block();
                    if (lane == 0) { s_pos = pos0 + total; __threadfence_block(); s_next = bin + 1; }
                }
                if (kpl) {
                    uint32_t wp = pos0 + x - nout;
#pragma unroll
                    for (int r = 0; r < KM; r++) if (cnt[r]) {
                        const uint32_t nbk = (uint32_t)(rec[r].w[0] >> 56);
                        for (uint32_t c = cnt[r]; c; ) {
                            const uint32_t w = c < WCAP ? c : WCAP;
                            DRec<RW> o = rec[r]; o.w[RW - 1] |= (uint64_t)(w - 1);
                            dd_store<RW>(part + (size_t)wp * RW, o); wp++;
                            ok += nbk; c -= w;
                        }
                    }
                }
                pos = pos0 + total;
                if (nch > 1) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // (the next chunk reuses the wave's LDS window and reads behind what was just written)
            }
        }
        __syncthreads();
        if (t == 0) rec_end[parts[bi].part] = rec_base[bi] + s_pos;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { ik += __shfl_down(ik, d, 64); ok += __shfl_down(ok, d, 64); }
    if (lane == 0 && ik) { atomicAdd(&totals[0], ik); atomicAdd(&totals[1], ok); }
}

// ------------------------------------------------------------------------------------------------ host orchestration
constexpr uint32_t PART_ALIGN = 256;            // a partition's slot range starts on a multiple of this many slots
// weight bits of a batch whose partitions all have at least min_bits sub-bucket bits (see the note at the top of the file). GKC_WEIGHT_BITS (tests, experiments)
// asks for a number; it is honoured as far as the keys stay valid.
static int weight_bits_env() { return gkc_tun().weight_bits; }
template <int KW> static uint32_t weight_bits_of(uint32_t k, uint32_t min_bits)
{
    const int env = weight_bits_env();
    const int stored = 64 * KW, drop_ok = (int)std::min<uint32_t>(min_bits, (uint32_t)WEIGHT_DROP_MAX);
    const int valid = std::min<int>(WEIGHT_BITS_MAX, stored + drop_ok - 2 * (int)k);               // >= WEIGHT_BITS_MIN for every k the key width is used for
    int wb = valid;
    wb = std::min<int>(wb, std::max<int>(WEIGHT_BITS_MIN, KTagBits<KW>::value + (int)min_bits - 2 * (int)k));   // the f64-tagged network is worth more than a weight bit
    if (env) wb = std::min(env, valid);
    return (uint32_t)std::max<int>(WEIGHT_BITS_MIN, wb);
}
constexpr int DEEP_FIXED = 4;                   // split levels launched unconditionally (a level with an empty queue returns at once); more only if the last one left work
constexpr int DEEP_COUNTERS = 8;                // per-level counter triples (next level's queue length, sort list length, item ticket), used cyclically
struct BatchBufs {
    DevBuf pd, keysA, keysB, cnt, cnt8, b_start, b_n, b_cons, l_big, l_wg, l_split, misc, nd, ns, off_d, off_s, chunk, q[2], sitems, giant, glist, rbase, rroot, rcnt, dd_arena, dd_base, dd_bins, dd_lg, dd_off, dd_ptr, dd_end, pidx, ptot, order, vpd, vstart, vtail, vdone;
    void release() { DevBuf* all[] = { &pd, &keysA, &keysB, &cnt, &cnt8, &b_start, &b_n, &b_cons, &l_big, &l_wg, &l_split, &misc, &nd, &ns, &off_d, &off_s, &chunk, &q[0], &q[1], &sitems, &giant, &glist, &rbase, &rroot, &rcnt, &dd_arena, &dd_base, &dd_bins, &dd_lg, &dd_off, &dd_ptr, &dd_end, &pidx, &ptot, &order, &vpd, &vstart, &vtail, &vdone };
                     for (DevBuf* d : all) d->release(); }
};

template <int KW, int RW>
static int count_batch(gkc_ctx* c, const uint32_t pass, const std::vector<Segment>& segments, const std::vector<uint32_t>& batch_parts, const std::vector<uint64_t>& part_keys,
                       const SegTable& segs, std::vector<void*>& outputs)
{   // (pass, segments: what this Stage B was started for — Stage A may have moved the context on to the next pass meanwhile: gkc_finish_pass_async)
    typedef typename KeyT<KW>::type key_t;
    const uint32_t nb = (uint32_t)batch_parts.size();
    const uint32_t k = c->k;
    // --- host-built tables (sizes are known exactly from Stage A)
    std::vector<PartDesc> pd(nb);
    std::vector<uint64_t> pidx(nb + 1);
    uint64_t n_slots = 0, n_sub = 0;
    const int dedupe_env = gkc_tun().dedupe;       // 0: never, 1: always, default: until a batch shows it does not pay
    // Sliced batch (see SliceTables): a partition beyond GKC_SLICE_MIN k-mers (default 8e6: twice and more what the batches and the drop-in's Configuration aim
    // at, and where the bins of the record deduplication are full) is expanded by up to 16 workgroups, so that an entry is 2e6 .. 4e6 k-mers like a planned
    // partition; GKC_SLICES=0 switches it off (tests lower the threshold). The record deduplication (one workgroup per partition, bins for <= 8e5 records: on such
    // partitions it costs more than it saves — 242 ms for 256 partitions of 1e8 reads) is skipped for a batch most of whose k-mers sit in sliced partitions.
    // Measured, 1e8 reads, k = 31, two lanes (profiles/r04_sliced_partitions.txt): 256 partitions 600 -> 375 ms per step (expand_count 109 -> 19 ms, expand_scatter
    // 238 -> 90 ms single lane; what is left of the gap to the 4096-partition step, 203 ms, is the split levels: every 2^13-th of such a partition is 5700 keys,
    // beyond the sort tiers), 64 partitions 387 ms, 1024 partitions 346 -> 323 ms.
    const uint64_t slice_min = gkc_tun().slice_min;
    const uint64_t slice_keys = std::max<uint64_t>(1, slice_min / 4);
    const bool slices_on = gkc_tun().slices;
    bool sliced = false; uint64_t heavy_keys = 0, batch_keys = 0;
    if (slices_on) for (uint32_t i = 0; i < nb; i++) { const uint64_t np = part_keys[batch_parts[i]]; batch_keys += np; if (np > slice_min) { sliced = true; heavy_keys += np; } }
    // (one heavy partition among a thousand planned ones — a repeat family under one minimizer — does not cost the batch its deduplication: its single dedupe workgroup
    // hides behind the others; the step is skipped where most of the batch's k-mers sit in such partitions)
    const bool dedupe = ((KW == 1 && RW == 2) || (KW == 2 && RW == 4 && k >= 32)) && dedupe_env != 0 && (dedupe_env == 1 || !c->dedupe_off) && nb > 0 && !(sliced && 2 * heavy_keys > batch_keys);
    // mean keys of a level-1 bucket, counted in k-mers BEFORE identical records are merged: with the merge on (8-byte keys: ~1.8x fewer keys on 30x reads) twice as
    // many — 12 sub-bucket bits instead of 13 for the partitions of the 1e8-read bench: first sort tier 46.8 -> 38.3 ms, the larger tiers +8, scatter -4: 220 -> 214 ms
    // (possible since the tagged sort carries 61 key bits: profiles/r04_weight_bits_experiment.txt)
    const uint32_t target = (KW == 1) ? (dedupe ? 2 * SUB_TARGET : SUB_TARGET) : SUB_TARGET / 2;
    const uint32_t max_bits1 = gkc_tun().max_sub_bits >= 0 ? (uint32_t)gkc_tun().max_sub_bits : (uint32_t)MAX_SUB_BITS;
    const uint32_t wb_goal = weight_bits_of<KW>(k, max_bits1);
    const int stored_or_mantissa = KTagBits<KW>::value;      // (16-byte keys: 125, which covers the 128 - 2k - wb >= -2 the dropped top bits need)
    const uint32_t need_goal = (uint32_t)std::min<int>((int)max_bits1, std::max<int>(0, 2 * (int)k + (int)wb_goal - stored_or_mantissa));
    const uint32_t need_min = (uint32_t)std::min<int>((int)max_bits1, std::max<int>(0, 2 * (int)k + WEIGHT_BITS_MIN - stored_or_mantissa));
    bool goal_ok = weight_bits_env() != 0;              // (weight bits asked for: the sub-bucket bits they need, whatever the sizes)
    {   uint64_t tot = 0; for (uint32_t i = 0; i < nb; i++) tot += part_keys[batch_parts[i]];
        const uint64_t mean = nb ? tot / nb : 0;
        uint32_t bm = 0; while (bm < max_bits1 && (mean >> bm) > target) bm++;
        goal_ok = goal_ok || bm + 1 >= need_goal;
    }
    for (uint32_t i = 0; i < nb; i++) {
        const uint64_t np = part_keys[batch_parts[i]];
        if (np >= (1ULL << 32)) GKC_FAIL(c, GKC_ERR_ARG, "partition %u holds %llu k-mers (>= 2^32): use more partitions", batch_parts[i], (unsigned long long)np);
        uint32_t bits = 0;
        while (bits < max_bits1 && bits < 2 * k && (np >> bits) > target) bits++;
        // 8-byte keys: a small partition still gets enough sub-buckets for what is left of a key below the sub-bucket index (k-mer + weight bits) to fit the 61 bits the
        // f64-tagged network orders, so that the whole batch sorts with it (one partition with fewer would switch the batch to the integer network: +30 %) — with the
        // weight bits the batch could have at best (wb_goal) if its MEAN partition is within one bit of what they need, else with the smallest weights. At k = 31
        // that is 4 / 3 bits (rounds 2-3, 52-bit tag: 13 / 12 bits whatever the partition's size — the 8-GPU share on one GPU, 32768 partitions of 4.5e5 k-mers,
        // 325 -> 278 ms when the tag was widened). (16-byte keys at k = 63: one sub-bucket bit at least, so that the third weight bit can push the key's top bit out.)
        if (gkc_tun().max_sub_bits < 0) bits = std::min<uint32_t>(std::max(bits, goal_ok ? need_goal : need_min), 2 * k);
        pd[i].part = batch_parts[i]; pd[i].sub_bits = bits; pd[i].shift = 2 * k - bits; pd[i].pad = 0; pd[i].aux = 0;
        pd[i].key_base = n_slots; pd[i].sub_base = n_sub;
        pidx[i] = n_sub;
        n_slots += (np + (3ull << bits) + PART_ALIGN - 1) / PART_ALIGN * PART_ALIGN;       // sub-buckets start on multiples of 4 slots (pair scatter)
        n_sub += (1ull << bits);
    }
    pidx[nb] = n_sub;
    if (n_sub >= (1ULL << 31)) GKC_FAIL(c, GKC_ERR_ARG, "too many sub-buckets in one batch");
    uint32_t min_bits1 = 64, max_bits_b = 0; for (uint32_t i = 0; i < nb; i++) { min_bits1 = std::min(min_bits1, pd[i].sub_bits); max_bits_b = std::max(max_bits_b, pd[i].sub_bits); }
    // the expansion's work list: one entry per partition, or (sliced batch) 2^s entries per partition, each a share of its records
    std::vector<PartDesc> vpd; std::vector<uint32_t> v_parent; uint64_t slice_words = 0;
    if (sliced) for (uint32_t i = 0; i < nb; i++) {
        const uint64_t np = part_keys[batch_parts[i]];
        uint32_t sb = 0; while (sb < 4 && np > slice_min && (np >> sb) > slice_keys) sb++;
        for (uint32_t w = 0; w < (1u << sb); w++) {
            PartDesc e = pd[i]; e.pad = (sb << 16) | w; e.aux = slice_words; slice_words += 1ull << e.sub_bits;
            vpd.push_back(e); v_parent.push_back(i);
        }
    }
    const uint32_t nv = sliced ? (uint32_t)vpd.size() : nb;
    const uint32_t wb = weight_bits_of<KW>(k, nb ? min_bits1 : 0u);
    const uint32_t drop = 2 * k + wb > 64u * KW ? 2 * k + wb - 64u * KW : 0u;      // top bits of a key that fall off the stored word (<= WEIGHT_DROP_MAX <= min_bits1)
    const uint32_t wcap = drop >= 2 ? (1u << wb) - 1u : (1u << wb);                // copies one merged record may stand for (see the note at the top of the file)
    constexpr uint32_t CAP1 = WaveCapT1<KW>::CAP, CAP2 = WaveCapHuge<KW>::CAP;
    constexpr int K1 = WaveCapHuge<KW>::KPL / 2;
    constexpr uint32_t C1 = 4 * 64 * K1;                                    // workgroup tier: 4 waves x 64 x K1 keys (4096 / 2048)
    // measured (ms per 1.2e10 keys, workgroup tier + split levels + their sorts): up to 4096 keys in the workgroup tier: 40, up to 8192: 44, none: 47
    const uint32_t wg_max = gkc_tun().wg_max ? std::min<uint32_t>(gkc_tun().wg_max, C1) : C1;
    const uint32_t cap3 = std::max(wg_max, CAP2);                           // sub-buckets beyond are split again
    const uint64_t list_cap = n_slots / CAP1 + nb + 1;                      // sub-buckets beyond the first tier / pieces beyond it at any split level
    BatchBufs B;
#define CB_TRY(expr) do { int rc__ = (expr); if (rc__ != GKC_OK) { B.release(); return rc__; } } while (0)
#define CB_HIP(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { B.release(); c->set_error(GKC_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); return GKC_ERR_HIP; } } while (0)
    CB_TRY(c->ensure(B.pd, nb * sizeof(PartDesc)));
    // the big working buffers are sized for the pass's batch budget, not for this batch: every batch then asks the allocator for exactly
    // the same blocks (a batch one partition larger or smaller would otherwise land in the next size class now and then)
    const uint64_t alloc_slots = std::max<uint64_t>(std::max<uint64_t>(n_slots, c->slots_hint), 4);
    CB_TRY(c->ensure(B.keysA, (size_t)alloc_slots * sizeof(key_t))); CB_TRY(c->ensure(B.keysB, (size_t)alloc_slots * sizeof(key_t)));
    CB_TRY(c->ensure(B.cnt, (size_t)alloc_slots * 4)); CB_TRY(c->ensure(B.cnt8, (size_t)alloc_slots));
    CB_TRY(c->ensure(B.b_start, (size_t)n_sub * 8)); CB_TRY(c->ensure(B.b_n, (size_t)n_sub * 4)); CB_TRY(c->ensure(B.b_cons, (size_t)n_sub));
    CB_TRY(c->ensure(B.l_big, (size_t)list_cap * 4)); CB_TRY(c->ensure(B.l_wg, (size_t)list_cap * 4)); CB_TRY(c->ensure(B.l_split, (size_t)list_cap * 4));
    CB_TRY(c->ensure(B.q[0], (size_t)list_cap * sizeof(DeepItem))); CB_TRY(c->ensure(B.q[1], (size_t)list_cap * sizeof(DeepItem)));
    CB_TRY(c->ensure(B.misc, 64 * 4));
    CB_TRY(c->ensure(B.nd, (size_t)std::max<uint64_t>(n_sub, 1) * 4));
    const bool all_solid = c->amin <= 1 && c->amax == 0x7fffffff;
    if (!all_solid) CB_TRY(c->ensure(B.ns, (size_t)std::max<uint64_t>(n_sub, 1) * 4));
    CB_TRY(c->ensure(B.off_d, (size_t)(n_sub + 1) * 8)); CB_TRY(c->ensure(B.off_s, (size_t)(n_sub + 1) * 8));
    CB_TRY(c->ensure(B.pidx, (size_t)(nb + 1) * 8)); CB_TRY(c->ensure(B.ptot, (size_t)(nb + 1) * 16));
    // the expansion kernels run one workgroup per partition: workgroup i takes the i-th LARGEST partition, so that the launch does not end on one long
    // workgroup (partition sizes spread 2-3x around their mean). Only the assignment changes: the layout of the batch stays in partition order.
    std::vector<uint32_t> order(nv);
    for (uint32_t i = 0; i < nv; i++) order[i] = i;
    const bool lpt = gkc_tun().batch_lpt;
    auto keys_of_entry = [&](uint32_t v) -> uint64_t { return sliced ? part_keys[batch_parts[v_parent[v]]] >> (vpd[v].pad >> 16) : part_keys[batch_parts[v]]; };
    if (lpt) std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return keys_of_entry(a) > keys_of_entry(b); });
    CB_TRY(c->ensure(B.order, (size_t)nv * 4));
    CB_HIP(hipMemcpyAsync(B.order.p, order.data(), (size_t)nv * 4, hipMemcpyHostToDevice, cur_stream(c)));
    CB_HIP(hipMemcpyAsync(B.pd.p, pd.data(), nb * sizeof(PartDesc), hipMemcpyHostToDevice, cur_stream(c)));
    SliceTables ST{ nullptr, nullptr, nullptr };
    if (sliced) {
        CB_TRY(c->ensure(B.vpd, (size_t)nv * sizeof(PartDesc))); CB_TRY(c->ensure(B.vstart, (size_t)slice_words * 4)); CB_TRY(c->ensure(B.vtail, (size_t)slice_words * 4)); CB_TRY(c->ensure(B.vdone, (size_t)nv * 4));
        CB_HIP(hipMemcpyAsync(B.vpd.p, vpd.data(), (size_t)nv * sizeof(PartDesc), hipMemcpyHostToDevice, cur_stream(c)));
        CB_HIP(hipMemsetAsync(B.vdone.p, 0, (size_t)nv * 4, cur_stream(c)));
        ST.start = (uint32_t*)B.vstart.p; ST.tail = (uint32_t*)B.vtail.p; ST.done = (uint32_t*)B.vdone.p;
    }
    const PartDesc* const d_entries = sliced ? (const PartDesc*)B.vpd.p : (const PartDesc*)B.pd.p;          // what the expansion kernels walk
    CB_HIP(hipMemcpyAsync(B.pidx.p, pidx.data(), (size_t)(nb + 1) * 8, hipMemcpyHostToDevice, cur_stream(c)));
    CB_HIP(hipMemsetAsync(B.misc.p, 0, 64 * 4, cur_stream(c)));
    CB_HIP(hipMemsetAsync(B.nd.p, 0, (size_t)std::max<uint64_t>(n_sub, 1) * 4, cur_stream(c)));
    if (!all_solid) CB_HIP(hipMemsetAsync(B.ns.p, 0, (size_t)std::max<uint64_t>(n_sub, 1) * 4, cur_stream(c)));
    uint32_t* const misc = (uint32_t*)B.misc.p;          // [0] big [1] wg [2] split list lengths [3] giants [4] root chunks [5] ticket, [8 ..) counter triples of the split levels (cyclic)
    TierLists T{};
    T.big_list = (uint32_t*)B.l_big.p; T.big_count = misc + 0; T.wg_list = (uint32_t*)B.l_wg.p; T.wg_count = misc + 1;
    T.split_list = (uint32_t*)B.l_split.p; T.split_count = misc + 2; T.cap1 = CAP1; T.cap2 = CAP2; T.cap3 = cap3;
    CB_TRY(c->ensure(B.glist, GIANT_MAX * 4));
    T.giant_list = (uint32_t*)B.glist.p; T.giant_count = misc + 3;

    // Identical super-k-mer records of a partition are merged first (8-byte keys; see k_dedupe_*): the expansion then reads the batch's own deduplicated copy
    SegTable segs_b = segs;
    std::vector<uint64_t> dd_base_h, dd_off_h; const void* dd_arena_h = nullptr;      // sources of asynchronous copies: alive until the batch is through
    if (dedupe) {
        unsigned long long* const dd_totals = reinterpret_cast<unsigned long long*>(misc + 40);     // k-mers into / out of the deduplication of this batch
        const uint32_t Pn = segs.P, p_first = batch_parts.front(), p_last = batch_parts.back();
        std::vector<uint64_t>& base = dd_base_h; std::vector<uint64_t>& off = dd_off_h; base.assign(nb + 1, 0); off.assign((size_t)Pn + 1, 0);
        {   uint64_t run = 0; uint32_t i = 0;
            for (uint32_t p = 0; p <= Pn; p++) {
                off[p] = run;
                if (p >= p_first && p <= p_last && i < nb && batch_parts[i] == p) {
                    uint64_t n = 0; for (const Segment& sg : segments) n += sg.rec_off[p + 1] - sg.rec_off[p];
                    base[i] = run; run += n; i++; base[i] = run;
                }
            }
        }
        const uint64_t total_recs = base[nb];
        bool fits = total_recs > 0; for (uint32_t i = 0; i < nb; i++) fits = fits && (base[i + 1] - base[i]) < (1ULL << 31);
        if (fits) {
            CB_TRY(c->ensure(B.dd_arena, (size_t)total_recs * RW * 8)); CB_TRY(c->ensure(B.dd_base, (size_t)(nb + 1) * 8)); CB_TRY(c->ensure(B.dd_bins, (size_t)nb * (DD_BINS_MAX + 1) * 4));
            CB_TRY(c->ensure(B.dd_lg, (size_t)nb * 4)); CB_TRY(c->ensure(B.dd_off, ((size_t)Pn + 1) * 8)); CB_TRY(c->ensure(B.dd_ptr, 8));
            const void*& arena_p = dd_arena_h; arena_p = B.dd_arena.p;
            CB_HIP(hipMemcpyAsync(B.dd_base.p, base.data(), (size_t)(nb + 1) * 8, hipMemcpyHostToDevice, cur_stream(c)));
            CB_HIP(hipMemcpyAsync(B.dd_off.p, off.data(), ((size_t)Pn + 1) * 8, hipMemcpyHostToDevice, cur_stream(c)));
            CB_HIP(hipMemcpyAsync(B.dd_ptr.p, &arena_p, 8, hipMemcpyHostToDevice, cur_stream(c)));
            DedupeTables DT{ (uint32_t*)B.dd_bins.p, (uint32_t*)B.dd_lg.p };
            {   ScopedTimer tm(c, "dedupe_bin");
                hipLaunchKernelGGL((k_dedupe_bin<RW>), dim3(std::min(nb, 512u)), dim3(DD_THREADS), 0, cur_stream(c), (const PartDesc*)B.pd.p, segs, k, (const uint64_t*)B.dd_base.p,
                                   (uint64_t*)B.dd_arena.p, DT, nb, misc + 5);
            }
            CB_TRY(c->ensure(B.dd_end, (size_t)Pn * 8));
            ScopedTimer tm(c, "dedupe_sort");
            hipLaunchKernelGGL((k_dedupe_sort<RW>), dim3(std::min(nb, 512u)), dim3(DDCap<RW>::WAVES * 64), 0, cur_stream(c), (uint64_t*)B.dd_arena.p, (const uint64_t*)B.dd_base.p, DT, (const PartDesc*)B.pd.p,
                               (uint64_t*)B.dd_end.p, nb, misc + 44, dd_totals, wcap);
            CB_HIP(hipGetLastError());
            segs_b.rec = (const uint8_t* const*)B.dd_ptr.p; segs_b.rec_off = (const uint64_t*)B.dd_off.p; segs_b.n_seg = 1; segs_b.rec_end = (const uint64_t*)B.dd_end.p;
        }
    }
    {   ScopedTimer tm(c, "expand_count");
        hipLaunchKernelGGL((k_expand_count<KW, RW>), dim3(nv), dim3(EXPAND_THREADS), 0, cur_stream(c), d_entries, segs_b, k,
                           (uint64_t*)B.b_start.p, (uint32_t*)B.b_n.p, (uint8_t*)B.b_cons.p, T, (const uint32_t*)B.order.p, nv, misc + 7, drop, ST);
        CB_HIP(hipGetLastError());
    }
    {   ScopedTimer tm(c, "expand_scatter");
        // The scatter is bound by the write requests the whole chip retires, not by its CUs (tools/scatter_bench: 63-127 workgroups write MORE than 254), and the
        // other Stage-B lane's kernel wants CUs: the launch takes 11/16 of them (persistent workgroups, partitions handed out largest first by a ticket).
        // Measured, two lanes, 1e8 reads: one workgroup per partition 266-273 ms per step, 160-192 workgroups 249-251, 128: 252, 96: 256.
        const uint32_t scatter_wgs = gkc_tun().scatter_wgs;
        if constexpr (KW == 1) {
            const size_t comb = (size_t)PAIR_THREADS * 8 + 16;                                            // 64 pairing words per wave (same-address relief)
            const size_t lds_max = (size_t)MAX_SUB * 12 + comb, lds = ((size_t)12 << max_bits_b) + comb;          // parking slots + cursors of the batch's largest sub-bucket count
            static std::once_flag once; std::call_once(once, [&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_expand_scatter_pair), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max); });
            hipLaunchKernelGGL(k_expand_scatter_pair, dim3(std::min(nv, scatter_wgs)), dim3(PAIR_THREADS), lds, cur_stream(c), d_entries, segs_b, k,
                               (const uint64_t*)B.b_start.p, (uint64_t*)B.keysA.p, (const uint32_t*)B.order.p, nv, misc + 6, wb, ST);
        } else {
            const size_t lds = (size_t)MAX_SUB * 20;                           // 160 KB: the whole LDS of a CU
            static std::once_flag once; std::call_once(once, [&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_expand_scatter_pair2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
            hipLaunchKernelGGL(k_expand_scatter_pair2, dim3(std::min(nv, scatter_wgs)), dim3(PAIR_THREADS), lds, cur_stream(c), d_entries, segs_b, k,
                               (const uint64_t*)B.b_start.p, (u128*)B.keysA.p, (const uint32_t*)B.order.p, nv, misc + 6, wb, ST);
        }
        CB_HIP(hipGetLastError());
    }
    SortOut O{};
    O.cnt8 = (uint8_t*)B.cnt8.p; O.cnt32 = (uint32_t*)B.cnt.p; O.histo = c->histo_of(pass); O.histo_max = c->histo_max;
    O.nd = (uint32_t*)B.nd.p; O.ns = all_solid ? (uint32_t*)B.nd.p : (uint32_t*)B.ns.p; O.amin = c->amin; O.amax = c->amax; O.all_solid = all_solid ? 1u : 0u;

    O.wb = wb;
    // --- the sort tiers, back to back: which sub-bucket goes where was decided by k_expand_count; no host round trip until the totals below
    // every bucket's keys share their top min_bits1 bits: when the rest fits a double's 52-bit mantissa the in-lane exchanges run as v_min/max_f64
    // (16-byte keys, round 4: the same tag on the key's top word — v_min/max_f64 there, selects on the low word: KTag<2>)
    const bool tag = 2 * k + wb <= (uint32_t)KTagBits<KW>::value + min_bits1 && !gkc_tun().no_f64;
    constexpr bool FT = true;
    key_t* const keysA = (key_t*)B.keysA.p; key_t* const keysB = (key_t*)B.keysB.p;
    const uint64_t* const bs = (const uint64_t*)B.b_start.p; const uint32_t* const bn = (const uint32_t*)B.b_n.p; const uint8_t* const bc = (const uint8_t*)B.b_cons.p;
    {   ScopedTimer tm(c, "bucket_sort");
        const uint64_t sgrid_env = 256 * 32;
        const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((n_sub + 3) / 4, sgrid_env));
        if (tag) hipLaunchKernelGGL((k_wave_sort<KW, FT>), dim3(grid), dim3(SORT_THREADS), 0, cur_stream(c), (const key_t*)keysA, keysA, bs, bn, (uint32_t)n_sub, O);
        else hipLaunchKernelGGL((k_wave_sort<KW, false>), dim3(grid), dim3(SORT_THREADS), 0, cur_stream(c), (const key_t*)keysA, keysA, bs, bn, (uint32_t)n_sub, O);
        CB_HIP(hipGetLastError());
    }
    {   ScopedTimer tm(c, "bucket_sort_big");                 // up to 2x the first tier: double-size wave network
        const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((list_cap + 3) / 4, 256 * 16));
        constexpr int KB = WaveCapHuge<KW>::KPL;
        if (tag) hipLaunchKernelGGL((k_wave_sort_big<KW, FT, KB>), dim3(grid), dim3(SORT_THREADS), 0, cur_stream(c), (const key_t*)keysA, keysA, bs, bn, (const uint32_t*)T.big_list, (const uint32_t*)T.big_count, O);
        else hipLaunchKernelGGL((k_wave_sort_big<KW, false, KB>), dim3(grid), dim3(SORT_THREADS), 0, cur_stream(c), (const key_t*)keysA, keysA, bs, bn, (const uint32_t*)T.big_list, (const uint32_t*)T.big_count, O);
        CB_HIP(hipGetLastError());
    }
    {   ScopedTimer tm(c, "bucket_sort_wg");                  // beyond one wave: workgroups of 4 waves, merges across waves through LDS
        static std::once_flag once;
        std::call_once(once, [&] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_wg_sort<KW, 4, K1, FT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(C1 * sizeof(key_t)));
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_wg_sort<KW, 4, K1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(C1 * sizeof(key_t)));
        });
        const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(list_cap, 256 * 4));
        if (tag) hipLaunchKernelGGL((k_wg_sort<KW, 4, K1, FT>), dim3(grid), dim3(256), C1 * sizeof(key_t), cur_stream(c), (const key_t*)keysA, keysA, bs, bn, (const uint32_t*)T.wg_list, (const uint32_t*)T.wg_count, O);
        else hipLaunchKernelGGL((k_wg_sort<KW, 4, K1, false>), dim3(grid), dim3(256), C1 * sizeof(key_t), cur_stream(c), (const key_t*)keysA, keysA, bs, bn, (const uint32_t*)T.wg_list, (const uint32_t*)T.wg_count, O);
        CB_HIP(hipGetLastError());
    }
    // split levels: level 1 takes the split list, level l > 1 the queue level l-1 filled; queue buffers alternate, the counters are used cyclically. Every level
    // is followed by the launch that sorts the pieces it listed (<= keys / 64 of them per level: the list is reused)
    const uint32_t deep_bits = gkc_tun().deep_bits;   // tests: few bits per level force many levels
    const uint64_t sort_cap = n_slots / (CAP1 / 4) + list_cap + 64 + (uint64_t)GIANT_MAX * MAX_SUB;     // an item of n keys lists <= 2 n / (cap1 / 2) + 1 runs and pieces
    CB_TRY(c->ensure(B.sitems, (size_t)sort_cap * sizeof(SortItem)));
    const unsigned deep_grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(list_cap, 256 * 8));
    auto counters_of = [&](int level) -> uint32_t* { return misc + 8 + 4 * (level % DEEP_COUNTERS); };     // [0] items for the next level [1] pieces to sort [2] ticket
    auto launch_deep = [&](int level) -> int {
        const uint32_t* roots = level == 1 ? (const uint32_t*)T.split_list : nullptr;
        const uint32_t* n_in = level == 1 ? (const uint32_t*)T.split_count : (const uint32_t*)counters_of(level - 1);
        const DeepItem* q_in = (const DeepItem*)B.q[(level - 1) & 1].p; DeepItem* q_out = (DeepItem*)B.q[level & 1].p;
        uint32_t* cn = counters_of(level);
        if (level > DEEP_FIXED) CB_HIP(hipMemsetAsync(cn, 0, 16, cur_stream(c)));                         // (the first ones were cleared with the whole block)
        const uint32_t bits_small = std::min<uint32_t>(deep_bits, DEEP_SMALL_BITS), bits_large = std::min<uint32_t>(deep_bits, (uint32_t)MAX_SUB_BITS);
        static std::once_flag once_deep; std::call_once(once_deep, [&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_deep_split<KW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)8 << MAX_SUB_BITS)); });
        hipLaunchKernelGGL((k_deep_split<KW>), dim3(deep_grid), dim3(DEEP_THREADS), (size_t)8 << bits_small, cur_stream(c), keysA, keysB, roots, bs, bn, bc, q_in, n_in, cn + 2, q_out, cn + 0,
                           (SortItem*)B.sitems.p, cn + 1, 2 * k + wb, bits_small, 0u, DEEP_SMALL_N, CAP1, O);
        hipLaunchKernelGGL((k_deep_split<KW>), dim3(std::min(deep_grid, 512u)), dim3(DEEP_THREADS), (size_t)8 << bits_large, cur_stream(c), keysA, keysB, roots, bs, bn, bc, q_in, n_in, cn + 3, q_out, cn + 0,
                           (SortItem*)B.sitems.p, cn + 1, 2 * k + wb, bits_large, DEEP_SMALL_N, 0xFFFFFFFFu, CAP1, O);
        const unsigned sgrid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((sort_cap + 3) / 4, 256 * 8));
        if (tag) hipLaunchKernelGGL((k_sort_items<KW, FT>), dim3(sgrid), dim3(SORT_THREADS), 0, cur_stream(c), keysA, (const key_t*)keysB, (const SortItem*)B.sitems.p, (const uint32_t*)(cn + 1), 0u, O);
        else hipLaunchKernelGGL((k_sort_items<KW, false>), dim3(sgrid), dim3(SORT_THREADS), 0, cur_stream(c), keysA, (const key_t*)keysB, (const SortItem*)B.sitems.p, (const uint32_t*)(cn + 1), 0u, O);
        CB_HIP(hipGetLastError());
        return GKC_OK;
    };
    {   ScopedTimer tm(c, "split_levels");
        {   // giants first: their pieces join level 1's sort list / queue
            const size_t gor_bytes = (size_t)GIANT_MAX * 32, tab_bytes = (size_t)GIANT_MAX * MAX_SUB * 4;
            CB_TRY(c->ensure(B.giant, gor_bytes + 2 * tab_bytes));
            CB_HIP(hipMemsetAsync(B.giant.p, 0, gor_bytes + tab_bytes, cur_stream(c)));                    // OR words + histograms (the cursors are written by k_giant_plan)
            GiantTables G{ (unsigned long long*)B.giant.p, (uint32_t*)((uint8_t*)B.giant.p + gor_bytes), (uint32_t*)((uint8_t*)B.giant.p + gor_bytes + tab_bytes),
                           (const uint32_t*)T.giant_list, (const uint32_t*)T.giant_count };
            uint32_t* cn = counters_of(1);
            hipLaunchKernelGGL((k_giant_or<KW>), dim3(GIANT_WGS, GIANT_MAX), dim3(GIANT_THREADS), 0, cur_stream(c), (const key_t*)keysA, G, bs, bn, (uint8_t*)B.cnt8.p, wb);
            hipLaunchKernelGGL((k_giant_hist<KW>), dim3(GIANT_WGS, GIANT_MAX), dim3(GIANT_THREADS), 0, cur_stream(c), (const key_t*)keysA, G, bs, bn, bc, 2 * k + wb, deep_bits, wb);
            hipLaunchKernelGGL((k_giant_plan<KW>), dim3(GIANT_MAX), dim3(GIANT_THREADS), 0, cur_stream(c), keysA, G, bs, bn, bc, (DeepItem*)B.q[1].p, cn + 0,
                               (SortItem*)B.sitems.p, cn + 1, 2 * k + wb, deep_bits, CAP1, O);
            hipLaunchKernelGGL((k_giant_scatter<KW>), dim3(GIANT_WGS, GIANT_MAX), dim3(GIANT_THREADS), 0, cur_stream(c), (const key_t*)keysA, keysB, G, bs, bn, bc, 2 * k + wb, deep_bits, wb);
            CB_HIP(hipGetLastError());
        }
        for (int level = 1; level <= DEEP_FIXED; level++) CB_TRY(launch_deep(level));
    }

    // --- dump: prefix over the per-bucket counts -> Count records
    uint64_t total_solid = 0;
    std::vector<uint64_t> ptot((size_t)(nb + 1) * 2);
    {   // "compact" times the KERNELS of the dump (two intervals: the prefix kernels, the gather kernels); the fetch of the prefix tables, the wait for it and the
        // allocation of the output block between them are host time — inside one interval they made the group look 2-3x its size wherever the host was busy (round 6)
        const uint32_t n_chunks = (uint32_t)((n_sub + SCAN2_CHUNK - 1) / SCAN2_CHUNK);
        if (n_chunks > (uint32_t)SCAN2_CHUNK) { B.release(); GKC_FAIL(c, GKC_ERR_ARG, "batch too large for the sub-bucket scan"); }
        CB_TRY(c->ensure(B.chunk, (size_t)std::max<uint32_t>(n_chunks, 1) * 16));
        uint64_t* ca = (uint64_t*)B.chunk.p; uint64_t* cb = ca + std::max<uint32_t>(n_chunks, 1);
        uint32_t h_misc[64];
        int level = DEEP_FIXED;
        for (uint64_t first = 1;; first = 0) {
            {   ScopedTimer tm(c, "compact", first);
            if (n_chunks) hipLaunchKernelGGL(k_scan2_chunks, dim3(n_chunks), dim3(1024), 0, cur_stream(c), (const uint32_t*)O.nd, (const uint32_t*)O.ns, n_sub, (uint64_t*)B.off_d.p, (uint64_t*)B.off_s.p, ca, cb);
            hipLaunchKernelGGL(k_scan2_totals, dim3(1), dim3(1024), 0, cur_stream(c), ca, cb, n_chunks, (uint64_t*)B.off_d.p, (uint64_t*)B.off_s.p, n_sub);
            if (n_chunks) hipLaunchKernelGGL(k_scan2_add, dim3(n_chunks), dim3(1024), 0, cur_stream(c), (uint64_t*)B.off_d.p, (uint64_t*)B.off_s.p, n_sub, (const uint64_t*)ca, (const uint64_t*)cb);
            hipLaunchKernelGGL(k_gather_u64, dim3((nb + 1 + 255) / 256), dim3(256), 0, cur_stream(c), (const uint64_t*)B.off_d.p, (const uint64_t*)B.off_s.p,
                               (const uint64_t*)B.pidx.p, nb + 1, (uint64_t*)B.ptot.p);
            }
            CB_HIP(hipGetLastError());
            CB_HIP(hipMemcpyAsync(ptot.data(), B.ptot.p, (size_t)(nb + 1) * 16, hipMemcpyDeviceToHost, cur_stream(c)));
            CB_HIP(hipMemcpyAsync(h_misc, B.misc.p, sizeof(h_misc), hipMemcpyDeviceToHost, cur_stream(c)));
            CB_HIP(hipStreamSynchronize(cur_stream(c)));
            if (h_misc[8 + 4 * (level % DEEP_COUNTERS)] == 0) break;                 // the last level launched left nothing: the counts are final
            level++;                                                               // pathological skew: one more level, then the prefix again
            if (level > 260) { B.release(); GKC_FAIL(c, GKC_ERR_HIP, "internal error: the split levels do not terminate"); }
            CB_TRY(launch_deep(level));
        }
        { std::lock_guard<std::mutex> lk(c->mu); c->pass_stats[pass].oversize_buckets += h_misc[2]; }
        if (dedupe) {                                                    // does merging identical records pay on this input? (it costs ~13 % of Stage B)
            unsigned long long dd[2]; memcpy(dd, h_misc + 40, 16);
            std::lock_guard<std::mutex> lk(c->mu);
            c->dedupe_in += dd[0]; c->dedupe_out += dd[1];
            c->pass_stats[pass].dedupe_kmers_in += dd[0]; c->pass_stats[pass].dedupe_keys_out += dd[1];
            if (dedupe_env != 1 && c->dedupe_in > 100000000ULL && (double)c->dedupe_out > 0.85 * (double)c->dedupe_in) c->dedupe_off = true;
            if (gkc_tun().verbose) fprintf(stderr, "[gkc] dedupe: %llu k-mers in the deduplicated bins -> %llu weighted keys (%.2fx)\n", dd[0], dd[1], dd[1] ? (double)dd[0] / (double)dd[1] : 0.0);
        }
        if (gkc_tun().verbose) fprintf(stderr, "[gkc] batch of %u partitions, %llu sub-buckets: %u in the double-size tier, %u in the workgroup tier, %u split (%d levels)\n",
                                           nb, (unsigned long long)n_sub, h_misc[0], h_misc[1], h_misc[2], level);
        total_solid = ptot[2 * nb + 1];
        constexpr int OW = (KW == 1) ? 2 : 4;
        void* out = c->dalloc((size_t)std::max<uint64_t>(total_solid, 1) * OW * 8);
        if (!out) { B.release(); return GKC_ERR_NOMEM; }
        { std::lock_guard<std::mutex> lk(c->mu); outputs.push_back(out); }
        if (n_sub) {
            ScopedTimer tm(c, "compact", 0);
            const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((n_sub + 255) / 256, 256 * 16));
            hipLaunchKernelGGL((k_gather_counts<KW>), dim3(grid), dim3(GATHER_THREADS), 0, cur_stream(c), (const key_t*)keysA, (const uint8_t*)B.cnt8.p, (const uint32_t*)B.cnt.p,
                               bs, bn, (const uint32_t*)O.nd, (const uint64_t*)B.off_s.p, (uint32_t)n_sub, cap3, c->amin, c->amax, O.all_solid, (uint64_t*)out, bc, 2 * k - drop);
            if (h_misc[2]) {                                                       // the split sub-buckets: records at their pieces' heads
                const uint32_t n_roots = h_misc[2];
                const uint64_t chunks_cap = n_slots / ROOT_CHUNK + n_roots + 1;
                CB_TRY(c->ensure(B.rbase, ((size_t)n_roots + 1) * 4)); CB_TRY(c->ensure(B.rroot, (size_t)chunks_cap * 4)); CB_TRY(c->ensure(B.rcnt, ((size_t)chunks_cap + 1) * 4));
                RootTables R{ (uint32_t*)B.rbase.p, (uint32_t*)B.rroot.p, (uint32_t*)B.rcnt.p, misc + 4 };
                const unsigned rgrid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((chunks_cap + 3) / 4, 256 * 8));
                hipLaunchKernelGGL(k_root_chunks, dim3(1), dim3(ROOT_THREADS), 0, cur_stream(c), (const uint32_t*)T.split_list, (const uint32_t*)T.split_count, bn, R);
                hipLaunchKernelGGL(k_root_count, dim3(rgrid), dim3(256), 0, cur_stream(c), (const uint8_t*)B.cnt8.p, (const uint32_t*)B.cnt.p, bs, bn, (const uint32_t*)T.split_list, R, c->amin, c->amax, O.all_solid);
                hipLaunchKernelGGL(k_root_scan, dim3(1), dim3(ROOT_THREADS), 0, cur_stream(c), R);
                hipLaunchKernelGGL((k_root_write<KW>), dim3(rgrid), dim3(256), 0, cur_stream(c), (const key_t*)keysA, (const uint8_t*)B.cnt8.p, (const uint32_t*)B.cnt.p, bs, bn, (const uint32_t*)T.split_list, R,
                                   (const uint64_t*)B.off_s.p, c->amin, c->amax, O.all_solid, (uint64_t*)out, bc, 2 * k - drop);
            }
            CB_HIP(hipGetLastError());
        }
        // streamed results: the batch's records go to the host sink on the copy stream while the lanes count the next batches — packed (7 bytes per record
        // instead of 16, expanded in place by host threads: gkc_sink.hip) when the keys are 8 bytes
        const uint8_t* h_base = nullptr; hipEvent_t landed = nullptr; const void* sink_batch = nullptr;
        bool room = false;
        if (c->sink && total_solid) {
            const uint64_t bytes = total_solid * OW * 8;
            std::lock_guard<std::mutex> lk(c->mu);
            if (c->sink_used + bytes > c->sink_cap) c->sink_overflow = true;          // the records stay on the device (gkc_partition_counts still serves them)
            else { h_base = (const uint8_t*)c->sink + c->sink_used; c->sink_used += bytes; room = true; }
        }
        if (room && gkc_sink_packed(c) && !gkc_sink_host_behind(c, total_solid)) {
            std::vector<uint64_t> solid_prefix(nb + 1);
            for (uint32_t i = 0; i <= nb; i++) solid_prefix[i] = ptot[2 * i + 1];
            sink_batch = gkc_sink_send_packed(c, out, (const uint64_t*)B.ptot.p, solid_prefix, (uint8_t*)h_base);      // (synchronizes the lane's stream)
        }
        CB_HIP(hipStreamSynchronize(cur_stream(c)));
        if (room && !sink_batch) {
            const uint64_t bytes = total_solid * OW * 8;
            std::lock_guard<std::mutex> lk(c->mu);
            if (gkc_sink_packed(c)) {                                  // a packing context whose batch travels plain: its bytes belong to what the library queued on the link
                c->sink_wire_bytes += bytes;
                if (gkc_tun().sink_debug || gkc_tun().verbose) fprintf(stderr, "[gkc sink] a batch of %llu records travels unpacked: %s\n", (unsigned long long)total_solid, gkc_sink_last_refusal());
            }
            {
                if (hipEventCreateWithFlags(&landed, hipEventDisableTiming) == hipSuccess &&
                    hipMemcpyAsync((void*)h_base, out, bytes, hipMemcpyDeviceToHost, c->copy_stream) == hipSuccess &&
                    hipEventRecord(landed, c->copy_stream) == hipSuccess) c->landed_events.push_back(landed);
                else { (void)hipGetLastError(); if (landed) (void)hipEventDestroy(landed); landed = nullptr; h_base = nullptr; c->sink_overflow = true; }
            }
        }
        {   std::lock_guard<std::mutex> lk(c->mu);
            for (uint32_t i = 0; i < nb; i++) {
                Dataset& D = c->datasets[(size_t)pass * c->nb_partitions + batch_parts[i]];
                const uint64_t s0 = ptot[2 * i + 1], s1 = ptot[2 * (i + 1) + 1];
                D.d_counts = (const uint8_t*)out + s0 * OW * 8;
                D.h_counts = h_base ? h_base + s0 * OW * 8 : nullptr; D.landed = landed; D.sink_batch = sink_batch;
                D.n_solid = s1 - s0; D.n_distinct = ptot[2 * (i + 1)] - ptot[2 * i]; D.n_kmers = part_keys[batch_parts[i]]; D.done = true;
                c->pass_stats[pass].kmers_nb_distinct += D.n_distinct; c->pass_stats[pass].kmers_nb_solid += D.n_solid;
            }
        }
        c->cv_done.notify_all();
    }
    B.release();
#undef CB_TRY
#undef CB_HIP
    return GKC_OK;
}

// pass / segments / lane0: the pass this Stage B counts, its segments and the stream of its first lane. Called in line by gkc_finish_pass (the context's own pass,
// segment list and stream) or on the worker thread of gkc_finish_pass_async with the DETACHED state of the pass (c->b_*) and a stream of its own, while the caller
// may already run Stage A of the next pass on the context's stream: nothing below reads c->pass, c->segments or stats_now().
int gkc_count_pass(gkc_ctx* c, const uint32_t pass, const std::vector<Segment>& segments, hipStream_t lane0, double reserve_bytes)
{
    const uint32_t Pn = c->nb_partitions;
    const uint32_t n_seg = (uint32_t)segments.size();
    c->t_stage_b0 = std::chrono::steady_clock::now();
    {   // a pass counted again (a retry after GKC_ERR_NOMEM, or gkc_finish_pass called twice) starts from a clean slate: what the
        // batches of the failed attempt added to the histogram, to the counters and to the result list must not be counted twice
        std::lock_guard<std::mutex> lk(c->mu);
        auto it = c->pass_outputs.find(pass);
        if (it != c->pass_outputs.end()) { for (void* p : it->second) c->dfree(p); it->second.clear(); }
        for (uint32_t p = 0; p < Pn; p++) c->datasets[(size_t)pass * Pn + p] = Dataset();
        gkc_stats& S = c->pass_stats[pass]; S.kmers_nb_distinct = 0; S.kmers_nb_solid = 0; S.oversize_buckets = 0; S.dedupe_kmers_in = 0; S.dedupe_keys_out = 0;
        // ... and the host sink starts over as well: the failed attempt's copies are drained, its records are overwritten
        if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
        for (hipEvent_t e : c->landed_events) (void)hipEventDestroy(e);
        gkc_sink_reset(c);
        c->landed_events.clear(); c->sink_used = 0; c->sink_overflow = false;
        GKC_HIP(c, hipMemsetAsync(c->histo_of(pass), 0, ((size_t)c->histo_max + 1) * 8, lane0));
    }
    std::vector<uint64_t> part_keys(Pn, 0);
    for (const Segment& s : segments) for (uint32_t p = 0; p < Pn; p++) part_keys[p] += s.nkmers[p];
    // device copy of the segment table
    DevBuf d_recptr, d_recoff;
    std::vector<const uint8_t*> ptrs(std::max<uint32_t>(n_seg, 1), nullptr);
    std::vector<uint64_t> offs((size_t)std::max<uint32_t>(n_seg, 1) * (Pn + 1), 0);
    for (uint32_t s = 0; s < n_seg; s++) {
        ptrs[s] = (const uint8_t*)segments[s].d_records;
        memcpy(&offs[(size_t)s * (Pn + 1)], segments[s].rec_off.data(), (size_t)(Pn + 1) * 8);
    }
    GKC_TRY(c->ensure(d_recptr, ptrs.size() * sizeof(void*)));
    int rc = c->ensure(d_recoff, offs.size() * 8);
    if (rc != GKC_OK) { d_recptr.release(); return rc; }
    hipError_t e1 = hipMemcpyAsync(d_recptr.p, ptrs.data(), ptrs.size() * sizeof(void*), hipMemcpyHostToDevice, lane0);      // (on the pass's own stream: a plain hipMemcpy would
    hipError_t e2 = hipMemcpyAsync(d_recoff.p, offs.data(), offs.size() * 8, hipMemcpyHostToDevice, lane0);                  //  wait for whatever Stage A of the next pass has queued)
    if (e1 == hipSuccess && e2 == hipSuccess) e1 = hipStreamSynchronize(lane0);
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
    int lanes = gkc_tun().lanes;
    if (lanes < 1) lanes = 1;
    if (lanes > 4) lanes = 4;
    if (total_keys < 50000000ULL || c->key_budget) lanes = 1;               // small inputs (and the tests' tiny forced budgets): one lane
    bool tight = false;                                                       // memory is running out: the extra lanes retire, one lane finishes the pass
    const double avail0 = [&] {                                              // memory this pass may use: free now + blocks parked in the caching allocator
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = (size_t)8 << 30;
        return std::max(0.0, (double)(free_b + c->pool.cached_bytes) - reserve_bytes);      // (overlapped passes: what Stage A of the next pass will allocate beside this Stage B)
    }();
    const size_t cap_env = (size_t)gkc_tun().batch_keys;
    // Few, large batches: every batch ends with the drain of ~12 kernels (the expand kernels run one 6 ms workgroup per partition) and
    // eight host round trips; 8 -> 4 batches per 1.2e10 keys: 320 -> 304 ms. Equal shares, a whole number of batches per lane.
    // With a host sink (streamed results) the batches are three times smaller: the first records start over PCIe sooner and the copy that is left
    // when the last batch has been counted is shorter (1e8 reads, abundance-min 2: 431 -> ms per step with everything landed; profiles/r02_*)
    const size_t cap_default = c->key_words == 1 ? (size_t)3200000000ULL : (size_t)1600000000ULL;
    const size_t cap_mode = c->sink ? cap_default / 3 : cap_default;       // (a host sink: three batches per lane keep the link busy)
    // gkc_set_batch_keys is an UPPER bound: never above the library's plan for the mode; GKC_BATCH_KEYS (developer switch) replaces the plan outright
    const size_t cap = cap_env ? cap_env : c->batch_cap ? std::min(cap_mode, std::max<size_t>(c->batch_cap, (size_t)1 << 24)) : cap_mode;   // the same with one lane or two
    uint64_t done_keys = 0, done_solid = 0;                                  // this pass: finished batches (keys, resident records) (guarded by plan_mu)
    // budget for a given solid-per-key ratio d. Deterministic in (free memory rounded to GB, total keys, d rounded up to 0.05): every
    // pass of a context plans the same sizes, so from the second pass on all blocks are parked already.
    const double avail_q = std::floor(avail0 / 1e9) * 1e9;
    int plan_lanes = (c->key_budget || total_keys < 50000000ULL) ? 1 : std::max(2, lanes);                 // (more than two lanes: planned for what will really run)                                  // one lane gets the same batches as two would (same blocks whichever way a pass runs) ...
    auto plan_budget = [&](double d) -> size_t {
        const double work = (double)plan_lanes * (double)work_per_key;
        const double dq = std::min(1.0, std::ceil(1.05 * d / 0.05) * 0.05);
        const double mem = (0.95 * avail_q - (double)total_keys * (double)rec_bytes * dq) / work;
        const uint64_t bmem = mem > (double)((size_t)1 << 20) ? (uint64_t)mem : ((uint64_t)1 << 20);
        if (c->nb_passes > 1) {
            // several passes: every pass plans the SAME batch size — half the cap plus a rounded partition — whatever its share of the k-mers (minimizer % nb_passes
            // does not cut them evenly: 5.2e9 and 6.8e9 keys for the two passes of 1e8 reads). Equal shares per pass gave every pass its own block sizes: the allocator was
            // trimmed and refilled at every pass, 4 s of hipMalloc for 0.15 s of counting (tools/twopass_probe.py).
            // (a small pass — the tests' — is one batch of its own size, rounded up to a power of two so that passes of similar size still plan alike)
            uint64_t mp = 1; while (mp < max_part) mp <<= 1;
            uint64_t tk = 1; while (tk < total_keys) tk <<= 1;
            return (size_t)(std::min<uint64_t>(std::min<uint64_t>(bmem, cap / 2), tk) + mp);
        }
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
            if (gkc_tun().pool_debug) fprintf(stderr, "[gkc plan] lane %d: budget %.3e does not fit (%.3e): done_solid %.3e committed %.1f GB\n", lane, (double)b, (double)fits, (double)done_solid, committed / 1e9);
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
    std::vector<void*>* outputs_p; { std::lock_guard<std::mutex> lk(c->mu); outputs_p = &c->pass_outputs[pass]; }      // (std::map nodes stay where they are)
    std::vector<void*>& outputs = *outputs_p;
    std::mutex plan_mu; uint32_t next_p = 0; int first_rc = GKC_OK;
    uint32_t lane_batches[4] = { 0, 0, 0, 0 }; uint64_t keys_left = total_keys;
    const size_t sink_first_div = gkc_tun().sink_first_div;
    auto carve = [&](std::vector<uint32_t>& batch, int lane) -> bool {   // next batch of consecutive partitions; false when nothing is left
        std::lock_guard<std::mutex> lk(plan_mu);
        batch.clear();
        if (first_rc != GKC_OK) return false;
        inflight[lane] = 0;
        size_t budget = budget_now(lane);
        if (tight && lane != 0) return false;
        if (probe_pending) { probe_pending = false; budget = probe_keys; }      // the pass's first batch is the small probe batch, every pass (same batches, same blocks)
        // streamed results: the link idles until the first batch has been counted and packed — the first batch of every lane is a quarter of the others (same
        // working buffers: they are sized for the budget), the copies start ~25 ms sooner
        // — and the last one as well: what is left when the last copy has landed is the expansion of the last batch on the host
        // Round 6: a RAMP instead of one small batch — 1/4, 1/2 of the budget, then whole ones. Stage B makes packed records ~2.6x faster than the link takes
        // them, so a batch twice the one before is ready before the link has drained; with one quarter batch per lane followed by whole ones (round 5) the link
        // sat idle between the end of the two small copies and the arrival of the first whole batches (GKC_SINK_DEBUG timeline: 13 ms at 5e7 reads).
        else if (c->sink && sink_first_div > 1 && !c->key_budget) {
            const size_t small = std::max<size_t>(budget / 4, (size_t)1 << 20), tail = small * (size_t)lanes;      // keys kept back for the small last batches
            const size_t ramp = sink_first_div >> std::min<uint32_t>(lane_batches[lane], 31u);                     // 4, 2, 1 (GKC_SINK_FIRST_DIV: the first divisor; measured 1/4: 483.8 ms per step, 1/8: 490.4, 1/16: 491.8; no ramp: 498.3)
            if (ramp > 1) budget = std::max<size_t>(budget / ramp, (size_t)1 << 20);
            else if (keys_left <= tail + small / 2) budget = small;
            else if (keys_left < budget + tail) budget = std::max<size_t>(small, (size_t)(keys_left - tail));
        }
        lane_batches[lane]++;
        uint64_t acc = 0;
        while (next_p < Pn) {
            const uint32_t p = next_p;
            if (part_keys[p] == 0) {                           // nothing to count (e.g. a partition another rank owns): an empty, finished dataset
                {   std::lock_guard<std::mutex> lk2(c->mu);       // (c->mu guards the datasets gkc_wait_partition looks at; plan_mu only the batch plan)
                    Dataset& D = c->datasets[(size_t)pass * Pn + p];
                    D.d_counts = nullptr; D.n_solid = 0; D.n_distinct = 0; D.n_kmers = 0; D.done = true;
                }
                c->cv_done.notify_all();
                next_p++; continue;
            }
            if (!batch.empty() && acc + part_keys[p] > budget) break;
            batch.push_back(p); acc += part_keys[p]; next_p++;
        }
        inflight[lane] = (double)acc * per_key_now();
        keys_left -= std::min<uint64_t>(keys_left, acc);
        return !batch.empty();
    };
    auto lane_main = [&](hipStream_t st, int lane) {
        (void)hipSetDevice(c->device);
        const hipStream_t tl_before = tl_stream_;
        tl_stream_ = st;
        std::vector<uint32_t> batch;
        while (carve(batch, lane)) {
            const int r = (c->key_words == 1) ? count_batch<1, 2>(c, pass, segments, batch, part_keys, segs, outputs) : count_batch<2, 4>(c, pass, segments, batch, part_keys, segs, outputs);
            std::lock_guard<std::mutex> lk(plan_mu);
            inflight[lane] = 0;
            if (r != GKC_OK) { if (first_rc == GKC_OK) first_rc = r; break; }
            for (uint32_t p : batch) { const Dataset& D = c->datasets[(size_t)pass * Pn + p]; done_keys += D.n_kmers; done_solid += D.n_solid; }
        }
        (void)hipStreamSynchronize(st);
        tl_stream_ = tl_before;
    };
    (void)hipStreamSynchronize(lane0);                                       // the table uploads (and, in line, Stage A on the same stream) are complete before the lanes start
    // d not known yet and the memory may bind: count a small PROBE batch first (the first partitions holding ~0.4 % of the keys) and take
    // its ratio. The context keeps that first estimate (until the configuration changes), so every later pass plans the same sizes.
    // In later passes the same small batch is simply the first one in the queue (it runs beside the other lane's first batch).
    probe_pending = !c->key_budget && plan_budget(1.0) < plan_budget(1e-9);
    if (probe_pending && c->d_hint <= 0) {
        std::vector<uint32_t> batch;
        fixed_budget = probe_keys; c->slots_hint = 0;
        if (carve(batch, 0)) {
            const int r = (c->key_words == 1) ? count_batch<1, 2>(c, pass, segments, batch, part_keys, segs, outputs) : count_batch<2, 4>(c, pass, segments, batch, part_keys, segs, outputs);
            inflight[0] = 0; last_b[0] = 0;
            if (r != GKC_OK) { d_recptr.release(); d_recoff.release(); return r; }
            for (uint32_t p : batch) { const Dataset& D = c->datasets[(size_t)pass * Pn + p]; done_keys += D.n_kmers; done_solid += D.n_solid; }
            if (done_keys) c->d_hint = std::max(1e-6, (double)done_solid / (double)done_keys);
        }
    }
    fixed_budget = c->d_hint > 0 ? plan_budget(c->d_hint) : plan_budget(1.0);   // without a ratio the memory does not bind even at d = 1
    if (gkc_tun().pool_debug) fprintf(stderr, "[gkc plan] avail %.1f GB, keys %.3e, d_hint %.4f, lanes %d, budget %.3e\n", avail0 / 1e9, (double)total_keys, c->d_hint, lanes, (double)fixed_budget);
    // another batch size than the last pass planned (another input, a host sink set or dropped, another solidity ratio): the blocks parked in the allocator have the
    // wrong sizes — keeping them would make every new block a failed hipMalloc followed by frees, one parked block at a time (seen: 1.5 s for a 0.26 s pass)
    if (c->last_plan_budget && (fixed_budget > c->last_plan_budget + c->last_plan_budget / 10 || fixed_budget + fixed_budget / 10 < c->last_plan_budget)) c->pool.trim();      // (the allocator reuses a block up to 25 % larger than asked)
    c->last_plan_budget = fixed_budget;
    if (!c->key_budget && fixed_budget < 250000000ULL) {                       // ... unless there is little room: one lane, larger batches
        lanes = 1; plan_lanes = 1; fixed_budget = c->d_hint > 0 ? plan_budget(c->d_hint) : plan_budget(1.0);
    }
    {   uint64_t nonempty = 0; for (uint64_t v : part_keys) nonempty += v != 0;
        const uint64_t avg_part = std::max<uint64_t>(total_keys / std::max<uint64_t>(nonempty, 1), 1);
        const uint64_t hint = (uint64_t)fixed_budget + ((uint64_t)fixed_budget / avg_part + 2) * ((3ull << MAX_SUB_BITS) + PART_ALIGN);
        c->slots_hint = c->key_budget ? 0 : (hint + PART_ALIGN - 1) / PART_ALIGN * PART_ALIGN;
    }
    for (int l = 1; l < lanes; l++)
        if (!c->lane_streams[l - 1] && hipStreamCreateWithFlags(&c->lane_streams[l - 1], hipStreamNonBlocking) != hipSuccess) { c->lane_streams[l - 1] = nullptr; lanes = l; break; }
    {
        std::vector<std::thread> extra;
        for (int l = 1; l < lanes; l++) extra.emplace_back(lane_main, c->lane_streams[l - 1], l);
        lane_main(lane0, 0);
        for (auto& t : extra) t.join();
    }
    rc = first_rc;
    c->cv_done.notify_all();
    if (gkc_tun().pool_debug) fprintf(stderr, "[gkc pool] mallocs %llu failed %llu trims %llu, %.1f ms in hipMalloc, cached %.2f GB\n", (unsigned long long)c->pool.n_malloc,
                                          (unsigned long long)c->pool.n_fail, (unsigned long long)c->pool.n_trim, c->pool.malloc_ms, (double)c->pool.cached_bytes / 1e9);
    d_recptr.release(); d_recoff.release();
    return rc;
}

// explicit result checksum entry (used by the C-ABI)
int gkc_result_checksum_impl(gkc_ctx* c, uint64_t* checksum, uint64_t* sum_abundance)
{
    DevBuf d; GKC_TRY(c->ensure(d, 16));
    GKC_HIP(c, hipMemsetAsync(d.p, 0, 16, cur_stream(c)));
    // datasets of one Stage-B batch lie one behind the other: one launch per contiguous run (a handful per pass), not one per dataset
    const size_t rb = c->key_words == 1 ? 16 : 32;
    const uint8_t* run = nullptr; uint64_t run_n = 0;
    auto flush = [&]() {
        if (!run_n) return;
        const unsigned grid = (unsigned)std::min<uint64_t>((run_n + 255) / 256, 8192);
        if (c->key_words == 1) hipLaunchKernelGGL((k_result_checksum<1>), dim3(grid), dim3(256), 0, cur_stream(c), (const uint64_t*)run, run_n, (unsigned long long*)d.p);
        else                   hipLaunchKernelGGL((k_result_checksum<2>), dim3(grid), dim3(256), 0, cur_stream(c), (const uint64_t*)run, run_n, (unsigned long long*)d.p);
        run_n = 0;
    };
    for (const Dataset& D : c->datasets) {
        if (!D.done || !D.n_solid) continue;
        if (run_n && (const uint8_t*)D.d_counts == run + run_n * rb) { run_n += D.n_solid; continue; }
        flush();
        run = (const uint8_t*)D.d_counts; run_n = D.n_solid;
    }
    flush();
    uint64_t h[2];
    hipError_t e = hipMemcpyAsync(h, d.p, 16, hipMemcpyDeviceToHost, cur_stream(c));
    if (e == hipSuccess) e = hipStreamSynchronize(cur_stream(c));
    d.release();
    if (e != hipSuccess) GKC_FAIL(c, GKC_ERR_HIP, "result checksum failed: %s", hipGetErrorString(e));
    *checksum = h[0]; *sum_abundance = h[1];
    return GKC_OK;
}
