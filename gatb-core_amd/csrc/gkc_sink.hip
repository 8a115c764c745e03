// gkc_sink.hip — streamed results, PACKED on the wire (gkc_set_host_sink, k <= 31): SURVEY §8(d) ends the clock when the last partition's Count[] is in host
// memory, and at abundance-min 1 that is 16 bytes per distinct k-mer over PCIe — 58 GB per 10^8 reads, 1.1 s at the 52 GB/s the link gives, five times the
// counting itself. The records of a partition are ascending keys with small abundances, so what crosses the link is
//     per block of PK_BLOCK records: the first key (8 bytes), then per record 6 bytes of key DELTA + 1 byte of abundance            = 7 bytes instead of 16
//     — and at abundance-min 1, where most records are the singletons of sequencing errors (84 % of the 30x input), the abundance byte travels only for the
//     records whose abundance is NOT 1: 6 bytes of delta + 1 bit in the block's bitmap + a byte in the batch's abundance stream for those = 6.3 bytes (PK6)
//     — and since round 6 the deltas of that format are bit-packed at the width of the largest delta of their sub-block of 128 records, one width byte per sub-block,
//     8- and 16-byte keys alike (PKV below): 5.8 bytes per record at k = 31, 14.5 of 32 at k = 63 (10^8 reads), no key escapes
// and library threads on the host expand it into the exact in-memory layout of Kmer<span>::Count ({u64 value; i32 abundance; pad}, Abundance.hpp:68-129) at its
// place in the caller's sink: what gkc_wait_partition hands out is byte for byte what the unpacked copy would have been (tests: the sink against
// gkc_partition_counts). Rare values leave through an exception list (record index, value): a delta of 2^48-1 or more (the delta field then holds the escape
// 0xFFFFFFFFFFFF), an abundance of 255 or more (escape 255). The reference's sink this stands in for is CountProcessorDump -> BagCache -> CollectionHDF5Patch
// (CountProcessorDump.hpp:148-152): the consumer of whole Count[] blocks.
//   device   k_pack_counts: one workgroup per block (blocks never straddle partitions, each has its own 16-byte-aligned 57344-byte slot), records -> 7-byte
//            entries staged through LDS and written as 16-byte words; reads the batch's Count[] once, writes 0.44x of it
//   link     ONE copy per Stage-B batch on the copy stream: [block bases | payload] then the exception entries, into a page-locked staging buffer of the library
//   host     a pool of unpack threads: the first to reach a batch waits for its copy (HIP event) and sorts the exceptions, then all of them take blocks off an
//            atomic counter (a block is independent of every other: base key + running sum of its deltas) and write the records with non-temporal 16-byte stores;
//            the last one marks the batch landed (gkc_wait_partition / gkc_finish_pass wait for that)
#include "gkc_common.hpp"
#include "gkc_device.hpp"
#include <algorithm>
#include <utility>
#include <atomic>
#include <deque>
#include <immintrin.h>
#include <pthread.h>
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>

namespace {
constexpr uint32_t PK_BLOCK = 8192, PK_THREADS = 256;
// entry = [key delta : W - 1 bytes][abundance : 1 byte], W = 7 where the partitions are dense (10^8 reads at abundance-min 1: 8.9e5 records per partition, 0.03 % of the
// deltas do not fit 48 bits), W = 8 where they are sparse (abundance-min 2: 1.4e5 per partition, 1 % would escape — and every escape is a sorted-list lookup on the host)
constexpr uint64_t pk_slot(int W) { return (uint64_t)PK_BLOCK * (uint64_t)W; }      // 57344 / 65536 bytes: multiples of 16
constexpr uint64_t pk_esc(int W) { return (1ull << (8 * (W - 1))) - 1ull; }
constexpr uint64_t PK_KEY_EXC = 1ull << 63;
constexpr uint64_t PK_DENSE = 300000;                                   // records per partition from which 6-byte deltas are used
// PKV (reported as width 6; round 6, it replaces the fixed 6-byte deltas of rounds 3-5): the deltas travel bit-packed at the width of the largest delta of their
// SUB-BLOCK of PKV_SUB = 128 records (64 sub-blocks per block of PK_BLOCK records, one width byte each in the header). Canonical k-mers thin out towards the top of
// the key space (density 2 (1 - x)): the gaps of a partition of 8.9e5 records average 2^42 and range from 2^41 at the bottom to 2^50 in its last blocks, so one width
// for all either wastes bits at the bottom or escapes at the top (48 bits + 0.03 % escapes before); the largest of 128 exponential gaps is 2.3 bits above their mean
// (of 8192: 3.2 + what the clusters of k-mers that start with their minimizer add): ~44.6 bits on average and NO key escapes. 128 W bits = 16 W bytes: every
// sub-block starts on a byte, a thread packs 8 records into W bytes. A block's payload = its sub-blocks back to back + a bitmap of PK_BLOCK bits (abundance != 1), at
// an offset of the batch's payload stream the block's workgroup reserves (u32 in 16-byte units in the header); the abundance bytes of the flagged records, in record
// order, sit in the batch's abundance stream from the block's offset on (u32 per block in the header) — both reserved by ONE atomic each: the order of the blocks in
// the streams is whatever it came out as. W > 56 (a host extraction reads 8 bytes at any bit offset: 7 + W <= 63) is sent as W = 64.
constexpr uint32_t PKV_CHUNK = 2048, PKV_SUB = 128, PKV_NSUB = PK_BLOCK / PKV_SUB;                              // records per pack iteration (256 threads x 8); per width; widths per block
constexpr uint64_t PKV_BITMAP = PK_BLOCK / 8, PKV_BLOCK_MAX = (uint64_t)PK_BLOCK * 8 + PKV_BITMAP;              // worst case of a block's payload (W = 64)
// PKV for 16-byte keys (reported as width 14; round 6): the same layout with 128-bit deltas — a sub-block's width W is 0..128 bits, a record's W bits are the low
// min(W, 64) bits of its delta followed by the W - 64 high ones; bases are 16 bytes per block. k = 63, 5.6e5 records per partition: gaps of 2^107 on average, 13.7 bytes
// per record where the fixed entries carry 15 or 16 (+ escapes).
constexpr uint64_t PKV2_BLOCK_MAX = (uint64_t)PK_BLOCK * 16 + PKV_BITMAP;
constexpr uint64_t pk_slot_of(int width) { return width == 6 ? PKV_BLOCK_MAX : width == 14 ? PKV2_BLOCK_MAX : pk_slot(width); }
// 16-byte keys (k >= 32; round 4, second session): the same scheme on 32-byte Count records {u128 value; i32 abundance; 12 bytes of padding} (Abundance.hpp:68-129 with
// LargeInt<2>): per block the first key (16 bytes), per record [key delta : 15 or 16 bytes][abundance : 1 byte] = widths 16 / 17 instead of 32. A partition of 5.6e5 records in a
// 126-bit key space has deltas of ~2^107 — but canonical k-mers thin out towards the top of the key space (density 2 (1 - x)), and with 14-byte deltas 0.1-0.4 % of them escaped
// (1e6 exception entries per batch of 2.6e8 records: measured, the batches fell back to plain copies): 15 bytes where the partitions are dense (a delta of 2^120 - 1 or more —
// a few per batch — escapes through TWO exception entries, low and high word), the full 16 bytes where they are sparse (no key escape at all).
constexpr uint64_t PK_KEY_EXC_HI = (1ull << 63) | (1ull << 62);
constexpr uint64_t PK2_DENSE = 100000;                                 // records per partition from which 15-byte deltas are used
}

struct PackPlan { const uint32_t* blk_first; /* [nb + 1] first block slot of every partition of the batch */ const uint64_t* ptot; /* [2 (nb + 1)] (distinct, solid) prefixes */ uint32_t nb; };

template <int W>
__global__ __launch_bounds__(PK_THREADS) void k_pack_counts(const uint64_t* __restrict__ recs, PackPlan P, uint64_t* __restrict__ bases, uint8_t* __restrict__ payload,
                                                            uint64_t* __restrict__ exc, unsigned long long* __restrict__ n_exc, uint32_t exc_cap)
{
    constexpr uint64_t PK_ESC = pk_esc(W), PK_SLOT = pk_slot(W);
    __shared__ __attribute__((aligned(16))) uint8_t s_out[PK_THREADS * W + 16];
    __shared__ uint32_t s_p;
    const uint32_t g = blockIdx.x, t = threadIdx.x;
    if (t == 0) {                                               // partition of block slot g: the largest p with blk_first[p] <= g
        uint32_t lo = 0, hi = P.nb;
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (P.blk_first[mid] <= g) lo = mid; else hi = mid; }
        s_p = lo;
    }
    __syncthreads();
    const uint32_t p = s_p, j = g - P.blk_first[p];
    const uint64_t s1 = P.ptot[2 * (p + 1) + 1], r0 = P.ptot[2 * p + 1] + (uint64_t)j * PK_BLOCK;
    const uint32_t n = (uint32_t)min((uint64_t)PK_BLOCK, s1 - r0);
    if (t == 0) bases[g] = recs[2 * r0];
    uint8_t* dstp = payload + (uint64_t)g * PK_SLOT;
    for (uint32_t i0 = 0; i0 < n; i0 += PK_THREADS) {
        const uint32_t i = i0 + t;
        uint64_t d = 0; uint32_t ab8 = 0;
        if (i < n) {
            const ulonglong2 me = *reinterpret_cast<const ulonglong2*>(recs + 2 * (r0 + i));
            const uint64_t prev = i ? recs[2 * (r0 + i - 1)] : me.x;
            d = me.x - prev;
            if (d >= PK_ESC) {
                const unsigned long long e = atomicAdd(n_exc, 1ull);
                if (e < exc_cap) { exc[2 * e] = PK_KEY_EXC | (r0 + i); exc[2 * e + 1] = me.x; }
                d = PK_ESC;
            }
            const uint32_t ab = (uint32_t)me.y;
            ab8 = ab;
            if (ab >= 255u) {
                const unsigned long long e = atomicAdd(n_exc, 1ull);
                if (e < exc_cap) { exc[2 * e] = r0 + i; exc[2 * e + 1] = ab; }
                ab8 = 255u;
            }
        }
        uint8_t* o = s_out + W * t;
#pragma unroll
        for (int b = 0; b < W - 1; b++) o[b] = (uint8_t)(d >> (8 * b));
        o[W - 1] = (uint8_t)ab8;
        __syncthreads();
        if (t < PK_THREADS * W / 16) reinterpret_cast<uint4*>(dstp + (uint64_t)i0 * W)[t] = reinterpret_cast<const uint4*>(s_out)[t];      // 1792 / 2048 bytes = 112 / 128 x 16
        __syncthreads();
    }
}

__global__ __launch_bounds__(PK_THREADS) void k_pack_counts6(const uint64_t* __restrict__ recs, PackPlan P, uint64_t* __restrict__ bases, uint32_t* __restrict__ cb_off,
                                                             uint32_t* __restrict__ pay_off16, uint8_t* __restrict__ wbits /* [nblk][PKV_NSUB] */,
                                                             uint8_t* __restrict__ payload, unsigned long long* __restrict__ pay_cursor /* bytes */,
                                                             uint8_t* __restrict__ cb_stream, unsigned long long* __restrict__ cb_cursor,
                                                             uint64_t* __restrict__ exc, unsigned long long* __restrict__ n_exc, uint32_t exc_cap)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_out[PK_THREADS * 64];                  // one chunk's entries: 16 sub-blocks of 16 W bytes
    __shared__ uint64_t s_key[PKV_CHUNK + 1];                                                 // the chunk's keys, [0] = the key before the chunk
    __shared__ __attribute__((aligned(16))) unsigned long long s_bits[PK_BLOCK / 64];
    __shared__ __attribute__((aligned(16))) uint8_t s_cb[PK_BLOCK];
    __shared__ unsigned long long s_wmax[PKV_NSUB];                                           // largest delta of every sub-block
    __shared__ uint32_t s_w[PKV_NSUB], s_off[PKV_NSUB + 1];                                   // its width, the byte offset of its entries in the block's payload
    __shared__ uint32_t s_p, s_wcnt[PK_THREADS / 64];
    __shared__ unsigned long long s_base, s_pay;
    const uint32_t g = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) {                                               // partition of block slot g: the largest p with blk_first[p] <= g
        uint32_t lo = 0, hi = P.nb;
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (P.blk_first[mid] <= g) lo = mid; else hi = mid; }
        s_p = lo;
    }
    for (uint32_t i = t; i < PK_BLOCK / 64; i += PK_THREADS) s_bits[i] = 0ull;
    if (t < PKV_NSUB) s_wmax[t] = 0ull;
    __syncthreads();
    const uint32_t p = s_p, j = g - P.blk_first[p];
    const uint64_t s1 = P.ptot[2 * (p + 1) + 1], r0 = P.ptot[2 * p + 1] + (uint64_t)j * PK_BLOCK;
    const uint32_t n = (uint32_t)min((uint64_t)PK_BLOCK, s1 - r0);
    if (t == 0) bases[g] = recs[2 * r0];
    // ---- pass 1 (coalesced): the largest delta of every sub-block -> its width; the abundance side (bitmap, stream bytes, escapes of abundances >= 255) as before
    uint32_t run = 0;                                           // flagged records of the rounds before this one (the same in every thread)
    for (uint32_t i0 = 0; i0 < n; i0 += PK_THREADS) {
        const uint32_t i = i0 + t;
        uint32_t ab8 = 1; uint64_t d = 0;
        if (i < n) {
            const ulonglong2 me = *reinterpret_cast<const ulonglong2*>(recs + 2 * (r0 + i));
            const uint64_t prev = i ? recs[2 * (r0 + i - 1)] : me.x;
            d = me.x - prev;
            const uint32_t ab = (uint32_t)me.y;
            ab8 = ab;
            if (ab >= 255u) {
                const unsigned long long e = atomicAdd(n_exc, 1ull);
                if (e < exc_cap) { exc[2 * e] = r0 + i; exc[2 * e + 1] = ab; }
                ab8 = 255u;
            }
        }
#pragma unroll
        for (int d_ = 32; d_ >= 1; d_ >>= 1) { const uint64_t y = (uint64_t)__shfl_xor((unsigned long long)d, d_, 64); d = y > d ? y : d; }      // (a wave's 64 records lie in one sub-block)
        if (lane == 0 && d) atomicMax(&s_wmax[(i0 >> 7) + (wave >> 1)], (unsigned long long)d);
        const bool flag = ab8 != 1u;
        const unsigned long long bal = __ballot(flag);
        if (lane == 0) { s_bits[(i0 >> 6) + wave] = bal; s_wcnt[wave] = (uint32_t)__popcll(bal); }
        __syncthreads();
        uint32_t before = run, total = 0;
#pragma unroll
        for (int w = 0; w < PK_THREADS / 64; w++) { if (w < (int)wave) before += s_wcnt[w]; total += s_wcnt[w]; }
        if (flag) s_cb[before + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = (uint8_t)ab8;
        run += total;
        __syncthreads();
    }
    if (t < PKV_NSUB) {                                         // (wave 0) widths, and the exclusive prefix of the sub-blocks' 16 W bytes
        const uint64_t m = s_wmax[t];
        uint32_t W = m ? 64u - (uint32_t)__clzll((long long)m) : 0u;
        if (W > 56u) W = 64u;
        if (t * PKV_SUB >= n) W = 0u;
        uint32_t x = 16u * W;
#pragma unroll
        for (int d_ = 1; d_ < 64; d_ <<= 1) { const uint32_t y = __shfl_up(x, d_, 64); if ((int)lane >= d_) x += y; }
        s_w[t] = W; s_off[t] = x - 16u * W;
        if (t == PKV_NSUB - 1) s_off[PKV_NSUB] = x;
        wbits[(uint64_t)g * PKV_NSUB + t] = (uint8_t)W;
    }
    __syncthreads();
    if (t == 0) {
        const uint64_t bytes = (uint64_t)s_off[PKV_NSUB] + PKV_BITMAP;                        // a multiple of 16
        s_pay = atomicAdd(pay_cursor, (unsigned long long)bytes);
        pay_off16[g] = (uint32_t)(s_pay >> 4);
        s_base = run ? atomicAdd(cb_cursor, (unsigned long long)run) : 0ull; cb_off[g] = (uint32_t)s_base;      // (the stream is shorter than 2^32 bytes: one byte per record at most)
    }
    __syncthreads();
    uint8_t* dstp = payload + s_pay;
    // ---- pass 2 (the block's records again: L2): chunks of 2048 keys through LDS, every thread packs 8 consecutive deltas into the W bytes of its sub-block's width,
    //      the chunk (16 sub-blocks back to back) leaves as 16-byte words
    for (uint32_t c0 = 0; c0 < n; c0 += PKV_CHUNK) {
        for (uint32_t i = t; i < PKV_CHUNK; i += PK_THREADS) s_key[1 + i] = c0 + i < n ? recs[2 * (r0 + c0 + i)] : 0ull;
        if (t == 0) s_key[0] = c0 ? recs[2 * (r0 + c0 - 1)] : recs[2 * r0];
        __syncthreads();
        const uint32_t sub0 = c0 / PKV_SUB, sub = sub0 + (t >> 4), W = s_w[sub], cbase = s_off[sub0];
        {
            unsigned __int128 acc = 0; uint32_t nbits = 0;
            uint8_t* o = s_out + (s_off[sub] - cbase) + (size_t)(t & 15u) * W;
            uint64_t prev = s_key[8 * t];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const uint32_t i = c0 + 8 * t + q;
                const uint64_t key = s_key[1 + 8 * t + q];
                const uint64_t d = i < n ? key - prev : 0ull;                               // (beyond the block's records: zero bits; W = 0: nothing is written)
                prev = key;
                acc |= (unsigned __int128)d << nbits; nbits += W;
                while (nbits >= 8) { *o++ = (uint8_t)acc; acc >>= 8; nbits -= 8; }
            }
        }
        __syncthreads();
        const uint32_t cwords = (s_off[sub0 + PKV_CHUNK / PKV_SUB] - cbase) >> 4;             // (sub0 + 16 <= 64)
        uint4* dst = reinterpret_cast<uint4*>(dstp + cbase);
        for (uint32_t w = t; w < cwords; w += PK_THREADS) dst[w] = reinterpret_cast<const uint4*>(s_out)[w];
        __syncthreads();
    }
    if (t < PKV_BITMAP / 16) reinterpret_cast<uint4*>(dstp + s_off[PKV_NSUB])[t] = reinterpret_cast<const uint4*>(s_bits)[t];      // the bitmap: 1024 bytes = 64 x 16
    uint8_t* cb = cb_stream + s_base;
    for (uint32_t i = t; i < run; i += PK_THREADS) cb[i] = s_cb[i];
}

// PKV, 16-byte keys (see PKV2_BLOCK_MAX above): recs = 4 words per record (value low, value high, abundance, 0); bases = 2 words per block
__global__ __launch_bounds__(PK_THREADS) void k_pack_pkv2(const uint64_t* __restrict__ recs, PackPlan P, uint64_t* __restrict__ bases, uint32_t* __restrict__ cb_off,
                                                          uint32_t* __restrict__ pay_off16, uint8_t* __restrict__ wbits /* [nblk][PKV_NSUB] */,
                                                          uint8_t* __restrict__ payload, unsigned long long* __restrict__ pay_cursor /* bytes */,
                                                          uint8_t* __restrict__ cb_stream, unsigned long long* __restrict__ cb_cursor,
                                                          uint64_t* __restrict__ exc, unsigned long long* __restrict__ n_exc, uint32_t exc_cap)
{
    typedef unsigned __int128 u128;
    __shared__ __attribute__((aligned(16))) uint8_t s_out[PK_THREADS * 128];                 // one chunk's entries: 16 sub-blocks of 16 W bytes, W <= 128
    __shared__ uint64_t s_klo[PKV_CHUNK + 1], s_khi[PKV_CHUNK + 1];                           // the chunk's keys, [0] = the key before the chunk
    __shared__ __attribute__((aligned(16))) unsigned long long s_bits[PK_BLOCK / 64];
    __shared__ __attribute__((aligned(16))) uint8_t s_cb[PK_BLOCK];
    __shared__ uint32_t s_w[PKV_NSUB], s_off[PKV_NSUB + 1];                                   // a sub-block's width (bits of its largest delta), the byte offset of its entries
    __shared__ uint32_t s_p, s_wcnt[PK_THREADS / 64];
    __shared__ unsigned long long s_base, s_pay;
    const uint32_t g = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) {
        uint32_t lo = 0, hi = P.nb;
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (P.blk_first[mid] <= g) lo = mid; else hi = mid; }
        s_p = lo;
    }
    for (uint32_t i = t; i < PK_BLOCK / 64; i += PK_THREADS) s_bits[i] = 0ull;
    if (t < PKV_NSUB) s_w[t] = 0u;
    __syncthreads();
    const uint32_t p = s_p, j = g - P.blk_first[p];
    const uint64_t s1 = P.ptot[2 * (p + 1) + 1], r0 = P.ptot[2 * p + 1] + (uint64_t)j * PK_BLOCK;
    const uint32_t n = (uint32_t)min((uint64_t)PK_BLOCK, s1 - r0);
    if (t == 0) { bases[2 * (uint64_t)g] = recs[4 * r0]; bases[2 * (uint64_t)g + 1] = recs[4 * r0 + 1]; }
    uint32_t run = 0;
    for (uint32_t i0 = 0; i0 < n; i0 += PK_THREADS) {
        const uint32_t i = i0 + t;
        uint32_t ab8 = 1, wd = 0;
        if (i < n) {
            const ulonglong2 me = *reinterpret_cast<const ulonglong2*>(recs + 4 * (r0 + i));
            ulonglong2 pv = me;
            if (i) pv = *reinterpret_cast<const ulonglong2*>(recs + 4 * (r0 + i - 1));
            const u128 d = (((u128)me.y << 64) | me.x) - (((u128)pv.y << 64) | pv.x);
            const uint64_t dh = (uint64_t)(d >> 64), dl = (uint64_t)d;
            wd = dh ? 128u - (uint32_t)__clzll((long long)dh) : dl ? 64u - (uint32_t)__clzll((long long)dl) : 0u;
            const uint32_t ab = (uint32_t)recs[4 * (r0 + i) + 2];
            ab8 = ab;
            if (ab >= 255u) {
                const unsigned long long e = atomicAdd(n_exc, 1ull);
                if (e < exc_cap) { exc[2 * e] = r0 + i; exc[2 * e + 1] = ab; }
                ab8 = 255u;
            }
        }
#pragma unroll
        for (int d_ = 32; d_ >= 1; d_ >>= 1) { const uint32_t y = __shfl_xor(wd, d_, 64); wd = y > wd ? y : wd; }      // (a wave's 64 records lie in one sub-block)
        if (lane == 0 && wd) atomicMax(&s_w[(i0 >> 7) + (wave >> 1)], wd);
        const bool flag = ab8 != 1u;
        const unsigned long long bal = __ballot(flag);
        if (lane == 0) { s_bits[(i0 >> 6) + wave] = bal; s_wcnt[wave] = (uint32_t)__popcll(bal); }
        __syncthreads();
        uint32_t before = run, total = 0;
#pragma unroll
        for (int w = 0; w < PK_THREADS / 64; w++) { if (w < (int)wave) before += s_wcnt[w]; total += s_wcnt[w]; }
        if (flag) s_cb[before + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = (uint8_t)ab8;
        run += total;
        __syncthreads();
    }
    if (t < PKV_NSUB) {
        uint32_t W = s_w[t];
        if (t * PKV_SUB >= n) W = 0u;
        uint32_t x = 16u * W;
#pragma unroll
        for (int d_ = 1; d_ < 64; d_ <<= 1) { const uint32_t y = __shfl_up(x, d_, 64); if ((int)lane >= d_) x += y; }
        s_w[t] = W; s_off[t] = x - 16u * W;
        if (t == PKV_NSUB - 1) s_off[PKV_NSUB] = x;
        wbits[(uint64_t)g * PKV_NSUB + t] = (uint8_t)W;
    }
    __syncthreads();
    if (t == 0) {
        const uint64_t bytes = (uint64_t)s_off[PKV_NSUB] + PKV_BITMAP;
        s_pay = atomicAdd(pay_cursor, (unsigned long long)bytes);
        pay_off16[g] = (uint32_t)(s_pay >> 4);
        s_base = run ? atomicAdd(cb_cursor, (unsigned long long)run) : 0ull; cb_off[g] = (uint32_t)s_base;
    }
    __syncthreads();
    uint8_t* dstp = payload + s_pay;
    for (uint32_t c0 = 0; c0 < n; c0 += PKV_CHUNK) {
        for (uint32_t i = t; i < PKV_CHUNK; i += PK_THREADS) {
            ulonglong2 q = make_ulonglong2(0, 0);
            if (c0 + i < n) q = *reinterpret_cast<const ulonglong2*>(recs + 4 * (r0 + c0 + i));
            s_klo[1 + i] = q.x; s_khi[1 + i] = q.y;
        }
        if (t == 0) { const uint64_t rp = c0 ? r0 + c0 - 1 : r0; s_klo[0] = recs[4 * rp]; s_khi[0] = recs[4 * rp + 1]; }
        __syncthreads();
        const uint32_t sub0 = c0 / PKV_SUB, sub = sub0 + (t >> 4), W = s_w[sub], cbase = s_off[sub0];
        {
            const uint32_t wl = W < 64u ? W : 64u, wh = W - wl;
            u128 acc = 0; uint32_t nbits = 0;
            uint8_t* o = s_out + (s_off[sub] - cbase) + (size_t)(t & 15u) * W;
            u128 prev = ((u128)s_khi[8 * t] << 64) | s_klo[8 * t];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const uint32_t i = c0 + 8 * t + q;
                const u128 key = ((u128)s_khi[1 + 8 * t + q] << 64) | s_klo[1 + 8 * t + q];
                const u128 d = i < n ? key - prev : (u128)0;
                prev = key;
                acc |= (u128)(uint64_t)d << nbits; nbits += wl;                             // (the delta's low 64 bits hold nothing above wl bits unless W > 64, and then wl = 64)
                while (nbits >= 8) { *o++ = (uint8_t)acc; acc >>= 8; nbits -= 8; }
                if (wh) { acc |= (u128)(uint64_t)(d >> 64) << nbits; nbits += wh; while (nbits >= 8) { *o++ = (uint8_t)acc; acc >>= 8; nbits -= 8; } }
            }
        }
        __syncthreads();
        const uint32_t cwords = (s_off[sub0 + PKV_CHUNK / PKV_SUB] - cbase) >> 4;
        uint4* dst = reinterpret_cast<uint4*>(dstp + cbase);
        for (uint32_t w = t; w < cwords; w += PK_THREADS) dst[w] = reinterpret_cast<const uint4*>(s_out)[w];
        __syncthreads();
    }
    if (t < PKV_BITMAP / 16) reinterpret_cast<uint4*>(dstp + s_off[PKV_NSUB])[t] = reinterpret_cast<const uint4*>(s_bits)[t];
    uint8_t* cb = cb_stream + s_base;
    for (uint32_t i = t; i < run; i += PK_THREADS) cb[i] = s_cb[i];
}

// 16-byte keys: recs = 4 words per record (value low, value high, abundance, 0); bases = 2 words per block; W = 16 (15-byte deltas) or 17 (16-byte deltas)
template <int W>
__global__ __launch_bounds__(PK_THREADS) void k_pack_counts2(const uint64_t* __restrict__ recs, PackPlan P, uint64_t* __restrict__ bases, uint8_t* __restrict__ payload,
                                                             uint64_t* __restrict__ exc, unsigned long long* __restrict__ n_exc, uint32_t exc_cap)
{
    typedef unsigned __int128 u128;
    constexpr uint64_t PK_SLOT = pk_slot(W);
    constexpr int WORDS16 = PK_THREADS * W / 16;                // 256 / 272 16-byte words per chunk of PK_THREADS entries
    __shared__ __attribute__((aligned(16))) uint8_t s_out[PK_THREADS * W + 16];
    __shared__ uint32_t s_p;
    const uint32_t g = blockIdx.x, t = threadIdx.x;
    if (t == 0) {
        uint32_t lo = 0, hi = P.nb;
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (P.blk_first[mid] <= g) lo = mid; else hi = mid; }
        s_p = lo;
    }
    __syncthreads();
    const uint32_t p = s_p, j = g - P.blk_first[p];
    const uint64_t s1 = P.ptot[2 * (p + 1) + 1], r0 = P.ptot[2 * p + 1] + (uint64_t)j * PK_BLOCK;
    const uint32_t n = (uint32_t)min((uint64_t)PK_BLOCK, s1 - r0);
    if (t == 0) { bases[2 * (uint64_t)g] = recs[4 * r0]; bases[2 * (uint64_t)g + 1] = recs[4 * r0 + 1]; }
    uint8_t* dstp = payload + (uint64_t)g * PK_SLOT;
    for (uint32_t i0 = 0; i0 < n; i0 += PK_THREADS) {
        const uint32_t i = i0 + t;
        uint64_t d_lo = 0, d_hi = 0; uint32_t ab8 = 0;
        if (i < n) {
            const ulonglong2 me = *reinterpret_cast<const ulonglong2*>(recs + 4 * (r0 + i));
            const uint32_t ab = (uint32_t)recs[4 * (r0 + i) + 2];
            ulonglong2 pv = me;
            if (i) pv = *reinterpret_cast<const ulonglong2*>(recs + 4 * (r0 + i - 1));
            const u128 d = (((u128)me.y << 64) | me.x) - (((u128)pv.y << 64) | pv.x);
            d_lo = (uint64_t)d; d_hi = (uint64_t)(d >> 64);
            if (W == 16 && (d_hi >> 56) != 0) d_hi = ~0ull, d_lo = ~0ull;                        // does not fit 120 bits: escape below
            if (W == 16 && d_lo == ~0ull && (d_hi & 0xFFFFFFFFFFFFFFull) == 0xFFFFFFFFFFFFFFull) {  // the escape pattern (also a true delta of exactly 2^120 - 1)
                const unsigned long long e = atomicAdd(n_exc, 2ull);
                if (e + 1 < exc_cap) { exc[2 * e] = PK_KEY_EXC | (r0 + i); exc[2 * e + 1] = me.x; exc[2 * e + 2] = PK_KEY_EXC_HI | (r0 + i); exc[2 * e + 3] = me.y; }
            }
            ab8 = ab;
            if (ab >= 255u) {
                const unsigned long long e = atomicAdd(n_exc, 1ull);
                if (e < exc_cap) { exc[2 * e] = r0 + i; exc[2 * e + 1] = ab; }
                ab8 = 255u;
            }
        }
        uint8_t* o = s_out + W * t;
#pragma unroll
        for (int b = 0; b < 8; b++) o[b] = (uint8_t)(d_lo >> (8 * b));
#pragma unroll
        for (int b = 0; b < W - 9; b++) o[8 + b] = (uint8_t)(d_hi >> (8 * b));
        o[W - 1] = (uint8_t)ab8;
        __syncthreads();
        for (int w = t; w < WORDS16; w += PK_THREADS) reinterpret_cast<uint4*>(dstp + (uint64_t)i0 * W)[w] = reinterpret_cast<const uint4*>(s_out)[w];
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ host side
struct SinkBatch {
    int users = 0;                               // workers between picking this batch and their last access to it (under gkc_unpacker::mu): a batch is deleted only at users == 0
    hipEvent_t copied = nullptr;                 // the batch's packed bytes are in the staging buffer
    hipEvent_t copy_start = nullptr;             // GKC_SINK_DEBUG: when the copy stream got to it
    const uint8_t* stage = nullptr;              // [bases: 8 x nblk, padded to 64][payload: nblk x PK_SLOT][exceptions: 16 x n_exc]
    uint64_t nblk = 0, n_exc = 0, pay_off = 0, exc_off = 0; int width = 7;
    uint64_t cboff_off = 0, cb_off = 0, n_cb = 0;        // width 6: the blocks' offsets into the abundance stream (u32 each), the stream, its length
    uint64_t pay16_off = 0, wbits_off = 0, pay_bytes = 0; // width 6: the blocks' payload offsets (u32, 16-byte units) and bit widths (u8) in the header; bytes of the payload stream
    std::vector<uint64_t> blk_rec0; std::vector<uint32_t> blk_n;       // per block: first record (index in the batch), records
    uint8_t* dest = nullptr;                     // the batch's records in the caller's sink
    void* d_packed = nullptr;                    // device buffer, given back once copied
    std::vector<std::pair<uint64_t, uint64_t>> exc;                    // sorted by (kind | record index)
    bool ready = false, syncing = false;         // copy completed + exceptions sorted (under the pool's lock)
    std::atomic<uint64_t> next{0}, finished{0};
    std::atomic<bool> done{false};
    std::chrono::steady_clock::time_point t_queued, t_ready;           // GKC_SINK_DEBUG
    double pack_ms = 0;
};
#define g_sink_debug (gkc_tun().sink_debug)

struct gkc_unpacker {
    gkc_ctx* c = nullptr;
    std::vector<std::thread> threads;
    std::mutex mu; std::condition_variable cv, cv_done;
    std::deque<SinkBatch*> queue;                // batches whose blocks are not all taken yet, oldest first
    std::vector<SinkBatch*> all;                 // every batch of the pass (owned)
    bool stop = false;
    uint8_t* staging = nullptr; uint64_t staging_cap = 0, staging_used = 0;
    std::atomic<uint64_t> n_decisions{0}, n_adaptive_raw{0}, max_batch_records{0};     // batches of this context that travelled raw because the host was behind (gkc_sink_host_behind)

    static uint64_t lookup(const std::vector<std::pair<uint64_t, uint64_t>>& exc, uint64_t tag)
    {
        auto it = std::lower_bound(exc.begin(), exc.end(), std::make_pair(tag, (uint64_t)0));
        return it != exc.end() && it->first == tag ? it->second : 0;
    }
    template <int W> static void unpack_block_w(const SinkBatch& B, uint64_t g)
    {
        constexpr uint64_t PK_ESC = pk_esc(W);
        const uint8_t* pay = B.stage + B.pay_off + g * pk_slot(W);
        const uint64_t r0 = B.blk_rec0[g]; const uint32_t n = B.blk_n[g];
        uint64_t key = reinterpret_cast<const uint64_t*>(B.stage)[g];
        __m128i* out = reinterpret_cast<__m128i*>(B.dest + r0 * 16);
        for (uint32_t i = 0; i < n; i++) {
            uint64_t w; memcpy(&w, pay + W * (size_t)i, 8);      // (W = 7: one byte beyond the entry; the staging buffer is padded)
            const uint64_t d = w & PK_ESC; uint32_t ab = (uint32_t)(w >> (8 * (W - 1))) & 255u;
            if (i) key = d == PK_ESC ? lookup(B.exc, PK_KEY_EXC | (r0 + i)) : key + d;
            if (ab == 255u) ab = (uint32_t)lookup(B.exc, r0 + i);
            _mm_stream_si128(out + i, _mm_set_epi64x((long long)(uint64_t)ab, (long long)key));      // {u64 value; i32 abundance; 4 bytes of padding = 0}
        }
    }
    // One sub-block of PKV (<= 128 records at width W), W a template constant: 8 records = W bytes, so inside a group every byte offset and shift is a constant
    // (the generic loop with a running bit position expanded 1.0e10 records/s on 24 threads — level with the link; this one keeps the margin of the fixed 6-byte format)
    template <int W> static void pkv_sub(const SinkBatch& B, const uint8_t* pay, const uint32_t cnt, const uint64_t rec0, uint64_t& key, const uint64_t* bits /* the sub-block's 2 words */,
                                         const uint8_t*& cb, __m128i* out)
    {
        constexpr uint64_t mask = W >= 64 ? ~0ull : ((1ull << (W & 63)) - 1ull);
        auto one = [&](const uint32_t i, const uint64_t w, const uint32_t sh, const uint32_t f) {
            key += (w >> sh) & mask;
            uint32_t ab = 1u + f * ((uint32_t)*cb - 1u); cb += f;            // 16 % of the records, at random: no branch on it (the byte under the cursor is read either way; padding follows the stream)
            if (ab == 255u) ab = (uint32_t)lookup(B.exc, rec0 + i);
            _mm_stream_si128(out + i, _mm_set_epi64x((long long)(uint64_t)ab, (long long)key));
        };
        uint32_t i = 0;
        for (; i + 8 <= cnt; i += 8) {
            const uint8_t* q = pay + (size_t)(i >> 3) * W;
            const uint32_t m = (uint32_t)(bits[i >> 6] >> (i & 63)) & 255u;
#pragma unroll
            for (int j = 0; j < 8; j++) { uint64_t w; memcpy(&w, q + ((j * W) >> 3), 8); one(i + j, w, (uint32_t)((j * W) & 7), (m >> j) & 1u); }      // (up to 7 bytes beyond the group: the next one / the bitmap / padding)
        }
        for (uint64_t bit = (uint64_t)i * W; i < cnt; i++, bit += W) { uint64_t w; memcpy(&w, pay + (bit >> 3), 8); one(i, w, (uint32_t)(bit & 7), (uint32_t)(bits[i >> 6] >> (i & 63)) & 1u); }
    }
    typedef void (*pkv_fn)(const SinkBatch&, const uint8_t*, uint32_t, uint64_t, uint64_t&, const uint64_t*, const uint8_t*&, __m128i*);
    template <size_t... I> static const pkv_fn* pkv_table(std::index_sequence<I...>) { static const pkv_fn t[] = { &pkv_sub<(int)I>... }; return t; }
    static void unpack_block_6(const SinkBatch& B, uint64_t g)                       // PKV: one delta width per sub-block of 128 records, no key escapes
    {
        static const pkv_fn* const table = pkv_table(std::make_index_sequence<65>());
        const uint64_t r0 = B.blk_rec0[g]; const uint32_t n = B.blk_n[g];
        const uint8_t* wb = B.stage + B.wbits_off + g * PKV_NSUB;
        const uint8_t* pay = B.stage + B.pay_off + ((uint64_t)reinterpret_cast<const uint32_t*>(B.stage + B.pay16_off)[g] << 4);
        uint32_t total = 0; for (uint32_t s = 0; s < PKV_NSUB; s++) total += 16u * wb[s];
        const uint64_t* bits = reinterpret_cast<const uint64_t*>(pay + total);
        const uint8_t* cb = B.stage + B.cb_off + reinterpret_cast<const uint32_t*>(B.stage + B.cboff_off)[g];
        uint64_t key = reinterpret_cast<const uint64_t*>(B.stage)[g];           // (a block's first delta is 0)
        __m128i* out = reinterpret_cast<__m128i*>(B.dest + r0 * 16);
        for (uint32_t s0 = 0; s0 < n; s0 += PKV_SUB) {
            const uint32_t W = std::min<uint32_t>(wb[s0 / PKV_SUB], 64u);
            table[W](B, pay, std::min<uint32_t>(PKV_SUB, n - s0), r0 + s0, key, bits + (s0 >> 6), cb, out + s0);
            pay += 16u * W;
        }
    }
    static void unpack_block_pkv2(const SinkBatch& B, uint64_t g)                     // PKV, 16-byte keys: 32-byte records {value low, value high, abundance, 0}
    {
        typedef unsigned __int128 u128;
        const uint64_t r0 = B.blk_rec0[g]; const uint32_t n = B.blk_n[g];
        const uint8_t* wb = B.stage + B.wbits_off + g * PKV_NSUB;
        const uint8_t* pay = B.stage + B.pay_off + ((uint64_t)reinterpret_cast<const uint32_t*>(B.stage + B.pay16_off)[g] << 4);
        uint32_t total = 0; for (uint32_t s = 0; s < PKV_NSUB; s++) total += 16u * wb[s];
        const uint64_t* bits = reinterpret_cast<const uint64_t*>(pay + total);
        const uint8_t* cb = B.stage + B.cb_off + reinterpret_cast<const uint32_t*>(B.stage + B.cboff_off)[g];
        const uint64_t* b2 = reinterpret_cast<const uint64_t*>(B.stage) + 2 * g;
        u128 key = ((u128)b2[1] << 64) | b2[0];                                  // (a block's first delta is 0)
        __m128i* out = reinterpret_cast<__m128i*>(B.dest + r0 * 32);
        for (uint32_t s0 = 0; s0 < n; s0 += PKV_SUB) {
            const uint32_t W = wb[s0 / PKV_SUB], wl = W < 64u ? W : 64u, wh = W - wl;
            const uint64_t ml = wl >= 64 ? ~0ull : (1ull << wl) - 1ull, mh = wh >= 64 ? ~0ull : (1ull << wh) - 1ull;
            uint64_t bit = 0;
            const uint32_t e = std::min<uint32_t>(n, s0 + PKV_SUB);
            for (uint32_t i = s0; i < e; i++) {
                u128 x; memcpy(&x, pay + (bit >> 3), 16);                        // (16 bytes from any byte: 7 + 64 bits lie inside; up to 15 bytes beyond the entries: bitmap / padding)
                const uint64_t lo = (uint64_t)(x >> (bit & 7)) & ml; bit += wl;
                uint64_t hi = 0;
                if (wh) { memcpy(&x, pay + (bit >> 3), 16); hi = (uint64_t)(x >> (bit & 7)) & mh; bit += wh; }
                key += ((u128)hi << 64) | lo;
                const uint32_t f = (uint32_t)(bits[i >> 6] >> (i & 63)) & 1u;
                uint32_t ab = 1u + f * ((uint32_t)*cb - 1u); cb += f;
                if (ab == 255u) ab = (uint32_t)lookup(B.exc, r0 + i);
                _mm_stream_si128(out + 2 * (size_t)i, _mm_set_epi64x((long long)(uint64_t)(key >> 64), (long long)(uint64_t)key));
                _mm_stream_si128(out + 2 * (size_t)i + 1, _mm_set_epi64x(0ll, (long long)(uint64_t)ab));
            }
            pay += 16u * W;
        }
    }
    template <int W> static void unpack_block_2(const SinkBatch& B, uint64_t g)         // 16-byte keys: 32-byte records {value low, value high, abundance, 0}
    {
        typedef unsigned __int128 u128;
        const uint8_t* pay = B.stage + B.pay_off + g * pk_slot(W);
        const uint64_t r0 = B.blk_rec0[g]; const uint32_t n = B.blk_n[g];
        const uint64_t* b2 = reinterpret_cast<const uint64_t*>(B.stage) + 2 * g;
        u128 key = ((u128)b2[1] << 64) | b2[0];
        __m128i* out = reinterpret_cast<__m128i*>(B.dest + r0 * 32);
        for (uint32_t i = 0; i < n; i++) {
            uint64_t lo, hi; memcpy(&lo, pay + W * (size_t)i, 8); memcpy(&hi, pay + W * (size_t)i + 8, 8);      // (W = 16: the 8th byte of `hi` is the abundance)
            uint32_t ab = pay[W * (size_t)i + W - 1];
            if (W == 16) hi &= 0xFFFFFFFFFFFFFFull;
            if (i) {
                if (W == 16 && lo == ~0ull && hi == 0xFFFFFFFFFFFFFFull) key = ((u128)lookup(B.exc, PK_KEY_EXC_HI | (r0 + i)) << 64) | lookup(B.exc, PK_KEY_EXC | (r0 + i));
                else key += ((u128)hi << 64) | lo;
            }
            if (ab == 255u) ab = (uint32_t)lookup(B.exc, r0 + i);
            _mm_stream_si128(out + 2 * (size_t)i, _mm_set_epi64x((long long)(uint64_t)(key >> 64), (long long)(uint64_t)key));
            _mm_stream_si128(out + 2 * (size_t)i + 1, _mm_set_epi64x(0ll, (long long)(uint64_t)ab));
        }
    }
    static void unpack_block(const SinkBatch& B, uint64_t g)
    {
        if (B.width == 6) unpack_block_6(B, g); else if (B.width == 14) unpack_block_pkv2(B, g); else if (B.width == 7) unpack_block_w<7>(B, g); else if (B.width == 8) unpack_block_w<8>(B, g);
        else if (B.width == 16) unpack_block_2<16>(B, g); else unpack_block_2<17>(B, g);
    }
    void worker()
    {
        (void)hipSetDevice(c->device);
        for (;;) {
            SinkBatch* B = nullptr;
            {   std::unique_lock<std::mutex> lk(mu);
                for (;;) {
                    if (stop) return;
                    if (!queue.empty()) {
                        B = queue.front();
                        if (B->ready) break;
                        if (!B->syncing) { B->syncing = true; break; }       // this thread waits for the copy
                    }
                    cv.wait(lk);
                }
                B->users++;                                                  // (gkc_sink_reset deletes a batch only when nobody is inside it any more)
            }
            if (!B->ready) {
                (void)hipEventSynchronize(B->copied);
                if (B->n_exc) {
                    const uint64_t* e = reinterpret_cast<const uint64_t*>(B->stage + B->exc_off);
                    B->exc.resize(B->n_exc);
                    for (uint64_t i = 0; i < B->n_exc; i++) B->exc[i] = { e[2 * i], e[2 * i + 1] };
                    std::sort(B->exc.begin(), B->exc.end());
                }
                if (B->d_packed) { c->dfree(B->d_packed); B->d_packed = nullptr; }
                B->t_ready = std::chrono::steady_clock::now();
                { std::lock_guard<std::mutex> lk(mu); B->ready = true; }
                cv.notify_all();
            }
            for (;;) {
                const uint64_t g = B->next.fetch_add(1);
                if (g >= B->nblk) break;
                unpack_block(*B, g);
                if (B->finished.fetch_add(1) + 1 == B->nblk) {
                    _mm_sfence();
                    if (g_sink_debug) {
                        const auto now = std::chrono::steady_clock::now();
                        float copy_ms = -1; if (B->copy_start) (void)hipEventElapsedTime(&copy_ms, B->copy_start, B->copied);
                        const auto t00 = all.empty() ? B->t_queued : all.front()->t_queued;
                        fprintf(stderr, "[gkc sink] +%.1f ms: batch of %llu blocks (%.2f GB packed, %llu exceptions): pack %.1f ms, queued -> copied %.1f ms (the copy itself %.1f ms), unpack %.1f ms\n",
                                std::chrono::duration<double, std::milli>(B->t_queued - t00).count(), (unsigned long long)B->nblk,
                                (double)((B->width == 6 || B->width == 14 ? B->pay_bytes : B->nblk * pk_slot_of(B->width)) + B->n_cb) / 1e9, (unsigned long long)B->n_exc, B->pack_ms, std::chrono::duration<double, std::milli>(B->t_ready - B->t_queued).count(), copy_ms,
                                std::chrono::duration<double, std::milli>(now - B->t_ready).count());
                    }
                    { std::lock_guard<std::mutex> lk(mu); B->done.store(true); }
                    cv_done.notify_all(); c->cv_done.notify_all();
                }
            }
            {   std::lock_guard<std::mutex> lk(mu);                              // every block of B has been taken: the next batch becomes the front
                if (!queue.empty() && queue.front() == B && B->next.load() >= B->nblk) queue.pop_front();
                B->users--;                                                      // the last access of this thread to B
            }
            cv.notify_all(); cv_done.notify_all();
        }
    }
};

// The unpack threads run on the cores of the NUMA node that holds the buffers they stream through (the page-locked staging buffer and the caller's sink are allocated
// by the thread that set the sink up, on its node): measured on the 2-socket host of the MI355X box, the same 16-24 threads expand a batch in 20 ms or in 50-80 ms
// depending on where the scheduler happened to put them. get_mempolicy(MPOL_F_NODE | MPOL_F_ADDR) names the node of a page; /sys lists its cores.
static int numa_node_of(const void* p)
{
    int node = -1;
    if (syscall(SYS_get_mempolicy, &node, nullptr, 0ul, const_cast<void*>(p), 3ul /* MPOL_F_NODE | MPOL_F_ADDR */) != 0) return -1;
    return node;
}
static bool cpus_of_node(int node, cpu_set_t* set)
{
    char path[96]; snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE* f = fopen(path, "r"); if (!f) return false;
    char buf[4096]; const bool got = fgets(buf, sizeof buf, f) != nullptr; fclose(f);
    if (!got) return false;
    CPU_ZERO(set); int n = 0;
    for (char* q = buf; *q; ) {
        char* e; const long a = strtol(q, &e, 10); if (e == q) break;
        long b = a; if (*e == '-') { q = e + 1; b = strtol(q, &e, 10); }
        for (long i = a; i <= b && i < CPU_SETSIZE; i++) { CPU_SET((int)i, set); n++; }
        q = *e == ',' ? e + 1 : e; if (*e != ',') break;
    }
    return n > 0;
}
static void pin_unpackers(gkc_unpacker* U);

static gkc_unpacker* unpacker_of(gkc_ctx* c)
{
    if (c->unpacker) return c->unpacker;
    gkc_unpacker* U = new gkc_unpacker(); U->c = c;
    // measured on the 2 x 64-core host of the MI355X box (tools/hostmem_probe/unpack_probe): 16 threads expand 14e9 records/s (100 GB/s read + 230 GB/s of non-temporal
    // writes) with or without a device -> host copy running beside them; 64 threads fall to 6e9/s beside the copy stream, 128 to 4e9/s even alone
    int n = gkc_tun().unpack_threads > 0 ? gkc_tun().unpack_threads : (int)std::min<unsigned>(24u, std::max(2u, std::thread::hardware_concurrency() / 2));
    // Several ranks of one job share the host (a communicator of W ranks on this context = W processes, taken to be spread evenly over the host's NUMA nodes): the host
    // expands 1.2-1.4e10 records/s in all however many ranks ask, and FEWER threads reach it — 8 ranks x 24 threads get 5.2e9 records/s, 8 x 3 threads 1.33e10 (round 6,
    // tools/hostmem_probe/unpack_ranks_probe on the 2 x 64-core host: profiles/r06_host_unpack_ceiling.txt) — so every rank takes its share of 12 threads per node.
    if (gkc_tun().unpack_threads <= 0 && c->comm_world > 1) {
        int nodes = 0; cpu_set_t tmp; while (nodes < 64 && cpus_of_node(nodes, &tmp)) nodes++;
        if (nodes < 1) nodes = 1;
        const int per_node = (c->comm_world + nodes - 1) / nodes;
        n = std::max(2, std::min(n, 12) / std::max(1, per_node));             // 12 per node: 2 ranks on 2 nodes 1.11e10 records/s with 12 or 24 each; 8 ranks: 3 each 1.33e10, 6: 1.01e10, 24: 5.2e9
    }
    if (n < 1) n = 1;
    for (int i = 0; i < n; i++) U->threads.emplace_back([U] { U->worker(); });
    c->unpacker = U;
    return U;
}

// whether the sink of this context takes packed batches (8-byte keys; GKC_SINK_PACKED=0 keeps the plain copies)
bool gkc_sink_packed(gkc_ctx* c)
{
    const bool off = !gkc_tun().sink_packed;
    const bool off2 = !gkc_tun().sink_packed2;       // (16-byte keys only)
    return c->sink && !c->sink_raw && (c->key_words == 1 || !off2) && !off && ((uintptr_t)c->sink & 15) == 0;
}

// the staging buffer holds the packed stream of ONE pass (like the sink holds one pass of records): 7/16 of the sink + the block slack of every partition
int gkc_sink_prepare(gkc_ctx* c)
{
    if (!gkc_sink_packed(c)) return GKC_OK;
    gkc_unpacker* U = unpacker_of(c);
    const uint64_t want = c->key_words == 1 ? c->sink_cap / 16 * 8 + (uint64_t)c->nb_partitions * (pk_slot(8) + 8) + ((uint64_t)64 << 20)
                                            : c->sink_cap / 32 * 17 + (uint64_t)c->nb_partitions * (pk_slot(17) + 16) + ((uint64_t)64 << 20);
    if (U->staging_cap < want) {
        if (U->staging) (void)hipHostFree(U->staging);
        U->staging = nullptr; U->staging_cap = 0;
        void* p = nullptr;
        if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return GKC_OK; }      // no staging buffer: the batches travel unpacked
        U->staging = (uint8_t*)p; U->staging_cap = want;
    }
    pin_unpackers(U);
    return GKC_OK;
}

static void pin_unpackers(gkc_unpacker* U)
{
    if (!U->staging) return;
    const int node = numa_node_of(U->c->sink ? U->c->sink : (const void*)U->staging);
    cpu_set_t set;
    if (node < 0 || !cpus_of_node(node, &set)) return;
    for (std::thread& t : U->threads) (void)pthread_setaffinity_np(t.native_handle(), sizeof(set), &set);
    if (g_sink_debug) fprintf(stderr, "[gkc sink] %zu unpack threads on the cores of NUMA node %d (where the sink lives)\n", U->threads.size(), node);
}

// start of a pass / a pass counted again: nothing of the previous one is in flight any more
void gkc_sink_reset(gkc_ctx* c)
{
    gkc_unpacker* U = c->unpacker;
    if (!U) return;
    {   std::unique_lock<std::mutex> lk(U->mu);
        U->cv_done.wait(lk, [&] { for (SinkBatch* B : U->all) if (!B->done.load() || B->users != 0) return false; return true; });
        U->queue.clear();
    }
    for (SinkBatch* B : U->all) { if (B->copied) (void)hipEventDestroy(B->copied); if (B->copy_start) (void)hipEventDestroy(B->copy_start); if (B->d_packed) c->dfree(B->d_packed); delete B; }
    U->all.clear(); U->staging_used = 0; c->sink_wire_bytes = 0; U->max_batch_records = 0;
}
void gkc_sink_drain(gkc_ctx* c)
{
    gkc_unpacker* U = c->unpacker;
    if (!U) return;
    std::unique_lock<std::mutex> lk(U->mu);
    U->cv_done.wait(lk, [&] { for (SinkBatch* B : U->all) if (!B->done.load()) return false; return true; });
}
void gkc_sink_shutdown(gkc_ctx* c)
{
    gkc_unpacker* U = c->unpacker;
    if (!U) return;
    gkc_sink_reset(c);
    { std::lock_guard<std::mutex> lk(U->mu); U->stop = true; }
    U->cv.notify_all();
    for (std::thread& t : U->threads) t.join();
    if (U->staging) (void)hipHostFree(U->staging);
    delete U; c->unpacker = nullptr;
}
void gkc_sink_wait_batch(gkc_ctx* c, const void* batch)
{
    gkc_unpacker* U = c->unpacker;
    if (!U || !batch) return;
    const SinkBatch* B = static_cast<const SinkBatch*>(batch);
    std::unique_lock<std::mutex> lk(U->mu);
    U->cv_done.wait(lk, [&] { return B->done.load(); });
}

// Round 6 — several ranks on one host: is the HOST behind? Records whose copy has landed in the staging buffer and that no thread has expanded yet, against what this
// batch holds. One rank alone expands a batch in 2/3 of the time its copy takes (the link is the bound: at most the batch in hand is pending); 2-8 ranks sharing the
// host's memory controllers get 1.0-1.4e10 records/s between them (profiles/r06_host_unpack_ceiling.txt) and their staging buffers fill with landed, unexpanded
// batches. A batch queued then travels RAW instead (16 B per record on this rank's own link, no host work) — the link is busy 2.6x longer with it and the expansion
// threads catch up: every rank balances its link against its share of the host by itself, batch by batch. The sink ends up byte for byte the same either way.
static thread_local const char* g_sink_why = "";                   // why the last batch of this thread did not travel packed (GKC_SINK_DEBUG)
bool gkc_sink_host_behind(gkc_ctx* c, uint64_t n_records)
{
    gkc_unpacker* U = c->unpacker;
    if (!U || !gkc_tun().sink_adaptive) return false;
    uint64_t pending = 0;
    {   std::lock_guard<std::mutex> lk(U->mu);
        for (SinkBatch* B : U->all) {
            if (B->done.load()) continue;
            if (B->ready || hipEventQuery(B->copied) == hipSuccess) pending += B->nblk - std::min<uint64_t>(B->finished.load(), B->nblk);
        }
    }
    (void)hipGetLastError();                                        // (hipErrorNotReady of the queries)
    // (against the LARGEST batch of the pass so far, not this one: the small batches at the end of a ramp would otherwise see the whole batch in hand of the
    //  expansion threads as "1.5 batches behind" and travel raw — 16 instead of 6 bytes per record on the link at the very end of the step)
    uint64_t ref = U->max_batch_records.load(); if (n_records > ref) { U->max_batch_records = n_records; ref = n_records; }
    const bool behind = gkc_tun().sink_adaptive == 2 ? (U->n_decisions++ & 1) != 0                       // (tests: packed and raw batches alternate in one pass)
                                                     : pending * PK_BLOCK > std::max<uint64_t>(ref + ref / 2, (uint64_t)1 << 22);
    if (behind) { g_sink_why = "the host is behind with the expansion (landed, unexpanded records beyond 1.5 batches): this batch travels raw"; U->n_adaptive_raw++; }
    return behind;
}

// One Stage-B batch: d_out = its Count[] (total records, partition i = [solid_prefix[i], solid_prefix[i+1])), d_ptot = the (distinct, solid) prefixes on the device,
// h_dest = where the records belong in the sink. Runs on the calling lane's stream up to the point where the copy can be queued; returns the batch handle
// (nullptr: not packed — no staging room, too many exceptions — the caller sends the plain records).
const char* gkc_sink_last_refusal() { return g_sink_why; }
void* gkc_sink_send_packed(gkc_ctx* c, const void* d_out, const uint64_t* d_ptot, const std::vector<uint64_t>& solid_prefix, uint8_t* h_dest)
{
    gkc_unpacker* U = c->unpacker;
    g_sink_why = "no staging buffer";
    if (!U || !U->staging) return nullptr;
    const uint32_t nb = (uint32_t)solid_prefix.size() - 1;
    std::vector<uint32_t> blk_first(nb + 1);
    uint64_t nblk = 0;
    for (uint32_t i = 0; i < nb; i++) { blk_first[i] = (uint32_t)nblk; nblk += (solid_prefix[i + 1] - solid_prefix[i] + PK_BLOCK - 1) / PK_BLOCK; }
    blk_first[nb] = (uint32_t)nblk;
    g_sink_why = "no records / too many blocks";
    if (nblk == 0 || nblk >= (1ull << 31)) return nullptr;
    // width of an entry: 8 where the partitions are sparse, 7 where dense, "6" = per-block delta widths + bitmap + abundance stream (PKV) where dense at abundance-min 1
    // (most abundances are 1: sequencing errors), checked batch by batch: a batch that came out above 7 bytes per record switches the context back to 7
    const bool no6 = !gkc_tun().sink_width6;
    const uint64_t dense_min = gkc_tun().sink_dense ? gkc_tun().sink_dense : PK_DENSE;      // (tests: 1 = every batch is "dense")
    const bool wide = c->key_words == 2;
    const bool dense = solid_prefix[nb] / std::max<uint32_t>(nb, 1) >= (wide ? std::min<uint64_t>(dense_min, PK2_DENSE) : dense_min);
    const bool pkv_ok = !no6 && !c->sink_no6 && solid_prefix[nb] < (1ull << 32);
    const int width = wide ? (pkv_ok ? 14 : dense ? 16 : 17) : !dense ? 8 : (c->amin <= 1 && pkv_ok) ? 6 : 7;
    const bool pkv = width == 6 || width == 14;
    const uint64_t n_rec = solid_prefix[nb];
    const uint64_t bases_bytes = (nblk * (wide ? 16 : 8) + 63) / 64 * 64, cboff_bytes = pkv ? (nblk * 4 + 63) / 64 * 64 : 0;
    const uint64_t wbits_bytes = pkv ? nblk * PKV_NSUB : 0, hdr_bytes = bases_bytes + 2 * cboff_bytes + wbits_bytes;      // width 6: [bases | abundance-stream offsets | payload offsets | widths]
    const uint64_t pay_bytes = nblk * pk_slot_of(width) + 64, cb_cap = pkv ? (n_rec + 63) / 64 * 64 : 0;                           // (width 6: the worst case — every block at 64 bits; what is copied is what was used)
    const uint32_t exc_cap = 1u << 20;
    g_sink_why = "no device memory for the packed copy";
    DevBuf d_first; if (c->ensure(d_first, (size_t)(nb + 1) * 4) != GKC_OK) return nullptr;
    uint8_t* d_packed = (uint8_t*)c->dalloc((size_t)(hdr_bytes + pay_bytes + cb_cap + (uint64_t)exc_cap * 16 + 64));
    if (!d_packed) { d_first.release(); return nullptr; }
    hipStream_t st = cur_stream(c);
    const auto t_pack0 = std::chrono::steady_clock::now();
    uint8_t* const d_pay = d_packed + hdr_bytes; uint8_t* const d_cb = d_pay + pay_bytes; uint8_t* const d_exc = d_cb + cb_cap;
    unsigned long long* d_nexc = reinterpret_cast<unsigned long long*>(d_exc + (uint64_t)exc_cap * 16);        // [0] exceptions [1] bytes of the abundance stream [2] bytes of the payload stream (width 6)
    unsigned long long h_cnt[3] = {0, 0, 0};
    bool ok = hipMemcpyAsync(d_first.p, blk_first.data(), (size_t)(nb + 1) * 4, hipMemcpyHostToDevice, st) == hipSuccess
           && hipMemsetAsync(d_nexc, 0, 24, st) == hipSuccess;
    if (ok) {
        PackPlan P{ (const uint32_t*)d_first.p, d_ptot, nb };
        if (width == 6) hipLaunchKernelGGL(k_pack_counts6, dim3((unsigned)nblk), dim3(PK_THREADS), 0, st, (const uint64_t*)d_out, P, (uint64_t*)d_packed, (uint32_t*)(d_packed + bases_bytes),
                                           (uint32_t*)(d_packed + bases_bytes + cboff_bytes), d_packed + bases_bytes + 2 * cboff_bytes, d_pay, d_nexc + 2, d_cb, d_nexc + 1, (uint64_t*)d_exc, d_nexc, exc_cap);
        else if (width == 14) hipLaunchKernelGGL(k_pack_pkv2, dim3((unsigned)nblk), dim3(PK_THREADS), 0, st, (const uint64_t*)d_out, P, (uint64_t*)d_packed, (uint32_t*)(d_packed + bases_bytes),
                                                 (uint32_t*)(d_packed + bases_bytes + cboff_bytes), d_packed + bases_bytes + 2 * cboff_bytes, d_pay, d_nexc + 2, d_cb, d_nexc + 1, (uint64_t*)d_exc, d_nexc, exc_cap);
        else if (width == 16) hipLaunchKernelGGL((k_pack_counts2<16>), dim3((unsigned)nblk), dim3(PK_THREADS), 0, st, (const uint64_t*)d_out, P, (uint64_t*)d_packed, d_pay,
                                                 (uint64_t*)d_exc, d_nexc, exc_cap);
        else if (width == 17) hipLaunchKernelGGL((k_pack_counts2<17>), dim3((unsigned)nblk), dim3(PK_THREADS), 0, st, (const uint64_t*)d_out, P, (uint64_t*)d_packed, d_pay,
                                                 (uint64_t*)d_exc, d_nexc, exc_cap);
        else if (width == 7) hipLaunchKernelGGL((k_pack_counts<7>), dim3((unsigned)nblk), dim3(PK_THREADS), 0, st, (const uint64_t*)d_out, P, (uint64_t*)d_packed, d_pay,
                                                (uint64_t*)d_exc, d_nexc, exc_cap);
        else hipLaunchKernelGGL((k_pack_counts<8>), dim3((unsigned)nblk), dim3(PK_THREADS), 0, st, (const uint64_t*)d_out, P, (uint64_t*)d_packed, d_pay,
                                (uint64_t*)d_exc, d_nexc, exc_cap);
        ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(h_cnt, d_nexc, 24, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
    }
    d_first.release();
    const unsigned long long h_nexc = h_cnt[0], h_ncb = h_cnt[1];
    const uint64_t pay_used = pkv ? ((uint64_t)h_cnt[2] + 63) / 64 * 64 : pay_bytes;               // bytes of the payload that travel (and are staged)
    g_sink_why = !ok ? "pack launch failed" : "too many exceptions";
    if (!ok || h_nexc > exc_cap || h_ncb > cb_cap || (pkv && h_cnt[2] > pay_bytes - 64)) { (void)hipGetLastError(); c->dfree(d_packed); return nullptr; }
    // (a batch whose per-block widths + abundance stream came out above the 7 bytes per record of the fixed entries — wide gaps AND few abundances of 1 — switches the
    //  context to those; this batch still travels as it was packed)
    if (pkv && (double)(h_cnt[2] + h_ncb) > (wide ? 16.0 : 7.0) * (double)n_rec) c->sink_no6 = true;
    SinkBatch* B = new SinkBatch();
    const uint64_t cb_stage = (h_ncb + 63) / 64 * 64;
    const uint64_t need = hdr_bytes + pay_used + cb_stage + h_nexc * 16 + 64;
    {   std::lock_guard<std::mutex> lk(c->mu);
        if (U->staging_used + need > U->staging_cap) { g_sink_why = "staging buffer full"; delete B; c->dfree(d_packed); return nullptr; }
        B->stage = U->staging + U->staging_used; U->staging_used += (need + 63) / 64 * 64;
        c->sink_wire_bytes += hdr_bytes + pay_used + h_ncb + h_nexc * 16;
    }
    B->nblk = nblk; B->n_exc = h_nexc; B->width = width; B->pay_off = hdr_bytes; B->cboff_off = bases_bytes; B->cb_off = hdr_bytes + pay_used; B->n_cb = h_ncb;
    B->pay16_off = bases_bytes + cboff_bytes; B->wbits_off = bases_bytes + 2 * cboff_bytes; B->pay_bytes = pay_used;
    B->exc_off = hdr_bytes + pay_used + cb_stage; B->dest = h_dest; B->d_packed = d_packed;
    B->blk_rec0.resize(nblk); B->blk_n.resize(nblk);
    for (uint32_t i = 0; i < nb; i++) {
        const uint64_t s0 = solid_prefix[i], s1 = solid_prefix[i + 1];
        for (uint64_t r = s0, g = blk_first[i]; r < s1; r += PK_BLOCK, g++) { B->blk_rec0[g] = r; B->blk_n[g] = (uint32_t)std::min<uint64_t>(PK_BLOCK, s1 - r); }
    }
    if (g_sink_debug && hipEventCreate(&B->copy_start) == hipSuccess) (void)hipEventRecord(B->copy_start, c->copy_stream);
    bool queued = hipEventCreateWithFlags(&B->copied, g_sink_debug ? hipEventDefault : hipEventDisableTiming) == hipSuccess
               && hipMemcpyAsync((void*)B->stage, d_packed, (size_t)(hdr_bytes + pay_used), hipMemcpyDeviceToHost, c->copy_stream) == hipSuccess
               && (h_ncb == 0 || hipMemcpyAsync((void*)(B->stage + B->cb_off), d_cb, (size_t)h_ncb, hipMemcpyDeviceToHost, c->copy_stream) == hipSuccess)
               && (h_nexc == 0 || hipMemcpyAsync((void*)(B->stage + B->exc_off), d_exc, (size_t)h_nexc * 16, hipMemcpyDeviceToHost, c->copy_stream) == hipSuccess)
               && hipEventRecord(B->copied, c->copy_stream) == hipSuccess;
    if (!queued) { g_sink_why = "copy could not be queued"; (void)hipGetLastError(); (void)hipStreamSynchronize(c->copy_stream); if (B->copied) (void)hipEventDestroy(B->copied); delete B; c->dfree(d_packed); return nullptr; }
    B->t_queued = std::chrono::steady_clock::now(); B->pack_ms = std::chrono::duration<double, std::milli>(B->t_queued - t_pack0).count();
    if (g_sink_debug) { std::lock_guard<std::mutex> lk(U->mu); if (U->all.empty()) fprintf(stderr, "[gkc sink] first batch queued %.1f ms after Stage B began\n", std::chrono::duration<double, std::milli>(B->t_queued - c->t_stage_b0).count()); }
    { std::lock_guard<std::mutex> lk(U->mu); U->all.push_back(B); U->queue.push_back(B); }
    U->cv.notify_all();
    return B;
}
