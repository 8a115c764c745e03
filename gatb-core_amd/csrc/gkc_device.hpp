// gkc_device.hpp — device-side arithmetic shared by the kernels (gfx950, wave64).
// Semantics restated from the reference (file:line under /root/reference/gatb-core/src/gatb/).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned __int128 u128;

// ---- A1: Data::ConvertASCII (tools/misc/api/Data.hpp:185) ----
__device__ __forceinline__ uint32_t nt_code(uint32_t c) { return (c >> 1) & 3u; }
__device__ __forceinline__ uint32_t nt_valid(uint32_t c)
{
    // ACGTacgt : (c & 0xDF) - 'A' in {0 (A), 2 (C), 6 (G), 19 (T)}
    uint32_t idx = (c & 0xDFu) - 0x41u;
    return (idx < 32u) ? ((0x00080045u >> idx) & 1u) : 0u;
}

// ---- LargeInt1.pri:137-154 revcomp64 ----
__device__ __forceinline__ uint64_t revcomp64(uint64_t x, uint32_t k)
{
    uint64_t r = x;
    r = ((r >> 2) & 0x3333333333333333ULL) | ((r & 0x3333333333333333ULL) << 2);
    r = ((r >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((r & 0x0F0F0F0F0F0F0F0FULL) << 4);
    r = __builtin_bswap64(r);
    r ^= 0xAAAAAAAAAAAAAAAAULL;
    return r >> (2u * (32u - k));     // k in [1,32]
}
// LargeInt2.pri:168-197 (k in [1,63])
__device__ __forceinline__ u128 revcomp128(u128 x, uint32_t k)
{
    uint64_t hi = (uint64_t)(x >> 64), lo = (uint64_t)x;
    if (k <= 32) return (u128)revcomp64(lo, k);
    uint32_t nb_hi = k - 32;
    uint64_t rhi = revcomp64(hi, nb_hi);
    uint64_t rlo = revcomp64(lo, 32);
    return (((u128)rlo) << (2 * nb_hi)) + rhi;
}

// ---- LargeInt1.pri:157-170 hash64 ----
__device__ __forceinline__ uint64_t hash64(uint64_t key, uint64_t seed)
{
    uint64_t h = seed;
    h ^= (h << 7) ^ key * (h >> 3) ^ (~((h << 11) + (key ^ (h >> 5))));
    h = (~h) + (h << 21);
    h = h ^ (h >> 24);
    h = (h + (h << 3)) + (h << 8);
    h = h ^ (h >> 14);
    h = (h + (h << 2)) + (h << 4);
    h = h ^ (h >> 28);
    h = h + (h << 31);
    return h;
}

// order-independent checksum mixer (splitmix64 finaliser); host twin in gkc.py / tests
__device__ __host__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// ---- key types: 8-byte (k<=31) and 16-byte (k<=63) canonical k-mers ----
template <int KW> struct KeyT;
template <> struct KeyT<1> {
    typedef uint64_t type;
    static __device__ __forceinline__ uint64_t max() { return ~0ULL; }
    static __device__ __forceinline__ uint64_t revcomp(uint64_t x, uint32_t k) { return revcomp64(x, k); }
    static __device__ __forceinline__ uint64_t mask(uint32_t k) { return (k >= 32) ? ~0ULL : ((1ULL << (2 * k)) - 1); }
    static __device__ __forceinline__ uint64_t mixv(uint64_t x) { return mix64(x); }
};
template <> struct KeyT<2> {
    typedef u128 type;
    static __device__ __forceinline__ u128 max() { return ~(u128)0; }
    static __device__ __forceinline__ u128 revcomp(u128 x, uint32_t k) { return revcomp128(x, k); }
    static __device__ __forceinline__ u128 mask(uint32_t k) { return (k >= 64) ? ~(u128)0 : ((((u128)1) << (2 * k)) - 1); }
    static __device__ __forceinline__ uint64_t mixv(u128 x) { return mix64((uint64_t)x) ^ mix64(~(uint64_t)(x >> 64)); }
};

// ---- device super-k-mer record (internal bucket format; NOT the reference wire format) ----
// RW 64-bit words, most significant first: word0 = [nbK:8][nt 0..27], word j>0 = nt 28+32(j-1) .. 28+32j-1,
// nucleotide i left-to-right in read order, 2 bits each, big-endian inside a word. Unused trailing bits are zero.
// 16 B (RW=2) holds 60 nt (k<=31, nbK<=28 -> k+nbK-1 <= 58); 32 B (RW=4) holds 124 nt (k<=63, nbK<=60 -> <=122).
template <int RW> struct RecT { uint64_t w[RW]; };

// wave64 helpers
__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// 16-byte store of data that is not read again soon (keys on their way to the sort, records, results). GKC_NT_STORES=1 (build flag): non-temporal
#ifndef GKC_NT_STORES
#define GKC_NT_STORES 0
#endif
typedef unsigned long long gkc_v2u64 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store16(void* dst, unsigned long long a, unsigned long long b)
{
    gkc_v2u64 v; v.x = a; v.y = b;
#if GKC_NT_STORES
    __builtin_nontemporal_store(v, reinterpret_cast<gkc_v2u64*>(dst));
#else
    *reinterpret_cast<gkc_v2u64*>(dst) = v;
#endif
}
