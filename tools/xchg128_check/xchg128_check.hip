// Is ds_wrxchg2_rtn_b64 a usable 128-bit LDS exchange? Every thread of a 1024-thread workgroup swaps consistent (lo, hi = lo ^ MAGIC) pairs
// through a handful of hot 16-byte slots, many rounds. Atomic per lane <=> every pair that ever comes out is consistent, and the multiset of
// pairs is conserved (what went in = what came out + what is left in the slots).
// build: hipcc -O3 --offload-arch=gfx950 xchg128_check.hip -o xchg128_check ; run: ./xchg128_check
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
constexpr uint64_t MAGIC = 0x9E3779B97F4A7C15ULL;
__device__ __forceinline__ void lds_xchg128(unsigned long long* slot, uint64_t in_lo, uint64_t in_hi, uint64_t& out_lo, uint64_t& out_hi)
{
    const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)slot;
    v4u r;
    asm volatile("ds_wrxchg2_rtn_b64 %0, %1, %2, %3 offset1:1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r) : "v"(addr), "v"(in_lo), "v"(in_hi) : "memory");
    out_lo = (uint64_t)r.x | ((uint64_t)r.y << 32); out_hi = (uint64_t)r.z | ((uint64_t)r.w << 32);
}
__global__ __launch_bounds__(1024) void k(unsigned long long* bad, unsigned long long* xsum, int nslots, int rounds)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long s[];
    for (int i = threadIdx.x; i < nslots; i += blockDim.x) { s[2 * i] = 0; s[2 * i + 1] = 0 ^ MAGIC; }
    __syncthreads();
    uint64_t lo = ((uint64_t)(blockIdx.x * 1024 + threadIdx.x + 1) << 20), hi = lo ^ MAGIC;
    unsigned long long nbad = 0;
    uint32_t rng = threadIdx.x * 2654435761u + blockIdx.x;
    for (int r = 0; r < rounds; r++) {
        rng = rng * 1664525u + 1013904223u;
        const int q = (rng >> 8) % nslots;
        uint64_t ol, oh;
        lds_xchg128(&s[2 * q], lo, hi, ol, oh);
        if ((ol ^ MAGIC) != oh) nbad++;
        lo = ol + 1; hi = lo ^ MAGIC;                       // a new value every round, derived from what came out
    }
    if (nbad) atomicAdd(bad, nbad);
    // conservation: sum over (values held at the end + slot contents) of lo must equal initial sum + rounds (each round adds 1) per thread
    atomicAdd(xsum, (unsigned long long)lo);
    __syncthreads();
    for (int i = threadIdx.x; i < nslots; i += blockDim.x) { if ((s[2 * i] ^ MAGIC) != s[2 * i + 1]) atomicAdd(bad, 1ULL); atomicAdd(xsum, s[2 * i]); }
}
int main()
{
    unsigned long long *d, h[2];
    hipMalloc(&d, 16);
    for (int nslots : {1, 3, 64, 1024, 8192}) {
        const int rounds = 20000, blocks = 64;
        hipMemset(d, 0, 16);
        hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, nslots * 16);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(1024), nslots * 16, 0, d, d + 1, nslots, rounds);
        hipDeviceSynchronize();
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        unsigned long long want = 0;
        for (int b = 0; b < blocks; b++) for (int t = 0; t < 1024; t++) want += ((unsigned long long)(b * 1024 + t + 1) << 20) + (unsigned long long)rounds;
        printf("nslots %5d: inconsistent pairs %llu, conserved %s\n", nslots, h[0], h[1] == want ? "yes" : "NO");
    }
    return 0;
}
