// scatter_bench — what does a scattered store of W bytes per lane cost on this GPU, in the access pattern of Stage B's key scatter?
//
// One workgroup per region of REGION_MB megabytes (like k_expand_scatter_*: one workgroup per partition's key range); the region is cut
// into NCUR buckets with one LDS cursor each; every thread picks a pseudo-random bucket, advances its cursor by W bytes (LDS atomic) and
// stores W contiguous bytes there (W/16 dwordx4 stores of one lane; W = 8: one dwordx2). The regions together exceed the 256 MB
// Infinity Cache many times. Printed: GB/s of useful bytes for W = 8..256 and NCUR = 2048 / 4096 / 8192 — the calibration
// the roofline / PMC discussion of the scatter kernel needs (run under rocprofv3 --pmc WRITE_SIZE to calibrate that counter too).
//   hipcc -O3 --offload-arch=gfx950 scatter_bench.hip -o scatter_bench && ./scatter_bench [total_GB]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ULL; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31);
}

template <int W>
__global__ __launch_bounds__(1024) void k_scatter(uint8_t* __restrict__ out, uint64_t region_bytes, uint32_t ncur, uint32_t iters)
{
    extern __shared__ uint32_t s_cur[];                     // byte cursor of every bucket, relative to the bucket start
    for (uint32_t i = threadIdx.x; i < ncur; i += blockDim.x) s_cur[i] = 0;
    __syncthreads();
    uint8_t* region = out + (uint64_t)blockIdx.x * region_bytes;
    const uint32_t bucket_bytes = (uint32_t)(region_bytes / ncur);
    uint64_t h = mix64(((uint64_t)blockIdx.x << 32) | threadIdx.x);
    for (uint32_t it = 0; it < iters; it++) {
        h = mix64(h);
        const uint32_t q = (uint32_t)(h >> 40) % ncur;
        const uint32_t pos = atomicAdd(&s_cur[q], (uint32_t)W);
        if (pos + W > bucket_bytes) continue;               // a bucket that filled up early (random fill): skip
        uint8_t* dst = region + (uint64_t)q * bucket_bytes + pos;
        if (W == 8) *reinterpret_cast<uint64_t*>(dst) = h;
        else {
#pragma unroll
            for (int j = 0; j < W / 16; j++) reinterpret_cast<ulonglong2*>(dst)[j] = make_ulonglong2(h, h + j);
        }
    }
}

// Round 6 ("fold k_dedupe_bin into Stage A's emit"): what a DIRECT emission of records to (partition, hash bin) slots costs. The fan-out — 4096
// partitions x 2048 bins = 2^23 open buckets — is far beyond any workgroup's LDS, so the cursors live in global memory: every record is one returning global
// atomic on a random cursor + one 16-byte store behind it. NCUR_TOTAL cursors over the whole buffer, every workgroup scatters over all of them.
__global__ __launch_bounds__(1024) void k_scatter_global(uint8_t* __restrict__ out, uint32_t* __restrict__ cur, uint64_t bucket_bytes, uint32_t ncur_total, uint32_t iters)
{
    uint64_t h = mix64(((uint64_t)blockIdx.x << 32) | threadIdx.x);
    for (uint32_t it = 0; it < iters; it++) {
        h = mix64(h);
        const uint32_t q = (uint32_t)((h >> 24) % ncur_total);
        const uint32_t pos = atomicAdd(&cur[q], 16u);
        if (pos + 16 > bucket_bytes) continue;
        *reinterpret_cast<ulonglong2*>(out + (uint64_t)q * bucket_bytes + pos) = make_ulonglong2(h, ~h);
    }
}

// coalesced streaming write of the same bytes: the practical ceiling on this box
__global__ void k_stream(ulonglong2* __restrict__ out, uint64_t n16)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) out[i] = make_ulonglong2(i, ~i);
}

template <int W> static double run(uint8_t* buf, uint32_t regions, uint64_t region_bytes, uint32_t ncur, double fill)
{
    const uint64_t stores = (uint64_t)((double)region_bytes * fill / W);
    const uint32_t iters = (uint32_t)(stores / 1024);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_scatter<W>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    hipLaunchKernelGGL(k_scatter<W>, dim3(regions), dim3(1024), ncur * 4, 0, buf, region_bytes, ncur, iters);   // warm
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k_scatter<W>, dim3(regions), dim3(1024), ncur * 4, 0, buf, region_bytes, ncur, iters);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    return (double)regions * iters * 1024.0 * W / (ms * 1e-3) / 1e9;
}

int main(int argc, char** argv)
{
    const double total_gb = argc > 1 ? atof(argv[1]) : 24.0;
    const uint64_t region_bytes = (uint64_t)24 << 20;                       // one partition's keys: 3e6 x 8 B
    const uint32_t regions = (uint32_t)(total_gb * 1e9 / region_bytes);
    uint8_t* buf; CK(hipMalloc(&buf, (size_t)regions * region_bytes));
    CK(hipMemset(buf, 0, (size_t)regions * region_bytes));
    {
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        const uint64_t n16 = (uint64_t)regions * region_bytes / 16;
        hipLaunchKernelGGL(k_stream, dim3(256 * 8), dim3(256), 0, 0, (ulonglong2*)buf, n16);
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k_stream, dim3(256 * 8), dim3(256), 0, 0, (ulonglong2*)buf, n16);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
        printf("stream write: %.0f GB/s (%u regions of %llu MB)\n", n16 * 16.0 / (ms * 1e-3) / 1e9, regions, (unsigned long long)(region_bytes >> 20));
    }
    const uint32_t ncurs[] = {1024, 2048, 4096, 8192};
    printf("%8s", "W\\ncur");
    for (uint32_t n : ncurs) printf(" %8u", n);
    printf("   (GB/s of useful bytes, fill 0.7)\n");
#define ROW(W) do { printf("%8d", W); for (uint32_t n : ncurs) printf(" %8.0f", run<W>(buf, regions, region_bytes, n, 0.7)); printf("\n"); fflush(stdout); } while (0)
    ROW(8); ROW(16); ROW(32); ROW(64); ROW(128); ROW(256);
    {   // direct scatter with global cursors (16-byte records), fan-out 2^12 .. 2^23 over the whole buffer
        const uint64_t total = (uint64_t)regions * region_bytes;
        printf("16-byte records through GLOBAL cursors (one returning atomic + one store per record), GB/s of useful bytes / records per s:\n");
        for (uint32_t lg : {12u, 16u, 18u, 20u, 23u}) {
            const uint32_t nc = 1u << lg; const uint64_t bucket = total / nc / 16 * 16;
            uint32_t* cur; CK(hipMalloc(&cur, (size_t)nc * 4));
            const uint64_t stores = (uint64_t)((double)total * 0.7 / 16); const uint32_t wgs = 256 * 4, iters = (uint32_t)(stores / ((uint64_t)wgs * 1024));
            hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
            float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                CK(hipMemset(cur, 0, (size_t)nc * 4)); CK(hipDeviceSynchronize());
                CK(hipEventRecord(a));
                hipLaunchKernelGGL(k_scatter_global, dim3(wgs), dim3(1024), 0, 0, buf, cur, bucket, nc, iters);
                CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
            }
            const double recs = (double)wgs * 1024.0 * iters;
            printf("  2^%-2u cursors: %7.0f GB/s  %.2e records/s  (%.1f ms for 1.07e9 records)\n", lg, recs * 16 / (ms * 1e-3) / 1e9, recs / (ms * 1e-3), 1.07e9 / (recs / (ms * 1e-3)) * 1e3);
            CK(hipFree(cur));
        }
    }
    CK(hipFree(buf));
    return 0;
}
