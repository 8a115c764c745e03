// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access patterns of this library (the x2 of MI355X_MICROARCH.md is established for 16-byte-per-lane
// coalesced streaming reads only).   hipcc -O2 --offload-arch=gfx950 fetch_calib.hip -o fetch_calib
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -- ./fetch_calib      then tools/rocprof_summary.py out/.../*.db
// Every kernel reads a 4 GiB buffer (16x the Infinity Cache) exactly once (streams) or 2^28 random elements of it (gathers) and adds what it read into one word.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <typename T> __device__ uint64_t fold(T v);
template <> __device__ uint64_t fold<uint8_t>(uint8_t v) { return v; }
template <> __device__ uint64_t fold<uint32_t>(uint32_t v) { return v; }
template <> __device__ uint64_t fold<uint64_t>(uint64_t v) { return v; }
template <> __device__ uint64_t fold<ulonglong2>(ulonglong2 v) { return v.x ^ v.y; }
template <typename T> __global__ void k_stream(const T* __restrict__ p, uint64_t n, unsigned long long* out)
{
    uint64_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) acc ^= fold<T>(p[i]);
    if (acc == 0x1234567) atomicAdd(out, 1ull);
}
__device__ __forceinline__ uint64_t mix(uint64_t z) { z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31); }
template <typename T> __global__ void k_gather(const T* __restrict__ p, uint64_t n_elems, uint64_t n_reads, unsigned long long* out)
{
    uint64_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_reads; i += (uint64_t)gridDim.x * blockDim.x) acc ^= fold<T>(p[mix(i) % n_elems]);
    if (acc == 0x1234567) atomicAdd(out, 1ull);
}
// a bucket of 256 consecutive 8-byte keys per WAVE at a random place (the first sort tier's reads), 4 keys per lane
__global__ void k_buckets8(const uint64_t* __restrict__ p, uint64_t n_elems, uint64_t n_buckets, unsigned long long* out)
{
    uint64_t acc = 0; const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((uint64_t)gridDim.x * blockDim.x) >> 6; const int lane = threadIdx.x & 63;
    for (uint64_t b = wave; b < n_buckets; b += nw) { const uint64_t s = (mix(b) % (n_elems / 256)) * 256; for (int r = 0; r < 4; r++) acc ^= p[s + r * 64 + lane]; }
    if (acc == 0x1234567) atomicAdd(out, 1ull);
}
int main()
{
    const size_t bytes = (size_t)4 << 30; void* buf; unsigned long long* out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(buf, 1, bytes); (void)hipMemset(out, 0, 8); (void)hipDeviceSynchronize();
    const dim3 g(256 * 16), b(256);
    hipLaunchKernelGGL(k_stream<ulonglong2>, g, b, 0, 0, (const ulonglong2*)buf, bytes / 16, out);
    hipLaunchKernelGGL(k_stream<uint64_t>, g, b, 0, 0, (const uint64_t*)buf, bytes / 8, out);
    hipLaunchKernelGGL(k_stream<uint32_t>, g, b, 0, 0, (const uint32_t*)buf, bytes / 4, out);
    hipLaunchKernelGGL(k_stream<uint8_t>, g, b, 0, 0, (const uint8_t*)buf, bytes / 4, out);                 // (the first GiB, one byte per lane)
    const uint64_t nr = 1ull << 28;
    hipLaunchKernelGGL(k_gather<ulonglong2>, g, b, 0, 0, (const ulonglong2*)buf, bytes / 16, nr, out);
    hipLaunchKernelGGL(k_gather<uint64_t>, g, b, 0, 0, (const uint64_t*)buf, bytes / 8, nr, out);
    hipLaunchKernelGGL(k_gather<uint32_t>, g, b, 0, 0, (const uint32_t*)buf, bytes / 4, nr, out);
    hipLaunchKernelGGL(k_gather<uint8_t>, g, b, 0, 0, (const uint8_t*)buf, bytes, nr, out);
    hipLaunchKernelGGL(k_buckets8, g, b, 0, 0, (const uint64_t*)buf, bytes / 8, (uint64_t)1 << 21, out);      // 2^21 buckets x 2 KiB = 4 GiB
    (void)hipDeviceSynchronize();
    printf("expected bytes: stream16 / stream8 / stream4 %zu each, stream1 %zu; gathers: 2^28 reads of 16 / 8 / 4 / 1 bytes (%.2f / %.2f / %.2f / %.2f GB of payload, %.2f GB if a 64-byte line each); buckets8 %zu\n",
           bytes, bytes / 4, nr * 16 / 1e9, nr * 8 / 1e9, nr * 4 / 1e9, nr * 1 / 1e9, nr * 64 / 1e9, bytes);
    return 0;
}
