// Probe of the HIP virtual-memory API on the box: can a reserved address range be grown chunk by chunk while kernels write into it, and what does a chunk cost?
//   hipcc --offload-arch=gfx950 -O2 -o vmm_probe vmm_probe.hip && ./vmm_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAIL %s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_fill(uint64_t* p, uint64_t n) { for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) p[i] = i * 0x9E3779B97F4A7C15ULL; }
__global__ void k_sum(const uint64_t* p, uint64_t n, unsigned long long* out) { unsigned long long s = 0; for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) s += p[i]; atomicAdd(out, s); }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv)
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int mode = argc > 1 ? atoi(argv[1]) : 0;      // 0: kernels only after the range is complete; 1: a kernel writes the mapped part while the next chunk is added
    int dev = 0; CK(hipSetDevice(dev));
    int vmm = 0; CK(hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported, dev));
    printf("virtual memory management supported: %d\n", vmm);
    hipMemAllocationProp prop{}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
    size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    size_t gmin = 0; CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
    printf("granularity: recommended %zu, minimum %zu\n", gran, gmin);
    const size_t VA = (size_t)160 << 30;
    void* base = nullptr; double t0 = now(); CK(hipMemAddressReserve(&base, VA, 0, nullptr, 0)); printf("reserve 160 GiB: %.3f ms\n", (now() - t0) * 1e3);
    hipMemAccessDesc acc{}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    unsigned long long* d_sum; CK(hipMalloc(&d_sum, 8));
    hipStream_t st, st2; CK(hipStreamCreate(&st)); CK(hipStreamCreate(&st2));
    uint64_t* busy; CK(hipMalloc(&busy, (size_t)4 << 30));
    for (size_t chunk : { (size_t)64 << 20, (size_t)512 << 20, (size_t)2 << 30 }) {
        std::vector<hipMemGenericAllocationHandle_t> hs; size_t off = 0;
        double tc = 0, tm = 0, ta = 0; const int N = 8;
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, st2, busy, ((size_t)4 << 30) / 8);          // another stream is busy meanwhile: do the calls wait for it?
        for (int i = 0; i < N; i++) {
            hipMemGenericAllocationHandle_t h; double a = now(); CK(hipMemCreate(&h, chunk, &prop, 0)); double b = now();
            CK(hipMemMap((char*)base + off, chunk, 0, h, 0)); double c = now();
            CK(hipMemSetAccess((char*)base + off, chunk, &acc, 1)); double d = now();
            tc += b - a; tm += c - b; ta += d - c; hs.push_back(h); off += chunk;
            // a kernel already running on the range mapped so far, while the next chunk is being added
            if (mode == 1) hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, st, (uint64_t*)base, off / 8);
        }
        if (mode == 2) CK(hipMemSetAccess(base, off, &acc, 1));
        hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, st, (uint64_t*)base, off / 8);
        CK(hipStreamSynchronize(st)); CK(hipStreamSynchronize(st2));
        CK(hipMemsetAsync(d_sum, 0, 8, st));
        hipLaunchKernelGGL(k_sum, dim3(2048), dim3(256), 0, st, (const uint64_t*)base, off / 8, d_sum);
        unsigned long long got = 0; CK(hipMemcpyAsync(&got, d_sum, 8, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
        unsigned long long want = 0; for (uint64_t i = 0; i < off / 8; i++) want += i * 0x9E3779B97F4A7C15ULL;
        // device -> pinned host straight from the mapped range, across a chunk boundary
        void* h_pin; CK(hipHostMalloc(&h_pin, 2 * chunk > ((size_t)256 << 20) ? ((size_t)256 << 20) : 2 * chunk, 0));
        const size_t cp = 2 * chunk > ((size_t)256 << 20) ? ((size_t)256 << 20) : 2 * chunk; const size_t from = chunk - cp / 2;
        double a = now(); CK(hipMemcpyAsync(h_pin, (char*)base + from, cp, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); double b = now();
        bool okc = true; for (size_t i = 0; i < cp / 8; i += 4097) okc = okc && ((uint64_t*)h_pin)[i] == (from / 8 + i) * 0x9E3779B97F4A7C15ULL;
        CK(hipHostFree(h_pin));
        double tu = now();
        for (int i = 0; i < N; i++) { CK(hipMemUnmap((char*)base + (size_t)i * chunk, chunk)); CK(hipMemRelease(hs[i])); }
        tu = now() - tu;
        printf("chunk %5zu MiB: create %.3f ms, map %.3f ms, set access %.3f ms per chunk; unmap+release %.3f ms per chunk; kernel over %d chunks %s; D2H across a boundary %s (%.1f GB/s)\n",
               chunk >> 20, tc / N * 1e3, tm / N * 1e3, ta / N * 1e3, tu / N * 1e3, N, got == want ? "ok" : "WRONG", okc ? "ok" : "WRONG", cp / (b - a) * 1e-9);
    }
    // plain hipMalloc / hipFree of the same sizes, for scale
    for (size_t chunk : { (size_t)512 << 20, (size_t)8 << 30 }) { void* p; double a = now(); CK(hipMalloc(&p, chunk)); double b = now(); CK(hipFree(p)); double c = now(); printf("hipMalloc %zu MiB %.3f ms, hipFree %.3f ms\n", chunk >> 20, (b - a) * 1e3, (c - b) * 1e3); }
    CK(hipMemAddressFree(base, VA));
    printf("done\n");
    return 0;
}
