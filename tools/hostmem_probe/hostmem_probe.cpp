// hostmem_probe — what the host's cores get out of the kinds of page-locked memory HIP hands out (the packed result sink, csrc/gkc_sink.hip, reads a staging buffer the
// DMA engines wrote and writes the caller's sink): per kind, read bandwidth (sum of 8-byte words) and non-temporal write bandwidth with 1 / 16 / 64 threads.
//   hipcc -O3 -o hostmem_probe hostmem_probe.cpp -lpthread ; ./hostmem_probe [GB=2]
#include <hip/hip_runtime.h>
#include <immintrin.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void run(const char* name, uint8_t* p, size_t n)
{
    memset(p, 1, n);
    for (int nt : { 1, 16, 64 }) {
        std::vector<std::thread> th; std::vector<uint64_t> sums(nt);
        double t0 = now();
        for (int t = 0; t < nt; t++) th.emplace_back([&, t] { const uint64_t* q = (const uint64_t*)(p + n / nt * t); uint64_t s = 0; for (size_t i = 0; i < n / nt / 8; i++) s += q[i]; sums[t] = s; });
        for (auto& x : th) x.join();
        double rd = n / (now() - t0) / 1e9;
        th.clear(); t0 = now();
        for (int t = 0; t < nt; t++) th.emplace_back([&, t] { __m128i* q = (__m128i*)(p + n / nt * t); const __m128i v = _mm_set1_epi64x(t); for (size_t i = 0; i < n / nt / 16; i++) _mm_stream_si128(q + i, v); _mm_sfence(); });
        for (auto& x : th) x.join();
        double wr = n / (now() - t0) / 1e9;
        printf("%-44s %2d threads: read %7.1f GB/s   nt-write %7.1f GB/s   (%llu)\n", name, nt, rd, wr, (unsigned long long)sums[0]);
    }
}
int main(int argc, char** argv)
{
    const size_t n = (size_t)(argc > 1 ? atof(argv[1]) : 2.0) * (1ull << 30);
    void* p = nullptr;
    p = aligned_alloc(4096, n); run("malloc (pageable)", (uint8_t*)p, n);
    if (hipHostRegister(p, n, hipHostRegisterDefault) == hipSuccess) { run("malloc + hipHostRegister", (uint8_t*)p, n); (void)hipHostUnregister(p); } else printf("hipHostRegister failed\n");
    free(p);
    struct { const char* name; unsigned flags; } kinds[] = { { "hipHostMalloc default", hipHostMallocDefault }, { "hipHostMalloc NonCoherent", hipHostMallocNonCoherent }, { "hipHostMalloc Coherent", hipHostMallocCoherent },
                                                             { "hipHostMalloc Portable|Mapped", hipHostMallocPortable | hipHostMallocMapped }, { "hipHostMalloc WriteCombined", hipHostMallocWriteCombined } };
    for (auto& k : kinds) {
        if (hipHostMalloc(&p, n, k.flags) != hipSuccess) { printf("%s: allocation failed\n", k.name); (void)hipGetLastError(); continue; }
        run(k.name, (uint8_t*)p, n);
        // a device -> host copy into it, for the link rate
        void* d = nullptr; (void)hipMalloc(&d, n); (void)hipMemset(d, 2, n); (void)hipDeviceSynchronize();
        double t0 = now(); (void)hipMemcpy(p, d, n, hipMemcpyDeviceToHost); printf("%-44s device -> host copy %.1f GB/s\n", k.name, n / (now() - t0) / 1e9);
        (void)hipFree(d); (void)hipHostFree(p);
    }
    return 0;
}
