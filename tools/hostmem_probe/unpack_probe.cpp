// unpack_probe — the host loop of csrc/gkc_sink.hip (7-byte entries -> 16-byte Count records, non-temporal stores) on page-locked buffers, alone and beside a
// device -> host copy stream, with T threads taking 8192-record blocks off an atomic counter.   ./unpack_probe [threads=64] [GB of records=5]
#include <hip/hip_runtime.h>
#include <immintrin.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void unpack(const uint8_t* pay, uint64_t base, uint32_t n, uint8_t* dest)
{
    uint64_t key = base; __m128i* out = (__m128i*)dest;
    for (uint32_t i = 0; i < n; i++) {
        uint64_t w; memcpy(&w, pay + 7 * (size_t)i, 8);
        const uint64_t d = w & 0xFFFFFFFFFFFFull; const uint32_t ab = (uint32_t)(w >> 48) & 255u;
        if (i) key += d;
        _mm_stream_si128(out + i, _mm_set_epi64x((long long)(uint64_t)ab, (long long)key));
    }
}
int main(int argc, char** argv)
{
    const int nt = argc > 1 ? atoi(argv[1]) : 64;
    const size_t B = 8192, nblk = (size_t)((argc > 2 ? atof(argv[2]) : 5.0) * 1e9 / 16 / B);
    uint8_t *stage, *sink, *other; void* d;
    hipHostMalloc((void**)&stage, nblk * B * 7 + 64, 0); hipHostMalloc((void**)&sink, nblk * B * 16, 0); hipHostMalloc((void**)&other, (size_t)4 << 30, 0);
    hipMalloc(&d, (size_t)4 << 30); hipMemset(d, 3, (size_t)4 << 30); hipDeviceSynchronize();
    for (int mode = 0; mode < 4; mode++) {
        // 0: staging written by the CPU; 1: staging written by a device -> host copy; 2: like 1, with another device -> host copy running beside the unpack; 3: like 0, sink = pageable memory
        if (mode == 0 || mode == 3) memset(stage, 5, nblk * B * 7); else for (size_t o = 0; o < nblk * B * 7; o += (size_t)4 << 30) hipMemcpy(stage + o, d, std::min<size_t>((size_t)4 << 30, nblk * B * 7 - o), hipMemcpyDeviceToHost);
        uint8_t* dst = mode == 3 ? (uint8_t*)aligned_alloc(4096, nblk * B * 16) : sink;
        if (mode == 3) memset(dst, 0, nblk * B * 16);
        std::atomic<bool> stop{false};
        hipStream_t st; hipStreamCreate(&st);
        std::thread dma; if (mode == 2) dma = std::thread([&] { while (!stop.load()) { hipMemcpyAsync(other, d, (size_t)4 << 30, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st); } });
        std::atomic<size_t> next{0};
        std::vector<std::thread> th;
        const double t0 = now();
        for (int t = 0; t < nt; t++) th.emplace_back([&] { for (;;) { const size_t g = next.fetch_add(1); if (g >= nblk) break; unpack(stage + g * B * 7, g, B, dst + g * B * 16); } _mm_sfence(); });
        for (auto& x : th) x.join();
        const double dt = now() - t0;
        stop.store(true); if (dma.joinable()) dma.join();
        printf("mode %d, %d threads: %.2f G records/s (%.1f GB/s read + %.1f GB/s written), %.1f ms for %.1f GB of records\n", mode, nt, nblk * B / dt / 1e9, nblk * B * 7 / dt / 1e9, nblk * B * 16 / dt / 1e9, dt * 1e3, nblk * B * 16 / 1e9);
        hipStreamDestroy(st);
    }
    return 0;
}
