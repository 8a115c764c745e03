// unpack_numa_probe — does the host expansion of csrc/gkc_sink.hip (7-byte entries -> 16-byte Count records, non-temporal stores) get faster when the SINK is
// interleaved over the NUMA nodes of the box and the threads run on all of them? One device -> host copy stream runs beside every measurement.
//   ./unpack_numa_probe [GB of records=5]      prints records/s for: sink local / interleaved  x  threads 16, 24, 32, 48, 64  x  pinned to the staging node / to all nodes
#include <hip/hip_runtime.h>
#include <immintrin.h>
#include <pthread.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <unistd.h>
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
static int node_of(const void* p) { int node = -1; if (syscall(SYS_get_mempolicy, &node, nullptr, 0ul, const_cast<void*>(p), 3ul) != 0) return -1; return node; }
static bool cpus_of_node(int node, cpu_set_t* set)
{
    char path[96]; snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE* f = fopen(path, "r"); if (!f) return false;
    char buf[4096]; const bool got = fgets(buf, sizeof buf, f) != nullptr; fclose(f); if (!got) return false;
    CPU_ZERO(set);
    for (char* q = buf; *q; ) { char* e; const long a = strtol(q, &e, 10); if (e == q) break; long b = a; if (*e == '-') { q = e + 1; b = strtol(q, &e, 10); }
        for (long i = a; i <= b && i < CPU_SETSIZE; i++) CPU_SET((int)i, set); q = *e == ',' ? e + 1 : e; if (*e != ',') break; }
    return true;
}
int main(int argc, char** argv)
{
    const size_t B = 8192, nblk = (size_t)((argc > 1 ? atof(argv[1]) : 5.0) * 1e9 / 16 / B);
    int n_nodes = 0; { cpu_set_t s; while (cpus_of_node(n_nodes, &s)) n_nodes++; }
    printf("NUMA nodes: %d\n", n_nodes);
    uint8_t *stage, *sink_local, *sink_il = nullptr, *other; void* d;
    hipHostMalloc((void**)&stage, nblk * B * 7 + 64, 0); hipHostMalloc((void**)&sink_local, nblk * B * 16, 0); hipHostMalloc((void**)&other, (size_t)4 << 30, 0);
    {   // interleaved over all nodes: anonymous pages with an MPOL_INTERLEAVE binding, touched, then page-locked for the device (hipHostRegister) —
        // (hipHostMalloc with hipHostMallocNumaUser under set_mempolicy(MPOL_INTERLEAVE) put every page on the caller's node: measured, first version of this probe)
        const size_t bytes = nblk * B * 16;
        void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        unsigned long mask = (1ul << n_nodes) - 1ul;
        const long rc = p == MAP_FAILED ? -1 : syscall(SYS_mbind, p, bytes, 3 /* MPOL_INTERLEAVE */, &mask, sizeof(mask) * 8, 0u);
        if (p != MAP_FAILED) { const double t0 = now(); memset(p, 0, bytes); const double t1 = now();
            const hipError_t e = hipHostRegister(p, bytes, hipHostRegisterPortable); const double t2 = now();
            printf("mbind rc %ld, touch %.2f s, hipHostRegister %s in %.2f s; pages of the interleaved sink on nodes:", rc, t1 - t0, hipGetErrorString(e), t2 - t1);
            sink_il = (uint8_t*)p;
            for (int i = 0; i < 8; i++) printf(" %d", node_of(sink_il + (size_t)i * 4096)); }
        printf("; stage on node %d, local sink on node %d\n", node_of(stage), node_of(sink_local));
    }
    hipMalloc(&d, (size_t)4 << 30); hipMemset(d, 3, (size_t)4 << 30); hipDeviceSynchronize();
    for (size_t o = 0; o < nblk * B * 7; o += (size_t)4 << 30) hipMemcpy(stage + o, d, std::min<size_t>((size_t)4 << 30, nblk * B * 7 - o), hipMemcpyDeviceToHost);
    const int stage_node = node_of(stage);
    for (int il = 0; il < 2; il++) for (int pin = 0; pin < 2; pin++) for (int nt : {16, 24, 32, 40, 48, 64}) {
        uint8_t* dst = il ? sink_il : sink_local; if (!dst) continue;
        std::atomic<bool> stop{false};
        hipStream_t st; hipStreamCreate(&st);
        std::thread dma([&] { while (!stop.load()) { hipMemcpyAsync(other, d, (size_t)1 << 30, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st); } });
        std::atomic<size_t> next{0};
        std::vector<std::thread> th;
        cpu_set_t set; bool have = false;
        if (pin == 0) have = cpus_of_node(stage_node, &set);
        const double t0 = now();
        for (int t = 0; t < nt; t++) { th.emplace_back([&] { for (;;) { const size_t g = next.fetch_add(1); if (g >= nblk) break; unpack(stage + g * B * 7, g, B, dst + g * B * 16); } _mm_sfence(); });
                                       if (have) pthread_setaffinity_np(th.back().native_handle(), sizeof(set), &set); }
        for (auto& x : th) x.join();
        const double dt = now() - t0;
        stop.store(true); dma.join();
        printf("sink %-11s threads %-22s %2d: %5.2f G records/s (%5.1f GB/s read + %5.1f GB/s written), %5.1f ms per 3.1e8 records\n", il ? "interleaved" : "local", pin == 0 ? "on the staging node" : "anywhere", nt,
               nblk * B / dt / 1e9, nblk * B * 7 / dt / 1e9, nblk * B * 16 / dt / 1e9, dt * 1e3 * 3.1e8 / (nblk * B));
        hipStreamDestroy(st);
    }
    return 0;
}
