// unpack_ranks_probe — what does ONE host give R ranks that all expand packed result batches at the same time? (VERDICT r5 "missing" #2)
//
// At N GPUs on one node every rank lands its own partitions through its own PCIe link, but the expansion of the wire format (csrc/gkc_sink.hip: 6-byte key deltas +
// abundance bitmap + abundance stream -> 16-byte Count records, non-temporal stores) runs on the CPU threads of the ONE host: per record 6.3 B read + 16 B written by
// cores, on top of 6.3 B of DMA writes. This probe runs R "fake ranks" side by side — each with its own staging buffer and sink, first-touched and expanded by T
// threads bound to the NUMA node bench.py would bind the rank to (bind_to_gpu_numa_node: the node of the rank's GPU; GPUs g = 0..7 are taken to hang on node
// g * n_nodes / 8) — and reports the host's aggregate records/s: the ceiling of `value` at N = R with the packed sink. The inner loop is the library's
// (unpack_block_6). Beside it: the aggregate rate of plain non-temporal 16-byte stores by the same threads (what the host's memory controllers take as writes: the
// bound of the RAW sink mode, whose bytes arrive by DMA and are touched by no core).
//   g++ -O3 -march=native -pthread unpack_ranks_probe.cpp -o unpack_ranks_probe ;  ./unpack_ranks_probe [records per rank = 2.5e8] [threads per rank = 24] [rounds = 4]
#include <immintrin.h>
#include <pthread.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static bool cpus_of_node(int node, cpu_set_t* set)
{
    char path[96]; snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE* f = fopen(path, "r"); if (!f) return false;
    char buf[4096]; const bool got = fgets(buf, sizeof buf, f) != nullptr; fclose(f); if (!got) return false;
    CPU_ZERO(set); int n = 0;
    for (char* q = buf; *q; ) { char* e; const long a = strtol(q, &e, 10); if (e == q) break; long b = a; if (*e == '-') { q = e + 1; b = strtol(q, &e, 10); }
        for (long i = a; i <= b && i < CPU_SETSIZE; i++) { CPU_SET((int)i, set); n++; } q = *e == ',' ? e + 1 : e; if (*e != ',') break; }
    return n > 0;
}
constexpr uint32_t BLK = 8192; constexpr uint64_t ENTRIES = (uint64_t)BLK * 6, SLOT = ENTRIES + BLK / 8;
struct Rank {
    uint8_t* stage = nullptr; uint8_t* cb = nullptr; uint8_t* sink = nullptr; uint64_t nblk = 0; std::vector<uint32_t> cb_off; std::atomic<uint64_t> next{0};
};
static void unpack_block_6(const Rank& R, uint64_t g)      // csrc/gkc_sink.hip unpack_block_6 without the (rare) exception look-ups
{
    const uint8_t* pay = R.stage + g * SLOT;
    const uint64_t* bits = reinterpret_cast<const uint64_t*>(pay + ENTRIES);
    const uint8_t* cb = R.cb + R.cb_off[g];
    uint64_t key = g * 0x9E3779B97F4A7C15ull >> 2;
    __m128i* out = reinterpret_cast<__m128i*>(R.sink + g * BLK * 16);
    for (uint32_t i0 = 0; i0 < BLK; i0 += 64) {
        uint64_t m = bits[i0 >> 6];
        for (uint32_t i = i0; i < i0 + 64; i++, m >>= 1) {
            uint64_t w; memcpy(&w, pay + 6 * (size_t)i, 8);
            key += w & 0xFFFFFFFFFFFFull;
            const uint32_t f = (uint32_t)(m & 1ull);
            const uint32_t ab = 1u + f * ((uint32_t)*cb - 1u); cb += f;
            _mm_stream_si128(out + i, _mm_set_epi64x((long long)(uint64_t)ab, (long long)key));
        }
    }
}
static void* numa_alloc(size_t bytes) { void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0); return p == MAP_FAILED ? nullptr : p; }

int main(int argc, char** argv)
{
    const uint64_t n_rec = (uint64_t)(argc > 1 ? atof(argv[1]) : 2.5e8) / BLK * BLK;
    const int T0 = argc > 2 ? atoi(argv[2]) : (int)std::min<unsigned>(24u, std::max(2u, std::thread::hardware_concurrency() / 2));
    const int rounds = argc > 3 ? atoi(argv[3]) : 4;
    int n_nodes = 0; { cpu_set_t s; while (cpus_of_node(n_nodes, &s)) n_nodes++; }
    if (n_nodes == 0) n_nodes = 1;
    printf("host: %u hardware threads, %d NUMA node(s); up to %d unpack threads per rank (the library's default for one rank), %.2e records per rank and round, %d rounds\n",
           std::thread::hardware_concurrency(), n_nodes, T0, (double)n_rec, rounds);
    printf("%-6s %-12s %-8s %14s %14s %12s %12s | %18s\n", "ranks", "rank->node", "thr/rank", "records/s", "per rank", "GB/s read", "GB/s written", "NT stores only GB/s");
    for (int spread = 0; spread < 2; spread++)
    for (int R : {1, 2, 4, 8}) {
        if (spread && (R == 8 || n_nodes == 1)) continue;
        std::vector<Rank> ranks((size_t)R);
        std::vector<int> node((size_t)R);
        char map[128]; int mo = 0;
        for (int r = 0; r < R; r++) { node[r] = spread ? (r * n_nodes / R) : (r * n_nodes / 8); mo += snprintf(map + mo, sizeof map - mo, "%d", node[r]); }
        // buffers first-touched on the rank's node
        {   std::vector<std::thread> init;
            for (int r = 0; r < R; r++) init.emplace_back([&, r] {
                cpu_set_t set; if (cpus_of_node(node[r], &set)) (void)pthread_setaffinity_np(pthread_self(), sizeof set, &set);
                Rank& K = ranks[r]; K.nblk = n_rec / BLK;
                K.stage = (uint8_t*)numa_alloc(K.nblk * SLOT + 64); K.cb = (uint8_t*)numa_alloc(n_rec / 4 + 4096); K.sink = (uint8_t*)numa_alloc(n_rec * 16);
                if (!K.stage || !K.cb || !K.sink) { fprintf(stderr, "allocation failed\n"); exit(1); }
                uint64_t x = 0x1234567ull + r; K.cb_off.resize(K.nblk); uint64_t cbo = 0;
                for (uint64_t g = 0; g < K.nblk; g++) {
                    uint8_t* pay = K.stage + g * SLOT; uint64_t* bits = (uint64_t*)(pay + ENTRIES);
                    for (uint32_t i = 0; i < BLK; i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; const uint64_t d = x & 0x3FFFFFFFFFFull; memcpy(pay + 6 * (size_t)i, &d, 6); }
                    uint32_t flagged = 0;
                    for (uint32_t wq = 0; wq < BLK / 64; wq++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; uint64_t y = x; x ^= x << 13; x ^= x >> 7; x ^= x << 17; const uint64_t mm = x & y & (x >> 21 | y << 5); bits[wq] = mm; flagged += (uint32_t)__builtin_popcountll(mm); }   // ~16-19 % flagged
                    K.cb_off[g] = (uint32_t)cbo; memset(K.cb + cbo, 2, flagged); cbo += flagged;
                }
                memset(K.sink, 0, n_rec * 16);
            });
            for (auto& t : init) t.join();
        }
        for (int T : {T0, T0 / 2, T0 / 4, T0 / 8}) {
        if (T < 1 || (T != T0 && R == 1)) continue;
        double rate = 0, rd = 0, wr = 0, nt = 0;
        for (int what = 0; what < 2; what++) {      // 0: the unpack, 1: non-temporal stores only
            for (Rank& K : ranks) K.next.store(0);
            std::atomic<int> ready{0}; std::atomic<bool> go{false};
            std::vector<std::thread> th;
            for (int r = 0; r < R; r++) for (int t = 0; t < T; t++) th.emplace_back([&, r] {
                cpu_set_t set; if (cpus_of_node(node[r], &set)) (void)pthread_setaffinity_np(pthread_self(), sizeof set, &set);
                Rank& K = ranks[r];
                ready.fetch_add(1); while (!go.load()) { }
                for (;;) {
                    const uint64_t g = K.next.fetch_add(1); if (g >= K.nblk * (uint64_t)rounds) break;
                    const uint64_t b = g % K.nblk;
                    if (what == 0) unpack_block_6(K, b);
                    else { __m128i* out = reinterpret_cast<__m128i*>(K.sink + b * BLK * 16); const __m128i v = _mm_set_epi64x(1, (long long)b); for (uint32_t i = 0; i < BLK; i++) _mm_stream_si128(out + i, v); }
                }
                _mm_sfence();
            });
            while (ready.load() < R * T) { }
            const double t0 = now(); go.store(true);
            for (auto& t : th) t.join();
            const double dt = now() - t0, recs = (double)n_rec * rounds * R;
            if (what == 0) { rate = recs / dt; rd = recs * 6.3 / dt / 1e9; wr = recs * 16 / dt / 1e9; } else nt = recs * 16 / dt / 1e9;
        }
        printf("%-6d %-12s %-8d %14.3e %14.3e %12.1f %12.1f | %18.1f\n", R, map, T, rate, rate / R, rd, wr, nt);
        fflush(stdout);
        }
        for (Rank& K : ranks) { munmap(K.stage, K.nblk * SLOT + 64); munmap(K.cb, n_rec / 4 + 4096); munmap(K.sink, n_rec * 16); }
    }
    return 0;
}
