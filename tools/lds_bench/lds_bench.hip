// LDS micro-benchmark for gfx950: throughput and dependent latency of the LDS operations the scatter kernels are built from,
// random addresses over a 64 KB table (8192 x 8 B), 1024 threads per workgroup, one workgroup per CU.
//   hipcc --offload-arch=gfx950 -O3 -o lds_bench lds_bench.hip && ./lds_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int THREADS = 1024, NSLOT = 8192, ITER = 4096;
__device__ __forceinline__ uint32_t rnd(uint32_t& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

// MODE 0 read64, 1 write64, 2 add32 no return, 3 add32 return, 4 cas64 return, 5 exch64 return, 6 read64 + cas64 dependent (the pair protocol),
// 7 add32 return on a 32 KB table, 8 cas32 return
template <int MODE, int ILP>
__global__ __launch_bounds__(THREADS) void k_lds(uint64_t* out, int dep, int active = 64)
{
    __shared__ unsigned long long tab[NSLOT];
    __shared__ uint32_t tab32[NSLOT];
    for (int i = threadIdx.x; i < NSLOT; i += THREADS) { tab[i] = i; tab32[i] = 0; }
    __syncthreads();
    uint32_t s = threadIdx.x * 2654435761u + blockIdx.x;
    unsigned long long acc = 0;
    for (int it = 0; it < ITER; it++) {
        uint32_t q[ILP]; unsigned long long v[ILP] = {};
#pragma unroll
        for (int g = 0; g < ILP; g++) { q[g] = (rnd(s) + (dep ? (uint32_t)acc : 0u)) & (NSLOT - 1); }
#pragma unroll
        for (int g = 0; g < ILP; g++) if ((int)(threadIdx.x & 63) < active) {
            if (MODE == 0) v[g] = *reinterpret_cast<volatile unsigned long long*>(&tab[q[g]]);
            else if (MODE == 1) { *reinterpret_cast<volatile unsigned long long*>(&tab[q[g]]) = s; v[g] = 0; }
            else if (MODE == 2) { atomicAdd(&tab32[q[g]], 1u); v[g] = 0; }
            else if (MODE == 3) v[g] = atomicAdd(&tab32[q[g]], 1u);
            else if (MODE == 4) v[g] = atomicCAS(&tab[q[g]], (unsigned long long)q[g], (unsigned long long)q[g]);
            else if (MODE == 5) v[g] = atomicExch(&tab[q[g]], (unsigned long long)q[g]);
            else if (MODE == 6) { const unsigned long long c = *reinterpret_cast<volatile unsigned long long*>(&tab[q[g]]); v[g] = atomicCAS(&tab[q[g]], c, c); }
            else if (MODE == 8) v[g] = atomicCAS(&tab32[q[g]], 0u, 0u);
        }
#pragma unroll
        for (int g = 0; g < ILP; g++) acc += ((int)(threadIdx.x & 63) < active) ? v[g] : 0;
    }
    out[blockIdx.x * THREADS + threadIdx.x] = acc + tab[threadIdx.x] + tab32[threadIdx.x];
}

template <int MODE, int ILP> int run(const char* name, uint64_t* d_out, int dep)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256;
    hipLaunchKernelGGL((k_lds<MODE, ILP>), dim3(grid), dim3(THREADS), 0, 0, d_out, dep);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_lds<MODE, ILP>), dim3(grid), dim3(THREADS), 0, 0, d_out, dep);
    hipEventRecord(e1); CHECK(hipEventSynchronize(e1));
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double)grid * THREADS * ITER * ILP;          // lane-ops
    const double clk = ms * 1e-3 * 2.4e9;                            // per CU (one workgroup per CU)
    printf("%-28s ILP %d dep %d: %8.3f ms  %7.2f lane-ops/clk/CU  %7.1f clk per wave-instr (16 waves/CU)\n", name, ILP, dep, ms, ops / grid / clk,
           clk / ((double)THREADS / 64 * ITER * ILP) * 1.0);
    return 0;
}
template <int MODE> void sweep(const char* name, uint64_t* d_out)
{
    for (int a : {64, 32, 16, 8, 4, 1}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL((k_lds<MODE, 1>), dim3(256), dim3(THREADS), 0, 0, d_out, 0, a);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_lds<MODE, 1>), dim3(256), dim3(THREADS), 0, 0, d_out, 0, a);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-16s active lanes %2d: %7.3f ms  %6.1f clk per wave-instr\n", name, a, ms, ms * 1e-3 * 2.4e9 / ((double)THREADS / 64 * ITER));
    }
}

int main()
{
    uint64_t* d_out; CHECK(hipMalloc(&d_out, 256 * THREADS * 8));
#define RUN(M, name) run<M, 1>(name, d_out, 0); run<M, 4>(name, d_out, 0); run<M, 1>(name, d_out, 1);
    RUN(0, "read64"); RUN(1, "write64"); RUN(2, "add32 noret"); RUN(3, "add32 ret"); RUN(4, "cas64 ret"); RUN(5, "exch64 ret"); RUN(6, "read64+cas64"); RUN(8, "cas32 ret");
    sweep<5>("exch64 ret", d_out); sweep<3>("add32 ret", d_out); sweep<2>("add32 noret", d_out); sweep<0>("read64", d_out);
    return 0;
}
