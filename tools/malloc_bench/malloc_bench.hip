// How expensive is device memory on this box, and does it depend on the API? (the Stage-B planner's rule "never allocate in steady state" exists because of this)
// build: hipcc -O2 --offload-arch=gfx950 malloc_bench.hip -o malloc_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv)
{
    const bool async_first = argc > 1;
    const size_t GB = (size_t)1 << 30;
    (void)hipFree(0);
    if (!async_first) for (size_t gb : {1, 8, 32, 64}) {
        void* p = nullptr; double t0 = now();
        hipError_t e = hipMalloc(&p, gb * GB); double t1 = now();
        (void)hipMemset(p, 1, gb * GB); (void)hipDeviceSynchronize(); double t2 = now();
        (void)hipMemset(p, 2, gb * GB); (void)hipDeviceSynchronize(); double t3 = now();
        (void)hipFree(p); double t4 = now();
        printf("hipMalloc %2zu GB: %7.1f ms (%d), first memset %6.1f ms, second %6.1f ms, hipFree %6.1f ms\n", gb, t1 - t0, (int)e, t2 - t1, t3 - t2, t4 - t3);
    }
    {   hipStream_t s; (void)hipStreamCreate(&s);
        for (size_t gb : {32, 64}) {
            void* p = nullptr; double t0 = now();
            hipError_t e = hipMallocAsync(&p, gb * GB, s); (void)hipStreamSynchronize(s); double t1 = now();
            (void)hipMemsetAsync(p, 1, gb * GB, s); (void)hipStreamSynchronize(s); double t2 = now();
            (void)hipFreeAsync(p, s); (void)hipStreamSynchronize(s); double t3 = now();
            void* q = nullptr; e = hipMallocAsync(&q, gb * GB, s); (void)hipStreamSynchronize(s); double t4 = now();
            (void)hipFreeAsync(q, s); (void)hipStreamSynchronize(s);
            printf("hipMallocAsync %2zu GB: %7.1f ms (%d), first memset %6.1f ms, free %6.1f ms, malloc again %6.1f ms\n", gb, t1 - t0, (int)e, t2 - t1, t3 - t2, t4 - t3);
        }
    }
    {   // second hipMalloc of the same size after a free: does the driver keep the pages?
        void* p = nullptr; (void)hipMalloc(&p, 16 * GB); (void)hipFree(p);
        double t0 = now(); (void)hipMalloc(&p, 16 * GB); double t1 = now(); (void)hipFree(p);
        printf("hipMalloc 16 GB right after freeing 16 GB: %7.1f ms\n", t1 - t0);
    }
    return 0;
}
