// Where do the seconds of a process's first pass go? (round 5) hipMalloc vs the virtual-memory API, first touch vs second, first process on the box vs later ones.
//   hipcc -O2 --offload-arch=gfx950 alloc_probe.hip -o alloc_probe;  ./alloc_probe [malloc|vmm] [GB per block] [blocks]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv)
{
    const bool vmm = argc > 1 && !strcmp(argv[1], "vmm");
    const size_t gb = argc > 2 ? atoi(argv[2]) : 16, nblk = argc > 3 ? atoi(argv[3]) : 8, GB = (size_t)1 << 30;
    double t00 = now(); (void)hipFree(0); printf("[%s] runtime init %.1f ms\n", vmm ? "vmm" : "malloc", now() - t00);
    std::vector<void*> ps; double ta = 0, t1 = 0, t2 = 0;
    hipMemAllocationProp prop{}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    std::vector<std::vector<hipMemGenericAllocationHandle_t>> hs;
    for (size_t b = 0; b < nblk; b++) {
        void* p = nullptr; double a = now();
        if (!vmm) { if (hipMalloc(&p, gb * GB) != hipSuccess) { printf("hipMalloc failed\n"); return 1; } }
        else {
            if (hipMemAddressReserve(&p, gb * GB, 0, nullptr, 0) != hipSuccess) { printf("reserve failed\n"); return 1; }
            hs.emplace_back();
            for (size_t o = 0; o < gb; o++) { hipMemGenericAllocationHandle_t h; if (hipMemCreate(&h, GB, &prop, 0) != hipSuccess || hipMemMap((char*)p + o * GB, GB, 0, h, 0) != hipSuccess) { printf("create/map failed\n"); return 1; } hs.back().push_back(h); }
            hipMemAccessDesc acc{}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
            if (hipMemSetAccess(p, gb * GB, &acc, 1) != hipSuccess) { printf("set access failed\n"); return 1; }
        }
        double c = now(); ta += c - a;
        (void)hipMemset(p, 1, gb * GB); (void)hipDeviceSynchronize(); double d = now(); t1 += d - c;
        (void)hipMemset(p, 2, gb * GB); (void)hipDeviceSynchronize(); t2 += now() - d;
        ps.push_back(p);
    }
    printf("[%s] %zu blocks of %zu GB: allocate %.1f ms, first memset %.1f ms, second memset %.1f ms (totals)\n", vmm ? "vmm" : "malloc", nblk, gb, ta, t1, t2);
    double f = now();
    for (size_t b = 0; b < nblk; b++) {
        if (!vmm) (void)hipFree(ps[b]);
        else { for (size_t o = 0; o < gb; o++) { (void)hipMemUnmap((char*)ps[b] + o * GB, GB); (void)hipMemRelease(hs[b][o]); } (void)hipMemAddressFree(ps[b], gb * GB); }
    }
    printf("[%s] free %.1f ms\n", vmm ? "vmm" : "malloc", now() - f);
    // again, same process
    double a = now(); void* q = nullptr; (void)hipMalloc(&q, gb * nblk / 2 * GB); double c = now(); (void)hipMemset(q, 1, gb * nblk / 2 * GB); (void)hipDeviceSynchronize(); double d = now();
    printf("[%s] then hipMalloc of %zu GB in the same process: %.1f ms, first memset %.1f ms\n", vmm ? "vmm" : "malloc", gb * nblk / 2, c - a, d - c);
    (void)hipFree(q);
    return 0;
}
