// developer probe: verifies on the GPU that the DPP / permlane-swap sequences equal __shfl_xor for every mask the sort uses
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int M> __device__ unsigned lx(unsigned v) {
    const unsigned lane = threadIdx.x & 63;
    if constexpr (M == 4) {
        unsigned r = __builtin_amdgcn_update_dpp(v, v, 0x104, 0xF, 0x5, false);   // row_shl:4 into banks 0,2 (lane reads lane+4)
        return __builtin_amdgcn_update_dpp(r, v, 0x114, 0xF, 0xA, false);          // row_shr:4 into banks 1,3 (lane reads lane-4)
    } else if constexpr (M == 16) {
        auto p = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        return (lane & 16) ? p[0] : p[1];
    } else if constexpr (M == 32) {
        auto p = __builtin_amdgcn_permlane32_swap(v, v, false, false);
        return (lane & 32) ? p[0] : p[1];
    } else if constexpr (M == 31) {
        unsigned m = (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x140, 0xF, 0xF, true);
        return lx<16>(m);
    } else if constexpr (M == 63) {
        return lx<32>(lx<31>(v));
    } else return 0;
}
__global__ void k(unsigned* out) {
    unsigned v = threadIdx.x * 7 + 3;
    out[0 * 64 + threadIdx.x] = lx<4>(v) ^ (unsigned)__shfl_xor((int)v, 4, 64);
    out[1 * 64 + threadIdx.x] = lx<16>(v) ^ (unsigned)__shfl_xor((int)v, 16, 64);
    out[2 * 64 + threadIdx.x] = lx<32>(v) ^ (unsigned)__shfl_xor((int)v, 32, 64);
    out[3 * 64 + threadIdx.x] = lx<31>(v) ^ (unsigned)__shfl_xor((int)v, 31, 64);
    out[4 * 64 + threadIdx.x] = lx<63>(v) ^ (unsigned)__shfl_xor((int)v, 63, 64);
}
int main() {
    unsigned* d; hipMalloc(&d, 5 * 64 * 4); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned h[320]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const int masks[5] = {4, 16, 32, 31, 63};
    for (int m = 0; m < 5; m++) { int bad = 0; for (int i = 0; i < 64; i++) bad += h[m * 64 + i] != 0; printf("xor %d: %s\n", masks[m], bad ? "MISMATCH" : "ok"); }
    return 0;
}
