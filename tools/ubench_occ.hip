// VALU issue rate of the DP's instruction mix (v_min_u16 / v_add_u16, VOP2) as a
// function of wavefronts per SIMD and of the number of independent chains.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_occ tools/ubench_occ.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int CH>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t b, uint32_t c) {
    uint32_t a[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) a[i] = threadIdx.x * 0x10003u + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < CH; ++i) asm volatile("v_min_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#pragma unroll
            for (int i = 0; i < CH; ++i) asm volatile("v_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        }
    }
    uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < CH; ++i) x ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = x;
}

template <int CH>
void run(uint32_t* out, int bpc) {
    const int blocks = 256 * bpc, iters = 4096 * 16 / CH / 2;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<CH>, dim3(blocks), dim3(256), 0, 0, out, iters, 0x00030005u, 0x7ff07ff1u);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    double lane_instr = (double)blocks * 256 * iters * 8 * CH * 2;
    printf("chains %2d  waves/SIMD %d : %8.3f ms  %6.1f lanes/clk/CU @2.4GHz\n", CH, bpc, best,
           lane_instr / (best * 1e-3) / 256 / 2.4e9);
}

int main() {
    uint32_t* out;
    if (hipMalloc(&out, 256 * 8 * 256 * 4) != hipSuccess) return 1;
    for (int bpc : {1, 2, 3, 4, 5, 6, 8}) run<16>(out, bpc);
    for (int bpc : {1, 2, 3, 4, 8}) run<4>(out, bpc);
    for (int bpc : {1, 2, 3, 4, 8}) run<2>(out, bpc);
    for (int bpc : {1, 2, 4, 8}) run<1>(out, bpc);
    return 0;
}
