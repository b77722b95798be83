// Does the VOP2 issue rate depend on which registers the two sources name?
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_bank tools/ubench_bank.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

// MODE 0: d = op(d, b)           (one fixed second source)
// MODE 1: d_i = op(d_i, d_{i+S}) (two distinct sources, register distance S)
template <int MODE, int S>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t b, uint32_t c) {
    uint32_t a[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) a[i] = threadIdx.x * 0x10003u + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (MODE == 0) asm volatile("v_min_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                else asm volatile("v_min_u16 %0, %1, %2" : "=v"(a[i]) : "v"(a[(i + S) & 31]), "v"(a[(i + 2 * S + 1) & 31]));
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (MODE == 0) asm volatile("v_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                else asm volatile("v_add_u16 %0, %1, %2" : "=v"(a[16 + i]) : "v"(a[(i + S) & 31]), "v"(a[(i + 3 * S + 2) & 31]));
            }
        }
    }
    uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) x ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = x;
}

template <int MODE, int S>
void run(uint32_t* out, int bpc, const char* name) {
    const int blocks = 256 * bpc, iters = 2048;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<MODE, S>), dim3(blocks), dim3(256), 0, 0, out, iters, 0x00030005u, 0x7ff07ff1u);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    double lane_instr = (double)blocks * 256 * iters * 4 * 32;
    printf("%-28s waves/SIMD %d : %8.3f ms  %6.1f lanes/clk/CU @2.4GHz\n", name, bpc, best,
           lane_instr / (best * 1e-3) / 256 / 2.4e9);
}

int main() {
    uint32_t* out;
    if (hipMalloc(&out, 256 * 8 * 256 * 4) != hipSuccess) return 1;
    for (int bpc : {2, 4, 8}) {
        run<0, 0>(out, bpc, "d=op(d,const)");
        run<1, 1>(out, bpc, "d=op(s,s') stride 1");
        run<1, 2>(out, bpc, "d=op(s,s') stride 2");
        run<1, 4>(out, bpc, "d=op(s,s') stride 4");
        run<1, 5>(out, bpc, "d=op(s,s') stride 5");
        run<1, 8>(out, bpc, "d=op(s,s') stride 8");
    }
    return 0;
}
