// Issue-rate probe for single VALU instructions on gfx950.
// Each kernel runs one instruction form on 16 independent registers per lane,
// 8 workgroups x 4 wavefronts per CU, and reports lane-instructions per clock
// per CU at an assumed 2.4 GHz plus the wall rate.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_valu tools/ubench_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define KERNEL(NAME, ASM)                                                              \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, int iters, uint32_t b, uint32_t c) { \
        uint32_t a[16];                                                                \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0x10003u + i; \
        for (int it = 0; it < iters; ++it) {                                           \
            _Pragma("unroll") for (int u = 0; u < 8; ++u) {                            \
                _Pragma("unroll") for (int i = 0; i < 16; ++i)                         \
                    asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c));                   \
            }                                                                          \
        }                                                                              \
        uint32_t x = 0;                                                                \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) x ^= a[i];                      \
        out[blockIdx.x * 256 + threadIdx.x] = x;                                       \
    }

KERNEL(k_pk_add_u16, "v_pk_add_u16 %0, %0, %1")
KERNEL(k_pk_min_i16, "v_pk_min_i16 %0, %0, %1")
KERNEL(k_pk_min_u16, "v_pk_min_u16 %0, %0, %1")
KERNEL(k_pk_mad_u16, "v_pk_mad_u16 %0, %0, %1, %2")
KERNEL(k_pk_mul_lo_u16, "v_pk_mul_lo_u16 %0, %0, %1")
KERNEL(k_pk_sub_u16, "v_pk_sub_u16 %0, %0, %1")
KERNEL(k_pk_max_i16, "v_pk_max_i16 %0, %0, %1")
KERNEL(k_pk_lshlrev_b16, "v_pk_lshlrev_b16 %0, 1, %0")
KERNEL(k_pk_add_u16_opsel, "v_pk_add_u16 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]")
KERNEL(k_add_u32, "v_add_u32 %0, %0, %1")
KERNEL(k_min_i32, "v_min_i32 %0, %0, %1")
KERNEL(k_min3_i32, "v_min3_i32 %0, %0, %1, %2")
KERNEL(k_add3_u32, "v_add3_u32 %0, %0, %1, %2")
KERNEL(k_xor_b32, "v_xor_b32 %0, %0, %1")
KERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %1, 16")
KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2")
KERNEL(k_lshl_or, "v_lshl_or_b32 %0, %0, 16, %1")
KERNEL(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
KERNEL(k_min_i16, "v_min_i16 %0, %0, %1")
KERNEL(k_add_u16, "v_add_u16 %0, %0, %1")
KERNEL(k_min3_i16, "v_min3_i16 %0, %0, %1, %2")
KERNEL(k_mad_u16, "v_mad_u16 %0, %0, %1, %2")
KERNEL(k_mov_dpp, "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL(k_min_i32_dpp, "v_min_i32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL(k_add_u32_dpp, "v_add_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL(k_min_i16_sdwa, "v_min_i16_sdwa %0, %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_1")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL(k_bfi, "v_bfi_b32 %0, %1, %0, %2")
KERNEL(k_mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL(k_xad_u32, "v_xad_u32 %0, %0, %1, %2")
KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, 1, %1")
KERNEL(k_sad_u16, "v_sad_u16 %0, %0, %1, %2")
KERNEL(k_pk_fma_f16, "v_pk_fma_f16 %0, %0, %1, %2")
KERNEL(k_fma_f32, "v_fma_f32 %0, %0, %1, %2")
KERNEL(k_pk_add_min, "v_pk_add_u16 %0, %0, %1\n\tv_pk_min_i16 %0, %0, %2")
KERNEL(k_max3_i16, "v_max3_i16 %0, %0, %1, %2")
KERNEL(k_med3_i32, "v_med3_i32 %0, %0, %1, %2")
KERNEL(k_min_u16, "v_min_u16 %0, %0, %1")
KERNEL(k_max_i16, "v_max_i16 %0, %0, %1")
KERNEL(k_sub_u16, "v_sub_u16 %0, %0, %1")
KERNEL(k_mul_lo_u16, "v_mul_lo_u16 %0, %0, %1")
KERNEL(k_and_b32, "v_and_b32 %0, %0, %1")
KERNEL(k_or_b32, "v_or_b32 %0, %0, %1")
KERNEL(k_lshrrev_b32, "v_lshrrev_b32 %0, 1, %0")
KERNEL(k_lshlrev_b16, "v_lshlrev_b16 %0, 1, %0")
KERNEL(k_mov_b32, "v_mov_b32 %0, %1")
KERNEL(k_sub_u32, "v_sub_u32 %0, %0, %1")
KERNEL(k_max_i32, "v_max_i32 %0, %0, %1")
KERNEL(k_min_u32, "v_min_u32 %0, %0, %1")
KERNEL(k_bfe_u32, "v_bfe_u32 %0, %0, 8, 8")
KERNEL(k_and_sdwa, "v_and_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD")
KERNEL(k_min_i16_sgpr, "v_min_i16 %0, s10, %0")
KERNEL(k_add_u16_lit, "v_add_u16 %0, 12, %0")
KERNEL(k_add_u16_lit32, "v_add_u16 %0, 0x123, %0")
KERNEL(k_cndmask_s, "v_cndmask_b32 %0, %0, %1, s[10:11]")
KERNEL(k_mul_u32_u24, "v_mul_u32_u24 %0, %0, %1")
KERNEL(k_min_f32, "v_min_f32 %0, %0, %1")
KERNEL(k_add_f32, "v_add_f32 %0, %0, %1")
KERNEL(k_min_f16, "v_min_f16 %0, %0, %1")
KERNEL(k_add_f16, "v_add_f16 %0, %0, %1")
KERNEL(k_xor_add16, "v_xor_b32 %0, %0, %1\n\tv_add_u16 %0, %0, %2")

typedef void (*kfn)(uint32_t*, int, uint32_t, uint32_t);
struct Entry { const char* name; kfn fn; int per; };

int main() {
    Entry es[] = {
#define E(n) {#n, n, 1}
        E(k_pk_add_u16), E(k_pk_min_i16), E(k_pk_min_u16), E(k_pk_mad_u16), E(k_pk_mul_lo_u16), E(k_pk_sub_u16),
        E(k_pk_max_i16), E(k_pk_lshlrev_b16), E(k_pk_add_u16_opsel), E(k_add_u32), E(k_min_i32), E(k_min3_i32),
        E(k_add3_u32), E(k_xor_b32), E(k_alignbit), E(k_perm), E(k_lshl_or), E(k_and_or), E(k_min_i16), E(k_add_u16),
        E(k_min3_i16), E(k_mad_u16), E(k_mov_dpp), E(k_min_i32_dpp), E(k_add_u32_dpp), E(k_min_i16_sdwa), E(k_cndmask),
        E(k_bfi), E(k_mad_u32_u24), E(k_xad_u32), E(k_lshl_add), E(k_sad_u16), E(k_pk_fma_f16), E(k_fma_f32),
        {"k_pk_add_min(2 instr)", k_pk_add_min, 2}, E(k_max3_i16), E(k_med3_i32),
        E(k_min_u16), E(k_max_i16), E(k_sub_u16), E(k_mul_lo_u16), E(k_and_b32), E(k_or_b32), E(k_lshrrev_b32),
        E(k_lshlrev_b16), E(k_mov_b32), E(k_sub_u32), E(k_max_i32), E(k_min_u32), E(k_bfe_u32), E(k_and_sdwa),
        E(k_min_i16_sgpr), E(k_add_u16_lit), E(k_add_u16_lit32), E(k_cndmask_s), E(k_mul_u32_u24), E(k_min_f32),
        E(k_add_f32), E(k_min_f16), E(k_add_f16), {"k_xor_add16(2 instr)", k_xor_add16, 2},
    };
    const int blocks = 256 * 8, iters = 2048;
    uint32_t* out;
    if (hipMalloc(&out, blocks * 256 * 4) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (auto& e : es) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, iters, 0x00030005u, 0x7ff07ff1u);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep && ms < best) best = ms;
        }
        if (hipGetLastError() != hipSuccess) { printf("%-28s launch error\n", e.name); continue; }
        double lane_instr = (double)blocks * 256 * iters * 8 * 16 * e.per;
        double rate = lane_instr / (best * 1e-3);
        printf("%-28s %8.3f ms  %7.2f T lane-instr/s  %6.1f lanes/clk/CU @2.4GHz\n", e.name, best, rate / 1e12,
               rate / 256 / 2.4e9);
    }
    return 0;
}
