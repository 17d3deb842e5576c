// valu_probe.hip -- per-opcode VALU issue-rate probe for gfx950 (wave64 instructions per cycle per SIMD).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define OPS_LIST \
  X(add_u32,      "v_add_u32 %0, %0, %1") \
  X(max_i32,      "v_max_i32 %0, %0, %1") \
  X(min_i32,      "v_min_i32 %0, %0, %1") \
  X(max3_i32,     "v_max3_i32 %0, %0, %1, %1") \
  X(cndmask_vcc,  "v_cndmask_b32 %0, %0, %1, vcc") \
  X(cndmask_sgpr, "v_cndmask_b32_e64 %0, %0, %1, s[10:11]") \
  X(cmp_eq_u32,   "v_cmp_eq_u32 vcc, %0, %1") \
  X(cmp_eq_u16,   "v_cmp_eq_u16 vcc, %0, %1") \
  X(cmp_sgpr,     "v_cmp_eq_u32_e64 s[10:11], %0, %1") \
  X(cmp_cnd_pair, "v_cmp_eq_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc") \
  X(cmpx_pair,    "v_cmp_eq_u32_e64 s[10:11], %0, %1\n v_cndmask_b32_e64 %0, %0, %1, s[10:11]") \
  X(or_b32,       "v_or_b32 %0, %0, %1") \
  X(lshlrev_b32,  "v_lshlrev_b32 %0, 1, %0") \
  X(lshrrev_b32,  "v_lshrrev_b32 %0, 1, %0") \
  X(ashrrev_i32,  "v_ashrrev_i32 %0, 1, %0") \
  X(add_u16,      "v_add_u16 %0, %0, %1") \
  X(sub_u16,      "v_sub_u16 %0, %0, %1") \
  X(max_i16,      "v_max_i16 %0, %0, %1") \
  X(min_i16,      "v_min_i16 %0, %0, %1") \
  X(max_u16,      "v_max_u16 %0, %0, %1") \
  X(min_u16,      "v_min_u16 %0, %0, %1") \
  X(lshlrev_b16,  "v_lshlrev_b16 %0, 1, %0") \
  X(mul_lo_u16,   "v_mul_lo_u16 %0, %0, %1") \
  X(max_f32,      "v_max_f32 %0, %0, %1") \
  X(add_f32,      "v_add_f32 %0, %0, %1") \
  X(max_f16,      "v_max_f16 %0, %0, %1") \
  X(add_f16,      "v_add_f16 %0, %0, %1") \
  X(and_or_b32,   "v_and_or_b32 %0, %0, %1, %1") \
  X(lshl_add_u32, "v_lshl_add_u32 %0, %0, 5, %1") \
  X(bfi_b32,      "v_bfi_b32 %0, %0, %1, %1") \
  X(med3_i32,     "v_med3_i32 %0, %0, %1, %1") \
  X(add_co_u32,   "v_add_co_u32 %0, vcc, %0, %1") \
  X(subrev_u32,   "v_subrev_u32 %0, %0, %1") \
  X(mad_u16,      "v_mad_u16 %0, %0, %1, %1") \
  X(sub_u16_clamp,"v_sub_u16_e64 %0, %0, %1 clamp") \
  X(max_i16_sdwa, "v_max_i16_sdwa %0, %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1") \
  X(add_u16_sdwa, "v_add_u16_sdwa %0, %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1") \
  X(pk_max_i16,   "v_pk_max_i16 %0, %0, %1") \
  X(perm_b32,     "v_perm_b32 %0, %0, %1, %1") \
  X(bfe_u32,      "v_bfe_u32 %0, %0, %1, 8") \
  X(bfe_i32,      "v_bfe_i32 %0, %0, %1, 8") \
  X(xor_b32,      "v_xor_b32 %0, %0, %1") \
  X(and_b32,      "v_and_b32 %0, %0, %1") \
  X(alignbit_b32, "v_alignbit_b32 %0, %0, %1, %1") \
  X(alignbyte_b32,"v_alignbyte_b32 %0, %0, %1, %1") \
  X(lshrrev_b16,  "v_lshrrev_b16 %0, %1, %0") \
  X(lshrrev_b32v, "v_lshrrev_b32 %0, %1, %0") \
  X(perm_add_pair,"v_perm_b32 %0, %0, %1, %1\n v_add_u16 %0, %0, %1") \
  X(cmp_cnd_add,  "v_cmp_eq_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_add_u16 %0, %0, %1") \
  X(mov_dpp,      "v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf") \
  X(xor_subclamp, "v_xor_b32 %0, %0, %1\n v_sub_u16_e64 %0, %1, %0 clamp") \
  X(cnd_dpp_vcc,  "v_cndmask_b32_dpp %0, %0, %1, vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0") \
  X(smov_cnd_dpp, "s_mov_b64 vcc, s[10:11]\n v_cndmask_b32_dpp %0, %0, %1, vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0") \
  X(smov_3cnd_dpp,"s_mov_b64 vcc, s[10:11]\n v_cndmask_b32_dpp %0, %0, %1, vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_cndmask_b32_dpp %0, %0, %1, vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_cndmask_b32_dpp %0, %0, %1, vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0") \
  X(dpp_cnd_sgpr, "v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_cndmask_b32_e64 %0, %0, %1, s[10:11]") \
  X(add_u16_dpp,  "v_add_u16_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0") \
  X(max_i16_dpp,  "v_max_i16_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0") \
  X(mad_u32_u24,  "v_mad_u32_u24 %0, %0, %1, %1") \
  X(s_nop_add,    "s_nop 0\n v_add_u16 %0, %0, %1") \
  X(add_u16_sgpr, "v_add_u16 %0, s10, %0") \
  X(smov_add,     "s_mov_b64 vcc, s[10:11]\n v_add_u16 %0, %0, %1") \
  X(max_i16_inl0, "v_max_i16 %0, 0, %0") \
  X(add_u16_inl7, "v_add_u16 %0, 7, %0") \
  X(max_i16_sgpr, "v_max_i16 %0, s10, %0") \
  X(perm_sgpr,    "v_perm_b32 %0, %0, %1, s10") \
  X(mov_dpp_rshl, "v_mov_b32_dpp %0, %1 row_shl:3 row_mask:0xf bank_mask:0xf bound_ctrl:0") \
  X(and_lit,      "v_and_b32 %0, 0x30000, %0") \
  X(mul_u32_u24,  "v_mul_u32_u24 %0, %0, %1") \
  X(lshl_or,      "v_lshl_or_b32 %0, %0, 3, %1") \
  X(add_lshl,     "v_add_lshl_u32 %0, %0, %1, 1") \
  X(add3,         "v_add3_u32 %0, %0, %1, %1") \
  X(add_i16_clamp,"v_add_i16 %0, %0, %1 clamp") \
  X(sub_i16_clamp,"v_sub_i16 %0, %0, %1 clamp") \
  X(add_u16_clamp,"v_add_u16_e64 %0, %0, %1 clamp") \
  X(add_i16,      "v_add_i16 %0, %0, %1") \
  X(max_i16_e64,  "v_max_i16_e64 %0, %0, %1") \
  X(mov_b32,      "v_mov_b32 %0, %1") \
  X(swap_b32,     "v_swap_b32 %0, %1")

#define X(name, str) \
__global__ void __launch_bounds__(256) k_##name(uint32_t* out, int iters) { \
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    uint32_t b = blockIdx.x | 1; \
    for (int i = 0; i < iters; ++i) { \
        _Pragma("unroll") for (int u = 0; u < 8; ++u) { \
            asm volatile(str : "+v"(a0) : "v"(b) : "vcc"); asm volatile(str : "+v"(a1) : "v"(b) : "vcc"); \
            asm volatile(str : "+v"(a2) : "v"(b) : "vcc"); asm volatile(str : "+v"(a3) : "v"(b) : "vcc"); \
            asm volatile(str : "+v"(a4) : "v"(b) : "vcc"); asm volatile(str : "+v"(a5) : "v"(b) : "vcc"); \
            asm volatile(str : "+v"(a6) : "v"(b) : "vcc"); asm volatile(str : "+v"(a7) : "v"(b) : "vcc"); } } \
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7; }
OPS_LIST
#undef X

int main()
{
    const int blocks = 256 * 8, iters = 2000;            // 8 blocks of 4 waves per CU = 8 waves/SIMD
    uint32_t* out; CHECK(hipMalloc(&out, size_t(blocks) * 256 * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("%-14s %9s %12s %16s\n", "op", "ms", "Twave-op/s", "cyc/instr/SIMD@2.4");
#define X(name, str) { \
    k_##name<<<blocks, 256>>>(out, 10); CHECK(hipDeviceSynchronize()); \
    CHECK(hipEventRecord(e0)); k_##name<<<blocks, 256>>>(out, iters); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); \
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); \
    const double winstr = double(blocks) * 4 * iters * 64; \
    printf("%-14s %9.3f %12.4f %16.2f\n", #name, ms, winstr / ms / 1e9, 1024.0 * 2.4e9 / (winstr / (ms * 1e-3))); }
    OPS_LIST
#undef X
    return 0;
}
