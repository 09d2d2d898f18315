// valu_rate.hip -- what one gfx950 SIMD issues per cycle: cycles per wave64 vector instruction for the instruction
// classes the render kernel is made of, at 1 / 2 / 3 / 4 waves per SIMD, as INDEPENDENT streams (8 accumulators per wave,
// so nothing waits for a result) and, for v_fma_f32, as one dependent chain (latency).  Settles the peak that
// bench.py's roofline prices the render kernel against (the round-3 verdict, item 1b: the hardware guide says 2 cycles
// per wave64 VALU instruction once two waves share a SIMD, the round-3 build assumed 4).
//
// Method: one block per CU (100 KB of dynamic LDS keeps a second block off the CU), 256 w threads = w waves per SIMD;
// every wave brackets N instructions of ONE kind with s_memtime (shader cycles) and s_memrealtime (100 MHz), after a
// block barrier.  cycles per wave-instruction per SIMD = (t1 - t0) / (w N): what the SIMD spends per instruction when
// w waves feed it.  Each wave also records HW_ID / XCC_ID so the host can verify the placement (w waves on each SIMD).
// Two grids: 1 block (an otherwise idle chip: no power management in the way) and 256 blocks (the whole chip busy).
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate scripts/valu_rate.hip && /tmp/valu_rate > profiles/r04_valu_issue_rate.txt
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Stamp { uint64_t cycles, real; uint32_t hw_id, xcc_id; };

// One asm statement holds the whole loop body (16 x 8 instructions): the compiler places nothing between them
// (between separate asm statements it pads with s_nop, which costs an issue slot).  %0..%7 = the 8 accumulators,
// %8 / %9 = two loop-invariant sources (%10 / %11: their low dwords, for 32-bit sources of 64-bit instructions).
#define BODY8(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)
#define REP16(X) X X X X X X X X X X X X X X X X
#define UNROLL 16   // 16 x 8 = 128 instructions per loop trip (1 KiB of code: far inside the instruction cache)

// T = the register type the instruction's destination needs (uint32_t: one VGPR, uint64_t: a pair)
#define DEFK(NAME, T, I)                                                                                         \
  __global__ __launch_bounds__(1024) void NAME(int iters, Stamp* out) {                                          \
    T a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    T x = (T)0x3c003c00u + threadIdx.x, y = (T)0x38003800u;                                                      \
    __syncthreads();                                                                                             \
    const uint64_t r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();                     \
    for (int i = 0; i < iters; i++) {                                                                            \
      asm volatile(REP16(BODY8(I))                                                                               \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)               \
                   : "v"(x), "v"(y), "v"((uint32_t)x), "v"((uint32_t)y) : "vcc", "s20", "s21", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");                                                             \
    }                                                                                                            \
    const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();                     \
    T s = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                                                                 \
    if ((threadIdx.x & 63) == 0 || s == (T)0x12345679u) {                                                       \
      Stamp st{t1 - t0, r1 - r0, (uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | 4),                            \
               (uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | 20)};                                            \
      out[blockIdx.x * 16 + (threadIdx.x >> 6)] = st;                                                            \
    }                                                                                                            \
  }

#define S_FMA(d) "v_fma_f32 %" #d ", %8, %9, %" #d "\n"
#define S_FMA_DEP(d) "v_fma_f32 %0, %0, %8, %9\n"
#define S_MUL(d) "v_mul_f32 %" #d ", %8, %" #d "\n"
#define S_ADDU(d) "v_add_u32 %" #d ", %8, %" #d "\n"
#define S_LSHL(d) "v_lshlrev_b32 %" #d ", 1, %" #d "\n"
#define S_AND(d) "v_and_b32 %" #d ", %8, %" #d "\n"
#define S_XOR(d) "v_xor_b32 %" #d ", %8, %" #d "\n"
#define S_MOV(d) "v_mov_b32 %" #d ", %8\n"
#define S_CNDMASK(d) "v_cndmask_b32 %" #d ", %" #d ", %8, vcc\n"
#define S_MAD24(d) "v_mad_u32_u24 %" #d ", %8, %9, %" #d "\n"
#define S_MULLO(d) "v_mul_lo_u32 %" #d ", %8, %" #d "\n"
#define S_MADU64(d) "v_mad_u64_u32 %" #d ", vcc, %10, %11, %" #d "\n"
#define S_BITOP3(d) "v_bitop3_b32 %" #d ", %" #d ", %8, %9 bitop3:0x96\n"
#define S_ANDOR(d) "v_and_or_b32 %" #d ", %" #d ", %8, %9\n"
#define S_LSHLADD(d) "v_lshl_add_u32 %" #d ", %" #d ", 1, %8\n"
#define S_PKFMA16(d) "v_pk_fma_f16 %" #d ", %8, %9, %" #d "\n"
#define S_PKMUL16(d) "v_pk_mul_f16 %" #d ", %8, %" #d "\n"
#define S_PKMAX16(d) "v_pk_max_f16 %" #d ", %8, %" #d "\n"
#define S_PKADD16(d) "v_pk_add_f16 %" #d ", %8, %" #d "\n"
#define S_CVTPK(d) "v_cvt_pk_f16_f32 %" #d ", %8, %" #d "\n"
#define S_CVTPKRTZ(d) "v_cvt_pkrtz_f16_f32 %" #d ", %8, %" #d "\n"
#define S_FMAMIX(d) "v_fma_mix_f32 %" #d ", %8, %9, %" #d " op_sel_hi:[1,1,0]\n"
#define S_FMAMIXLO(d) "v_fma_mixlo_f16 %" #d ", %8, %9, %" #d " op_sel_hi:[1,1,0]\n"
#define S_CVTU32(d) "v_cvt_u32_f32 %" #d ", %" #d "\n"
#define S_CVTF32(d) "v_cvt_f32_u32 %" #d ", %" #d "\n"
#define S_FRACT(d) "v_fract_f32 %" #d ", %" #d "\n"
#define S_FLOOR(d) "v_floor_f32 %" #d ", %" #d "\n"
#define S_MED3(d) "v_med3_f32 %" #d ", %" #d ", %8, %9\n"
#define S_MAXF(d) "v_max_f32 %" #d ", %8, %" #d "\n"
#define S_EXP(d) "v_exp_f32 %" #d ", %" #d "\n"
#define S_RCP(d) "v_rcp_f32 %" #d ", %" #d "\n"
#define S_PKFMA32(d) "v_pk_fma_f32 %" #d ", %8, %9, %" #d "\n"
#define S_PKMUL32(d) "v_pk_mul_f32 %" #d ", %8, %" #d "\n"
#define S_FMA64(d) "v_fma_f64 %" #d ", %8, %9, %" #d "\n"
#define S_ADD64(d) "v_add_f64 %" #d ", %8, %" #d "\n"
#define S_SNOP(d) "s_nop 0\n"
#define S_CNDMASK64(d) "v_cndmask_b32_e64 %" #d ", %" #d ", %8, s[20:21]\n"
#define S_CNDMASK_NODEP(d) "v_cndmask_b32 %" #d ", %8, %9, vcc\n"
#define S_CMPGT(d) "v_cmp_gt_f32 vcc, %8, %" #d "\n"
#define S_CMPGT64(d) "v_cmp_gt_f32_e64 s[20:21], %8, %" #d "\n"
#define S_CMP_CND(d) "v_cmp_gt_f32 vcc, %8, %" #d "\n v_cndmask_b32 %" #d ", %" #d ", %9, vcc\n"
#define S_ADDF(d) "v_add_f32 %" #d ", %8, %" #d "\n"
#define S_SUBF(d) "v_sub_f32 %" #d ", %8, %" #d "\n"
#define S_MINF(d) "v_min_f32 %" #d ", %8, %" #d "\n"
#define S_SUBU(d) "v_sub_u32 %" #d ", %8, %" #d "\n"
#define S_OR(d) "v_or_b32 %" #d ", %8, %" #d "\n"
#define S_MINU(d) "v_min_u32 %" #d ", %8, %" #d "\n"
#define S_MAXU(d) "v_max_u32 %" #d ", %8, %" #d "\n"
#define S_ADD3(d) "v_add3_u32 %" #d ", %" #d ", %8, %9\n"
#define S_LSHR(d) "v_lshrrev_b32 %" #d ", 1, %" #d "\n"
#define S_ASHR(d) "v_ashrrev_i32 %" #d ", 1, %" #d "\n"
#define S_BFE(d) "v_bfe_u32 %" #d ", %" #d ", 3, 5\n"
#define S_BFI(d) "v_bfi_b32 %" #d ", %8, %9, %" #d "\n"
#define S_PERM(d) "v_perm_b32 %" #d ", %" #d ", %8, %9\n"
#define S_ALIGNBIT(d) "v_alignbit_b32 %" #d ", %" #d ", %8, 16\n"
#define S_FFBL(d) "v_ffbl_b32 %" #d ", %" #d "\n"
#define S_BCNT(d) "v_bcnt_u32_b32 %" #d ", %8, %" #d "\n"
#define S_MBCNT(d) "v_mbcnt_lo_u32_b32 %" #d ", %8, %" #d "\n"
#define S_MULU24(d) "v_mul_u32_u24 %" #d ", %8, %" #d "\n"
#define S_CVTI32(d) "v_cvt_i32_f32 %" #d ", %" #d "\n"
#define S_PKADDU16(d) "v_pk_add_u16 %" #d ", %8, %" #d "\n"
#define S_PKMAXI16(d) "v_pk_max_i16 %" #d ", %8, %" #d "\n"
#define S_PKASHR16(d) "v_pk_ashrrev_i16 %" #d ", 15, %" #d "\n"
#define S_MAXF16(d) "v_max_f16 %" #d ", %8, %" #d "\n"
#define S_FMAF16(d) "v_fma_f16 %" #d ", %8, %9, %" #d "\n"
#define S_DOT2(d) "v_dot2_f32_f16 %" #d ", %8, %9, %" #d "\n"
#define S_DOT2C(d) "v_dot2c_f32_f16 %" #d ", %8, %9\n"
#define S_CVTF16F32(d) "v_cvt_f16_f32 %" #d ", %" #d "\n"
#define S_CVTF32F16(d) "v_cvt_f32_f16 %" #d ", %" #d "\n"
#define S_LDEXP(d) "v_ldexp_f32 %" #d ", %" #d ", 1\n"
#define S_ACCW(d) "v_accvgpr_write_b32 a" #d ", %" #d "\n"
#define S_ACCR(d) "v_accvgpr_read_b32 %" #d ", a" #d "\n"
#define S_MOVDPP(d) "v_mov_b32_dpp %" #d ", %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define S_ADDDPP(d) "v_add_f32_dpp %" #d ", %8, %" #d " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define S_ADDSDWA(d) "v_add_u32_sdwa %" #d ", %8, %" #d " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n"
#define S_MAX3(d) "v_max3_f32 %" #d ", %" #d ", %8, %9\n"
#define S_XAD(d) "v_xad_u32 %" #d ", %" #d ", %8, %9\n"
#define S_LSHLOR(d) "v_lshl_or_b32 %" #d ", %" #d ", 3, %8\n"
#define S_ADDLSHL(d) "v_add_lshl_u32 %" #d ", %" #d ", %8, 3\n"
#define S_MULHI24(d) "v_mul_hi_u32_u24 %" #d ", %8, %" #d "\n"
#define S_PKMULLO16(d) "v_pk_mul_lo_u16 %" #d ", %8, %" #d "\n"
#define S_PKMAD16(d) "v_pk_mad_u16 %" #d ", %8, %9, %" #d "\n"

DEFK(k_fma_f32, uint32_t, S_FMA)
DEFK(k_fma_f32_dependent, uint32_t, S_FMA_DEP)
DEFK(k_mul_f32, uint32_t, S_MUL)
DEFK(k_add_u32, uint32_t, S_ADDU)
DEFK(k_lshlrev_b32, uint32_t, S_LSHL)
DEFK(k_and_b32, uint32_t, S_AND)
DEFK(k_xor_b32, uint32_t, S_XOR)
DEFK(k_mov_b32, uint32_t, S_MOV)
DEFK(k_cndmask_b32, uint32_t, S_CNDMASK)
DEFK(k_mad_u32_u24, uint32_t, S_MAD24)
DEFK(k_mul_lo_u32, uint32_t, S_MULLO)
DEFK(k_mad_u64_u32, uint64_t, S_MADU64)
DEFK(k_bitop3_b32, uint32_t, S_BITOP3)
DEFK(k_and_or_b32, uint32_t, S_ANDOR)
DEFK(k_lshl_add_u32, uint32_t, S_LSHLADD)
DEFK(k_pk_fma_f16, uint32_t, S_PKFMA16)
DEFK(k_pk_mul_f16, uint32_t, S_PKMUL16)
DEFK(k_pk_max_f16, uint32_t, S_PKMAX16)
DEFK(k_pk_add_f16, uint32_t, S_PKADD16)
DEFK(k_cvt_pk_f16_f32, uint32_t, S_CVTPK)
DEFK(k_cvt_pkrtz_f16_f32, uint32_t, S_CVTPKRTZ)
DEFK(k_fma_mix_f32, uint32_t, S_FMAMIX)
DEFK(k_fma_mixlo_f16, uint32_t, S_FMAMIXLO)
DEFK(k_cvt_u32_f32, uint32_t, S_CVTU32)
DEFK(k_cvt_f32_u32, uint32_t, S_CVTF32)
DEFK(k_fract_f32, uint32_t, S_FRACT)
DEFK(k_floor_f32, uint32_t, S_FLOOR)
DEFK(k_med3_f32, uint32_t, S_MED3)
DEFK(k_max_f32, uint32_t, S_MAXF)
DEFK(k_exp_f32, uint32_t, S_EXP)
DEFK(k_rcp_f32, uint32_t, S_RCP)
DEFK(k_pk_fma_f32, uint64_t, S_PKFMA32)
DEFK(k_pk_mul_f32, uint64_t, S_PKMUL32)
DEFK(k_fma_f64, uint64_t, S_FMA64)
DEFK(k_add_f64, uint64_t, S_ADD64)
DEFK(k_s_nop, uint32_t, S_SNOP)
DEFK(k_cndmask64, uint32_t, S_CNDMASK64)
DEFK(k_cndmask_nodep, uint32_t, S_CNDMASK_NODEP)
DEFK(k_cmp_gt, uint32_t, S_CMPGT)
DEFK(k_cmp_gt64, uint32_t, S_CMPGT64)
DEFK(k_cmp_cnd, uint32_t, S_CMP_CND)
DEFK(k_add_f32, uint32_t, S_ADDF)
DEFK(k_sub_f32, uint32_t, S_SUBF)
DEFK(k_min_f32, uint32_t, S_MINF)
DEFK(k_max3_f32, uint32_t, S_MAX3)
DEFK(k_ldexp, uint32_t, S_LDEXP)
DEFK(k_sub_u32, uint32_t, S_SUBU)
DEFK(k_or_b32, uint32_t, S_OR)
DEFK(k_min_u32, uint32_t, S_MINU)
DEFK(k_max_u32, uint32_t, S_MAXU)
DEFK(k_add3, uint32_t, S_ADD3)
DEFK(k_xad, uint32_t, S_XAD)
DEFK(k_lshl_or, uint32_t, S_LSHLOR)
DEFK(k_add_lshl, uint32_t, S_ADDLSHL)
DEFK(k_lshr, uint32_t, S_LSHR)
DEFK(k_ashr, uint32_t, S_ASHR)
DEFK(k_bfe, uint32_t, S_BFE)
DEFK(k_bfi, uint32_t, S_BFI)
DEFK(k_perm, uint32_t, S_PERM)
DEFK(k_alignbit, uint32_t, S_ALIGNBIT)
DEFK(k_ffbl, uint32_t, S_FFBL)
DEFK(k_bcnt, uint32_t, S_BCNT)
DEFK(k_mbcnt, uint32_t, S_MBCNT)
DEFK(k_mul_u24, uint32_t, S_MULU24)
DEFK(k_mulhi24, uint32_t, S_MULHI24)
DEFK(k_cvt_i32, uint32_t, S_CVTI32)
DEFK(k_pk_add_u16, uint32_t, S_PKADDU16)
DEFK(k_pk_max_i16, uint32_t, S_PKMAXI16)
DEFK(k_pk_ashr16, uint32_t, S_PKASHR16)
DEFK(k_pk_mullo16, uint32_t, S_PKMULLO16)
DEFK(k_pk_mad16, uint32_t, S_PKMAD16)
DEFK(k_max_f16, uint32_t, S_MAXF16)
DEFK(k_fma_f16, uint32_t, S_FMAF16)
DEFK(k_dot2, uint32_t, S_DOT2)
DEFK(k_dot2c, uint32_t, S_DOT2C)
DEFK(k_cvt_f16_f32, uint32_t, S_CVTF16F32)
DEFK(k_cvt_f32_f16, uint32_t, S_CVTF32F16)
DEFK(k_accw, uint32_t, S_ACCW)
DEFK(k_accr, uint32_t, S_ACCR)
DEFK(k_mov_dpp, uint32_t, S_MOVDPP)
DEFK(k_add_dpp, uint32_t, S_ADDDPP)
DEFK(k_add_sdwa, uint32_t, S_ADDSDWA)

// v_permlane32_swap exchanges two registers' halves: 4 independent register pairs
__global__ __launch_bounds__(1024) void k_permlane32_swap(int iters, Stamp* out) {
  uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  __syncthreads();
  const uint64_t r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < UNROLL * 2; u++) {
      asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a0), "+v"(a1));
      asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a2), "+v"(a3));
      asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a4), "+v"(a5));
      asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a6), "+v"(a7));
    }
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  uint32_t s = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
  if ((threadIdx.x & 63) == 0 || s == 0x12345679u) {
    Stamp st{t1 - t0, r1 - r0, (uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | 4), (uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | 20)};
    out[blockIdx.x * 16 + (threadIdx.x >> 6)] = st;
  }
}

// The render kernel's MLP phase in miniature: one v_mfma_f32_32x32x16_f16 followed by NFILL independent VALU instructions
// (v_pk_fma_f16 and v_cvt_pk_f16_f32 alternating), 8 MFMAs on 2 accumulators per trip.  Cycles per MFMA per SIMD says how
// many vector instructions ride along an MFMA for free when w waves share the SIMD.
#define F_PKFMA16(a) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(a) : "v"(x), "v"(y));
#define F_CVTPK(a) asm volatile("v_cvt_pk_f16_f32 %0, %1, %0" : "+v"(a) : "v"(x));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
template <int NFILL>
__global__ __launch_bounds__(1024) void k_mfma_fill(int iters, Stamp* out) {
  uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  uint32_t x = 0x3c003c00u + threadIdx.x, y = 0x38003800u;
  half8 A, B;
  for (int k = 0; k < 8; k++) { A[k] = (_Float16)(threadIdx.x * 0.001f + k); B[k] = (_Float16)(k * 0.5f); }
  float16v c0 = {}, c1 = {};
  __syncthreads();
  const uint64_t r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      if (u & 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c1) : "v"(A), "v"(B));
      else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c0) : "v"(A), "v"(B));
#pragma unroll
      for (int f = 0; f < NFILL; f++) {
        switch (f & 7) {
          case 0: F_PKFMA16(a0) break;
          case 1: F_CVTPK(a1) break;
          case 2: F_PKFMA16(a2) break;
          case 3: F_CVTPK(a3) break;
          case 4: F_PKFMA16(a4) break;
          case 5: F_CVTPK(a5) break;
          case 6: F_PKFMA16(a6) break;
          default: F_CVTPK(a7) break;
        }
      }
    }
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  uint32_t s = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
  float cs = 0.f;
  for (int k = 0; k < 16; k++) cs += c0[k] + c1[k];
  if ((threadIdx.x & 63) == 0 || s == 0x12345679u || cs == 1.2345f) {
    Stamp st{t1 - t0, r1 - r0, (uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | 4), (uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | 20)};
    out[blockIdx.x * 16 + (threadIdx.x >> 6)] = st;
  }
}

struct Result { double cyc_per_inst_simd, clock_ghz; bool placement_ok; double worst_wave_cycles; };

typedef void (*kern_t)(int, Stamp*);

static Result run(kern_t k, int w, int blocks, int iters, double insts_per_iter, Stamp* dout, std::vector<Stamp>& h) {
  const size_t lds = 100 * 1024;
  CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CHECK(hipMemset(dout, 0, sizeof(Stamp) * 16 * blocks));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256 * w), lds, 0, iters / 8 + 1, dout);  // warm (clocks, instruction cache)
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256 * w), lds, 0, iters, dout);
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemcpy(h.data(), dout, sizeof(Stamp) * 16 * blocks, hipMemcpyDeviceToHost));
  // the slowest wave of the launch prices the SIMD: all waves start together (barrier) and run the same stream
  double sum_c = 0, sum_r = 0, worst = 0;
  std::map<uint64_t, int> per_simd;
  int n = 0;
  for (int b = 0; b < blocks; b++)
    for (int v = 0; v < 4 * w; v++) {
      const Stamp& s = h[b * 16 + v];
      sum_c += (double)s.cycles; sum_r += (double)s.real; n++;
      worst = std::max(worst, (double)s.cycles);
      // HW_ID: simd_id [5:4], cu_id [11:8], sh_id [12], se_id [15:13]; XCC_ID [3:0]
      const uint64_t key = ((uint64_t)(s.xcc_id & 0xf) << 32) | (s.hw_id & 0xfff0u & ~0xc0u);
      per_simd[key]++;
    }
  bool ok = (int)per_simd.size() == blocks * 4;
  for (auto& kv : per_simd) ok = ok && kv.second == w;
  Result r;
  r.cyc_per_inst_simd = (sum_c / n) / (w * insts_per_iter * iters);
  r.clock_ghz = sum_c / sum_r * 0.1;  // s_memrealtime ticks at 100 MHz
  r.placement_ok = ok;
  r.worst_wave_cycles = worst / (w * insts_per_iter * iters);
  return r;
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  printf("# valu_rate: %s, %d CUs, clock %d MHz (max)\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1000);
  printf("# cycles per wave64 instruction per SIMD = (s_memtime delta of a wave) / (w waves per SIMD x instructions per wave); mean over waves [slowest wave]\n");
  printf("# grid 1 = one CU busy (idle chip); grid 256 = every CU busy.  clk = shader clock under the load, GHz (s_memtime / s_memrealtime)\n");
  printf("# placement '!' = the waves were NOT spread w per SIMD, one block per CU (figure not comparable)\n");
  const int max_blocks = 256;
  Stamp* dout;
  CHECK(hipMalloc(&dout, sizeof(Stamp) * 16 * max_blocks));
  std::vector<Stamp> h(16 * max_blocks);
  struct Entry { const char* name; kern_t k; double per_iter; };
  const double per = UNROLL * 8.0;
  std::vector<Entry> entries = {
      {"v_fma_f32", k_fma_f32, per}, {"v_fma_f32 (one dependent chain)", k_fma_f32_dependent, per}, {"v_mul_f32", k_mul_f32, per},
      {"v_max_f32", k_max_f32, per}, {"v_med3_f32", k_med3_f32, per}, {"v_fract_f32", k_fract_f32, per}, {"v_floor_f32", k_floor_f32, per},
      {"v_cvt_u32_f32", k_cvt_u32_f32, per}, {"v_cvt_f32_u32", k_cvt_f32_u32, per},
      {"v_add_u32", k_add_u32, per}, {"v_lshlrev_b32", k_lshlrev_b32, per}, {"v_and_b32", k_and_b32, per}, {"v_xor_b32", k_xor_b32, per},
      {"v_mov_b32", k_mov_b32, per}, {"v_cndmask_b32", k_cndmask_b32, per}, {"v_bitop3_b32", k_bitop3_b32, per},
      {"v_and_or_b32", k_and_or_b32, per}, {"v_lshl_add_u32", k_lshl_add_u32, per}, {"v_mad_u32_u24", k_mad_u32_u24, per},
      {"v_mul_lo_u32", k_mul_lo_u32, per}, {"v_mad_u64_u32", k_mad_u64_u32, per},
      {"v_pk_fma_f16", k_pk_fma_f16, per}, {"v_pk_mul_f16", k_pk_mul_f16, per}, {"v_pk_add_f16", k_pk_add_f16, per},
      {"v_pk_max_f16", k_pk_max_f16, per}, {"v_cvt_pk_f16_f32", k_cvt_pk_f16_f32, per}, {"v_cvt_pkrtz_f16_f32", k_cvt_pkrtz_f16_f32, per},
      {"v_fma_mix_f32", k_fma_mix_f32, per}, {"v_fma_mixlo_f16", k_fma_mixlo_f16, per},
      {"v_permlane32_swap", k_permlane32_swap, per},
      {"v_exp_f32", k_exp_f32, per}, {"v_rcp_f32", k_rcp_f32, per},
      {"v_pk_fma_f32", k_pk_fma_f32, per}, {"v_pk_mul_f32", k_pk_mul_f32, per}, {"v_fma_f64", k_fma_f64, per}, {"v_add_f64", k_add_f64, per},
      {"v_cndmask_b32_e64 (SGPR-pair mask)", k_cndmask64, per},
      {"v_cndmask_b32 vcc (no dst dependence)", k_cndmask_nodep, per},
      {"v_cmp_gt_f32 -> vcc", k_cmp_gt, per},
      {"v_cmp_gt_f32_e64 -> SGPR pair", k_cmp_gt64, per},
      {"v_cmp_gt_f32 + v_cndmask_b32 (pair = 2 instructions)", k_cmp_cnd, per},
      {"v_add_f32", k_add_f32, per},
      {"v_sub_f32", k_sub_f32, per},
      {"v_min_f32", k_min_f32, per},
      {"v_max3_f32", k_max3_f32, per},
      {"v_ldexp_f32", k_ldexp, per},
      {"v_sub_u32", k_sub_u32, per},
      {"v_or_b32", k_or_b32, per},
      {"v_min_u32", k_min_u32, per},
      {"v_max_u32", k_max_u32, per},
      {"v_add3_u32", k_add3, per},
      {"v_xad_u32", k_xad, per},
      {"v_lshl_or_b32", k_lshl_or, per},
      {"v_add_lshl_u32", k_add_lshl, per},
      {"v_lshrrev_b32", k_lshr, per},
      {"v_ashrrev_i32", k_ashr, per},
      {"v_bfe_u32", k_bfe, per},
      {"v_bfi_b32", k_bfi, per},
      {"v_perm_b32", k_perm, per},
      {"v_alignbit_b32", k_alignbit, per},
      {"v_ffbl_b32", k_ffbl, per},
      {"v_bcnt_u32_b32", k_bcnt, per},
      {"v_mbcnt_lo_u32_b32", k_mbcnt, per},
      {"v_mul_u32_u24", k_mul_u24, per},
      {"v_mul_hi_u32_u24", k_mulhi24, per},
      {"v_cvt_i32_f32", k_cvt_i32, per},
      {"v_pk_add_u16", k_pk_add_u16, per},
      {"v_pk_max_i16", k_pk_max_i16, per},
      {"v_pk_ashrrev_i16", k_pk_ashr16, per},
      {"v_pk_mul_lo_u16", k_pk_mullo16, per},
      {"v_pk_mad_u16", k_pk_mad16, per},
      {"v_max_f16", k_max_f16, per},
      {"v_fma_f16", k_fma_f16, per},
      {"v_dot2_f32_f16", k_dot2, per},
      {"v_dot2c_f32_f16", k_dot2c, per},
      {"v_cvt_f16_f32", k_cvt_f16_f32, per},
      {"v_cvt_f32_f16", k_cvt_f32_f16, per},
      {"v_accvgpr_write_b32", k_accw, per},
      {"v_accvgpr_read_b32", k_accr, per},
      {"v_mov_b32_dpp quad_perm", k_mov_dpp, per},
      {"v_add_f32_dpp row_shr", k_add_dpp, per},
      {"v_add_u32_sdwa", k_add_sdwa, per},
      {"s_nop 0", k_s_nop, per},
      {"v_mfma_f32_32x32x16_f16 alone (per MFMA)", k_mfma_fill<0>, 8.0},
      {"v_mfma + 2 VALU (per MFMA)", k_mfma_fill<2>, 8.0}, {"v_mfma + 4 VALU (per MFMA)", k_mfma_fill<4>, 8.0},
      {"v_mfma + 6 VALU (per MFMA)", k_mfma_fill<6>, 8.0}, {"v_mfma + 8 VALU (per MFMA)", k_mfma_fill<8>, 8.0},
      {"v_mfma + 12 VALU (per MFMA)", k_mfma_fill<12>, 8.0}, {"v_mfma + 16 VALU (per MFMA)", k_mfma_fill<16>, 8.0},
      {"v_mfma + 24 VALU (per MFMA)", k_mfma_fill<24>, 8.0},
  };
  const int only_grid = argc > 1 ? atoi(argv[1]) : 0;
  for (int blocks : {1, 256}) {
    if (only_grid && blocks != only_grid) continue;
    printf("\n## grid %d block%s\n", blocks, blocks > 1 ? "s (one per CU)" : " (one CU)");
    printf("%-44s %22s %22s %22s %22s\n", "instruction", "1 wave/SIMD", "2 waves/SIMD", "3 waves/SIMD", "4 waves/SIMD");
    for (auto& e : entries) {
      printf("%-44s", e.name);
      for (int w = 1; w <= 4; w++) {
        const bool mfma = e.per_iter == 8.0;
        const int iters = mfma ? 2048 / w : 4096 / w;
        Result r = run(e.k, w, blocks, iters, e.per_iter, dout, h);
        printf("  %6.2f [%6.2f] clk %4.2f%s", r.cyc_per_inst_simd, r.worst_wave_cycles, r.clock_ghz, r.placement_ok ? " " : "!");
      }
      printf("\n");
      fflush(stdout);
    }
  }
  CHECK(hipFree(dout));
  return 0;
}
