// Register-resident recurrent mat-vec machinery shared by the persistent kernels (attention cluster, LSTM):
// y[n] = sum_k x[k] W[k][n] as v_mfma_f32_16x16x32_bf16 with the bf16 weight tiles held in registers for the whole
// launch and the fp32 input vector split exactly into three bf16 rows of the (otherwise empty) 16-row A operand.
#pragma once
#include "common.h"

namespace {

// Row stride pad (bf16) of the split A-operand images in LDS.  A ds_read_b128 is served in four 16-lane groups, each holding the
// lanes of rows 0, 1, 2, 3 at one k quarter plus row-3 lanes of another quarter; with a row stride that is a multiple of 256 bytes
// rows 0..3 of a quarter share their four banks (4-way conflict on every A fetch: SQ_LDS_BANK_CONFLICT was 30% of
// lstm_bwd_mfma_k's cycles).  64 bytes of pad put row r, quarter q at byte (64 r + 16 q) mod 256: sixteen distinct slots.
constexpr int APAD = 32;
// the same for a row of `tiles` 32-element K tiles with a run-time count: an odd tile count makes the stride 64 or 192 mod 256 bytes
__host__ __device__ constexpr int a_stride(int tiles) { return (tiles | 1) * 32; }

// exact 3-way bf16 split of an fp32 value into rows 0..2 of the MFMA A-operand staging array xs[4][XS] (row 3 = 0)
__device__ __forceinline__ void xs_put(uint16_t* xs, int XS, int i, float v) {
  const uint16_t h = f2bf(v); const float r1 = v - bf2f(h);
  const uint16_t m = f2bf(r1); const float r2 = r1 - bf2f(m);
  xs[i] = h; xs[XS + i] = m; xs[2 * XS + i] = f2bf(r2);
}

// MFMA with the B operand pinned to the accumulation-register half of the unified register file: the resident
// weight slice must never compete with (and be spilled by) the working VGPRs of the other phases.
typedef __attribute__((ext_vector_type(4))) int i32x4_t;
// Hazard cover of the inline-asm MFMAs, measured on gfx950 (scratch/mfma_nops.hip, scratch/mfma_test.hip): s_nop N
// costs N+1 issue slots of 4 cycles; v_mfma_f32_16x16x32_bf16 is a 4-pass instruction (16 cycles of pipe time) whose
// result needs 7 wait states before a VALU read; an operand written by a VALU instruction needs 2 before the MFMA.
#define SATT_MFMA "v_mfma_f32_16x16x32_bf16 "
#define SATT_PRE "s_nop 2\n\t"
#define SATT_POST "s_nop 7\n\ts_nop 0"
// LAST = false ("chained"): the trailing cover is left out.  Legal only when the next instructions that touch the block's
// accumulators are the MFMAs of the next block of the same chain (SrcC = vDst: interlocked by the hardware) and nothing but LDS
// reads of the next A operands, waits and scalar work sits in between - tools/mfma_hazard_check.py verifies exactly that on the
// generated ISA.  The K loops of the recurrent mat-vecs are such chains: 36 cycles of cover per block that only the last one needs.
// FIRST = false: the leading cover (VALU write -> MFMA read, 2 wait states) is left out as well: legal when the block's operands
// come straight from LDS / the resident registers (checked by the same script: no VALU write of an operand within 2 wait states).
// Measured A/B on one box: it pays only in lstm_bwd_mfma_k (192 -> 187 us); the forward kernels and the attention kernels run 1 - 4 %
// SLOWER without the two idle slots in front of a block, so everything else keeps them.
#define SATT_PRE_C "; chained\n\t"
#define SATT_ASM(BODY, ...)                                                                          \
  do {                                                                                               \
    if constexpr (LAST && FIRST) asm volatile(SATT_PRE BODY SATT_POST __VA_ARGS__);                  \
    else if constexpr (LAST) asm volatile(SATT_PRE_C BODY SATT_POST __VA_ARGS__);                    \
    else if constexpr (FIRST) asm volatile(SATT_PRE BODY "; chained" __VA_ARGS__);                   \
    else asm volatile(SATT_PRE_C BODY "; chained" __VA_ARGS__);                                      \
  } while (0)
template <bool LAST = true, bool FIRST = true>
__device__ __forceinline__ void mfma_bf16_areg(f32x4_t& acc, const bf16x8_t& a, const i32x4_t& b_areg) {
  // The compiler cannot see the MFMA inside the asm statement, so the software-managed hazard "XDL write VGPR ->
  // VALU read" (11 wait states for this 8-pass MFMA) is covered inside the statement: whatever the compiler puts
  // next (a copy, the next MFMA of the chain, the final read) is safe.
  // The leading nops cover "VALU write VGPR -> MFMA read" for operands the compiler produced just before.
  SATT_ASM(SATT_MFMA "%0, %1, %2, %0\n\t", : "+v"(acc) : "v"(a), "a"(b_areg));
}

// the cover on its own, for a chain whose last block is not known at compile time: the accumulators pass THROUGH the statement,
// so no read of them can be scheduled in front of it
__device__ __forceinline__ void mfma_cover(f32x4_t& c0, f32x4_t& c1) { asm volatile(SATT_POST : "+v"(c0), "+v"(c1)); }
__device__ __forceinline__ void mfma_cover(f32x4_t& c0) { asm volatile(SATT_POST : "+v"(c0)); }

// asm MFMA with the B operand in ordinary VGPRs (tiles of the slice that live in LDS); same hazard cover as above
template <bool LAST = true, bool FIRST = true>
__device__ __forceinline__ void mfma_bf16_vreg(f32x4_t& acc, const bf16x8_t& a, const i32x4_t& b) {
  SATT_ASM(SATT_MFMA "%0, %1, %2, %0\n\t", : "+v"(acc) : "v"(a), "v"(b));
}
// Batched forms: several MFMAs inside ONE asm statement (dependent ones back to back: the hardware interlocks the
// SrcC = vDst chain), with the operand / result hazard cover paid once per block instead of once per instruction.
// two K tiles x two N tiles: acc0 += a0*b00 + a1*b10, acc1 += a0*b01 + a1*b11
#define SATT_DEF_BLOCK22(NAME, BC)                                                                                     \
  template <bool LAST = true, bool FIRST = true> __device__ __forceinline__ void NAME(f32x4_t& acc0, f32x4_t& acc1, const bf16x8_t& a0, const bf16x8_t& a1,          \
                                       const i32x4_t& b00, const i32x4_t& b01, const i32x4_t& b10, const i32x4_t& b11) { \
    SATT_ASM(SATT_MFMA "%0, %2, %4, %0\n\t" SATT_MFMA "%1, %2, %5, %1\n\t" SATT_MFMA "%0, %3, %6, %0\n\t"   \
                 SATT_MFMA "%1, %3, %7, %1\n\t",                                                               \
                 : "+v"(acc0), "+v"(acc1) : "v"(a0), "v"(a1), BC(b00), BC(b01), BC(b10), BC(b11));                      \
  }
// one K tile x two N tiles
#define SATT_DEF_BLOCK12(NAME, BC)                                                                                     \
  template <bool LAST = true, bool FIRST = true> __device__ __forceinline__ void NAME(f32x4_t& acc0, f32x4_t& acc1, const bf16x8_t& a0, const i32x4_t& b00,           \
                                       const i32x4_t& b01) {                                                           \
    SATT_ASM(SATT_MFMA "%0, %2, %3, %0\n\t" SATT_MFMA "%1, %2, %4, %1\n\t",                        \
                 : "+v"(acc0), "+v"(acc1) : "v"(a0), BC(b00), BC(b01));                                                 \
  }
// two K tiles x one N tile
#define SATT_DEF_BLOCK21(NAME, BC)                                                                                     \
  template <bool LAST = true, bool FIRST = true> __device__ __forceinline__ void NAME(f32x4_t& acc0, const bf16x8_t& a0, const bf16x8_t& a1, const i32x4_t& b0,       \
                                       const i32x4_t& b1) {                                                            \
    SATT_ASM(SATT_MFMA "%0, %1, %3, %0\n\t" SATT_MFMA "%0, %2, %4, %0\n\t",                        \
                 : "+v"(acc0) : "v"(a0), "v"(a1), BC(b0), BC(b1));                                                      \
  }
// one K tile (shared A) x four N tiles: four independent accumulators
#define SATT_DEF_BLOCK14(NAME, BC)                                                                                     \
  template <bool LAST = true, bool FIRST = true> __device__ __forceinline__ void NAME(f32x4_t& c0, f32x4_t& c1, f32x4_t& c2, f32x4_t& c3, const bf16x8_t& a0,         \
                                       const i32x4_t& b0, const i32x4_t& b1, const i32x4_t& b2, const i32x4_t& b3) {   \
    SATT_ASM(SATT_MFMA "%0, %4, %5, %0\n\t" SATT_MFMA "%1, %4, %6, %1\n\t" SATT_MFMA "%2, %4, %7, %2\n\t"   \
                 SATT_MFMA "%3, %4, %8, %3\n\t",                                                               \
                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a0), BC(b0), BC(b1), BC(b2), BC(b3));                   \
  }
// same with zero accumulators (SrcC = 0): the four results are fresh registers, nothing to initialise
#define SATT_DEF_BLOCK14Z(NAME, BC)                                                                                    \
  template <bool LAST = true, bool FIRST = true> __device__ __forceinline__ void NAME(f32x4_t& c0, f32x4_t& c1, f32x4_t& c2, f32x4_t& c3, const bf16x8_t& a0,         \
                                       const i32x4_t& b0, const i32x4_t& b1, const i32x4_t& b2, const i32x4_t& b3) {   \
    SATT_ASM(SATT_MFMA "%0, %4, %5, 0\n\t" SATT_MFMA "%1, %4, %6, 0\n\t" SATT_MFMA "%2, %4, %7, 0\n\t"    \
                 SATT_MFMA "%3, %4, %8, 0\n\t",                                                                \
                 : "=&v"(c0), "=&v"(c1), "=&v"(c2), "=&v"(c3) : "v"(a0), BC(b0), BC(b1), BC(b2), BC(b3));               \
  }
#define SATT_DEF_BLOCK12Z(NAME, BC)                                                                                    \
  template <bool LAST = true, bool FIRST = true> __device__ __forceinline__ void NAME(f32x4_t& c0, f32x4_t& c1, const bf16x8_t& a0, const i32x4_t& b0,                \
                                       const i32x4_t& b1) {                                                            \
    SATT_ASM(SATT_MFMA "%0, %2, %3, 0\n\t" SATT_MFMA "%1, %2, %4, 0\n\t",                          \
                 : "=&v"(c0), "=&v"(c1) : "v"(a0), BC(b0), BC(b1));                                                      \
  }
#define SATT_BC_A(x) "a"(x)
#define SATT_BC_V(x) "v"(x)
SATT_DEF_BLOCK22(mfma22_a, SATT_BC_A)
SATT_DEF_BLOCK22(mfma22_v, SATT_BC_V)
SATT_DEF_BLOCK12(mfma12_a, SATT_BC_A)
SATT_DEF_BLOCK12(mfma12_v, SATT_BC_V)
SATT_DEF_BLOCK21(mfma21_a, SATT_BC_A)
SATT_DEF_BLOCK21(mfma21_v, SATT_BC_V)
SATT_DEF_BLOCK14(mfma14_a, SATT_BC_A)
SATT_DEF_BLOCK14(mfma14_v, SATT_BC_V)
SATT_DEF_BLOCK14Z(mfma14z_a, SATT_BC_A)
SATT_DEF_BLOCK14Z(mfma14z_v, SATT_BC_V)
SATT_DEF_BLOCK12Z(mfma12z_a, SATT_BC_A)
SATT_DEF_BLOCK12Z(mfma12z_v, SATT_BC_V)

// two K tiles (a0, a1) x four N tiles: c_j += a0*b_j + a1*b_(4+j); Z: the first pass starts from SrcC = 0
#define SATT_DEF_BLOCK24(NAME, OUTC, C0, C1, C2, C3, BC0, BC1, BC2, BC3, BC4, BC5, BC6, BC7)                            \
  template <bool LAST = true, bool FIRST = true> __device__ __forceinline__ void NAME(f32x4_t& c0, f32x4_t& c1, f32x4_t& c2, f32x4_t& c3, const bf16x8_t& a0,         \
                                       const bf16x8_t& a1, const i32x4_t& b0, const i32x4_t& b1, const i32x4_t& b2,     \
                                       const i32x4_t& b3, const i32x4_t& b4, const i32x4_t& b5, const i32x4_t& b6,      \
                                       const i32x4_t& b7) {                                                             \
    SATT_ASM(SATT_MFMA "%0, %4, %6, " C0 "\n\t" SATT_MFMA "%1, %4, %7, " C1 "\n\t"                          \
                 SATT_MFMA "%2, %4, %8, " C2 "\n\t" SATT_MFMA "%3, %4, %9, " C3 "\n\t"                                   \
                 SATT_MFMA "%0, %5, %10, %0\n\t" SATT_MFMA "%1, %5, %11, %1\n\t"                                         \
                 SATT_MFMA "%2, %5, %12, %2\n\t" SATT_MFMA "%3, %5, %13, %3\n\t",                               \
                 : OUTC(c0), OUTC(c1), OUTC(c2), OUTC(c3)                                                               \
                 : "v"(a0), "v"(a1), BC0(b0), BC1(b1), BC2(b2), BC3(b3), BC4(b4), BC5(b5), BC6(b6), BC7(b7));           \
  }
#define SATT_OUT_ACC(x) "+v"(x)
#define SATT_OUT_NEW(x) "=&v"(x)
SATT_DEF_BLOCK24(mfma24_a, SATT_OUT_ACC, "%0", "%1", "%2", "%3", SATT_BC_A, SATT_BC_A, SATT_BC_A, SATT_BC_A, SATT_BC_A,
                 SATT_BC_A, SATT_BC_A, SATT_BC_A)
SATT_DEF_BLOCK24(mfma24z_a, SATT_OUT_NEW, "0", "0", "0", "0", SATT_BC_A, SATT_BC_A, SATT_BC_A, SATT_BC_A, SATT_BC_A,
                 SATT_BC_A, SATT_BC_A, SATT_BC_A)
// the first two B tiles in accumulation registers, the other six in ordinary VGPRs (tiles staged from LDS)
SATT_DEF_BLOCK24(mfma24_aav6, SATT_OUT_ACC, "%0", "%1", "%2", "%3", SATT_BC_A, SATT_BC_A, SATT_BC_V, SATT_BC_V, SATT_BC_V,
                 SATT_BC_V, SATT_BC_V, SATT_BC_V)
// four K tiles x one N tile on one accumulator (a dependent chain: the hardware interlocks SrcC = vDst)
#define SATT_DEF_BLOCK41(NAME, OUTC, C0)                                                                                \
  template <bool LAST = true, bool FIRST = true> __device__ __forceinline__ void NAME(f32x4_t& c0, const bf16x8_t& a0, const bf16x8_t& a1, const bf16x8_t& a2,        \
                                       const bf16x8_t& a3, const i32x4_t& b0, const i32x4_t& b1, const i32x4_t& b2,     \
                                       const i32x4_t& b3) {                                                             \
    SATT_ASM(SATT_MFMA "%0, %1, %5, " C0 "\n\t" SATT_MFMA "%0, %2, %6, %0\n\t" SATT_MFMA "%0, %3, %7, %0\n\t" \
                 SATT_MFMA "%0, %4, %8, %0\n\t",                                                               \
                 : OUTC(c0) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3));                  \
  }
SATT_DEF_BLOCK41(mfma41_v, SATT_OUT_ACC, "%0")
SATT_DEF_BLOCK41(mfma41z_v, SATT_OUT_NEW, "0")

// two independent two-step chains from zero: c0 = a00*b00 + a01*b01, c1 = a10*b10 + a11*b11 (interleaved issue)
template <bool LAST = true, bool FIRST = true>
__device__ __forceinline__ void mfma_2chains_z(f32x4_t& c0, f32x4_t& c1, const bf16x8_t& a00, const bf16x8_t& a01,
                                               const bf16x8_t& a10, const bf16x8_t& a11, const i32x4_t& b00,
                                               const i32x4_t& b01, const i32x4_t& b10, const i32x4_t& b11) {
  SATT_ASM(SATT_MFMA "%0, %2, %6, 0\n\t" SATT_MFMA "%1, %4, %8, 0\n\t" SATT_MFMA "%0, %3, %7, %0\n\t"
               SATT_MFMA "%1, %5, %9, %1\n\t",
               : "=&v"(c0), "=&v"(c1)
               : "v"(a00), "v"(a01), "v"(a10), "v"(a11), "v"(b00), "v"(b01), "v"(b10), "v"(b11));
}
// one two-step chain from zero
template <bool LAST = true, bool FIRST = true>
__device__ __forceinline__ void mfma_chain2_z(f32x4_t& c0, const bf16x8_t& a0, const bf16x8_t& a1, const i32x4_t& b0,
                                              const i32x4_t& b1) {
  SATT_ASM(SATT_MFMA "%0, %1, %3, 0\n\t" SATT_MFMA "%0, %2, %4, %0\n\t",
               : "=&v"(c0) : "v"(a0), "v"(a1), "v"(b0), "v"(b1));
}

// exact 3-way bf16 split of 8 consecutive fp32 values (two float4) into three B/A operand vectors
__device__ __forceinline__ void split8(const float (&v)[8], i32x4_t& hi, i32x4_t& mid, i32x4_t& lo) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    uint32_t h[2], m[2], l[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const float x = v[2 * q + e];
      const uint16_t hh = f2bf(x); const float r1 = x - bf2f(hh);
      const uint16_t mm = f2bf(r1); const float r2 = r1 - bf2f(mm);
      h[e] = hh; m[e] = mm; l[e] = f2bf(r2);
    }
    hi[q] = (int)(h[0] | (h[1] << 16)); mid[q] = (int)(m[0] | (m[1] << 16)); lo[q] = (int)(l[0] | (l[1] << 16));
  }
}

}  // namespace
