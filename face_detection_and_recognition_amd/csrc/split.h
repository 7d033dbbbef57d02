// split.h — fp32 operands on the bf16 matrix cores: the exact three-way split ("bf16x6") shared by the *_x6 kernels.
//
// gfx950 has no TF32 / xf32 MFMA, and its fp32 MFMA runs at the fp32 VECTOR rate on the vector ALU (64 FLOP/clk/SIMD,
// 157 TFLOP/s; FINDINGS.md finding 18: it does not even overlap with VALU work), while v_mfma_f32_16x16x32_bf16 delivers
// 1024 FLOP/clk/SIMD on the matrix core beside the VALU.  An fp32 number has 24 significant bits = three bf16 pieces
// of 8 bits each, cut by ROUND-TO-NEAREST-EVEN conversions (v_cvt_pk_bf16_f32, one instruction per two values):
//     h = bf16(x),  m = bf16(x - h),  l = bf16((x - h) - m)        x == h + m + l  EXACTLY
// Both subtractions are exact in fp32 (Sterbenz-type: x - h is the rounding error of an 8-bit rounding of a 24-bit
// number, it has at most 16 significant bits; the second remainder at most 8, so the last conversion does not round), and
//     |m| <= 2^-8 |x| (1 + 2^-8),   |l| <= 2^-16 |x| (1 + 2^-8)      (half an ulp of an 8-bit significand, twice).
// A product a*b of two fp32 numbers is then the sum of nine bf16 x bf16 products (each exact in the MFMA's fp32
// accumulator); the kernels keep the six whose weight is >= 2^-16 of the product,
//     ah*bh + (ah*bm + am*bh) + (ah*bl + al*bh + am*bm),
// and drop am*bl + al*bm + al*bl, whose magnitude is <= (2 * 2^-24 + 2^-32)(1 + 2^-7) |a*b| < 2^-22.99 |a*b|: ONE fp32
// rounding unit, and -- because the remainders of a rounding carry either sign -- not a one-sided bias.  (Rounds 1-3 cut
// the pieces by TRUNCATION: |m| < 2^-7, |l| < 2^-15, dropped terms < 2^-21 |a*b| and always of the product's sign; the
// round-to-nearest cut costs the same number of VALU instructions, 9 per pair against 11.)  Accumulation is fp32.
// Values within 2^-8 of FLT_MAX would round to a bf16 infinity; activations and weights are nowhere near.
// Measured through the whole Mobile-FaceNet (tests/test_split_precision.py, fp64 as the truth): the embedding error of
// this scheme is not above the fp32 fmaf chain's; adversarial operands (all-ones mantissas, same-sign rows, K up to
// 1152) in tests/test_gpu_parity.py (test_split_gemms_adversarial_vs_fp64).  A two-piece split (three products) gives
// 3.5e-5 and is NOT used.  Cost: 6 MFMAs of ~18 cycles per 16x16x32 block (tools/lab/coexec_bf16_lab.hip) against 16
// of 32 cycles (16x16x4 f32) = 4.7x fewer matrix cycles, and each MFMA holds the SIMD's issue port for only ~8 of them:
// the VALU of the partner wave gets about two instructions in per MFMA.
// Weights are split once on the host (plan.py split3_bf16, the same roundings); activations in registers where they are
// produced.
#pragma once
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// two floats -> one dword holding bf16(a) in its low half, bf16(b) in its high half (round to nearest even: v_cvt_pk_bf16_f32)
__device__ __forceinline__ unsigned fp_pack_bf16_rn(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

// Two floats -> their three bf16 planes, each plane a dword (element 0 in the low half).
__device__ __forceinline__ void fp_split_pair(float v0, float v1, unsigned& h, unsigned& m, unsigned& l) {
  h = fp_pack_bf16_rn(v0, v1);
  const float r0 = v0 - __builtin_bit_cast(float, h << 16);
  const float r1 = v1 - __builtin_bit_cast(float, h & 0xffff0000u);
  m = fp_pack_bf16_rn(r0, r1);
  const float s0 = r0 - __builtin_bit_cast(float, m << 16);
  const float s1 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
  l = fp_pack_bf16_rn(s0, s1);      // exact: the second remainder has at most 8 significant bits
}

// One float -> its three bf16 pieces (bit patterns in the low 16 bits).
__device__ __forceinline__ void fp_split_one(float v, unsigned& h, unsigned& m, unsigned& l) {
  h = fp_pack_bf16_rn(v, v) & 0xffffu;
  const float r = v - __builtin_bit_cast(float, h << 16);
  m = fp_pack_bf16_rn(r, r) & 0xffffu;
  const float s = r - __builtin_bit_cast(float, m << 16);
  l = fp_pack_bf16_rn(s, s) & 0xffffu;
}

// Eight consecutive k of one row (two float4) -> the three 8 x bf16 MFMA fragments.
struct fp_frag3 {
  u32x4 h, m, l;
};
__device__ __forceinline__ fp_frag3 fp_split8(const f32x4 a, const f32x4 b) {
  unsigned h[4], m[4], l[4];
  fp_split_pair(a[0], a[1], h[0], m[0], l[0]);
  fp_split_pair(a[2], a[3], h[1], m[1], l[1]);
  fp_split_pair(b[0], b[1], h[2], m[2], l[2]);
  fp_split_pair(b[2], b[3], h[3], m[3], l[3]);
  fp_frag3 f;
  f.h = u32x4{h[0], h[1], h[2], h[3]};
  f.m = u32x4{m[0], m[1], m[2], m[3]};
  f.l = u32x4{l[0], l[1], l[2], l[3]};
  return f;
}

#define FP_BF(x) __builtin_bit_cast(bf16x8, (x))
// acc += A * B with both operands split: the six products, smallest first.  A / B follow the operand order of the MFMA
// (A: rows, B: columns); each is (h, m, l) of the same 16 x 32 / 32 x 16 fragment.
__device__ __forceinline__ f32x4 fp_mfma_x6(const u32x4 ah, const u32x4 am, const u32x4 al, const u32x4 bh, const u32x4 bm,
                                            const u32x4 bl, f32x4 acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FP_BF(am), FP_BF(bm), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FP_BF(al), FP_BF(bh), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FP_BF(ah), FP_BF(bl), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FP_BF(am), FP_BF(bh), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FP_BF(ah), FP_BF(bm), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FP_BF(ah), FP_BF(bh), acc, 0, 0, 0);
  return acc;
}

// Two independent accumulators interleaved: a 16x16x32 bf16 MFMA that accumulates into the previous MFMA's result issues
// every ~24 cycles instead of 16 (tools/lab/x6_lab.hip: 96 chained MFMAs took 2300 cycles), two alternating chains run
// at the issue rate.  FP_MFMA_ORDER() pins the source order (hipcc regroups MFMAs by accumulator).
#define FP_X6_STEP(A0, B0, A1, B1)                                                             \
  acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FP_BF(A0), FP_BF(B0), acc0, 0, 0, 0);         \
  FP_MFMA_ORDER();                                                                             \
  acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FP_BF(A1), FP_BF(B1), acc1, 0, 0, 0);         \
  FP_MFMA_ORDER();
// same A (h, m, l), two B operands
__device__ __forceinline__ void fp_mfma_x6_2b(const u32x4 ah, const u32x4 am, const u32x4 al, const fp_frag3& b0,
                                              const fp_frag3& b1, f32x4& acc0, f32x4& acc1) {
  FP_X6_STEP(am, b0.m, am, b1.m)
  FP_X6_STEP(al, b0.h, al, b1.h)
  FP_X6_STEP(ah, b0.l, ah, b1.l)
  FP_X6_STEP(am, b0.h, am, b1.h)
  FP_X6_STEP(ah, b0.m, ah, b1.m)
  FP_X6_STEP(ah, b0.h, ah, b1.h)
}
// two A operands, same B
__device__ __forceinline__ void fp_mfma_x6_2a(const fp_frag3& a0, const fp_frag3& a1, const u32x4 bh, const u32x4 bm,
                                              const u32x4 bl, f32x4& acc0, f32x4& acc1) {
  FP_X6_STEP(a0.m, bm, a1.m, bm)
  FP_X6_STEP(a0.l, bh, a1.l, bh)
  FP_X6_STEP(a0.h, bl, a1.h, bl)
  FP_X6_STEP(a0.m, bh, a1.m, bh)
  FP_X6_STEP(a0.h, bm, a1.h, bm)
  FP_X6_STEP(a0.h, bh, a1.h, bh)
}
// two independent products
__device__ __forceinline__ void fp_mfma_x6_2(const fp_frag3& a0, const fp_frag3& b0, const fp_frag3& a1, const fp_frag3& b1,
                                             f32x4& acc0, f32x4& acc1) {
  FP_X6_STEP(a0.m, b0.m, a1.m, b1.m)
  FP_X6_STEP(a0.l, b0.h, a1.l, b1.h)
  FP_X6_STEP(a0.h, b0.l, a1.h, b1.l)
  FP_X6_STEP(a0.m, b0.h, a1.m, b1.h)
  FP_X6_STEP(a0.h, b0.m, a1.h, b1.m)
  FP_X6_STEP(a0.h, b0.h, a1.h, b1.h)
}
