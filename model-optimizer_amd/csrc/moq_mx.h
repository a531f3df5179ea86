// moq_mx.h -- the MX element formats: rounding onto an ExMy / fixed-point grid, block scales (E8M0 and element-format
// scales with an optional tensor-wide amax), quantize-dequantize of one element.  Shared by the streaming MX kernels
// (moq_formats.hip) and the GPTQ column sweep (moq_sparse.hip).
#pragma once

#include "moq_common.h"

namespace moq {

// ================================================================================================
// MX formats (a8)
// ================================================================================================
// Small ExMy / fixed-point element formats described by (mantissa bits M, exponent of the smallest
// normal EMIN, max value, tie rule).  Rounding a non-negative finite value onto the grid:
//   normal range   : round the fp32 mantissa to M bits (RNE, or half-up for E3M0) with integer ops;
//   below 2^EMIN   : fixed quantum 2^(EMIN-M)  (the format's subnormals / its uniform low segment);
//   then saturate to the format maximum.
// This reproduces the reference's value/bound tables (tensor_quant_mx.h:42-71): every bound is the exact
// midpoint of its neighbours and "tie -> even table index" (h:104-122) is RNE on the code's mantissa LSB;
// E3M0 uses "tie -> away" (h:92-101).
struct MxFmt {
  int m;        // mantissa bits
  int emin;     // exponent of min normal
  float maxv;   // saturation value
  int half_up;  // 1: ties away from zero (E3M0)
  int kind;     // 0 table-like (inf/NaN saturate to max), 1 fp8 (NaN stays NaN), 2 int8
};
__host__ __device__ inline MxFmt mx_fmt(int t) {
  switch (t) {
    case MOQ_E2M1: return {1, 0, 6.0f, 0, 0};
    case MOQ_E1M2: return {2, 1, 3.5f, 0, 0};     // uniform 0.5 steps below 2, then 2..3.5
    case MOQ_E0M3: return {3, 3, 7.0f, 0, 0};     // integers 0..7 (everything below 8 is "subnormal")
    case MOQ_E3M0: return {0, -2, 16.0f, 1, 0};   // powers of two 0.25..16
    case MOQ_E3M2: return {2, -2, 28.0f, 0, 0};
    case MOQ_E2M3: return {3, 0, 7.5f, 0, 0};
    case MOQ_E4M3: return {3, -6, 448.0f, 0, 1};
    case MOQ_E5M2: return {2, -14, 57344.0f, 0, 1};
    case MOQ_INT8: return {0, 0, 127.0f, 0, 2};
    default: return {0, 0, 0.0f, 0, -1};
  }
}
__device__ __forceinline__ float mx_round_abs(float a, const MxFmt f) {
  // a >= 0 (may be inf / NaN).  Both segments are evaluated and one is selected: no data-dependent branch (a divergent
  // branch per element costs more than the dozen VALU instructions of the segment not taken).
  if (f.kind == 2) {  // convert_int8_saturating (h:74-84); the format is workgroup-uniform
    float r = __builtin_rintf(a);
    return r > 127.0f ? 127.0f : r;
  }
  const float min_normal = __builtin_ldexpf(1.0f, f.emin);
  // normal range: mantissa rounded with integer ops.  inf stays inf here (and saturates below); a carry out of the
  // largest finite exponent gives inf, which saturates too
  const int shift = 23 - f.m;
  uint32_t u = __float_as_uint(a);
  const uint32_t half = 1u << (shift - 1);
  u += f.half_up ? half : (half - 1u + ((u >> shift) & 1u));
  u &= ~((1u << shift) - 1u);
  const float qn = __uint_as_float(u);
  // below 2^emin: fixed quantum
  const float inv_quantum = __builtin_ldexpf(1.0f, f.m - f.emin), quantum = __builtin_ldexpf(1.0f, f.emin - f.m);
  const float t = a * inv_quantum;  // exact (power of two)
  const float qs = (f.half_up ? __builtin_floorf(t + 0.5f) : __builtin_rintf(t)) * quantum;
  float q = a >= min_normal ? qn : qs;
  q = q > f.maxv ? f.maxv : q;
  return a != a ? (f.kind == 1 ? a : f.maxv) : q;  // NaN: stays NaN for the fp8 kinds, saturates for the table kinds
}
// compute_scale_e8m0_NV (tensor_quant_mx.cu:105-137): unscale = 2^ceil(log2(amax / fmt_max))
__device__ __forceinline__ void mx_scale_e8m0(float amax, float fmt_max, float& scale, float& unscale) {
  const bool bad = amax == 0.0f || amax != amax || __float_as_uint(amax) == 0x7F800000u;  // cu:143-145: scale 1
  const float ratio = amax / fmt_max;
  const uint32_t u = __float_as_uint(ratio), ef = (u >> 23) & 0xFFu, mf = u & 0x7FFFFFu;
  const int ue = (mf > 0 && ef != 0xFE && !(ef == 0 && mf <= 0x400000u)) ? (int)ef - 126 : (int)ef - 127;
  scale = bad ? 1.0f : __builtin_ldexpf(1.0f, -ue);
  unscale = bad ? 1.0f : __builtin_ldexpf(1.0f, ue);
}
__device__ __forceinline__ float mx_qdq(float x, float scale, float unscale, const MxFmt f) {
  // quantize() (cu:36-55); the reference leaves `sign` uninitialised for 0 / NaN inputs -- taken as 0
  const float sign = x < 0.0f ? -1.0f : (x > 0.0f ? 1.0f : 0.0f);
  return sign * (mx_round_abs(__builtin_fabsf(x) * scale, f) * unscale);
}
// |x| clamped to FLT_MAX before the block max (compute_max_warp/block, cu:185-226); NaN is dropped by fmaxf
__device__ __forceinline__ float mx_abs_clamped(float x) {
  float a = __builtin_fabsf(x);
  return a > 3.402823466e+38f ? 3.402823466e+38f : a;
}

// MXFP4 (E2M1 elements, E8M0 block scales) on the chip's own converters.  gfx950 has scaled FP4 conversions:
//   v_cvt_scalef32_pk_fp4_f32  : two f32, divided by the scale 2^k, to two E2M1 nibbles (round to nearest even, saturating)
//   v_cvt_scalef32_pk_f32_fp4  : two nibbles back to f32, multiplied by 2^k
// Together they ARE the reference's quantize-dequantize of an element, sign(x) * (round_E2M1(|x| * 2^-k) * 2^k): checked for every
// bf16 pattern x every block exponent k in [-126, 126] (tools/exp/fp4_probe.hip on the MI355X: 0 differences), with two
// exceptions the caller handles -- a NaN (the reference's uninitialised sign makes it 0; a block that holds one is found by
// its abs-max pattern and takes the general path) and the input -0.0 (reference: +0, since its sign is "0"; `x + 0.0f` makes
// it +0 before the conversion and changes nothing else).  The element rounding by integer ops (mx_round_abs) costs ~20 vector
// instructions per element -- the MX kernels were bound by vector issue, not by HBM (0.755 of 8 TB/s); this path costs ~5.
// v: the V elements of a packet (already rounded to the storage dtype); amax_bits: abs-max PATTERN of the packet's block.
// Returns false when the block must take the general path (NaN inside; block exponent at the edge of the fp32 range).
typedef float mx_f2 __attribute__((ext_vector_type(2)));
template <int V>
__device__ __forceinline__ bool mx_e2m1_hw(float (&v)[8], uint32_t amax_bits) {
  if (amax_bits > 0x7F800000u) return false;  // every NaN pattern sorts above +inf
  // |x| clamped to FLT_MAX before the block max (compute_max_warp, cu:185-226): an inf counts as FLT_MAX
  const float am = __uint_as_float(amax_bits < 0x7F7FFFFFu ? amax_bits : 0x7F7FFFFFu);
  float sc, un;
  mx_scale_e8m0(am, 6.0f, sc, un);
  const uint32_t field = __float_as_uint(un) >> 23;  // un = 2^k: k + 127
  if (field < 1u || field > 253u) return false;
  uint32_t q = 0u;
  q = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(q, v[0] + 0.0f, v[1] + 0.0f, un, 0);
  q = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(q, v[2] + 0.0f, v[3] + 0.0f, un, 1);
  if constexpr (V == 8) {
    q = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(q, v[4] + 0.0f, v[5] + 0.0f, un, 2);
    q = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(q, v[6] + 0.0f, v[7] + 0.0f, un, 3);
  }
  mx_f2 r = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(q, un, 0);
  v[0] = r.x; v[1] = r.y;
  r = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(q, un, 1);
  v[2] = r.x; v[3] = r.y;
  if constexpr (V == 8) {
    r = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(q, un, 2);
    v[4] = r.x; v[5] = r.y;
    r = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(q, un, 3);
    v[6] = r.x; v[7] = r.y;
  }
  return true;
}

// compute_scale / compute_scale_with_global (tensor_quant_mx.cu:139-183) for block-scale formats other than E8M0
// (NVFP4-style: E2M1 elements, E4M3 block scales, optional tensor-wide amax).  The reference mixes float and double
// steps; they are kept one by one: float divisions, the product and the reciprocal in double, results narrowed to
// float when the tuple<float, float> is formed.
__device__ __forceinline__ bool mx_bad_amax(float v) { return v == 0.0f || v != v || __float_as_uint(v) == 0x7F800000u; }
__device__ __forceinline__ void mx_scale_general(float amax, float emax, const MxFmt sf, const float* global,
                                                 float& scale, float& unscale) {
  scale = 1.0f;
  unscale = 1.0f;
  if (global != nullptr) {
    const float g = *global;
    if (mx_bad_amax(amax) || mx_bad_amax(g)) return;
    const float local_unscale = amax / emax;
    const double two_level = (double)sf.maxv * (double)(emax / g);
    const float arg = (float)((double)local_unscale * two_level);
    const double q = (double)mx_round_abs(arg, sf) / two_level;
    scale = (float)(1.0 / q);
    unscale = (float)q;
  } else {
    if (mx_bad_amax(amax)) return;
    const double s = (double)(emax / amax);
    const double inv = (double)mx_round_abs((float)(1.0 / s), sf);
    scale = (float)(1.0 / inv);
    unscale = (float)inv;
  }
}

}  // namespace moq
