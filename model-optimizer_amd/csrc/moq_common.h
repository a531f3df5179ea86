// moq_common.h -- shared device/host helpers for libmoquant (gfx950 / CDNA4 only).
//
// Layout conventions used by every kernel in this directory:
//   * tensors are contiguous; a lane moves 16 bytes per memory instruction (8 x bf16/f16 or 4 x f32),
//     a wave64 therefore moves 1 KiB per instruction, fully coalesced;
//   * all arithmetic is fp32 (as in the reference kernels), storage dtype conversion is RNE through
//     the gfx950 hardware converters (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32);
//   * the library is built with -ffp-contract=off: the reference's results depend on separately rounded
//     multiply / round / divide, so no FMA contraction may happen behind our back.
#pragma once

#include <stdlib.h>

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/moquant.h"

// Tuning knobs (A/B switches between CORRECT variants, grid shapes) exist in the experiment build only
// (MOQ_EXPERIMENTS=1 build.sh -> libmoquant_exp.so, never the library that ships); there they are read on every call so
// that one process can compare settings on the same allocations.  The release library reads ONE environment variable,
// MOQ_TUNE_GEMM_GEO (its two error-GEMM loop structures are both release kernels with identical results).
static inline long long moq_tune(const char* name, long long dflt) {
#ifdef MOQ_EXPERIMENTS
  const char* e = getenv(name);
  return e ? atoll(e) : dflt;
#else
  (void)name;
  return dflt;
#endif
}

namespace moq {

constexpr int kWave = 64;
constexpr int kBlock = 256;          // threads per workgroup for the streaming kernels
constexpr float kAmaxEps = 1.0f / (1 << 24);  // "amax <= 2^-24" rule (tensor_quant.py:629, :50)

// ---------------------------------------------------------------- error plumbing (host)
void set_error(const char* fmt, ...);
int check_launch(const char* what);
bool lds_opt_in(const void* kernel, int bytes, const char* kernel_name);  // > 64 KiB of dynamic LDS; refusal -> check_launch

// grid for grid-stride streaming kernels: enough workgroups to fill 256 CUs x 8 blocks, never more than
// the work available (cdna_hip_programming.md guideline 11).
inline int stream_grid(int64_t work_items_per_block_iter, int64_t total_items) {
  int64_t need = (total_items + work_items_per_block_iter - 1) / work_items_per_block_iter;
  if (need < 1) need = 1;
  const int64_t cap = 256 * 8;
  return (int)(need < cap ? need : cap);
}

// ---------------------------------------------------------------- element traits
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

template <int DT>
struct Elem;
template <>
struct Elem<MOQ_F32> {
  using storage = float;
  static constexpr int kVec = 4;  // elements per 16-byte lane access
};
template <>
struct Elem<MOQ_F16> {
  using storage = uint16_t;
  static constexpr int kVec = 8;
};
template <>
struct Elem<MOQ_BF16> {
  using storage = uint16_t;
  static constexpr int kVec = 8;
};

// A 16-byte register packet.
struct alignas(16) Pack16 {
  uint32_t w[4];
};

__device__ __forceinline__ Pack16 load16(const void* p) {
  return *reinterpret_cast<const Pack16*>(p);
}
__device__ __forceinline__ void store16(void* p, const Pack16& v) {
  *reinterpret_cast<Pack16*>(p) = v;
}
// streaming (read-once / write-once) variants: non-temporal hint keeps L2/MALL for data that is reused.  The pointer
// is cast to the global address space: tensors always live in HBM, and pointers that reach a kernel through a
// segment table (multi-tensor launches) would otherwise compile to FLAT instructions (aperture check, both counters).
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const u32x4_t* gptr_c16;
typedef __attribute__((address_space(1))) u32x4_t* gptr_16;
__device__ __forceinline__ Pack16 load16_nt(const void* p) {
  Pack16 r;
  const u32x4_t v = __builtin_nontemporal_load((gptr_c16)(uintptr_t)p);
  r.w[0] = v.x; r.w[1] = v.y; r.w[2] = v.z; r.w[3] = v.w;
  return r;
}
__device__ __forceinline__ void store16_nt(void* p, const Pack16& r) {
  u32x4_t v = {r.w[0], r.w[1], r.w[2], r.w[3]};
  __builtin_nontemporal_store(v, (gptr_16)(uintptr_t)p);
}

// unpack a 16-byte packet into kVec floats
template <int DT>
__device__ __forceinline__ void unpack(const Pack16& p, float* f);
template <>
__device__ __forceinline__ void unpack<MOQ_F32>(const Pack16& p, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) f[i] = __uint_as_float(p.w[i]);
}
template <>
__device__ __forceinline__ void unpack<MOQ_BF16>(const Pack16& p, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __uint_as_float(p.w[i] << 16);
    f[2 * i + 1] = __uint_as_float(p.w[i] & 0xFFFF0000u);
  }
}
template <>
__device__ __forceinline__ void unpack<MOQ_F16>(const Pack16& p, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f16x2 h = *reinterpret_cast<const f16x2*>(&p.w[i]);
    f[2 * i] = (float)h.x;
    f[2 * i + 1] = (float)h.y;
  }
}

// pack kVec floats (RNE) into a 16-byte packet
template <int DT>
__device__ __forceinline__ Pack16 pack(const float* f);
template <>
__device__ __forceinline__ Pack16 pack<MOQ_F32>(const float* f) {
  Pack16 p;
#pragma unroll
  for (int i = 0; i < 4; ++i) p.w[i] = __float_as_uint(f[i]);
  return p;
}
template <>
__device__ __forceinline__ Pack16 pack<MOQ_BF16>(const float* f) {
  Pack16 p;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f32x2 v = {f[2 * i], f[2 * i + 1]};
    bf16x2 b = __builtin_convertvector(v, bf16x2);  // v_cvt_pk_bf16_f32 (RNE, NaN-preserving)
    p.w[i] = *reinterpret_cast<uint32_t*>(&b);
  }
  return p;
}
// fp32 -> f16 conversions take their operand through `fp32_value`: an empty asm that forces the value to exist as a
// ROUNDED fp32 number.  Without it the compiler folds a preceding fp32 multiply / add into v_fma_mixlo_f16 /
// v_fma_mixhi_f16, which rounds the exact product ONCE to f16, while the reference rounds twice (fp32 result,
// then .to(float16)) -- a rare one-ulp difference that -ffp-contract=off does not cover.
__device__ __forceinline__ float fp32_value(float v) {
  asm volatile("" : "+v"(v));
  return v;
}
template <>
__device__ __forceinline__ Pack16 pack<MOQ_F16>(const float* f) {
  Pack16 p;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f32x2 v = {fp32_value(f[2 * i]), fp32_value(f[2 * i + 1])};
    f16x2 h = __builtin_convertvector(v, f16x2);  // v_cvt_pk_f16_f32 (RNE)
    p.w[i] = *reinterpret_cast<uint32_t*>(&h);
  }
  return p;
}

// scalar element access (tails, strided kernels)
template <int DT>
__device__ __forceinline__ float load1(const void* base, int64_t i);
template <>
__device__ __forceinline__ float load1<MOQ_F32>(const void* b, int64_t i) {
  return reinterpret_cast<const float*>(b)[i];
}
template <>
__device__ __forceinline__ float load1<MOQ_BF16>(const void* b, int64_t i) {
  return __uint_as_float((uint32_t) reinterpret_cast<const uint16_t*>(b)[i] << 16);
}
template <>
__device__ __forceinline__ float load1<MOQ_F16>(const void* b, int64_t i) {
  return (float) reinterpret_cast<const _Float16*>(b)[i];
}
template <int DT>
__device__ __forceinline__ void store1(void* base, int64_t i, float v);
template <>
__device__ __forceinline__ void store1<MOQ_F32>(void* b, int64_t i, float v) {
  reinterpret_cast<float*>(b)[i] = v;
}
template <>
__device__ __forceinline__ void store1<MOQ_BF16>(void* b, int64_t i, float v) {
  reinterpret_cast<__bf16*>(b)[i] = (__bf16)v;
}
template <>
__device__ __forceinline__ void store1<MOQ_F16>(void* b, int64_t i, float v) {
  reinterpret_cast<_Float16*>(b)[i] = (_Float16)fp32_value(v);
}
// round an fp32 value to the storage dtype and back (the value a store+load would produce)
template <int DT>
__device__ __forceinline__ float round_to_dtype(float v);
template <>
__device__ __forceinline__ float round_to_dtype<MOQ_F32>(float v) { return v; }
template <>
__device__ __forceinline__ float round_to_dtype<MOQ_BF16>(float v) { return (float)(__bf16)v; }
template <>
__device__ __forceinline__ float round_to_dtype<MOQ_F16>(float v) { return (float)(_Float16)fp32_value(v); }

// ---------------------------------------------------------------- abs-max on bit patterns
// |x| as an unsigned pattern orders exactly like the float for non-NaN values and puts every NaN above
// +inf, so an unsigned max both reduces and propagates NaN (torch.max semantics) with integer ops only.
__device__ __forceinline__ uint32_t absbits(float v) { return __float_as_uint(v) & 0x7FFFFFFFu; }

// max of the two 16-bit abs patterns packed in a dword against a running packed max (v_pk_max_u16)
typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_absmax_u16(uint32_t acc, uint32_t w) {
  u16x2 a = *reinterpret_cast<u16x2*>(&acc);
  uint32_t m = w & 0x7FFF7FFFu;
  u16x2 b = *reinterpret_cast<u16x2*>(&m);
  u16x2 r = __builtin_elementwise_max(a, b);
  return *reinterpret_cast<uint32_t*>(&r);
}
// 16-bit storage abs pattern -> fp32 abs pattern (order preserving; NaN stays NaN)
template <int DT>
__device__ __forceinline__ uint32_t widen_abs16(uint32_t h);
template <>
__device__ __forceinline__ uint32_t widen_abs16<MOQ_BF16>(uint32_t h) { return h << 16; }
template <>
__device__ __forceinline__ uint32_t widen_abs16<MOQ_F16>(uint32_t h) {
  uint16_t hs = (uint16_t)h;
  return __float_as_uint((float)*reinterpret_cast<_Float16*>(&hs));
}

// abs-max pattern (fp32 pattern) of one 16-byte packet
template <int DT>
__device__ __forceinline__ uint32_t pack_absmax(const Pack16& p) {
  if constexpr (DT == MOQ_F32) {
    uint32_t m = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint32_t a = p.w[i] & 0x7FFFFFFFu;
      m = a > m ? a : m;
    }
    return m;
  } else {
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = pk_absmax_u16(acc, p.w[i]);
    uint32_t lo = acc & 0xFFFFu, hi = acc >> 16;
    return widen_abs16<DT>(lo > hi ? lo : hi);
  }
}

// True when a packet holds an element with 0 < |x| < 2^-100 (numerators for which the residual steps of a shared division
// could leave the normal range, see SharedDiv).  |x| - 1 as an unsigned pattern wraps a zero to the top, so one packed
// min over the packet and one compare decide it; f16 has no such values (its smallest subnormal is 2^-24).
template <int DT>
__device__ __forceinline__ bool pack_has_tiny_nonzero(const Pack16& p) {
  if constexpr (DT == MOQ_F16) {
    return false;
  } else if constexpr (DT == MOQ_F32) {
    uint32_t m = 0xFFFFFFFFu;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t a = (p.w[i] & 0x7FFFFFFFu) - 1u;
      m = a < m ? a : m;
    }
    return m < 0x0D800000u - 1u;
  } else {
    u16x2 m = {0xFFFF, 0xFFFF};
    const u16x2 one = {1, 1};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t a = p.w[i] & 0x7FFF7FFFu;
      m = __builtin_elementwise_min(m, *reinterpret_cast<const u16x2*>(&a) - one);
    }
    const uint16_t lo = m.x < m.y ? m.x : m.y;
    return lo < (uint16_t)(0x0D80u - 1u);  // bf16 pattern of 2^-100: exponent field 27
  }
}

// ---------------------------------------------------------------- cross-lane reductions (wave64)
// butterfly max over aligned sub-groups of `WIDTH` lanes (WIDTH power of two <= 64): every lane of the
// group ends with the group's max.  xor-shuffles of 1/2 lower to DPP quad_perm, 4/8 to DPP row ops,
// 16 to row_bcast/permlane, 32 to v_permlane32_swap / readlane on gfx950.
// DPP controls: quad_perm [1,0,3,2] / [2,3,0,1], row_half_mirror, row_mirror.  Once both lanes of a pair
// (then all 4 of a quad, all 8 of a half row) hold the same value, the mirrors act as xor-4 / xor-8, so a
// 16-lane max costs four full-rate VALU ops and never touches the LDS crossbar (ds_bpermute).
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false);
}
template <int WIDTH>
__device__ __forceinline__ uint32_t group_max_u32(uint32_t v) {
  uint32_t o;
  if constexpr (WIDTH >= 2) { o = dpp_u32<0xB1>(v); v = o > v ? o : v; }
  if constexpr (WIDTH >= 4) { o = dpp_u32<0x4E>(v); v = o > v ? o : v; }
  if constexpr (WIDTH >= 8) { o = dpp_u32<0x141>(v); v = o > v ? o : v; }
  if constexpr (WIDTH >= 16) { o = dpp_u32<0x140>(v); v = o > v ? o : v; }
#pragma unroll
  for (int off = 16; off < WIDTH; off <<= 1) {  // across 16-lane rows: LDS-crossbar shuffle
    o = (uint32_t)__shfl_xor((int)v, off, 64);
    v = o > v ? o : v;
  }
  return v;
}

// workgroup max (kBlock threads): returns the result in every lane of wave 0 (valid for threadIdx.x == 0)
__device__ __forceinline__ uint32_t block_max_u32(uint32_t v, uint32_t* smem /* kBlock/64 words */) {
  v = group_max_u32<64>(v);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) smem[wave] = v;
  __syncthreads();
  if (wave == 0) {
    uint32_t t = lane < (int)(blockDim.x >> 6) ? smem[lane] : 0u;
    v = group_max_u32<4>(t);  // kBlock = 256 -> 4 waves
  }
  return v;
}

// ---------------------------------------------------------------- quantize-dequantize cores
struct IntQ {
  float lo, hi;  // clamp bounds as floats (integers)
};
__host__ __device__ __forceinline__ IntQ make_intq(int num_bits, int is_unsigned, int narrow) {
  // bound = 2^(bits-1+unsigned) - 1 (tensor_quant_gpu.cu:38-41)
  float bound = (float)((1 << (num_bits - 1 + (is_unsigned ? 1 : 0))) - 1);
  IntQ q;
  q.hi = bound;
  // eager: unsigned -> 0, narrow -> -bound, else -bound-1 (tensor_quant.py:622-628)
  q.lo = is_unsigned ? 0.0f : (narrow ? -bound : -bound - 1.0f);
  return q;
}
// scale for INT-k: bound / amax, 0 flags "amax <= eps" (outputs are then 0, tensor_quant.py:629-642)
__device__ __forceinline__ float int_scale(float amax, float bound) {
  return amax <= kAmaxEps ? 0.0f : bound / amax;
}
// Division of many numerators by ONE denominator (all elements of a quantization group share `scale`).
// The refined reciprocal y = rcp(d) + one Newton step is computed once; each quotient then costs five
// full-rate FMAs: q0 = n*y, two residual corrections -- the same Markstein sequence the compiler emits
// for `/` (v_div_scale / v_rcp / v_fma x4 / v_div_fmas / v_div_fixup) minus the per-element rcp (quarter
// rate) and scaling ops.  The final fma(r1, y, q1) is correctly rounded when no intermediate over- or
// underflows, which holds for d in [2^-60, 2^60] and |n| <= 2^16 (results are never subnormal there);
// outside that window the plain IEEE division is used.  Bit-equality with `/` is checked on the GPU by
// tests/test_gpu_parity.py::test_shared_division_exact.
struct SharedDiv {
  float d, y;
  bool fast;
};
__device__ __forceinline__ SharedDiv make_shared_div(float d) {
  SharedDiv r;
  r.d = d;
  const float a = __builtin_fabsf(d);
  r.fast = (a >= 0x1p-60f) && (a <= 0x1p60f);
  const float y0 = __builtin_amdgcn_rcpf(d);
  const float e = __builtin_fmaf(-d, y0, 1.0f);
  r.y = __builtin_fmaf(y0, e, y0);
  return r;
}
// the exact-window form on its own (caller has tested s.fast once for a whole packet)
__device__ __forceinline__ float shared_div_in_window(float n, const SharedDiv& s) {
  const float q0 = n * s.y;
  const float r0 = __builtin_fmaf(-s.d, q0, n);
  const float q1 = __builtin_fmaf(r0, s.y, q0);
  const float r1 = __builtin_fmaf(-s.d, q1, n);
  const float q = __builtin_fmaf(r1, s.y, q1);
  return q0 == 0.0f ? q0 : q;  // (see shared_div)
}
__device__ __forceinline__ float shared_div(float n, const SharedDiv& s) {
  if (!s.fast) return n / s.d;
  const float q0 = n * s.y;
  const float r0 = __builtin_fmaf(-s.d, q0, n);
  const float q1 = __builtin_fmaf(r0, s.y, q0);
  const float r1 = __builtin_fmaf(-s.d, q1, n);
  const float q = __builtin_fmaf(r1, s.y, q1);
  // a zero numerator: the residual steps turn -0 into +0 ((+0) + (-0) = +0 in round-to-nearest), IEEE division
  // keeps the sign (-0 / d = -0 for d > 0) -- and so does the first product
  return q0 == 0.0f ? q0 : q;
}
// INT-k QDQ with a group-shared scale: same arithmetic as qdq_int, division through SharedDiv
__device__ __forceinline__ float qdq_int_shared(float x, float scale, const SharedDiv& sd, const IntQ& q) {
  float p = x * scale;
  float t = __builtin_rintf(p);
  t = t < q.lo ? q.lo : t;  // torch.clamp's max(t, lo): keeps t when equal, so a -0 survives an unsigned lower bound of +0
  t = __builtin_fminf(t, q.hi);
  t = (p != p) ? p : t;
  return scale == 0.0f ? t : shared_div(t, sd);
}
// The same result with five VALU ops less per element, for groups where it is provably safe: every element of the
// group is finite (the group abs-max is finite), the scale is an ordinary positive number (amax > 2^-24) inside
// SharedDiv's exact window.  Then x * scale is finite, so no NaN can appear (no re-injection), the clamp can be the
// single v_med3_f32, the divide needs no fallback, and the only thing the residual steps of the shared division
// lose -- the sign of a zero quotient -- is put back by copying the sign of the (integer) numerator (v_bfi_b32).
__device__ __forceinline__ bool qdq_fast_ok(uint32_t amax_bits, float scale, const SharedDiv& sd) {
  return sd.fast && scale != 0.0f && amax_bits < 0x7F800000u;
}
__device__ __forceinline__ float qdq_int_fast(float x, float scale, const SharedDiv& sd, const IntQ& q) {
  float t = __builtin_rintf(x * scale);
  t = __builtin_amdgcn_fmed3f(t, q.lo, q.hi);
  const float q0 = t * sd.y;
  const float r0 = __builtin_fmaf(-sd.d, q0, t);
  const float q1 = __builtin_fmaf(r0, sd.y, q0);
  const float r1 = __builtin_fmaf(-sd.d, q1, t);
  return __builtin_copysignf(__builtin_fmaf(r1, sd.y, q1), t);
}
__device__ __forceinline__ float qdq_int(float x, float scale, const IntQ& q) {
  // rint(x*scale), clamp, then IEEE divide by the same scale; scale == 0 encodes the tiny-amax case where
  // the reference multiplies by 0 and divides by 1.  torch.clamp propagates NaN while fmaxf/fminf drop
  // it, so a NaN product (NaN input, or inf * 0) is re-injected.
  float p = x * scale;
  float t = __builtin_rintf(p);
  t = t < q.lo ? q.lo : t;  // as in qdq_int_shared
  t = __builtin_fminf(t, q.hi);
  t = (p != p) ? p : t;
  return scale == 0.0f ? t : t / scale;
}

// FP8-E4M3 (OCP e4m3fn) round trip through the gfx950 converters.  The input is already clamped to
// +-448 on the amax path, so saturation behaviour of the converter is irrelevant there.
__device__ __forceinline__ void e4m3_roundtrip2(float a, float b, float& ra, float& rb) {
  int p = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  ra = __builtin_amdgcn_cvt_f32_fp8(p, 0);
  rb = __builtin_amdgcn_cvt_f32_fp8(p, 1);
}
struct Fp8Scale {
  float s, inv;
};
__device__ __forceinline__ Fp8Scale fp8_scale(float amax) {
  // scale = 448 / where(amax <= eps, 1, amax); inv = 1 / scale (tensor_quant.py:49-54).
  // NB: in the eager reference `448.0 / tensor` is Tensor.__rtruediv__ = tensor.reciprocal() * 448.0,
  // two roundings -- the normative arithmetic (the oracle is pinned on it), so it is restated as such.
  float safe = amax <= kAmaxEps ? 1.0f : amax;
  Fp8Scale r;
  r.s = (1.0f / safe) * 448.0f;
  r.inv = 1.0f / r.s;
  return r;
}

}  // namespace moq

// dtype dispatch for host entry points
#define MOQ_DISPATCH_DTYPE(dt, ...)                                   \
  switch (dt) {                                                       \
    case MOQ_F32: { constexpr int DT = MOQ_F32; __VA_ARGS__; } break; \
    case MOQ_F16: { constexpr int DT = MOQ_F16; __VA_ARGS__; } break; \
    case MOQ_BF16: { constexpr int DT = MOQ_BF16; __VA_ARGS__; } break; \
    default: moq::set_error("unknown dtype code %d", (int)(dt)); return MOQ_ERR_INVALID; \
  }
