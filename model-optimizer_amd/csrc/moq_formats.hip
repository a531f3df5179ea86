// moq_formats.hip -- MX dynamic-block QDQ, histogram, 2:4 mask, real INT4 pack/unpack, export packer and
// the column-scale fold.  All HBM-bound streaming kernels with 16-byte lane accesses.
#include "moq_common.h"

namespace moq {

__device__ __forceinline__ bool al16(const void* p) {
  return (reinterpret_cast<uintptr_t>(p) & 15u) == 0;
}

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
  // a >= 0 (may be inf / NaN)
  if (f.kind == 2) {  // convert_int8_saturating (h:74-84)
    float r = __builtin_rintf(a);
    return r > 127.0f ? 127.0f : r;
  }
  if (a != a) return f.kind == 1 ? a : f.maxv;
  float q;
  const float min_normal = __builtin_ldexpf(1.0f, f.emin);
  if (a >= min_normal) {
    const int shift = 23 - f.m;
    uint32_t u = __float_as_uint(a);
    if (u >= 0x7F800000u) return f.maxv;  // inf saturates
    const uint32_t half = 1u << (shift - 1);
    u += f.half_up ? half : (half - 1u + ((u >> shift) & 1u));
    u &= ~((1u << shift) - 1u);
    q = __uint_as_float(u);
  } else {
    const float inv_quantum = __builtin_ldexpf(1.0f, f.m - f.emin), quantum = __builtin_ldexpf(1.0f, f.emin - f.m);
    const float t = a * inv_quantum;  // exact (power of two)
    q = (f.half_up ? __builtin_floorf(t + 0.5f) : __builtin_rintf(t)) * quantum;
  }
  return q > f.maxv ? f.maxv : q;
}
// compute_scale_e8m0_NV (tensor_quant_mx.cu:105-137): unscale = 2^ceil(log2(amax / fmt_max))
__device__ __forceinline__ void mx_scale_e8m0(float amax, float fmt_max, float& scale, float& unscale) {
  if (amax == 0.0f || amax != amax || __float_as_uint(amax) == 0x7F800000u) {  // cu:143-145
    scale = 1.0f;
    unscale = 1.0f;
    return;
  }
  const float ratio = amax / fmt_max;
  const uint32_t u = __float_as_uint(ratio), ef = (u >> 23) & 0xFFu, mf = u & 0x7FFFFFu;
  const int ue = (mf > 0 && ef != 0xFE && !(ef == 0 && mf <= 0x400000u)) ? (int)ef - 126 : (int)ef - 127;
  scale = __builtin_ldexpf(1.0f, -ue);
  unscale = __builtin_ldexpf(1.0f, ue);
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

// fast path: cols % block == 0, block % kVec == 0 -> an MX block is LPG adjacent lanes of one packet
template <int DT, int LPG>
__global__ __launch_bounds__(kBlock) void mx_kernel(const void* __restrict__ x, void* __restrict__ y,
                                                    int64_t n_packets, int fmt) {
  constexpr int V = Elem<DT>::kVec;
  const MxFmt f = mx_fmt(fmt);
  const char* xb = reinterpret_cast<const char*>(x);
  char* yb = reinterpret_cast<char*>(y);
  // whole waves iterate together so that the LPG-lane butterfly always has all its lanes
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  const int64_t n_round = (n_packets + 63) / 64 * 64;
  for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < n_round; p += stride) {
    const bool live = p < n_packets;
    float v[8];
    if (live) unpack<DT>(load16(xb + p * 16), v);
    else
      for (int i = 0; i < V; ++i) v[i] = 0.0f;
    float am = 0.0f;
#pragma unroll
    for (int i = 0; i < V; ++i) am = __builtin_fmaxf(am, mx_abs_clamped(v[i]));
    // non-negative floats order like their bit patterns
    am = __uint_as_float(group_max_u32<LPG>(__float_as_uint(am)));
    float sc, un;
    mx_scale_e8m0(am, f.maxv, sc, un);
#pragma unroll
    for (int i = 0; i < V; ++i) v[i] = mx_qdq(v[i], sc, un, f);
    if (live) store16(yb + p * 16, pack<DT>(v));
  }
}
// generic path: one thread per MX block, handles ragged last blocks (virtual zero padding) and any alignment
template <int DT>
__global__ void mx_generic_kernel(const void* __restrict__ x, void* __restrict__ y, int64_t rows,
                                  int64_t cols, int block, int fmt) {
  const MxFmt f = mx_fmt(fmt);
  const int64_t bpr = (cols + block - 1) / block;
  for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < rows * bpr;
       b += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = b / bpr, c0 = (b % bpr) * block;
    const int64_t c1 = c0 + block < cols ? c0 + block : cols;
    float am = 0.0f;
    for (int64_t c = c0; c < c1; ++c) am = __builtin_fmaxf(am, mx_abs_clamped(load1<DT>(x, r * cols + c)));
    float sc, un;
    mx_scale_e8m0(am, f.maxv, sc, un);
    for (int64_t c = c0; c < c1; ++c)
      store1<DT>(y, r * cols + c, mx_qdq(load1<DT>(x, r * cols + c), sc, un, f));
  }
}

// ================================================================================================
// histogram of |x| (a4)
// ================================================================================================
// LDS-privatised per workgroup: bins <= kHistMaxLdsBins live in LDS as u32 and are flushed once per
// workgroup with 64-bit global atomics; larger bin counts (after many growth steps) go straight to L2
// atomics.
constexpr int kHistMaxLdsBins = 16384;  // 64 KiB of the CU's 160 KiB LDS

__device__ __forceinline__ int hist_bin(float a, int bins, float max_edge, int skip_zeros) {
  // torch.histc: pos = (int)((v - min) * bins / (max - min)) in fp32, v == max -> last bin, outside -> skip
  if (!(a <= max_edge)) return -1;  // also drops NaN
  if (skip_zeros && a == 0.0f) return -1;
  int pos = (int)(a * (float)bins / max_edge);
  if (pos >= bins) pos = bins - 1;
  return pos;
}
template <int DT, bool LDS>
__global__ __launch_bounds__(kBlock) void hist_kernel(const void* __restrict__ x, int64_t n,
                                                      unsigned long long* __restrict__ counts, int bins,
                                                      float max_edge, int skip_zeros) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_hist[];
  constexpr int V = Elem<DT>::kVec;
  if (LDS) {
    for (int b = threadIdx.x; b < bins; b += kBlock) lds_hist[b] = 0;
    __syncthreads();
  }
  const bool fast = al16(x);
  const int64_t n_packets = (n + V - 1) / V;
  for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < n_packets;
       p += (int64_t)gridDim.x * kBlock) {
    float v[8];
    const int64_t e = p * V;
    if (fast && e + V <= n) {
      unpack<DT>(load16(reinterpret_cast<const char*>(x) + p * 16), v);
    } else {
      for (int i = 0; i < V; ++i) v[i] = e + i < n ? load1<DT>(x, e + i) : __uint_as_float(0x7FC00000u);
    }
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const int b = hist_bin(__builtin_fabsf(v[i]), bins, max_edge, skip_zeros);
      if (b >= 0) {
        if (LDS) atomicAdd(&lds_hist[b], 1u);
        else atomicAdd(&counts[b], 1ull);
      }
    }
  }
  if (LDS) {
    __syncthreads();
    for (int b = threadIdx.x; b < bins; b += kBlock) {
      const uint32_t c = lds_hist[b];
      if (c) atomicAdd(&counts[b], (unsigned long long)c);
    }
  }
}

// ================================================================================================
// 2:4 magnitude mask (a14)
// ================================================================================================
// reference pattern order (set(permutations(...)) under CPython 3.10, pinned by the golden test):
//   0:(0,1,0,1) 1:(1,1,0,0) 2:(0,1,1,0) 3:(1,0,1,0) 4:(1,0,0,1) 5:(0,0,1,1)
__device__ __forceinline__ uint32_t mask4(float a0, float a1, float a2, float a3) {
  // literal row-times-pattern dot products (torch.matmul(mat.abs(), patterns.t()), magnitude.py:79):
  // zero coefficients contribute a*0 so that inf/NaN poison a sum exactly like the matmul does.
  const float z0 = a0 * 0.0f, z1 = a1 * 0.0f, z2 = a2 * 0.0f, z3 = a3 * 0.0f;
  float s[6];
  s[0] = ((z0 + a1) + z2) + a3;
  s[1] = ((a0 + a1) + z2) + z3;
  s[2] = ((z0 + a1) + a2) + z3;
  s[3] = ((a0 + z1) + a2) + z3;
  s[4] = ((a0 + z1) + z2) + a3;
  s[5] = ((z0 + z1) + a2) + a3;
  int best = 0;
  float bv = s[0];
#pragma unroll
  for (int p = 1; p < 6; ++p) {
    const bool better = (bv == bv) && ((s[p] != s[p]) || s[p] > bv);  // argmax: NaN is max, first wins
    if (better) { best = p; bv = s[p]; }
  }
  // mask bytes for elements 0..3 packed little-endian
  const uint32_t tbl[6] = {0x01000100u, 0x00000101u, 0x00010100u, 0x00010001u, 0x01000001u, 0x01010000u};
  return tbl[best];
}
template <int DT>
__global__ __launch_bounds__(kBlock) void mask24_kernel(const void* __restrict__ w,
                                                        uint8_t* __restrict__ mask, int64_t n) {
  constexpr int V = Elem<DT>::kVec;  // 8 (two groups of 4) or 4 (one group)
  const bool fast = al16(w) && (reinterpret_cast<uintptr_t>(mask) & 7u) == 0;
  const int64_t n_packets = n / V;  // n % 4 == 0; a trailing half packet (bf16) is handled below
  for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < n_packets;
       p += (int64_t)gridDim.x * kBlock) {
    float v[8];
    if (fast) unpack<DT>(load16(reinterpret_cast<const char*>(w) + p * 16), v);
    else
      for (int i = 0; i < V; ++i) v[i] = load1<DT>(w, p * V + i);
#pragma unroll
    for (int i = 0; i < V; ++i) v[i] = __builtin_fabsf(v[i]);
    uint32_t m0 = mask4(v[0], v[1], v[2], v[3]);
    if constexpr (V == 8) {
      uint32_t m1 = mask4(v[4], v[5], v[6], v[7]);
      if (fast) {
        *reinterpret_cast<uint2*>(mask + p * 8) = make_uint2(m0, m1);
      } else {
        for (int i = 0; i < 4; ++i) { mask[p * 8 + i] = (m0 >> (8 * i)) & 1; mask[p * 8 + 4 + i] = (m1 >> (8 * i)) & 1; }
      }
    } else {
      if (fast) *reinterpret_cast<uint32_t*>(mask + p * 4) = m0;
      else
        for (int i = 0; i < 4; ++i) mask[p * 4 + i] = (m0 >> (8 * i)) & 1;
    }
  }
  // tail: n % V == 4 for 16-bit types
  if (blockIdx.x == 0 && threadIdx.x == 0 && (n % V) != 0) {
    const int64_t e = n_packets * V;
    const uint32_t m0 = mask4(__builtin_fabsf(load1<DT>(w, e)), __builtin_fabsf(load1<DT>(w, e + 1)),
                              __builtin_fabsf(load1<DT>(w, e + 2)), __builtin_fabsf(load1<DT>(w, e + 3)));
    for (int i = 0; i < 4; ++i) mask[e + i] = (m0 >> (8 * i)) & 1;
  }
}

// ================================================================================================
// real INT4 (a15) and the export packer
// ================================================================================================
template <int DT>
__global__ __launch_bounds__(kBlock) void int4_pack_kernel(const void* __restrict__ x,
                                                           const void* __restrict__ scales,
                                                           uint8_t* __restrict__ out, int64_t n, int g,
                                                           int rounding) {
  constexpr int V = Elem<DT>::kVec;
  const bool fast = al16(x) && (g % V) == 0 && (reinterpret_cast<uintptr_t>(out) & 3u) == 0 && (n % V) == 0;
  const int64_t n_packets = (n + V - 1) / V;
  for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < n_packets;
       p += (int64_t)gridDim.x * kBlock) {
    const int64_t e = p * V;
    float v[8];
    if (fast) unpack<DT>(load16(reinterpret_cast<const char*>(x) + p * 16), v);
    else
      for (int i = 0; i < V; ++i) v[i] = e + i < n ? load1<DT>(x, e + i) : 0.0f;
    uint32_t q[8];
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const float s = load1<DT>(scales, (e + i < n ? e + i : e) / g);
      float t = round_to_dtype<DT>(v[i] * s);  // arithmetic in the storage dtype, like the reference
      if (rounding == MOQ_ROUND_HALF_EVEN) {
        // qtensor/int4_tensor.py:72-76: round() (half-even), clamp [-8, 7], + 8
        float r = __builtin_rintf(t);
        r = __builtin_fminf(__builtin_fmaxf(r, -8.0f), 7.0f);
        q[i] = (uint32_t)(int)(r + 8.0f) & 0xFu;
      } else {
        // tensor_quant_gpu.cu:322-333: clamp first, roundf(v + 8) (half away), in the storage dtype
        t = __builtin_fminf(__builtin_fmaxf(t, -8.0f), 7.0f);
        t = round_to_dtype<DT>(t + 8.0f);
        q[i] = (uint32_t)(int)__builtin_roundf(t) & 0xFu;
      }
    }
    if (fast) {
      if constexpr (V == 8) {
        const uint32_t wv = ((q[0] << 4) | q[1]) | (((q[2] << 4) | q[3]) << 8) |
                            (((q[4] << 4) | q[5]) << 16) | (((q[6] << 4) | q[7]) << 24);
        *reinterpret_cast<uint32_t*>(out + p * 4) = wv;
      } else {
        const uint16_t hv = (uint16_t)(((q[0] << 4) | q[1]) | (((q[2] << 4) | q[3]) << 8));
        *reinterpret_cast<uint16_t*>(out + p * 2) = hv;
      }
    } else {
      for (int i = 0; i + 1 < V; i += 2)
        if (e + i + 1 < n) out[(e + i) / 2] = (uint8_t)((q[i] << 4) | q[i + 1]);
    }
  }
}
template <int DT>
__global__ __launch_bounds__(kBlock) void int4_unpack_kernel(const uint8_t* __restrict__ q,
                                                             const void* __restrict__ scales,
                                                             void* __restrict__ out, int64_t n_bytes,
                                                             int g) {
  // 4 bytes -> 8 elements per lane
  const int64_t n_words = (n_bytes + 3) / 4;
  const bool fast = (reinterpret_cast<uintptr_t>(q) & 3u) == 0 && (g % 8) == 0 && (n_bytes % 4) == 0 &&
                    al16(out) && DT != MOQ_F32;
  for (int64_t wi = (int64_t)blockIdx.x * kBlock + threadIdx.x; wi < n_words;
       wi += (int64_t)gridDim.x * kBlock) {
    float v[8];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int64_t bi = wi * 4 + b;
      const uint32_t byte = bi < n_bytes ? q[bi] : 0x88u;
      const float s = load1<DT>(scales, (bi < n_bytes ? 2 * bi : 0) / g);
      // tensor_quant_gpu.cu:275-281: (nibble - 8) / scale in the scale dtype
      v[2 * b] = (float)((int)(byte >> 4) - 8) / s;
      v[2 * b + 1] = (float)((int)(byte & 0xFu) - 8) / s;
    }
    if (fast) {
      if constexpr (DT != MOQ_F32) store16(reinterpret_cast<char*>(out) + wi * 16, pack<DT>(v));
    } else {
      for (int i = 0; i < 8; ++i)
        if (wi * 8 + i < 2 * n_bytes) store1<DT>(out, wi * 8 + i, v[i]);
    }
  }
}
// export packer: a lane owns 8 (4 for f32) adjacent columns of a row PAIR (2i, 2i+1): two 16-byte loads,
// one 8-byte (4-byte) store; rows of the pair are cols*elem bytes apart so both loads are fully coalesced.
template <int DT>
__global__ __launch_bounds__(kBlock) void int4_export_kernel(const void* __restrict__ w,
                                                             const float* __restrict__ wsf,
                                                             uint8_t* __restrict__ out, int64_t rows,
                                                             int64_t cols, int g) {
  constexpr int V = Elem<DT>::kVec;
  const int64_t spr = cols / g;
  const int64_t ppr = (cols + V - 1) / V;  // packets per row
  const int64_t n_items = (rows / 2) * ppr;
  const bool fast = al16(w) && (cols % V) == 0 && (g % V) == 0 && (reinterpret_cast<uintptr_t>(out) & 7u) == 0;
  for (int64_t it = (int64_t)blockIdx.x * kBlock + threadIdx.x; it < n_items;
       it += (int64_t)gridDim.x * kBlock) {
    const int64_t r2 = it / ppr, c0 = (it % ppr) * V;
    float a[8], b[8];
    if (fast) {
      unpack<DT>(load16(reinterpret_cast<const char*>(w) + ((2 * r2) * cols + c0) * (16 / V)), a);
      unpack<DT>(load16(reinterpret_cast<const char*>(w) + ((2 * r2 + 1) * cols + c0) * (16 / V)), b);
    } else {
      for (int i = 0; i < V; ++i) {
        a[i] = c0 + i < cols ? load1<DT>(w, (2 * r2) * cols + c0 + i) : 0.0f;
        b[i] = c0 + i < cols ? load1<DT>(w, (2 * r2 + 1) * cols + c0 + i) : 0.0f;
      }
    }
    uint32_t byte[8];
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const int64_t c = c0 + i < cols ? c0 + i : cols - 1;
      const float sa = wsf[(2 * r2) * spr + c / g], sb = wsf[(2 * r2 + 1) * spr + c / g];
      // quant_utils.py:800-805: (w / wsf).round().clamp(-8, 7) with fp32 division
      float qa = __builtin_rintf(a[i] / sa), qb = __builtin_rintf(b[i] / sb);
      qa = __builtin_fminf(__builtin_fmaxf(qa, -8.0f), 7.0f);
      qb = __builtin_fminf(__builtin_fmaxf(qb, -8.0f), 7.0f);
      byte[i] = ((uint32_t)(int)qa & 0xFu) | (((uint32_t)(int)qb & 0xFu) << 4);
    }
    uint8_t* dst = out + r2 * cols + c0;
    if (fast) {
      if constexpr (V == 8) {
        *reinterpret_cast<uint2*>(dst) = make_uint2(byte[0] | (byte[1] << 8) | (byte[2] << 16) | (byte[3] << 24),
                                                    byte[4] | (byte[5] << 8) | (byte[6] << 16) | (byte[7] << 24));
      } else {
        *reinterpret_cast<uint32_t*>(dst) = byte[0] | (byte[1] << 8) | (byte[2] << 16) | (byte[3] << 24);
      }
    } else {
      for (int i = 0; i < V; ++i)
        if (c0 + i < cols) dst[i] = (uint8_t)byte[i];
    }
  }
}

// ================================================================================================
// column scale fold (a11/a12 postprocess)
// ================================================================================================
template <int DT>
__global__ __launch_bounds__(kBlock) void scale_cols_kernel(const void* __restrict__ w,
                                                            const float* __restrict__ s,
                                                            void* __restrict__ y, int64_t rows,
                                                            int64_t cols) {
  constexpr int V = Elem<DT>::kVec;
  const int64_t n = rows * cols;
  const bool fast = al16(w) && al16(y) && (cols % V) == 0 && (reinterpret_cast<uintptr_t>(s) & 15u) == 0;
  const int64_t n_packets = (n + V - 1) / V;
  for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < n_packets;
       p += (int64_t)gridDim.x * kBlock) {
    const int64_t e = p * V;
    float v[8];
    if (fast) {
      unpack<DT>(load16(reinterpret_cast<const char*>(w) + p * 16), v);
      const int64_t c = e % cols;
      const float4 s0 = *reinterpret_cast<const float4*>(s + c);
      v[0] *= s0.x; v[1] *= s0.y; v[2] *= s0.z; v[3] *= s0.w;
      if constexpr (V == 8) {
        const float4 s1 = *reinterpret_cast<const float4*>(s + c + 4);
        v[4] *= s1.x; v[5] *= s1.y; v[6] *= s1.z; v[7] *= s1.w;
      }
      store16(reinterpret_cast<char*>(y) + p * 16, pack<DT>(v));
    } else {
      for (int i = 0; i < V; ++i)
        if (e + i < n) store1<DT>(y, e + i, load1<DT>(w, e + i) * s[(e + i) % cols]);
    }
  }
}

// y[r, c] = dtype((w[r, c] * mul[c]) / div[c]): _update_pre_quant_scale of the export resmooth step
// (export/quant_utils.py:1285-1296) -- fp32 multiply, fp32 IEEE divide, one rounding to the storage dtype.
template <int DT>
__global__ __launch_bounds__(kBlock) void rescale_cols_kernel(const void* __restrict__ w,
                                                              const float* __restrict__ mul,
                                                              const float* __restrict__ div,
                                                              void* __restrict__ y, int64_t rows,
                                                              int64_t cols) {
  constexpr int V = Elem<DT>::kVec;
  const int64_t n = rows * cols;
  const bool fast = al16(w) && al16(y) && (cols % V) == 0 && al16(mul) && al16(div);
  const int64_t n_packets = (n + V - 1) / V;
  for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < n_packets;
       p += (int64_t)gridDim.x * kBlock) {
    const int64_t e = p * V;
    if (fast) {
      float v[8];
      unpack<DT>(load16(reinterpret_cast<const char*>(w) + p * 16), v);
      const int64_t c = e % cols;
#pragma unroll
      for (int i = 0; i < V; ++i) v[i] = (v[i] * mul[c + i]) / div[c + i];
      store16(reinterpret_cast<char*>(y) + p * 16, pack<DT>(v));
    } else {
      for (int i = 0; i < V; ++i)
        if (e + i < n) store1<DT>(y, e + i, (load1<DT>(w, e + i) * mul[(e + i) % cols]) / div[(e + i) % cols]);
    }
  }
}

// y[a, r, c] = dtype(w[r, c] * s[a, c]) for a < n_scales: ONE read of w, n_scales writes (the 11 pre-scaled
// activation copies of an AWQ search step).  Requires the fast layout (checked by the host entry).
template <int DT>
__global__ __launch_bounds__(kBlock) void scale_cols_multi_kernel(const void* __restrict__ w,
                                                                  const float* __restrict__ s,
                                                                  void* __restrict__ y, int64_t rows,
                                                                  int64_t cols, int n_scales) {
  constexpr int V = Elem<DT>::kVec;
  const int64_t n = rows * cols;
  const int64_t n_packets = n / V;
  for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < n_packets;
       p += (int64_t)gridDim.x * kBlock) {
    float v[8];
    unpack<DT>(load16_nt(reinterpret_cast<const char*>(w) + p * 16), v);
    const int64_t c = (p * V) % cols;
    for (int a = 0; a < n_scales; ++a) {
      const float* sa = s + (int64_t)a * cols + c;
      float o[8];
      const float4 s0 = *reinterpret_cast<const float4*>(sa);
      o[0] = v[0] * s0.x; o[1] = v[1] * s0.y; o[2] = v[2] * s0.z; o[3] = v[3] * s0.w;
      if constexpr (V == 8) {
        const float4 s1 = *reinterpret_cast<const float4*>(sa + 4);
        o[4] = v[4] * s1.x; o[5] = v[5] * s1.y; o[6] = v[6] * s1.z; o[7] = v[7] * s1.w;
      }
      store16(reinterpret_cast<char*>(y) + ((int64_t)a * n_packets + p) * 16, pack<DT>(o));
    }
  }
}

}  // namespace moq

// ================================================================================================
// C-ABI
// ================================================================================================
using namespace moq;
static inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }

extern "C" int moq_mx_fused_amax_convert(const void* x, void* y, int64_t rows, int64_t cols, int block,
                                         int dt, int fmt, int scale_fmt, const float* global_amax,
                                         void* stream) {
  if (rows < 0 || cols < 0 || block <= 0 || (rows * cols > 0 && (x == nullptr || y == nullptr))) {
    set_error("moq_mx_fused_amax_convert: bad arguments");
    return MOQ_ERR_INVALID;
  }
  if (scale_fmt != MOQ_E8M0 || global_amax != nullptr) {
    set_error("moq_mx_fused_amax_convert: only E8M0 block scales without a global amax are implemented");
    return MOQ_ERR_UNSUPPORTED;
  }
  if (mx_fmt(fmt).kind < 0) {
    set_error("moq_mx_fused_amax_convert: unknown element format %d", fmt);
    return MOQ_ERR_INVALID;
  }
  const int64_t n = rows * cols;
  if (n == 0) return MOQ_OK;
  const int vec = dt == MOQ_F32 ? 4 : 8;
  const bool aligned = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15u) == 0;
  const int lpg = block / vec;
  if (aligned && cols % block == 0 && block % vec == 0 && lpg <= 64 && (lpg & (lpg - 1)) == 0) {
    const int64_t n_packets = n / vec;
    const int grid = stream_grid(kBlock, n_packets);
#define MOQ_MX_CASE(L) \
  case L: MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((mx_kernel<DT, L>), dim3(grid), dim3(kBlock), 0, S(stream), x, y, n_packets, fmt)); break;
    switch (lpg) {
      MOQ_MX_CASE(1) MOQ_MX_CASE(2) MOQ_MX_CASE(4) MOQ_MX_CASE(8) MOQ_MX_CASE(16) MOQ_MX_CASE(32) MOQ_MX_CASE(64)
      default: set_error("unreachable"); return MOQ_ERR_INVALID;
    }
#undef MOQ_MX_CASE
  } else {
    const int64_t nb = rows * ((cols + block - 1) / block);
    const int grid = stream_grid(kBlock, nb);
    MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((mx_generic_kernel<DT>), dim3(grid), dim3(kBlock), 0, S(stream),
                                              x, y, rows, cols, block, fmt));
  }
  return check_launch("moq_mx_fused_amax_convert");
}

extern "C" int moq_hist_abs(const void* x, int64_t n, int dt, unsigned long long* counts, int bins,
                            float max_edge, int skip_zeros, void* stream) {
  if (n < 0 || bins <= 0 || counts == nullptr || (n > 0 && x == nullptr)) {
    set_error("moq_hist_abs: bad arguments");
    return MOQ_ERR_INVALID;
  }
  if (n == 0) return MOQ_OK;
  const int vec = dt == MOQ_F32 ? 4 : 8;
  // one workgroup per ~64 KiB of input keeps the LDS flush (bins atomics per workgroup) amortised
  int64_t blocks = (n / vec + kBlock * 16 - 1) / (kBlock * 16);
  if (blocks < 1) blocks = 1;
  if (blocks > 1024) blocks = 1024;
  if (bins <= kHistMaxLdsBins) {
    MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((hist_kernel<DT, true>), dim3((int)blocks), dim3(kBlock),
                                              (size_t)bins * 4, S(stream), x, n, counts, bins, max_edge,
                                              skip_zeros));
  } else {
    MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((hist_kernel<DT, false>), dim3((int)blocks), dim3(kBlock), 0,
                                              S(stream), x, n, counts, bins, max_edge, skip_zeros));
  }
  return check_launch("moq_hist_abs");
}

extern "C" int moq_mask_2to4(const void* w, int64_t rows, int64_t cols, int dt, uint8_t* mask,
                             void* stream) {
  if (rows < 0 || cols < 0 || (rows * cols > 0 && (w == nullptr || mask == nullptr))) {
    set_error("moq_mask_2to4: bad arguments");
    return MOQ_ERR_INVALID;
  }
  if (cols % 4 != 0) {
    set_error("moq_mask_2to4: cols=%lld is not a multiple of 4 (pad on the host like reshape_1d)", (long long)cols);
    return MOQ_ERR_UNSUPPORTED;
  }
  const int64_t n = rows * cols;
  if (n == 0) return MOQ_OK;
  const int vec = dt == MOQ_F32 ? 4 : 8;
  const int grid = stream_grid(kBlock, (n + vec - 1) / vec);
  MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((mask24_kernel<DT>), dim3(grid), dim3(kBlock), 0, S(stream), w,
                                            mask, n));
  return check_launch("moq_mask_2to4");
}

extern "C" int moq_int4_pack(const void* x, const void* scales, uint8_t* out, int64_t n, int g, int dt,
                             int rounding, void* stream) {
  if (n < 0 || g <= 0 || (n > 0 && (x == nullptr || scales == nullptr || out == nullptr))) {
    set_error("moq_int4_pack: bad arguments");
    return MOQ_ERR_INVALID;
  }
  if (g % 2 != 0 || n % g != 0) {
    set_error("moq_int4_pack: need g even and n %% g == 0 (tensor_quant_gpu.cu:354-355)");
    return MOQ_ERR_INVALID;
  }
  if (rounding != MOQ_ROUND_HALF_EVEN && rounding != MOQ_ROUND_HALF_AWAY) {
    set_error("moq_int4_pack: unknown rounding %d", rounding);
    return MOQ_ERR_INVALID;
  }
  if (n == 0) return MOQ_OK;
  const int vec = dt == MOQ_F32 ? 4 : 8;
  const int grid = stream_grid(kBlock, (n + vec - 1) / vec);
  MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((int4_pack_kernel<DT>), dim3(grid), dim3(kBlock), 0, S(stream), x,
                                            scales, out, n, g, rounding));
  return check_launch("moq_int4_pack");
}

extern "C" int moq_int4_unpack(const uint8_t* q, const void* scales, void* out, int64_t n_bytes, int g,
                               int dt, void* stream) {
  if (n_bytes < 0 || g <= 0 || g % 2 != 0 || (n_bytes > 0 && (q == nullptr || scales == nullptr || out == nullptr))) {
    set_error("moq_int4_unpack: bad arguments");
    return MOQ_ERR_INVALID;
  }
  if (n_bytes == 0) return MOQ_OK;
  const int grid = stream_grid(kBlock, (n_bytes + 3) / 4);
  MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((int4_unpack_kernel<DT>), dim3(grid), dim3(kBlock), 0, S(stream),
                                            q, scales, out, n_bytes, g));
  return check_launch("moq_int4_unpack");
}

extern "C" int moq_int4_pack_export(const void* w, const float* wsf, uint8_t* out, int64_t rows,
                                    int64_t cols, int g, int dt, void* stream) {
  if (rows < 0 || cols < 0 || g <= 0 || (rows * cols > 0 && (w == nullptr || wsf == nullptr || out == nullptr))) {
    set_error("moq_int4_pack_export: bad arguments");
    return MOQ_ERR_INVALID;
  }
  if (rows % 2 != 0 || cols % g != 0) {
    set_error("moq_int4_pack_export: rows must be even and cols a multiple of g (quant_utils.py:795)");
    return MOQ_ERR_INVALID;
  }
  if (rows * cols == 0) return MOQ_OK;
  const int vec = dt == MOQ_F32 ? 4 : 8;
  const int grid = stream_grid(kBlock, (rows / 2) * ((cols + vec - 1) / vec));
  MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((int4_export_kernel<DT>), dim3(grid), dim3(kBlock), 0, S(stream),
                                            w, wsf, out, rows, cols, g));
  return check_launch("moq_int4_pack_export");
}

extern "C" int moq_scale_cols(const void* w, const float* s, void* y, int64_t rows, int64_t cols, int dt,
                              void* stream) {
  if (rows < 0 || cols < 0 || (rows * cols > 0 && (w == nullptr || s == nullptr || y == nullptr))) {
    set_error("moq_scale_cols: bad arguments");
    return MOQ_ERR_INVALID;
  }
  const int64_t n = rows * cols;
  if (n == 0) return MOQ_OK;
  const int vec = dt == MOQ_F32 ? 4 : 8;
  const int grid = stream_grid(kBlock, (n + vec - 1) / vec);
  MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((scale_cols_kernel<DT>), dim3(grid), dim3(kBlock), 0, S(stream), w,
                                            s, y, rows, cols));
  return check_launch("moq_scale_cols");
}

extern "C" int moq_scale_cols_multi(const void* w, const float* s, void* y, int64_t rows, int64_t cols,
                                    int n_scales, int dt, void* stream) {
  if (rows < 0 || cols <= 0 || n_scales < 1 || w == nullptr || s == nullptr || y == nullptr) {
    set_error("moq_scale_cols_multi: bad arguments");
    return MOQ_ERR_INVALID;
  }
  const int vec = dt == MOQ_F32 ? 4 : 8;
  if (cols % vec != 0 || ((reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(y) |
                            reinterpret_cast<uintptr_t>(s)) & 15u) != 0) {
    set_error("moq_scale_cols_multi: needs cols %% %d == 0 and 16-byte aligned pointers", vec);
    return MOQ_ERR_UNSUPPORTED;
  }
  const int64_t n = rows * cols;
  if (n == 0) return MOQ_OK;
  const int grid = stream_grid(kBlock, n / vec);
  MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((scale_cols_multi_kernel<DT>), dim3(grid), dim3(kBlock), 0, S(stream),
                                            w, s, y, rows, cols, n_scales));
  return check_launch("moq_scale_cols_multi");
}

extern "C" int moq_rescale_cols(const void* w, const float* mul, const float* div, void* y, int64_t rows,
                                int64_t cols, int dt, void* stream) {
  if (rows < 0 || cols < 0 || (rows * cols > 0 && (w == nullptr || mul == nullptr || div == nullptr || y == nullptr))) {
    set_error("moq_rescale_cols: bad arguments");
    return MOQ_ERR_INVALID;
  }
  const int64_t n = rows * cols;
  if (n == 0) return MOQ_OK;
  const int vec = dt == MOQ_F32 ? 4 : 8;
  const int grid = stream_grid(kBlock, (n + vec - 1) / vec);
  MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((rescale_cols_kernel<DT>), dim3(grid), dim3(kBlock), 0, S(stream), w,
                                            mul, div, y, rows, cols));
  return check_launch("moq_rescale_cols");
}
