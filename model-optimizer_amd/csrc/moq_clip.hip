// moq_clip.hip -- AWQ-clip block search (a13): awq_clip / _clip_search, block branch
// (quantization/model_calib.py:1800-1868).
//
// For every weight block (output channel r, block b of g input channels) and every clip ratio k the reference
// materialises  org[r,t,b] = sum_j x[t,j] w[r,j]  and  cur_k[r,t,b] = sum_j x[t,j] QDQ(w[r,j]; amax[r,b] * shrink_k)
// as broadcast products [co_batch, tokens, n_block, g] (11 + 1 of them per output-channel batch) and reduces
// mean_t (cur_k - org)^2.  These are n_block independent small contractions with K = g, i.e. dense MFMA work:
//
//   * a wave owns a tile of 32 output channels x one block and keeps the block's activations (<= 64 tokens x g)
//     as MFMA A-fragments in registers while it walks down the output channels;
//   * the 32 x g weight tile is read ONCE from HBM (16-byte lane loads, a lane owns one weight row); all clip
//     ratios are evaluated from those registers: the integer codes clamp(rint(w * scale_k)) are exact in
//     bf16 / f16 / f32 and go straight into the matrix cores as the B operand, the dequantisation divide is
//     applied to the 32 x tokens result (sum_j x_j q_j) / scale_k instead of to every element;
//   * C layout of v_mfma_f32_32x32x*: column = lane & 31 (our output channel), rows spread over the 16
//     accumulator registers and the lane half (our tokens) -> mean_t is an in-lane sum plus ONE cross-lane add.
//
// Arithmetic vs the reference: org / cur are rounded to the model dtype where the reference materialises them,
// their difference is rounded to the model dtype, squared and averaged in fp32 -- as in the reference.  Inside a
// block dot the reference rounds every product x*w to the model dtype before its fp32 summation; the matrix
// cores keep the products exact.  For fp32 models the two agree to summation order (rtol ~1e-5); for 16-bit
// models they differ at the level of the reference's own rounding noise (tests state the tolerance).
//
// Bound: VALU/MFMA co-issue, not HBM -- per weight element 11 x (mul, rint, clamp, convert) and 12 x 2 MFMA
// K-slices; the weight is streamed once (2 B/element) at ~1-2 TB/s, the whole Llama-3-8B search pass costs
// ~10 ms per calibration batch against ~100 ms for the model forward that feeds it.
#include "moq_common.h"

namespace moq {

typedef __bf16 c_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 c_f16x8 __attribute__((ext_vector_type(8)));
typedef float c_f32x16 __attribute__((ext_vector_type(16)));

constexpr int kClipMaxShrink = 32;

// one K-slice of a 32 x 32 tile: 16-bit types consume a whole 16-byte packet (8 k per lane half) per MFMA,
// f32 consumes one value per lane per MFMA (v_mfma_f32_32x32x2_f32: exact fp32 FMA chain)
template <int DT>
__device__ __forceinline__ c_f32x16 clip_mfma(const Pack16& a, const Pack16& b, c_f32x16 c) {
  if constexpr (DT == MOQ_BF16) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const c_bf16x8*>(&a),
                                                   *reinterpret_cast<const c_bf16x8*>(&b), c, 0, 0, 0);
  } else if constexpr (DT == MOQ_F16) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const c_f16x8*>(&a),
                                                  *reinterpret_cast<const c_f16x8*>(&b), c, 0, 0, 0);
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w[i]), __uint_as_float(b.w[i]), c, 0, 0, 0);
    return c;
  }
}

__device__ __forceinline__ Pack16 zero_pack() {
  Pack16 p;
  p.w[0] = p.w[1] = p.w[2] = p.w[3] = 0u;
  return p;
}

// NP = 16-byte packets per lane per block row = g / (2 * kVec): lane half h owns packets 2p + h.
// TT = token tiles (32 tokens each) held in registers per pass.
template <int DT, int NP, int TT, int ADT>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu((TT <= 2 && NP <= 8 && DT != MOQ_F32) ? 2 : 1, (TT <= 2 && NP <= 8 && DT != MOQ_F32) ? 2 : 1))) void awq_clip_kernel(const void* __restrict__ x, int64_t n_tok,
                                                          int64_t x_row_stride, const void* __restrict__ w,
                                                          int64_t cout, int64_t cin,
                                                          const float* __restrict__ amax,
                                                          const float* __restrict__ shrinks, int n_shrink,
                                                          float bound, int tiles_per_wave, float n_tok_total,
                                                          float* __restrict__ loss) {
  constexpr int V = Elem<DT>::kVec;
  constexpr int ES = 16 / V;
  constexpr int G = NP * 2 * V;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 31, h = lane >> 5;
  const int64_t blk = blockIdx.x;
  const int64_t nblk = gridDim.x;
  const int64_t col0 = blk * G;
  const int64_t n_tiles = (cout + 31) / 32;
  const int64_t tile0 = ((int64_t)blockIdx.y * (kBlock / 64) + wave) * tiles_per_wave;
  if (tile0 >= n_tiles) return;
  const float lo = -bound - 1.0f;
  const char* xb = reinterpret_cast<const char*>(x);
  const char* wb = reinterpret_cast<const char*>(w);

  {
    const int64_t t0 = 0;  // the host launches once per 32 * TT tokens
    // activations of this block as A fragments: row (token) = lane & 31, k-run = packet 2p + h
    Pack16 xa[TT][NP];
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      const int64_t t = t0 + tt * 32 + c;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int64_t col = col0 + (2 * p + h) * V;
        xa[tt][p] = (t < n_tok && col < cin) ? load16(xb + (t * x_row_stride + col) * ES) : zero_pack();
      }
    }
    for (int it = 0; it < tiles_per_wave; ++it) {
      const int64_t tile = tile0 + it;
      if (tile >= n_tiles) break;
      const int64_t r = tile * 32 + c;
      const bool valid = r < cout;
      Pack16 wp[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int64_t col = col0 + (2 * p + h) * V;
        wp[p] = (valid && col < cin) ? load16_nt(wb + (r * cin + col) * ES) : zero_pack();
      }
      const float am = valid ? amax[r * nblk + blk] : 0.0f;
      // org = x . w in the model dtype
      c_f32x16 acc[TT];
      float org[TT][16];
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[tt][e] = 0.0f;
      }
#pragma unroll
      for (int p = 0; p < NP; ++p) {
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) acc[tt] = clip_mfma<DT>(xa[tt][p], wp[p], acc[tt]);
      }
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
#pragma unroll
        for (int e = 0; e < 16; ++e) org[tt][e] = round_to_dtype<DT>(acc[tt][e]);
      }
      for (int k = 0; k < n_shrink; ++k) {
        // amax_k = w_amax * shrink in w_amax's dtype (tensor * python float, model_calib.py:1841)
        const float ak = round_to_dtype<ADT>(am * shrinks[k]);
        const float scale = int_scale(ak, bound);
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[tt][e] = 0.0f;
        }
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          float f[8];
          unpack<DT>(wp[p], f);
#pragma unroll
          for (int i = 0; i < V; ++i) {
            float t = __builtin_rintf(f[i] * scale);
            f[i] = __builtin_fminf(__builtin_fmaxf(t, lo), bound);  // integer codes: exact in every dtype
          }
          const Pack16 q = pack<DT>(f);
#pragma unroll
          for (int tt = 0; tt < TT; ++tt) acc[tt] = clip_mfma<DT>(xa[tt][p], q, acc[tt]);
        }
        // cur = (sum_j x_j q_j) / scale (the QDQ divide, applied once per output), loss = mean_t dt(cur - org)^2
        const SharedDiv sd = make_shared_div(scale == 0.0f ? 1.0f : scale);
        float s = 0.0f;
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const float cur = round_to_dtype<DT>(shared_div(acc[tt][e], sd));
            const float d = round_to_dtype<DT>(cur - org[tt][e]);
            s += d * d;
          }
        }
        s += __shfl_xor(s, 32, 64);
        if (h == 0 && valid) {
          float* dst = loss + ((int64_t)k * nblk + blk) * cout + r;  // [n_shrink, nblk, cout]: 128-B runs
          *dst += s / n_tok_total;  // .mean(dim=1): sum / tokens in fp32
        }
      }
    }
  }
}

}  // namespace moq

using namespace moq;
static inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }

extern "C" int moq_awq_clip_loss(const void* x, int64_t n_tok, int64_t x_row_stride, const void* w, int64_t cout,
                                 int64_t cin, int g, int dt, const float* amax, int amax_dt,
                                 const float* shrinks, int n_shrink, int num_bits, float* loss, void* stream) {
  if (n_tok <= 0 || cout <= 0 || cin <= 0 || g <= 0 || x == nullptr || w == nullptr || amax == nullptr ||
      shrinks == nullptr || loss == nullptr || n_shrink <= 0 || n_shrink > kClipMaxShrink || num_bits < 2 ||
      num_bits > 8) {
    set_error("moq_awq_clip_loss: bad arguments");
    return MOQ_ERR_INVALID;
  }
  const int vec = dt == MOQ_F32 ? 4 : 8;
  const int es = 16 / vec;
  if (cin % vec != 0 || x_row_stride % vec != 0 || x_row_stride < cin ||
      ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) & 15u) != 0) {
    set_error("moq_awq_clip_loss: needs cin %% %d == 0, x_row_stride %% %d == 0 and 16-byte aligned x / w", vec, vec);
    return MOQ_ERR_UNSUPPORTED;
  }
  if (amax_dt != MOQ_F32 && amax_dt != dt) {
    set_error("moq_awq_clip_loss: amax_dt must be MOQ_F32 or the weight dtype");
    return MOQ_ERR_INVALID;
  }
  const int np = g / (2 * vec);
  if (g % (2 * vec) != 0 || (np != 2 && np != 4 && np != 8 && np != 16)) {
    set_error("moq_awq_clip_loss: block size %d not supported (need g / %d in {2, 4, 8, 16})", g, 2 * vec);
    return MOQ_ERR_UNSUPPORTED;
  }
  const int64_t nblk = (cin + g - 1) / g;
  const int64_t n_tiles = (cout + 31) / 32;
  // >= ~2048 waves when the weight allows it, <= 8 tiles per wave so that the activations are re-read rarely
  int tpw = 8;
  while (tpw > 1 && nblk * ((n_tiles + tpw - 1) / tpw) < 2048) tpw >>= 1;
  const int64_t tiles_per_wg = (int64_t)tpw * (kBlock / 64);
  dim3 grid((unsigned)nblk, (unsigned)((n_tiles + tiles_per_wg - 1) / tiles_per_wg));
  const float bound = (float)((1 << (num_bits - 1)) - 1);
  // One launch covers up to 128 tokens (16-bit, g <= 128) / 64 tokens (fp32, g = 256) held as register fragments; the reference's
  // sub-sampling yields max_tokens .. 2 * max_tokens - 1 = 64..127 tokens per call, i.e. one launch.  More
  // tokens (a larger max_tokens_per_batch) are processed in chunks that all divide by the full token count.
  const int64_t max_tok = (dt == MOQ_F32 || np == 16) ? 64 : 128;
  const char* xb = reinterpret_cast<const char*>(x);
  for (int64_t t0 = 0; t0 < n_tok; t0 += max_tok) {
    const int64_t nt = n_tok - t0 < max_tok ? n_tok - t0 : max_tok;
    const void* xc = xb + t0 * x_row_stride * es;
    const int tt = nt <= 32 ? 1 : (nt <= 64 ? 2 : 4);
#define MOQ_CLIP_LAUNCH(DTV, NPV, TTV, ADTV)                                                                  \
  hipLaunchKernelGGL((awq_clip_kernel<DTV, NPV, TTV, ADTV>), grid, dim3(kBlock), 0, S(stream), xc, nt,         \
                     x_row_stride, w, cout, cin, amax, shrinks, n_shrink, bound, tpw, (float)n_tok, loss)
#define MOQ_CLIP_TT(DTV, NPV, ADTV)                                         \
  switch (tt) {                                                             \
    case 1: MOQ_CLIP_LAUNCH(DTV, NPV, 1, ADTV); break;                      \
    case 2: MOQ_CLIP_LAUNCH(DTV, NPV, 2, ADTV); break;                      \
    default:                                                                \
      if constexpr (DTV != MOQ_F32 && NPV < 16) { MOQ_CLIP_LAUNCH(DTV, NPV, 4, ADTV); } \
      break;                                                                \
  }
#define MOQ_CLIP_NP(DTV, ADTV)                        \
  switch (np) {                                       \
    case 2: MOQ_CLIP_TT(DTV, 2, ADTV); break;         \
    case 4: MOQ_CLIP_TT(DTV, 4, ADTV); break;         \
    case 8: MOQ_CLIP_TT(DTV, 8, ADTV); break;         \
    default: MOQ_CLIP_TT(DTV, 16, ADTV); break;       \
  }
    if (dt == MOQ_F32) {
      MOQ_CLIP_NP(MOQ_F32, MOQ_F32)
    } else if (dt == MOQ_BF16) {
      if (amax_dt == MOQ_F32) { MOQ_CLIP_NP(MOQ_BF16, MOQ_F32) } else { MOQ_CLIP_NP(MOQ_BF16, MOQ_BF16) }
    } else if (dt == MOQ_F16) {
      if (amax_dt == MOQ_F32) { MOQ_CLIP_NP(MOQ_F16, MOQ_F32) } else { MOQ_CLIP_NP(MOQ_F16, MOQ_F16) }
    } else {
      set_error("moq_awq_clip_loss: unknown dtype %d", dt);
      return MOQ_ERR_INVALID;
    }
#undef MOQ_CLIP_NP
#undef MOQ_CLIP_TT
#undef MOQ_CLIP_LAUNCH
  }
  return check_launch("moq_awq_clip_loss");
}
