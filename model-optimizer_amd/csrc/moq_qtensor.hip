// moq_qtensor.hip -- real (storage) quantisation of the path's formats other than INT4 (a15):
//   FP8QTensor.quantize / dequantize   (quantization/qtensor/fp8_tensor.py:40-151; also export's to_quantized_weight)
//   MXFP4QTensor.quantize / dequantize (quantization/qtensor/mxfp4_tensor.py:37-144)
// HBM-bound streaming kernels on the chunk skeleton (moq_chunk.h): 16-byte lane loads, all packets of a chunk in
// flight, non-temporal loads / stores.  Algorithmic bytes per element (bf16): FP8 pack 2 + 1, unpack 1 + 2;
// MXFP4 pack 2 + 0.5 + 1/32, unpack 0.5 + 1/32 + 2.
#include <type_traits>

#include "moq_common.h"
#include "moq_chunk.h"

namespace moq {

__device__ __forceinline__ void q_store8_nt(void* p, uint32_t a, uint32_t b) {
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  u32x2 v = {a, b};
  __builtin_nontemporal_store(v, reinterpret_cast<u32x2*>(p));
}

// torch's float -> float8_e4m3fn cast: RNE, NOT saturating: |v| > 464 (the midpoint above 448) and NaN give NaN
// (byte 0x7F | sign).  The hardware converter is fed values clamped to +-448, overflow is patched afterwards.
__device__ __forceinline__ uint32_t e4m3fn_bytes2(float a, float b) {
  const float ca = __builtin_fminf(__builtin_fmaxf(a, -448.0f), 448.0f);
  const float cb = __builtin_fminf(__builtin_fmaxf(b, -448.0f), 448.0f);
  uint32_t p = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(ca, cb, 0, false) & 0xFFFFu;
  const uint32_t na = ((__float_as_uint(a) >> 31) << 7) | 0x7Fu, nb = ((__float_as_uint(b) >> 31) << 7) | 0x7Fu;
  if (!(__builtin_fabsf(a) <= 464.0f)) p = (p & 0xFF00u) | na;
  if (!(__builtin_fabsf(b) <= 464.0f)) p = (p & 0x00FFu) | (nb << 8);
  return p;
}

// One packet -> V e4m3fn bytes (two dwords; the second is 0 for fp32 packets).  The quotient element / scale is rounded
// to the storage dtype first (ROUND) or kept in fp32 (promoted: fp32 scales).  Three levels, decided per packet from its
// abs-max pattern:
//   in range : scale > 0 inside SharedDiv's window and |x| <= 441 * scale for every element: every quotient stays below
//              448 after both roundings (441 * (1 + 2^-8) < 448) and nothing is NaN, so the converter needs neither the
//              clamp nor the overflow patch, and the one thing the residual steps of the shared division lose -- the sign
//              of a zero quotient -- comes back with one v_bfi (copysign).  Numerators too small for the residuals to stay
//              normal (|x| < 2^-100) have quotients below 2^-40: +-0 in e4m3 whatever their last bit.  About 10 VALU
//              instructions per element instead of 19 (the body was co-critical with the memory stream: 0.645 of 8 TB/s
//              in round 4).  With scale = amax / 448 only the packet holding the abs-max itself leaves this level.
//   shared   : the division still shares its reciprocal; clamp and NaN patch per pair
//   IEEE     : everything else
template <int DT, bool ROUND>
__device__ __forceinline__ void fp8_bytes_of_packet(const Pack16& in, float sc, const SharedDiv& sd, uint32_t& w0,
                                                    uint32_t& w1) {
  constexpr int V = Elem<DT>::kVec;
  float v[8];
  unpack<DT>(in, v);
  uint32_t b[4] = {0, 0, 0, 0};
  const uint32_t pmax = pack_absmax<DT>(in);
  const uint32_t lim = __float_as_uint(sc * 441.0f);
  if (sd.fast && sc > 0.0f && pmax <= 0x47800000u && pmax <= lim) {
    // the residual chain without its zero test: the sign is restored below
    auto quot = [&](float n) {
      const float q0 = n * sd.y;
      const float q1 = __builtin_fmaf(__builtin_fmaf(-sd.d, q0, n), sd.y, q0);
      return __builtin_fmaf(__builtin_fmaf(-sd.d, q1, n), sd.y, q1);
    };
    uint32_t w[2] = {0, 0};
#pragma unroll
    for (int i = 0; i < V; i += 4) {
      float q[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) q[j] = quot(v[i + j]);
      if constexpr (ROUND && DT == MOQ_BF16) {
        // two quotients per v_cvt_pk_bf16_f32; both signs come from the packet's own dword with one v_bfi
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int j = 0; j < 4; j += 2) {
          const f32x2 pr = {q[j], q[j + 1]};
          const bf16x2 rb = __builtin_convertvector(pr, bf16x2);
          uint32_t r = *reinterpret_cast<const uint32_t*>(&rb);
          r = (r & 0x7FFF7FFFu) | (in.w[(i + j) / 2] & 0x80008000u);
          q[j] = __uint_as_float(r << 16);
          q[j + 1] = __uint_as_float(r & 0xFFFF0000u);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          q[j] = __builtin_copysignf(q[j], v[i + j]);
          if constexpr (ROUND) q[j] = round_to_dtype<DT>(q[j]);
        }
      }
      int pk = __builtin_amdgcn_cvt_pk_fp8_f32(q[0], q[1], 0, false);
      pk = __builtin_amdgcn_cvt_pk_fp8_f32(q[2], q[3], pk, true);
      w[i / 4] = (uint32_t)pk;
    }
    w0 = w[0];
    w1 = w[1];
    return;
  } else {
    const bool shared = sd.fast && pmax <= 0x47800000u;
#pragma unroll
    for (int i = 0; i < V; i += 2) {
      float qa = shared ? shared_div_in_window(v[i], sd) : v[i] / sc;
      float qb = shared ? shared_div_in_window(v[i + 1], sd) : v[i + 1] / sc;
      if constexpr (ROUND) {
        qa = round_to_dtype<DT>(qa);
        qb = round_to_dtype<DT>(qb);
        if constexpr (DT == MOQ_BF16) {
          // torch's float -> bfloat16 conversion returns the canonical +NaN (0x7FC0) for every NaN; the hardware converter
          // keeps the sign, which the e4m3 NaN byte would then carry (0xFF instead of 0x7F).  Half keeps the sign in both.
          qa = (qa != qa) ? __uint_as_float(0x7FC00000u) : qa;
          qb = (qb != qb) ? __uint_as_float(0x7FC00000u) : qb;
        }
      }
      b[i / 2] = e4m3fn_bytes2(qa, qb);
    }
  }
  w0 = b[0] | (b[1] << 16);
  w1 = b[2] | (b[3] << 16);
}

// ---------------------------------------------------------------- FP8 pack / unpack
// scales have the storage dtype DT (the reference divides two tensors of the model dtype).  AXIS: one scale per
// `inner` consecutive elements, index = (e / inner) % axis_size (per-channel rows, 1-D blocks); else one scale.
template <int DT, bool AXIS, bool SF32>
__global__ __launch_bounds__(kBlock) void fp8_pack_kernel(const void* __restrict__ x,
                                                          const void* __restrict__ scales,
                                                          uint8_t* __restrict__ out, int64_t n, int64_t axis_size,
                                                          int64_t inner, int inner_shift, int wrap) {
  constexpr int V = Elem<DT>::kVec;
  constexpr int P = Chunk<DT>::kPackets;
  const int64_t n_chunks = (n + MOQ_MT_CHUNK - 1) / MOQ_MT_CHUNK;
  GroupIndex gi;
  gi.g = (uint32_t)inner;
  gi.shift = inner_shift;
  auto ld_scale = [&](int64_t i) { return SF32 ? reinterpret_cast<const float*>(scales)[i] : load1<DT>(scales, i); };
  const float s0 = AXIS ? 1.0f : ld_scale(0);
  const SharedDiv sd0 = make_shared_div(s0);
  for (int64_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    const int64_t e0 = c * MOQ_MT_CHUNK;
    if (AXIS) gi.seek(e0);
    // a chunk inside the tensor runs a copy of the body without per-packet bounds branches: with them hipcc waits for
    // every load on its own (one 16-byte load in flight per lane instead of the chunk's P)
    auto body = [&](auto FULL) {
    constexpr bool full = decltype(FULL)::value;
    Pack16 in[P];
    float sc[P];
#pragma unroll
    for (int u = 0; u < P; ++u) {
      const int64_t e = e0 + packet_off<DT>(u);
      sc[u] = s0;
      if (full || e < n) {
        in[u] = load16_nt(reinterpret_cast<const char*>(x) + e * (16 / V));
        if (AXIS) {
          int64_t row = gi.at((uint32_t)packet_off<DT>(u));
          if (wrap) row %= axis_size;
          sc[u] = ld_scale(row);
        }
      }
    }
      if constexpr (full) __builtin_amdgcn_sched_barrier(0);  // every load of the chunk is issued before the first use
#pragma unroll
    for (int u = 0; u < P; ++u) {
      const int64_t e = e0 + packet_off<DT>(u);
      if (!full && e >= n) continue;
      // the quotients of a packet share their denominator (AXIS: the packet's own scale -- one reciprocal per packet
      // instead of one per element); fp8_bytes_of_packet picks the cheapest exact form
      uint32_t w0, w1;
      fp8_bytes_of_packet<DT, true>(in[u], sc[u], AXIS ? make_shared_div(sc[u]) : sd0, w0, w1);
      if constexpr (V == 8) q_store8_nt(out + e, w0, w1);
      else __builtin_nontemporal_store(w0, reinterpret_cast<uint32_t*>(out + e));
    }
    };
    if (e0 + MOQ_MT_CHUNK <= n) body(std::true_type{});
    else body(std::false_type{});
  }
}
template <int DT, bool AXIS>
__global__ __launch_bounds__(kBlock) void fp8_unpack_kernel(const uint8_t* __restrict__ q,
                                                            const void* __restrict__ scales,
                                                            void* __restrict__ out, int64_t n, int64_t axis_size,
                                                            int64_t inner, int inner_shift, int wrap) {
  constexpr int V = Elem<DT>::kVec;
  constexpr int P = Chunk<DT>::kPackets;
  const int64_t n_chunks = (n + MOQ_MT_CHUNK - 1) / MOQ_MT_CHUNK;
  GroupIndex gi;
  gi.g = (uint32_t)inner;
  gi.shift = inner_shift;
  const float s0 = AXIS ? 1.0f : load1<DT>(scales, 0);
  for (int64_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    const int64_t e0 = c * MOQ_MT_CHUNK;
    if (AXIS) gi.seek(e0);
    auto body = [&](auto FULL) {  // (see fp8_pack_kernel)
    constexpr bool full = decltype(FULL)::value;
    uint32_t in[P][2];
    float sc[P];
#pragma unroll
    for (int u = 0; u < P; ++u) {
      const int64_t e = e0 + packet_off<DT>(u);
      sc[u] = s0;
      if (full || e < n) {
        if constexpr (V == 8) {
          typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
          const u32x2 t = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(q + e));
          in[u][0] = t.x;
          in[u][1] = t.y;
        } else {
          in[u][0] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(q + e));
          in[u][1] = 0;
        }
        if (AXIS) {
          int64_t row = gi.at((uint32_t)packet_off<DT>(u));
          if (wrap) row %= axis_size;
          sc[u] = load1<DT>(scales, row);
        }
      }
    }
      if constexpr (full) __builtin_amdgcn_sched_barrier(0);  // every load of the chunk is issued before the first use
#pragma unroll
    for (int u = 0; u < P; ++u) {
      const int64_t e = e0 + packet_off<DT>(u);
      if (!full && e >= n) continue;
      float v[8];
#pragma unroll
      for (int i = 0; i < V; ++i) {
        const int word = (int)in[u][i / 4];
        float f;
        switch (i & 3) {  // byte select must be an immediate
          case 0: f = __builtin_amdgcn_cvt_f32_fp8(word, 0); break;
          case 1: f = __builtin_amdgcn_cvt_f32_fp8(word, 1); break;
          case 2: f = __builtin_amdgcn_cvt_f32_fp8(word, 2); break;
          default: f = __builtin_amdgcn_cvt_f32_fp8(word, 3); break;
        }
        v[i] = f * sc[u];  // q.to(dtype) is exact (3 mantissa bits), product rounded once to dtype by pack()
      }
      store16_nt(reinterpret_cast<char*>(out) + e * (16 / V), pack<DT>(v));
    }
    };
    if (e0 + MOQ_MT_CHUNK <= n) body(std::true_type{});
    else body(std::false_type{});
  }
}

// ---------------------------------------------------------------- MXFP4 pack / unpack
// e = ceil(max(log2(descale), -127)) with torch's fp32 log2: the exact exponent / mantissa split, then the fp32
// addition k + log2(1 + f) decides whether a mantissa a few ulps above a power of two still rounds to the integer k.
__device__ __forceinline__ int mxfp4_exponent(float amax) {
  const float descale = amax / 6.0f;
  if (!(descale > 0.0f)) return -127;  // log2(0) = -inf -> max(-inf, -127)
  const uint32_t u = __float_as_uint(descale), ef = (u >> 23) & 0xFFu, mf = u & 0x7FFFFFu;
  if (ef == 0) return mf > 0x400000u ? -126 : -127;  // subnormal descale: 2^-127 is mf == 0x400000
  const int k = (int)ef - 127;
  if (mf == 0) return k;
  const float t = (float)k + (float)mf * 0x1p-23f * 1.44269504f;
  const int e = t > (float)k ? k + 1 : k;
  return e < -127 ? -127 : e;
}
__device__ __forceinline__ float pow2i(int e) {  // 2^e for e in [-127, 127]
  return e >= -126 ? __uint_as_float((uint32_t)(e + 127) << 23) : __uint_as_float(0x00400000u);
}
// ord = #{bound < |v|} over the 7 E2M1 rounding bounds 0.25, 0.75, 1.25, 1.75, 2.5, 3.5, 5 (strict: ties round down);
// sign_bit = (2 - sign(v)) // 2: zero gets 1.  Every bound has at most two mantissa bits, so "|v| > bound" is a comparison
// of the key (bits(|v|) + 2^21 - 1) >> 21 -- exponent field and top two mantissa bits, rounded up -- against the bound's
// own key: the seven compare / add pairs of the direct form (the kernel was VALU-bound on them, ~27 VALU instructions per
// element) become add, shift, clamp and a 19-entry table packed in one 64-bit constant.  NaN compares false everywhere.
// NO_NAN: the caller has seen the block's abs-max pattern at or below +inf -- no element is NaN, the test is dropped
template <bool NO_NAN = false>
__device__ __forceinline__ uint32_t mxfp4_nibble(float v) {
  const uint32_t u = __float_as_uint(v) & 0x7FFFFFFFu;
  int idx = (int)((u + 0x1FFFFFu) >> 21) - 500;  // key of 0.25 is 500, of 5.0 is 517
  idx = idx < 0 ? 0 : (idx > 18 ? 18 : idx);
  // idx: 0 -> 0 | 1..6 -> 1 | 7..9 -> 2 | 10, 11 -> 3 | 12, 13 -> 4 | 14, 15 -> 5 | 16, 17 -> 6 | 18 -> 7
  uint32_t ord = (uint32_t)(0x01F6B646D2449248ull >> (3 * idx)) & 7u;
  if constexpr (!NO_NAN) ord = u > 0x7F800000u ? 0u : ord;
  return ((v > 0.0f) ? 0u : 8u) + ord;
}
// FAST layout (host-checked): n % block == 0, block % kVec == 0, LPG = block / kVec a power of two <= 64.
template <int DT, int LPG>
__global__ __launch_bounds__(kBlock) void mxfp4_pack_kernel(const void* __restrict__ x,
                                                            uint8_t* __restrict__ packed,
                                                            uint8_t* __restrict__ e8m0, int64_t n,
                                                            int block_shift) {
  constexpr int V = Elem<DT>::kVec;
  constexpr int P = Chunk<DT>::kPackets;
  const int64_t n_chunks = (n + MOQ_MT_CHUNK - 1) / MOQ_MT_CHUNK;
  const int lane = threadIdx.x & 63;
  for (int64_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    const int64_t e0 = c * MOQ_MT_CHUNK;
    Pack16 in[P];
#pragma unroll
    for (int u = 0; u < P; ++u) {
      const int64_t e = e0 + packet_off<DT>(u);
      if (e < n) in[u] = load16_nt(reinterpret_cast<const char*>(x) + e * (16 / V));
      else in[u].w[0] = in[u].w[1] = in[u].w[2] = in[u].w[3] = 0u;
    }
#pragma unroll
    for (int u = 0; u < P; ++u) {
      const int64_t e = e0 + packet_off<DT>(u);
      const uint32_t amax_bits = group_max_u32<LPG>(pack_absmax<DT>(in[u]));
      const float amax = __uint_as_float(amax_bits);  // amax in fp32 (:70)
      const int ex = mxfp4_exponent(amax);
      const float inv = pow2i(-ex);  // x / 2^e == x * 2^-e exactly (power of two; same rounding when it underflows)
      float v[8];
      unpack<DT>(in[u], v);
      uint32_t word = 0;
      if (amax_bits <= 0x7F800000u) {  // every NaN pattern sorts above +inf: none in this block
#pragma unroll
        for (int i = 0; i < V; i += 2) {
          const uint32_t lo = mxfp4_nibble<true>(v[i] * inv), hi = mxfp4_nibble<true>(v[i + 1] * inv);
          word |= ((hi << 4) + lo) << (8 * (i / 2));
        }
      } else {
#pragma unroll
        for (int i = 0; i < V; i += 2) {
          const uint32_t lo = mxfp4_nibble(v[i] * inv), hi = mxfp4_nibble(v[i + 1] * inv);
          word |= ((hi << 4) + lo) << (8 * (i / 2));
        }
      }
      if (e < n) {
        if constexpr (V == 8) __builtin_nontemporal_store(word, reinterpret_cast<uint32_t*>(packed + e / 2));
        else *reinterpret_cast<uint16_t*>(packed + e / 2) = (uint16_t)word;
        if ((lane & (LPG - 1)) == 0) e8m0[e >> block_shift] = (uint8_t)(ex + 127);
      }
    }
  }
}
// the same over a segment table, with SmoothQuant's column fold in front (moq_mt_fold_mxfp4_pack): the block exponent and the
// nibbles are taken from dt(x * scale[col]), the value the separate fold would have written back.  Foldable tensors are
// walked in TILES (kPackets rows x kBlock * kVec columns: one scale read per thread and tile, see mt_fold_mx_kernel in
// moq_formats.hip); tensors without a fold, or whose shape does not tile, take the linear walk of mxfp4_pack_kernel.
template <int DT, int LPG>
__global__ __launch_bounds__(kBlock) void mt_fold_mxfp4_pack_kernel(const moq_seg* __restrict__ segs,
                                                                    const int64_t* __restrict__ blk_start,
                                                                    const moq_fold_seg* __restrict__ side, int n_seg,
                                                                    int64_t n_chunks, int block_shift) {
  constexpr int V = Elem<DT>::kVec;
  constexpr int P = Chunk<DT>::kPackets;
  constexpr int W = kBlock * V;
  if ((int64_t)blockIdx.x >= n_chunks) return;
  const int lane = threadIdx.x & 63;
  SegCursor cur;
  cur.init(segs, blk_start, n_seg, blockIdx.x);
  moq_fold_seg sd = side[cur.s];
  for (int64_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    if (cur.seek(c)) sd = side[cur.s];
    const int64_t j = c - cur.c_begin, n = cur.sg.n;
    const char* x = reinterpret_cast<const char*>(cur.sg.x);
    uint8_t* packed = reinterpret_cast<uint8_t*>(cur.sg.y);
    const bool fold = sd.scale != nullptr;  // (workgroup-uniform, like everything derived from the segment below)
    const uint32_t cols = (uint32_t)sd.cols;
    const bool tiled = fold && cols % W == 0 && (n / (int64_t)cols) % P == 0;
    Pack16 in[P];
    int64_t e[P];
    uint32_t col[P];
    if (tiled) {
      const uint32_t wpb = cols / W;
      const int64_t band = j / wpb;
      const uint32_t cl = (uint32_t)(j - band * wpb) * W + threadIdx.x * V;
#pragma unroll
      for (int u = 0; u < P; ++u) {
        e[u] = (band * P + u) * (int64_t)cols + cl;
        col[u] = cl;
      }
    } else {
      const int64_t e0 = j * MOQ_MT_CHUNK;
      const uint32_t c0 = fold ? (uint32_t)(e0 % (int64_t)cols) : 0u;
#pragma unroll
      for (int u = 0; u < P; ++u) {
        e[u] = e0 + packet_off<DT>(u);
        uint32_t t = c0 + (uint32_t)packet_off<DT>(u);
        col[u] = !fold ? 0u : (cols >= (uint32_t)MOQ_MT_CHUNK ? (t >= cols ? t - cols : t) : t % cols);
      }
    }
#pragma unroll
    for (int u = 0; u < P; ++u) {
      if (e[u] < n) in[u] = load16_nt(x + e[u] * (16 / V));
      else in[u].w[0] = in[u].w[1] = in[u].w[2] = in[u].w[3] = 0u;
    }
    float sf[8] = {1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f};
#pragma unroll
    for (int u = 0; u < P; ++u) {
      float v[8];
      unpack<DT>(in[u], v);
      uint32_t m;
      if (fold) {
        if (!tiled || u == 0) {
          const u32x4_t a = *((gptr_c16)(uintptr_t)(sd.scale + col[u]));  // (global, not FLAT: see fold_load in moq_formats.hip)
          sf[0] = __uint_as_float(a.x); sf[1] = __uint_as_float(a.y); sf[2] = __uint_as_float(a.z); sf[3] = __uint_as_float(a.w);
          if constexpr (V == 8) {
            const u32x4_t b = *((gptr_c16)(uintptr_t)(sd.scale + col[u] + 4));
            sf[4] = __uint_as_float(b.x); sf[5] = __uint_as_float(b.y); sf[6] = __uint_as_float(b.z); sf[7] = __uint_as_float(b.w);
          }
        }
        m = 0;
#pragma unroll
        for (int i = 0; i < V; ++i) {
          v[i] = round_to_dtype<DT>(v[i] * sf[i]);
          const uint32_t ab = absbits(v[i]);
          m = ab > m ? ab : m;
        }
      } else {
        m = pack_absmax<DT>(in[u]);
      }
      const uint32_t amax_bits = group_max_u32<LPG>(m);
      const int ex = mxfp4_exponent(__uint_as_float(amax_bits));
      const float inv = pow2i(-ex);
      uint32_t word = 0;
      if (amax_bits <= 0x7F800000u) {
#pragma unroll
        for (int i = 0; i < V; i += 2) {
          const uint32_t lo = mxfp4_nibble<true>(v[i] * inv), hi = mxfp4_nibble<true>(v[i + 1] * inv);
          word |= ((hi << 4) + lo) << (8 * (i / 2));
        }
      } else {
#pragma unroll
        for (int i = 0; i < V; i += 2) {
          const uint32_t lo = mxfp4_nibble(v[i] * inv), hi = mxfp4_nibble(v[i + 1] * inv);
          word |= ((hi << 4) + lo) << (8 * (i / 2));
        }
      }
      if (e[u] < n) {
        if constexpr (V == 8) __builtin_nontemporal_store(word, reinterpret_cast<uint32_t*>(packed + e[u] / 2));
        else *reinterpret_cast<uint16_t*>(packed + e[u] / 2) = (uint16_t)word;
        if ((lane & (LPG - 1)) == 0) sd.e8m0[e[u] >> block_shift] = (uint8_t)(ex + 127);
      }
    }
  }
}
template <int DT>
__global__ __launch_bounds__(kBlock) void mxfp4_generic_pack_kernel(const void* __restrict__ x,
                                                                    uint8_t* __restrict__ packed,
                                                                    uint8_t* __restrict__ e8m0,
                                                                    int64_t n_blocks, int block) {
  for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < n_blocks;
       b += (int64_t)gridDim.x * blockDim.x) {
    uint32_t am = 0;
    for (int j = 0; j < block; ++j) {
      const uint32_t a = absbits(load1<DT>(x, b * block + j));
      am = a > am ? a : am;
    }
    const int ex = mxfp4_exponent(__uint_as_float(am));
    const float inv = pow2i(-ex);
    e8m0[b] = (uint8_t)(ex + 127);
    for (int j = 0; j < block; j += 2) {
      const uint32_t lo = mxfp4_nibble(load1<DT>(x, b * block + j) * inv);
      const uint32_t hi = mxfp4_nibble(load1<DT>(x, b * block + j + 1) * inv);
      packed[(b * block + j) / 2] = (uint8_t)((hi << 4) + lo);
    }
  }
}
// 4 bytes -> 8 elements per lane (16-bit dtypes); generic per-element path otherwise
template <int DT, bool FASTL>
__global__ __launch_bounds__(kBlock) void mxfp4_unpack_kernel(const uint8_t* __restrict__ packed,
                                                              const uint8_t* __restrict__ e8m0,
                                                              void* __restrict__ out, int64_t n, int block,
                                                              int block_shift) {
  const float tbl[8] = {0.0f, 0.5f, 1.0f, 1.5f, 2.0f, 3.0f, 4.0f, 6.0f};
  if constexpr (FASTL && DT != MOQ_F32) {
    constexpr int P = 4;
    const int64_t n_chunks = (n + MOQ_MT_CHUNK - 1) / MOQ_MT_CHUNK;
    for (int64_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
      const int64_t e0 = c * MOQ_MT_CHUNK;
      uint32_t wv[P];
      uint32_t sb[P];
#pragma unroll
      for (int u = 0; u < P; ++u) {
        const int64_t e = e0 + packet_off<DT>(u);
        if (e < n) {
          wv[u] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(packed + e / 2));
          sb[u] = e8m0[e >> block_shift];
        }
      }
#pragma unroll
      for (int u = 0; u < P; ++u) {
        const int64_t e = e0 + packet_off<DT>(u);
        if (e >= n) continue;
        const float sc = pow2i((int)sb[u] - 127);
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const uint32_t nib = (wv[u] >> (4 * i)) & 0xFu;  // byte j: low nibble = even element, high = odd
          // E2M1 magnitude {0, .5, 1, 1.5, 2, 3, 4, 6} = exponent field (m >> 1), mantissa bit (m & 1)
          const uint32_t m = nib & 7u;
          const float mag = m < 2 ? 0.5f * (float)m : __uint_as_float(((m >> 1) + 126u) << 23 | (m & 1u) << 22);
          v[i] = ((nib & 8u) ? -mag : mag) * sc;
        }
        store16_nt(reinterpret_cast<char*>(out) + e * 2, pack<DT>(v));
      }
    }
  } else {
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n; e += (int64_t)gridDim.x * kBlock) {
      const uint8_t byte = packed[e / 2];
      const uint32_t nib = (e & 1) ? (byte >> 4) : (byte & 0xFu);
      const float sc = pow2i((int)e8m0[e / block] - 127);
      const float mag = tbl[nib & 7u];
      store1<DT>(out, e, ((nib & 8u) ? -mag : mag) * sc);
    }
  }
}


// ---------------------------------------------------------------- FP8 pack / unpack with one scale per br x bc tile
// FP8QTensor.quantize with block_sizes on both axes (fp8_tensor.py:60-112; the FP8 2-D blockwise weight-only export,
// export/quant_utils.py:874-877): the reference expands the [R/br, C/bc] scales with repeat_interleave and divides
// elementwise.  Here a packet (V elements of one row, inside one tile because bc % V == 0) looks its tile's scale up.
// PROMOTE: scales are fp32 while the tensor is 16-bit -> torch promotes the quotient to fp32 (dimensioned operand),
// so it is NOT rounded to the tensor dtype before the e4m3 cast; with scales of the tensor dtype it is.
// Indexing: blockIdx.y walks groups of kTileRows consecutive rows, blockIdx.x * kBlock + thread = packet column.  A lane
// keeps kTileRows packets (one per row of the group, same columns) in flight; no division per packet except the
// 32-bit col / bc (a shift when bc is a power of two), the row tile is workgroup-uniform.  (The first form -- a flat
// packet index split with three 64-bit divisions per packet -- was VALU-bound: 0.47 of the HBM roofline.)
constexpr int kTileRows = 4;
template <int DT, bool PROMOTE>
__global__ __launch_bounds__(kBlock) void fp8_pack_tile_kernel(const void* __restrict__ x,
                                                               const void* __restrict__ scales,
                                                               uint8_t* __restrict__ out, int64_t rows,
                                                               int64_t cols, int br, int bc, int bc_shift,
                                                               int64_t tiles_per_row) {
  constexpr int V = Elem<DT>::kVec;
  const uint32_t col = (blockIdx.x * kBlock + threadIdx.x) * V;
  if (col >= cols) return;
  const uint32_t tc = bc_shift >= 0 ? col >> bc_shift : col / (uint32_t)bc;
  auto ld_scale = [&](int64_t t) { return PROMOTE ? reinterpret_cast<const float*>(scales)[t] : load1<DT>(scales, t); };
  for (int64_t r0 = (int64_t)blockIdx.y * kTileRows; r0 < rows; r0 += (int64_t)gridDim.y * kTileRows) {
    auto body = [&](auto FULL) {
      constexpr bool full = decltype(FULL)::value;
      Pack16 in[kTileRows];
      float sc[kTileRows];
#pragma unroll
      for (int k = 0; k < kTileRows; ++k) {
        if (full || r0 + k < rows) {
          in[k] = load16_nt(reinterpret_cast<const char*>(x) + ((r0 + k) * cols + col) * (16 / V));
          sc[k] = ld_scale((int64_t)((uint32_t)(r0 + k) / (uint32_t)br) * tiles_per_row + tc);  // rows < 2^31 (host-checked)
        }
      }
#pragma unroll
      for (int k = 0; k < kTileRows; ++k) {
        if (!full && r0 + k >= rows) continue;
        uint32_t w0, w1;  // one tile scale per packet (fp8_bytes_of_packet)
        fp8_bytes_of_packet<DT, !PROMOTE>(in[k], sc[k], make_shared_div(sc[k]), w0, w1);
        uint8_t* o = out + (r0 + k) * cols + col;
        if constexpr (V == 8) q_store8_nt(o, w0, w1);
        else __builtin_nontemporal_store(w0, reinterpret_cast<uint32_t*>(o));
      }
    };
    if (r0 + kTileRows <= rows) body(std::true_type{});
    else body(std::false_type{});
  }
}

template <int DT>
__global__ __launch_bounds__(kBlock) void fp8_unpack_tile_kernel(const uint8_t* __restrict__ q,
                                                                 const void* __restrict__ scales,
                                                                 void* __restrict__ out, int64_t rows,
                                                                 int64_t cols, int br, int bc, int bc_shift,
                                                                 int64_t tiles_per_row) {
  constexpr int V = Elem<DT>::kVec;
  const uint32_t col = (blockIdx.x * kBlock + threadIdx.x) * V;
  if (col >= cols) return;
  const uint32_t tc = bc_shift >= 0 ? col >> bc_shift : col / (uint32_t)bc;
  for (int64_t r0 = (int64_t)blockIdx.y * kTileRows; r0 < rows; r0 += (int64_t)gridDim.y * kTileRows) {
    auto body = [&](auto FULL) {
      constexpr bool full = decltype(FULL)::value;
      uint32_t in[kTileRows][2];
      float sc[kTileRows];
#pragma unroll
      for (int k = 0; k < kTileRows; ++k) {
        if (full || r0 + k < rows) {
          const uint8_t* src = q + (r0 + k) * cols + col;
          if constexpr (V == 8) {
            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
            const u32x2 t = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(src));
            in[k][0] = t.x;
            in[k][1] = t.y;
          } else {
            in[k][0] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(src));
            in[k][1] = 0;
          }
          sc[k] = load1<DT>(scales, (int64_t)((uint32_t)(r0 + k) / (uint32_t)br) * tiles_per_row + tc);
        }
      }
#pragma unroll
      for (int k = 0; k < kTileRows; ++k) {
        if (!full && r0 + k >= rows) continue;
        float v[8];
#pragma unroll
        for (int i = 0; i < V; ++i) {
          const int word = (int)in[k][i / 4];
          float f;
          switch (i & 3) {
            case 0: f = __builtin_amdgcn_cvt_f32_fp8(word, 0); break;
            case 1: f = __builtin_amdgcn_cvt_f32_fp8(word, 1); break;
            case 2: f = __builtin_amdgcn_cvt_f32_fp8(word, 2); break;
            default: f = __builtin_amdgcn_cvt_f32_fp8(word, 3); break;
          }
          v[i] = f * sc[k];
        }
        store16_nt(reinterpret_cast<char*>(out) + ((r0 + k) * cols + col) * (16 / V), pack<DT>(v));
      }
    };
    if (r0 + kTileRows <= rows) body(std::true_type{});
    else body(std::false_type{});
  }
}


// element-wise forms of the two tile kernels for tiles narrower than a packet or unaligned tensors (the reference's
// own literal test vectors use 2 x 2 tiles, test_qtensor_cuda.py:190-236)
template <int DT, bool PROMOTE>
__global__ __launch_bounds__(kBlock) void fp8_pack_tile_elem_kernel(const void* __restrict__ x, const void* __restrict__ scales,
                                                                    uint8_t* __restrict__ out, int64_t n, int64_t cols,
                                                                    int br, int bc, int64_t tiles_per_row) {
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n; e += (int64_t)gridDim.x * kBlock) {
    const int64_t row = e / cols, col = e - row * cols;
    const int64_t t = (row / br) * tiles_per_row + col / bc;
    const float sc = PROMOTE ? reinterpret_cast<const float*>(scales)[t] : load1<DT>(scales, t);
    float q = load1<DT>(x, e) / sc;
    if constexpr (!PROMOTE) {
      q = round_to_dtype<DT>(q);
      if constexpr (DT == MOQ_BF16) q = (q != q) ? __uint_as_float(0x7FC00000u) : q;  // (see fp8_bytes_of_packet)
    }
    out[e] = (uint8_t)(e4m3fn_bytes2(q, 0.0f) & 0xFFu);
  }
}
template <int DT>
__global__ __launch_bounds__(kBlock) void fp8_unpack_tile_elem_kernel(const uint8_t* __restrict__ q, const void* __restrict__ scales,
                                                                      void* __restrict__ out, int64_t n, int64_t cols,
                                                                      int br, int bc, int64_t tiles_per_row) {
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n; e += (int64_t)gridDim.x * kBlock) {
    const int64_t row = e / cols, col = e - row * cols;
    const float sc = load1<DT>(scales, (row / br) * tiles_per_row + col / bc);
    store1<DT>(out, e, __builtin_amdgcn_cvt_f32_fp8((int)q[e], 0) * sc);
  }
}

}  // namespace moq

using namespace moq;
static inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }

static int fp8_args_ok(const char* who, const void* a, const void* b, const void* c, int64_t n, int amax_mode,
                       int64_t axis_size, int64_t inner) {
  if (n < 0 || (n > 0 && (a == nullptr || b == nullptr || c == nullptr))) {
    set_error("%s: null pointer or negative size", who);
    return MOQ_ERR_INVALID;
  }
  if (amax_mode != MOQ_AMAX_SCALAR && amax_mode != MOQ_AMAX_AXIS) {
    set_error("%s: unknown amax_mode %d", who, amax_mode);
    return MOQ_ERR_INVALID;
  }
  if (amax_mode == MOQ_AMAX_AXIS && (axis_size <= 0 || inner <= 0)) {
    set_error("%s: axis mode needs axis_size > 0 and inner > 0", who);
    return MOQ_ERR_INVALID;
  }
  return MOQ_OK;
}

extern "C" int moq_fp8_pack(const void* x, const void* scales, int scale_dt, uint8_t* out, int64_t n, int dt,
                            int amax_mode, int64_t axis_size, int64_t inner, void* stream) {
  if (scale_dt != MOQ_F32 && scale_dt != dt) {
    set_error("moq_fp8_pack: scale_dt must be MOQ_F32 or the tensor dtype");
    return MOQ_ERR_INVALID;
  }
  const bool sf32 = scale_dt == MOQ_F32 && dt != MOQ_F32;
  const int rc = fp8_args_ok("moq_fp8_pack", x, scales, out, n, amax_mode, axis_size, inner);
  if (rc != MOQ_OK || n == 0) return rc;
  const int vec = dt == MOQ_F32 ? 4 : 8;
  if ((reinterpret_cast<uintptr_t>(x) & 15u) != 0 || (reinterpret_cast<uintptr_t>(out) & 7u) != 0 || n % vec != 0 ||
      (amax_mode == MOQ_AMAX_AXIS && (inner % vec != 0 || inner >= (1LL << 31)))) {
    set_error("moq_fp8_pack: needs 16-byte aligned x, 8-byte aligned out, n and inner multiples of %d", vec);
    return MOQ_ERR_UNSUPPORTED;
  }
  const int grid = copy_grid((n + MOQ_MT_CHUNK - 1) / MOQ_MT_CHUNK);
  if (amax_mode == MOQ_AMAX_AXIS) {
    const int wrap = n > axis_size * inner ? 1 : 0;
    if (sf32) {
      MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((fp8_pack_kernel<DT, true, true>), dim3(grid), dim3(kBlock), copy_lds_1t(), S(stream),
                                                x, scales, out, n, axis_size, inner, log2_or_neg(inner), wrap));
    } else {
      MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((fp8_pack_kernel<DT, true, false>), dim3(grid), dim3(kBlock), copy_lds_1t(), S(stream),
                                                x, scales, out, n, axis_size, inner, log2_or_neg(inner), wrap));
    }
  } else if (sf32) {
    MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((fp8_pack_kernel<DT, false, true>), dim3(grid), dim3(kBlock), copy_lds_1t(), S(stream), x,
                                              scales, out, n, (int64_t)1, (int64_t)1, 0, 0));
  } else {
    MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((fp8_pack_kernel<DT, false, false>), dim3(grid), dim3(kBlock), copy_lds_1t(), S(stream), x,
                                              scales, out, n, (int64_t)1, (int64_t)1, 0, 0));
  }
  return check_launch("moq_fp8_pack");
}

// ---------------------------------------------------------------- INT8 weight pack (export)
// to_quantized_weight for INT8 SmoothQuant / weight-only (export/quant_utils.py:868-869):
//     (weight / weights_scaling_factor[:, None]).round().clamp(-128, 127).to(int8)
// with an fp32 scaling factor per output channel: the quotient is fp32 (a dimensioned fp32 operand promotes), rounded
// half to even, clamped.  One 16-byte packet never leaves its row (cols % V == 0), so the row's scale is one shared
// exact division per packet.
namespace moq {
// Chunk skeleton (moq_chunk.h): every packet of a chunk in flight, one 64-bit division per chunk for the row of its first
// element, a shift / 32-bit division per packet.  (The first form -- one packet per lane and grid step, row = e / cols in
// 64 bits per packet -- ran at 0.62 of 8 TB/s.)  Per packet, from its abs-max pattern:
//   in window : every element finite and <= 2^16, the scale inside SharedDiv's window: the shared division is exact, no NaN
//               exists, rint(clamp(q)) == clamp(rint(q)) for integer bounds, and the rounding itself is the fp32 addition
//               of 1.5 * 2^23 (round-to-nearest-even at integer spacing) whose low mantissa byte IS the two's-complement
//               int8 -- no float-to-int conversion, no masking; v_perm_b32 gathers the bytes.  A zero quotient's sign is
//               irrelevant to an integer.
//   otherwise : IEEE division, rint, clamp, NaN -> 0 (torch's cast)
template <int DT>
__global__ __launch_bounds__(kBlock) void int8_pack_rows_kernel(const void* __restrict__ w,
                                                                const float* __restrict__ scale,
                                                                int8_t* __restrict__ out, int64_t n, int64_t cols,
                                                                int cols_shift) {
  constexpr int V = Elem<DT>::kVec;
  constexpr int P = Chunk<DT>::kPackets;
  const int64_t n_chunks = (n + MOQ_MT_CHUNK - 1) / MOQ_MT_CHUNK;
  GroupIndex gi;
  gi.g = (uint32_t)cols;
  gi.shift = cols_shift;
  for (int64_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    const int64_t e0 = c * MOQ_MT_CHUNK;
    gi.seek(e0);
    auto body = [&](auto FULL) {  // (see fp8_pack_kernel)
      constexpr bool full = decltype(FULL)::value;
      Pack16 in[P];
      float sc[P];
#pragma unroll
      for (int u = 0; u < P; ++u) {
        const int64_t e = e0 + packet_off<DT>(u);
        sc[u] = 1.0f;
        if (full || e < n) {
          in[u] = load16_nt(reinterpret_cast<const char*>(w) + e * (16 / V));
          sc[u] = scale[gi.at((uint32_t)packet_off<DT>(u))];
        }
      }
      if constexpr (full) __builtin_amdgcn_sched_barrier(0);  // every load of the chunk is issued before the first use
#pragma unroll
      for (int u = 0; u < P; ++u) {
        const int64_t e = e0 + packet_off<DT>(u);
        if (!full && e >= n) continue;
        float v[8];
        unpack<DT>(in[u], v);
        const SharedDiv sd = make_shared_div(sc[u]);
        uint32_t b[2] = {0, 0};
        if (sd.fast && pack_absmax<DT>(in[u]) <= 0x47800000u) {
#pragma unroll
          for (int i = 0; i < V; i += 4) {
            uint32_t m[4];
#pragma unroll
            for (int j = 0; j < 4; j += 2) {  // two quotients per packed fp32 instruction (v_pk_mul / v_pk_fma)
              typedef float f32x2 __attribute__((ext_vector_type(2)));
              const f32x2 n2 = {v[i + j], v[i + j + 1]}, y2 = {sd.y, sd.y}, d2 = {-sd.d, -sd.d};
              const f32x2 q0 = n2 * y2;
              const f32x2 q1 = __builtin_elementwise_fma(__builtin_elementwise_fma(d2, q0, n2), y2, q0);
              const f32x2 q = __builtin_elementwise_fma(__builtin_elementwise_fma(d2, q1, n2), y2, q1);
              m[j] = __float_as_uint(__builtin_amdgcn_fmed3f(q.x, -128.0f, 127.0f) + 12582912.0f);
              m[j + 1] = __float_as_uint(__builtin_amdgcn_fmed3f(q.y, -128.0f, 127.0f) + 12582912.0f);
            }
            const uint32_t lo = __builtin_amdgcn_perm(m[1], m[0], 0x0c0c0400u);  // byte 0 of m0, byte 0 of m1
            const uint32_t hi = __builtin_amdgcn_perm(m[3], m[2], 0x04000c0cu);  // the same, placed in bytes 2 and 3
            b[i / 4] = lo | hi;
          }
        } else {
#pragma unroll
          for (int i = 0; i < V; ++i) {
            const float q = v[i] / sc[u];
            float t = __builtin_rintf(q);
            t = __builtin_fminf(__builtin_fmaxf(t, -128.0f), 127.0f);
            // torch: NaN -> int8 conversion is 0 on the host; clamp keeps NaN, the cast then gives 0
            const int ci = (q != q) ? 0 : (int)t;
            b[i / 4] |= ((uint32_t)ci & 0xFFu) << (8 * (i % 4));
          }
        }
        if constexpr (V == 8) q_store8_nt(reinterpret_cast<uint8_t*>(out) + e, b[0], b[1]);
        else __builtin_nontemporal_store(b[0], reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(out) + e));
      }
    };
    if (e0 + MOQ_MT_CHUNK <= n) body(std::true_type{});
    else body(std::false_type{});
  }
}
}  // namespace moq

extern "C" int moq_int8_pack_rows(const void* w, const float* scale, int8_t* out, int64_t rows, int64_t cols, int dt,
                                  void* stream) {
  if (rows < 0 || cols <= 0 || (rows > 0 && (w == nullptr || scale == nullptr || out == nullptr))) {
    set_error("moq_int8_pack_rows: bad arguments");
    return MOQ_ERR_INVALID;
  }
  if (rows == 0) return MOQ_OK;
  const int vec = dt == MOQ_F32 ? 4 : 8;
  if (cols % vec != 0 || (reinterpret_cast<uintptr_t>(w) & 15u) != 0 || (reinterpret_cast<uintptr_t>(out) & 7u) != 0) {
    set_error("moq_int8_pack_rows: needs cols %% %d == 0, a 16-byte aligned weight and an 8-byte aligned output", vec);
    return MOQ_ERR_UNSUPPORTED;
  }
  const int64_t n = rows * cols;
  if (cols > 0x7FFFFFFFLL - MOQ_MT_CHUNK) {
    set_error("moq_int8_pack_rows: cols must be below 2^31 - %d", (int)MOQ_MT_CHUNK);
    return MOQ_ERR_UNSUPPORTED;
  }
  const int grid = copy_grid((n + MOQ_MT_CHUNK - 1) / MOQ_MT_CHUNK);
  MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((int8_pack_rows_kernel<DT>), dim3(grid), dim3(kBlock), copy_lds_1t(32 * 1024), S(stream), w, scale,
                                            out, n, cols, log2_or_neg(cols)));
  return check_launch("moq_int8_pack_rows");
}

extern "C" int moq_fp8_unpack(const uint8_t* q, const void* scales, void* out, int64_t n, int dt, int amax_mode,
                              int64_t axis_size, int64_t inner, void* stream) {
  const int rc = fp8_args_ok("moq_fp8_unpack", q, scales, out, n, amax_mode, axis_size, inner);
  if (rc != MOQ_OK || n == 0) return rc;
  const int vec = dt == MOQ_F32 ? 4 : 8;
  if ((reinterpret_cast<uintptr_t>(out) & 15u) != 0 || (reinterpret_cast<uintptr_t>(q) & 7u) != 0 || n % vec != 0 ||
      (amax_mode == MOQ_AMAX_AXIS && (inner % vec != 0 || inner >= (1LL << 31)))) {
    set_error("moq_fp8_unpack: needs 16-byte aligned out, 8-byte aligned q, n and inner multiples of %d", vec);
    return MOQ_ERR_UNSUPPORTED;
  }
  const int grid = copy_grid((n + MOQ_MT_CHUNK - 1) / MOQ_MT_CHUNK);
  if (amax_mode == MOQ_AMAX_AXIS) {
    const int wrap = n > axis_size * inner ? 1 : 0;
    MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((fp8_unpack_kernel<DT, true>), dim3(grid), dim3(kBlock), copy_lds_1t(), S(stream), q,
                                              scales, out, n, axis_size, inner, log2_or_neg(inner), wrap));
  } else {
    MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((fp8_unpack_kernel<DT, false>), dim3(grid), dim3(kBlock), copy_lds_1t(), S(stream), q,
                                              scales, out, n, (int64_t)1, (int64_t)1, 0, 0));
  }
  return check_launch("moq_fp8_unpack");
}

extern "C" int moq_mt_fold_mxfp4_pack(const moq_seg* segs, const int64_t* blk_start, const moq_fold_seg* side, int n_seg,
                                      int64_t n_chunks, int block, int dt, void* stream) {
  if (n_seg < 0 || n_chunks < 0 || (n_seg > 0 && (segs == nullptr || blk_start == nullptr || side == nullptr))) {
    set_error("moq_mt_fold_mxfp4_pack: null pointer or negative size");
    return MOQ_ERR_INVALID;
  }
  const int vec = dt == MOQ_F32 ? 4 : 8;
  const int lpg = block / vec;
  if (block <= 0 || block % vec != 0 || lpg > 64 || (lpg & (lpg - 1)) != 0) {
    set_error("moq_mt_fold_mxfp4_pack: block sizes %d..%d (powers of two) are supported, got %d", vec, 64 * vec, block);
    return MOQ_ERR_UNSUPPORTED;
  }
  if (n_seg == 0 || n_chunks == 0) return MOQ_OK;
  const int grid = copy_grid(n_chunks);
  const int bs = log2_or_neg(block);
#define MOQ_MTFP_CASE(L) \
  case L: MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((mt_fold_mxfp4_pack_kernel<DT, L>), dim3(grid), dim3(kBlock), copy_lds_1t(), S(stream), segs, blk_start, side, n_seg, n_chunks, bs)); break;
  switch (lpg) {
    MOQ_MTFP_CASE(1) MOQ_MTFP_CASE(2) MOQ_MTFP_CASE(4) MOQ_MTFP_CASE(8) MOQ_MTFP_CASE(16) MOQ_MTFP_CASE(32) MOQ_MTFP_CASE(64)
    default: set_error("unreachable"); return MOQ_ERR_INVALID;
  }
#undef MOQ_MTFP_CASE
  return check_launch("moq_mt_fold_mxfp4_pack");
}

extern "C" int moq_mxfp4_pack(const void* x, uint8_t* packed, uint8_t* e8m0, int64_t n_blocks, int block, int dt,
                              void* stream) {
  if (n_blocks < 0 || block <= 0 || block % 2 != 0 ||
      (n_blocks > 0 && (x == nullptr || packed == nullptr || e8m0 == nullptr))) {
    set_error("moq_mxfp4_pack: bad arguments (block must be even)");
    return MOQ_ERR_INVALID;
  }
  if (n_blocks == 0) return MOQ_OK;
  const int64_t n = n_blocks * block;
  const int vec = dt == MOQ_F32 ? 4 : 8;
  const int lpg = block / vec;
  const bool fast = (reinterpret_cast<uintptr_t>(x) & 15u) == 0 && (reinterpret_cast<uintptr_t>(packed) & 3u) == 0 &&
                    block % vec == 0 && lpg >= 1 && lpg <= 64 && (lpg & (lpg - 1)) == 0;
  if (fast) {
    const int grid = copy_grid((n + MOQ_MT_CHUNK - 1) / MOQ_MT_CHUNK);
    const int bs = log2_or_neg(block);
#define MOQ_MXP_CASE(L) \
  case L: MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((mxfp4_pack_kernel<DT, L>), dim3(grid), dim3(kBlock), copy_lds_1t(), S(stream), x, packed, e8m0, n, bs)); break;
    switch (lpg) {
      MOQ_MXP_CASE(1) MOQ_MXP_CASE(2) MOQ_MXP_CASE(4) MOQ_MXP_CASE(8) MOQ_MXP_CASE(16) MOQ_MXP_CASE(32) MOQ_MXP_CASE(64)
      default: set_error("unreachable"); return MOQ_ERR_INVALID;
    }
#undef MOQ_MXP_CASE
  } else {
    MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((mxfp4_generic_pack_kernel<DT>), dim3(stream_grid(kBlock, n_blocks)),
                                              dim3(kBlock), 0, S(stream), x, packed, e8m0, n_blocks, block));
  }
  return check_launch("moq_mxfp4_pack");
}

extern "C" int moq_mxfp4_unpack(const uint8_t* packed, const uint8_t* e8m0, void* out, int64_t n_blocks, int block,
                                int dt, void* stream) {
  if (n_blocks < 0 || block <= 0 || block % 2 != 0 ||
      (n_blocks > 0 && (packed == nullptr || e8m0 == nullptr || out == nullptr))) {
    set_error("moq_mxfp4_unpack: bad arguments (block must be even)");
    return MOQ_ERR_INVALID;
  }
  if (n_blocks == 0) return MOQ_OK;
  const int64_t n = n_blocks * block;
  const int bs = log2_or_neg(block);
  const bool fast = dt != MOQ_F32 && bs >= 3 && (reinterpret_cast<uintptr_t>(packed) & 3u) == 0 &&
                    (reinterpret_cast<uintptr_t>(out) & 15u) == 0;
  if (fast) {
    const int grid = copy_grid((n + MOQ_MT_CHUNK - 1) / MOQ_MT_CHUNK);
    MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((mxfp4_unpack_kernel<DT, true>), dim3(grid), dim3(kBlock), copy_lds_1t(), S(stream),
                                              packed, e8m0, out, n, block, bs));
  } else {
    MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((mxfp4_unpack_kernel<DT, false>), dim3(stream_grid(kBlock, n)),
                                              dim3(kBlock), 0, S(stream), packed, e8m0, out, n, block, bs));
  }
  return check_launch("moq_mxfp4_unpack");
}

// x: packet columns of a row in workgroups of kBlock lanes; y: groups of kTileRows rows (grid-strided past 65535)
static dim3 tile_grid(int64_t rows, int64_t cols, int vec) {
  const int64_t gx = (cols / vec + kBlock - 1) / kBlock;
  int64_t gy = (rows + kTileRows - 1) / kTileRows;
  if (gy > 65535) gy = 65535;
  return dim3((unsigned)gx, (unsigned)gy);
}

static int fp8_tile_args_ok(const char* who, const void* a, const void* b, const void* c, int64_t rows, int64_t cols,
                            int br, int bc, int dt) {
  if (rows < 0 || cols < 0 || br <= 0 || bc <= 0 || (rows * cols > 0 && (a == nullptr || b == nullptr || c == nullptr))) {
    set_error("%s: bad arguments", who);
    return MOQ_ERR_INVALID;
  }
  (void)dt;
  if (rows % br != 0 || cols % bc != 0) {
    set_error("%s: needs rows %% br == 0 and cols %% bc == 0 (pad on the host like reduce_block_padding)", who);
    return MOQ_ERR_UNSUPPORTED;
  }
  return MOQ_OK;
}

extern "C" int moq_fp8_pack_tile(const void* x, const void* scales, int scale_dt, uint8_t* out, int64_t rows,
                                 int64_t cols, int br, int bc, int dt, void* stream) {
  if (scale_dt != MOQ_F32 && scale_dt != dt) {
    set_error("moq_fp8_pack_tile: scale_dt must be MOQ_F32 or the tensor dtype");
    return MOQ_ERR_INVALID;
  }
  const int rc = fp8_tile_args_ok("moq_fp8_pack_tile", x, scales, out, rows, cols, br, bc, dt);
  if (rc != MOQ_OK || rows * cols == 0) return rc;
  const int vec = dt == MOQ_F32 ? 4 : 8;
  const bool promote = scale_dt == MOQ_F32 && dt != MOQ_F32;
  if ((reinterpret_cast<uintptr_t>(x) & 15u) != 0 || (reinterpret_cast<uintptr_t>(out) & 7u) != 0 || bc % vec != 0) {
    const int64_t n = rows * cols;
    if (promote) {
      MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((fp8_pack_tile_elem_kernel<DT, true>), dim3(stream_grid(kBlock, n)), dim3(kBlock),
                                                0, S(stream), x, scales, out, n, cols, br, bc, cols / bc));
    } else {
      MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((fp8_pack_tile_elem_kernel<DT, false>), dim3(stream_grid(kBlock, n)), dim3(kBlock),
                                                0, S(stream), x, scales, out, n, cols, br, bc, cols / bc));
    }
    return check_launch("moq_fp8_pack_tile");
  }
  if (cols / vec > 0x7FFFFFFF / kBlock || rows > 0x7FFFFFFF) {
    set_error("moq_fp8_pack_tile: more than 2^31 rows or packets per row are not supported");
    return MOQ_ERR_UNSUPPORTED;
  }
  const dim3 grid = tile_grid(rows, cols, vec);
  if (promote) {
    MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((fp8_pack_tile_kernel<DT, true>), grid, dim3(kBlock), 0, S(stream), x,
                                              scales, out, rows, cols, br, bc, log2_or_neg(bc), cols / bc));
  } else {
    MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((fp8_pack_tile_kernel<DT, false>), grid, dim3(kBlock), 0, S(stream), x,
                                              scales, out, rows, cols, br, bc, log2_or_neg(bc), cols / bc));
  }
  return check_launch("moq_fp8_pack_tile");
}

extern "C" int moq_fp8_unpack_tile(const uint8_t* q, const void* scales, void* out, int64_t rows, int64_t cols, int br,
                                   int bc, int dt, void* stream) {
  const int rc = fp8_tile_args_ok("moq_fp8_unpack_tile", q, scales, out, rows, cols, br, bc, dt);
  if (rc != MOQ_OK || rows * cols == 0) return rc;
  const int vec = dt == MOQ_F32 ? 4 : 8;
  if ((reinterpret_cast<uintptr_t>(out) & 15u) != 0 || (reinterpret_cast<uintptr_t>(q) & 7u) != 0 || bc % vec != 0) {
    const int64_t n = rows * cols;
    MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((fp8_unpack_tile_elem_kernel<DT>), dim3(stream_grid(kBlock, n)), dim3(kBlock), 0,
                                              S(stream), q, scales, out, n, cols, br, bc, cols / bc));
    return check_launch("moq_fp8_unpack_tile");
  }
  if (cols / vec > 0x7FFFFFFF / kBlock || rows > 0x7FFFFFFF) {
    set_error("moq_fp8_unpack_tile: more than 2^31 rows or packets per row are not supported");
    return MOQ_ERR_UNSUPPORTED;
  }
  MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((fp8_unpack_tile_kernel<DT>), tile_grid(rows, cols, vec), dim3(kBlock), 0,
                                            S(stream), q, scales, out, rows, cols, br, bc, log2_or_neg(bc), cols / bc));
  return check_launch("moq_fp8_unpack_tile");
}
