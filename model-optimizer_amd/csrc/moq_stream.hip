// moq_stream.hip -- the HBM-streaming kernels of the PTQ hot path:
//   per-tensor abs-max, INT-k / FP8-E4M3 quantize-dequantize with scalar amax, fused per-group
//   abs-max + INT-k QDQ, and their multi-tensor (segment table) forms.
//
// One skeleton serves all of them.  A *chunk* is MOQ_MT_CHUNK = 8192 consecutive elements of one tensor;
// a 256-thread workgroup owns a chunk at a time and moves it with 16-byte lane accesses: packet u of
// thread t covers elements (u*256 + t)*kVec ... so each wave instruction touches one contiguous KiB.
// All loads of a chunk are issued before the first use (4 x 16 B per lane in flight for bf16, 8 for f32),
// which with 8 resident workgroups per CU keeps ~128 KiB per CU outstanding -- well past the ~25 KiB/CU
// that 6.3 TB/s x ~1 us of loaded HBM latency needs (MI355X_MICROARCH.md, HBM / per-instruction table).
//
// Roofline: every kernel here is HBM-bound.  Algorithmic bytes per element (bf16): amax 2; QDQ 4;
// fused group amax+QDQ 4 + 4/g.  VALU cost per element (~30 lane-ops incl. the IEEE divide) stays under
// the ~50 lane-ops/element a CU can issue at the HBM rate, so the divide is kept exact, not approximated.
#include <stdlib.h>

#include "moq_common.h"
#include "moq_chunk.h"
#include "moq_ops.h"

namespace moq {

// apply an elementwise operator to one chunk of one tensor
template <int DT, bool FAST, class Op>
__device__ __forceinline__ void chunk_apply(const void* x, void* y, int64_t e0, int64_t n, const Op& op) {
  constexpr int V = Elem<DT>::kVec;
  constexpr int P = Chunk<DT>::kPackets;
  Pack16 in[P];
#pragma unroll
  for (int u = 0; u < P; ++u) in[u] = ld_packet<DT, FAST>(x, e0 + packet_off<DT>(u), n);
#pragma unroll
  for (int u = 0; u < P; ++u) {
    float f[8];
    unpack<DT>(in[u], f);
    op(f, V);
    st_packet<DT, FAST>(y, e0 + packet_off<DT>(u), n, pack<DT>(f));
  }
}

// abs-max pattern of one chunk (per thread partial)
template <int DT, bool FAST, bool NT = true>
__device__ __forceinline__ uint32_t chunk_absmax(const void* x, int64_t e0, int64_t n, uint32_t acc) {
  constexpr int P = Chunk<DT>::kPackets;
  Pack16 in[P];
#pragma unroll
  for (int u = 0; u < P; ++u) in[u] = ld_packet<DT, FAST, NT>(x, e0 + packet_off<DT>(u), n);
#pragma unroll
  for (int u = 0; u < P; ++u) {
    uint32_t m = pack_absmax<DT>(in[u]);
    acc = m > acc ? m : acc;
  }
  return acc;
}

// ------------------------------------------------------------------------------------------------
// single-tensor kernels
// ------------------------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(kBlock) void amax_kernel(const void* __restrict__ x, int64_t n,
                                                      uint32_t* __restrict__ out_bits) {
  __shared__ uint32_t smem[kBlock / 64];
  const int64_t n_chunks = (n + MOQ_MT_CHUNK - 1) / MOQ_MT_CHUNK;
  const bool al = aligned16(x);
  // The value the running maximum had when this workgroup started: a maximum only grows, so a workgroup whose own
  // result does not exceed it has nothing to contribute and skips its atomic.  In a calibration loop (one running
  // abs-max per quantizer over many batches) that is nearly every workgroup of nearly every launch after the first
  // batches -- and the atomics matter at the sizes the flow presents: they all hit ONE address and retire at ~80 M/s
  // (MI355X_MICROARCH.md), 512 of them are a 6 us tail behind a 33 MB sweep that takes 6 us to stream.
  uint32_t seen = 0;
  if (threadIdx.x == 0) seen = __builtin_nontemporal_load(out_bits);
  uint32_t acc = 0;
  for (int64_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    const int64_t e0 = c * MOQ_MT_CHUNK;
    if (al && e0 + MOQ_MT_CHUNK <= n)
      acc = chunk_absmax<DT, true>(x, e0, n, acc);
    else
      acc = chunk_absmax<DT, false>(x, e0, n, acc);
  }
  acc = block_max_u32(acc, smem);
  if (threadIdx.x == 0 && acc > seen) atomicMax(out_bits, acc);  // non-negative float patterns order like uints
}

template <int DT, class Op>
__global__ __launch_bounds__(kBlock) void map_scalar_amax_kernel(const void* __restrict__ x,
                                                                 void* __restrict__ y, int64_t n,
                                                                 const float* __restrict__ amax,
                                                                 int num_bits, int is_unsigned,
                                                                 int narrow) {
  Op op;
  if constexpr (__is_same(Op, OpIntQdq)) {
    op.q = make_intq(num_bits, is_unsigned, narrow);
    op.set(amax[0]);
  } else if constexpr (__is_same(Op, OpFp8Qdq)) {
    op.sc = fp8_scale(amax[0]);
  }
  const int64_t n_chunks = (n + MOQ_MT_CHUNK - 1) / MOQ_MT_CHUNK;
  const bool al = aligned16(x) && aligned16(y);
  for (int64_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    const int64_t e0 = c * MOQ_MT_CHUNK;
    if (al && e0 + MOQ_MT_CHUNK <= n)
      chunk_apply<DT, true>(x, y, e0, n, op);
    else
      chunk_apply<DT, false>(x, y, e0, n, op);
  }
}

// QDQ with a per-axis amax: amax index = (i / inner) % axis_size.  Generic (any inner); the per-group
// case (inner = g) has the dedicated fused kernel below, per-channel rows (inner = Cin) take the
// "uniform per packet" fast path when inner % kVec == 0.
template <int DT, bool FP8>
__global__ __launch_bounds__(kBlock) void map_axis_amax_kernel(const void* __restrict__ x,
                                                               void* __restrict__ y, int64_t n,
                                                               const float* __restrict__ amax,
                                                               int64_t axis_size, int64_t inner,
                                                               int inner_shift, int wrap, int num_bits,
                                                               int is_unsigned, int narrow) {
  constexpr int V = Elem<DT>::kVec;
  constexpr int P = Chunk<DT>::kPackets;
  const IntQ q = make_intq(num_bits, is_unsigned, narrow);
  const int64_t n_chunks = (n + MOQ_MT_CHUNK - 1) / MOQ_MT_CHUNK;
  const bool al = aligned16(x) && aligned16(y);
  // a 16-byte packet never straddles two amax entries; the row index then costs one 64-bit division per chunk
  // (uniform) and a 32-bit shift / division per packet instead of a 64-bit div + mod per packet
  const bool uniform = (inner % V) == 0 && inner < (1LL << 31);
  GroupIndex gi;
  gi.g = (uint32_t)inner;
  gi.shift = inner_shift;
  for (int64_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    const int64_t e0 = c * MOQ_MT_CHUNK;
    const bool fast = al && e0 + MOQ_MT_CHUNK <= n;
    if (uniform) gi.seek(e0);
    Pack16 in[P];
    float am[P];
#pragma unroll
    for (int u = 0; u < P; ++u) {
      const int64_t e = e0 + packet_off<DT>(u);
      in[u] = fast ? ld_packet<DT, true>(x, e, n) : ld_packet<DT, false>(x, e, n);
      if (uniform) {
        int64_t row = gi.at((uint32_t)packet_off<DT>(u));
        if (wrap) row %= axis_size;
        am[u] = e < n ? amax[row] : 1.0f;
      }
    }
#pragma unroll
    for (int u = 0; u < P; ++u) {
      const int64_t e = e0 + packet_off<DT>(u);
      float f[8];
      unpack<DT>(in[u], f);
      if (uniform) {
        if constexpr (FP8) {
          OpFp8Qdq op;
          op.sc = fp8_scale(am[u]);
          op(f, V);
        } else {
          OpIntQdq op;
          op.q = q;
          op.set(am[u]);
          op(f, V);
        }
      } else {
#pragma unroll
        for (int i = 0; i < V; ++i) {
          const int64_t ei = e + i;
          const float a = ei < n ? amax[(ei / inner) % axis_size] : 1.0f;
          if constexpr (FP8) {
            const Fp8Scale sc = fp8_scale(a);
            float t = f[i] * sc.s;
            float ct = __builtin_fminf(__builtin_fmaxf(t, -448.0f), 448.0f);
            ct = (t != t) ? t : ct;
            float r0, r1;
            e4m3_roundtrip2(ct, 0.0f, r0, r1);
            f[i] = r0 * sc.inv;
          } else {
            f[i] = qdq_int(f[i], int_scale(a, q.hi), q);
          }
        }
      }
      const Pack16 o = pack<DT>(f);
      if (fast)
        st_packet<DT, true>(y, e, n, o);
      else
        st_packet<DT, false>(y, e, n, o);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// fused per-group abs-max + INT-k QDQ (the headline kernel)
// ------------------------------------------------------------------------------------------------
// LPG = lanes per group = g / kVec (power of two, 1..64).  A group's lanes are consecutive lanes of one
// wave for one packet index, so the group max is an LPG-wide butterfly in registers (DPP for <= 16
// lanes); the QDQ then runs from the registers the load filled: one HBM read, one HBM write.
// When QDQ is false only the group amax is produced (static per-group calibration).
// pre-scale: if s != nullptr the element is first multiplied by s[col] and rounded to the storage dtype
// (AWQ search: W * awq_scale in bf16, tensor_quantizer.py:1143-1144).
template <int DT, int LPG, bool QDQ, bool PRESCALE>
__device__ __forceinline__ void group_chunk(const void* __restrict__ x, void* __restrict__ y,
                                            float* __restrict__ amax_out, int64_t e0, const IntQ q,
                                            const void* __restrict__ s, int64_t cols) {
  constexpr int V = Elem<DT>::kVec;
  constexpr int P = Chunk<DT>::kPackets;
  constexpr int G = LPG * V;
  Pack16 in[P];
#pragma unroll
  for (int u = 0; u < P; ++u) in[u] = ld_packet<DT, true>(x, e0 + packet_off<DT>(u), 0);
  Pack16 sc[P];
  if constexpr (PRESCALE) {
#pragma unroll
    for (int u = 0; u < P; ++u) {
      const int64_t e = e0 + packet_off<DT>(u);
      sc[u] = ld_packet<DT, true>(s, e % cols, 0);  // cols % V == 0 so a packet stays inside one row
    }
  }
#pragma unroll
  for (int u = 0; u < P; ++u) {
    const int64_t e = e0 + packet_off<DT>(u);
    float f[8];
    uint32_t m;
    if constexpr (PRESCALE) {
      float sf[8];
      unpack<DT>(in[u], f);
      unpack<DT>(sc[u], sf);
      m = 0;
#pragma unroll
      for (int i = 0; i < V; ++i) {
        f[i] = round_to_dtype<DT>(f[i] * sf[i]);
        const uint32_t a = absbits(f[i]);
        m = a > m ? a : m;
      }
    } else {
      m = pack_absmax<DT>(in[u]);
      if constexpr (QDQ) unpack<DT>(in[u], f);
    }
    m = group_max_u32<LPG>(m);
    const float amax = __uint_as_float(m);
    if (amax_out != nullptr && (threadIdx.x & (LPG - 1)) == 0) amax_out[e / G] = amax;
    if constexpr (QDQ) {
      const float scale = int_scale(amax, q.hi);
      const SharedDiv sd = make_shared_div(scale);
      if (qdq_fast_ok(m, scale, sd)) {  // all but NaN / inf / tiny-amax groups: 5 VALU ops per element less
#pragma unroll
        for (int i = 0; i < V; ++i) f[i] = qdq_int_fast(f[i], scale, sd, q);
      } else {
#pragma unroll
        for (int i = 0; i < V; ++i) f[i] = qdq_int_shared(f[i], scale, sd, q);
      }
      st_packet<DT, true>(y, e, 0, pack<DT>(f));
    }
  }
}

// AWQ Gram search: the fp32 error weight of one candidate scale, E = dt(QDQ(dt(W * s))) * r - W (r = fp32 value of the
// dtype-rounded 1/s the input side uses), straight from one read of W -- plus its split-precision MFMA operand
// A = [E_hi | E_hi | E_lo] (bf16 [rows, 3 cols], E_hi = bf16(E), E_lo = bf16(E - E_hi)) for moq_awq_quadform.
// Replaces awq_scale_qdq + float() + mul + sub + two casts + a concatenation (ten elementwise passes, ~50 B/element)
// by one kernel: 2 B read, 4 + 6 B written per element.
template <int DT, int LPG>
__global__ __launch_bounds__(kBlock) void awq_err_weight_kernel(const void* __restrict__ w,
                                                                const void* __restrict__ s,
                                                                const float* __restrict__ r,
                                                                float* __restrict__ e_out,
                                                                uint16_t* __restrict__ a_out, int64_t n,
                                                                int64_t cols, int cols_shift, int num_bits,
                                                                int planes) {
  constexpr int V = Elem<DT>::kVec;
  constexpr int P = Chunk<DT>::kPackets;
  const IntQ q = make_intq(num_bits, 0, 0);
  const int64_t n_chunks = (n + MOQ_MT_CHUNK - 1) / MOQ_MT_CHUNK;
  GroupIndex gi;
  gi.g = (uint32_t)cols;
  gi.shift = cols_shift;
  for (int64_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    const int64_t e0 = c * MOQ_MT_CHUNK;
    gi.seek(e0);
    Pack16 in[P], sc[P];
    int64_t row[P];
    uint32_t col[P];
#pragma unroll
    for (int u = 0; u < P; ++u) {
      const int64_t e = e0 + packet_off<DT>(u);
      row[u] = gi.at((uint32_t)packet_off<DT>(u));
      col[u] = (uint32_t)(e - row[u] * cols);
      if (e < n) {
        in[u] = load16_nt(reinterpret_cast<const char*>(w) + e * (16 / V));
        sc[u] = load16(reinterpret_cast<const char*>(s) + (int64_t)col[u] * (16 / V));
      } else {
        in[u].w[0] = in[u].w[1] = in[u].w[2] = in[u].w[3] = 0u;
        sc[u] = in[u];
      }
    }
#pragma unroll
    for (int u = 0; u < P; ++u) {
      const int64_t e = e0 + packet_off<DT>(u);
      float w0[8], f[8], sf[8];
      unpack<DT>(in[u], w0);
      unpack<DT>(sc[u], sf);
      uint32_t m = 0;
#pragma unroll
      for (int i = 0; i < V; ++i) {
        f[i] = round_to_dtype<DT>(w0[i] * sf[i]);  // (W * s).to(dtype)
        const uint32_t a = absbits(f[i]);
        m = a > m ? a : m;
      }
      m = group_max_u32<LPG>(m);
      const float scale = int_scale(__uint_as_float(m), q.hi);
      const SharedDiv sd = make_shared_div(scale);
      if (e >= n) continue;
      float err[8], lo[8];
      uint32_t hi_bits[8];
#pragma unroll
      for (int i = 0; i < V; ++i) {
        const float what = round_to_dtype<DT>(qdq_int_shared(f[i], scale, sd, q));
        const float t = what * r[col[u] + i];
        err[i] = t - w0[i];
        const float hi = round_to_dtype<MOQ_BF16>(err[i]);
        hi_bits[i] = __float_as_uint(hi) >> 16;
        lo[i] = err[i] - hi;
      }
      float* ep = e_out + e;
      *reinterpret_cast<float4*>(ep) = make_float4(err[0], err[1], err[2], err[3]);
      if constexpr (V == 8) *reinterpret_cast<float4*>(ep + 4) = make_float4(err[4], err[5], err[6], err[7]);
      // planes 3: [hi | hi | lo], 2: [hi | lo], 1: [hi]
      uint16_t* ap = a_out + row[u] * planes * cols + col[u];
      const Pack16 plo = pack<MOQ_BF16>(lo);
      if constexpr (V == 8) {
        Pack16 phi;
#pragma unroll
        for (int i = 0; i < 4; ++i) phi.w[i] = hi_bits[2 * i] | (hi_bits[2 * i + 1] << 16);
        store16(ap, phi);
        if (planes == 3) store16(ap + cols, phi);
        if (planes >= 2) store16(ap + (planes - 1) * cols, plo);
      } else {
        const uint2 phi = make_uint2(hi_bits[0] | (hi_bits[1] << 16), hi_bits[2] | (hi_bits[3] << 16));
        *reinterpret_cast<uint2*>(ap) = phi;
        if (planes == 3) *reinterpret_cast<uint2*>(ap + cols) = phi;
        if (planes >= 2) *reinterpret_cast<uint2*>(ap + (planes - 1) * cols) = make_uint2(plo.w[0], plo.w[1]);
      }
    }
  }
}

// tail / unaligned groups: one thread per group, scalar (rare: only when n is not a chunk multiple or
// the base pointer is not 16-byte aligned)
template <int DT, bool QDQ, bool PRESCALE>
__device__ __forceinline__ void group_scalar(const void* x, void* y, float* amax_out, int64_t grp,
                                             int g, const IntQ q, const void* s, int64_t cols) {
  const int64_t e0 = grp * (int64_t)g;
  uint32_t m = 0;
  for (int i = 0; i < g; ++i) {
    float v = load1<DT>(x, e0 + i);
    if constexpr (PRESCALE) v = round_to_dtype<DT>(v * load1<DT>(s, (e0 + i) % cols));
    const uint32_t a = absbits(v);
    m = a > m ? a : m;
  }
  const float amax = __uint_as_float(m);
  if (amax_out != nullptr) amax_out[grp] = amax;
  if constexpr (QDQ) {
    const float scale = int_scale(amax, q.hi);
    for (int i = 0; i < g; ++i) {
      float v = load1<DT>(x, e0 + i);
      if constexpr (PRESCALE) v = round_to_dtype<DT>(v * load1<DT>(s, (e0 + i) % cols));
      store1<DT>(y, e0 + i, qdq_int(v, scale, q));
    }
  }
}

template <int DT, int LPG, bool QDQ, bool PRESCALE>
__global__ __launch_bounds__(kBlock) void group_kernel(const void* __restrict__ x, void* __restrict__ y,
                                                       float* __restrict__ amax_out, int64_t n_groups,
                                                       int num_bits, int is_unsigned, int narrow,
                                                       const void* __restrict__ s, int64_t cols) {
  constexpr int G = LPG * Elem<DT>::kVec;
  const IntQ q = make_intq(num_bits, is_unsigned, narrow);
  const int64_t n = n_groups * G;
  const bool al = aligned16(x) && (!QDQ || aligned16(y)) && (!PRESCALE || aligned16(s));
  const int64_t full_chunks = al ? n / MOQ_MT_CHUNK : 0;
  for (int64_t c = blockIdx.x; c < full_chunks; c += gridDim.x)
    group_chunk<DT, LPG, QDQ, PRESCALE>(x, y, amax_out, c * MOQ_MT_CHUNK, q, s, cols);
  // remainder groups (fewer than one chunk unless unaligned)
  const int64_t g0 = full_chunks * (MOQ_MT_CHUNK / G);
  for (int64_t grp = g0 + (int64_t)blockIdx.x * kBlock + threadIdx.x; grp < n_groups;
       grp += (int64_t)gridDim.x * kBlock)
    group_scalar<DT, QDQ, PRESCALE>(x, y, amax_out, grp, G, q, s, cols);
}

template <int DT, bool NT>
__global__ __launch_bounds__(kBlock) void mt_amax_kernel(const moq_seg* __restrict__ segs,
                                                         const int64_t* __restrict__ blk_start,
                                                         int n_seg, int64_t n_chunks) {
  __shared__ uint32_t smem[kBlock / 64];
  const ChunkRange r = block_range(n_chunks);
  if (r.begin >= r.end) return;
  SegCursor cur;
  cur.init(segs, blk_start, n_seg, r.begin);
  uint32_t acc = 0;
  for (int64_t c = r.begin; c < r.end; ++c) {
    if (c >= cur.c_end) {  // crossed into the next tensor: flush the finished one
      acc = block_max_u32(acc, smem);
      if (threadIdx.x == 0) atomicMax(reinterpret_cast<uint32_t*>(cur.sg.amax), acc);
      __syncthreads();
      acc = 0;
      cur.seek(c);
    }
    const int64_t e0 = (c - cur.c_begin) * MOQ_MT_CHUNK;
    if (cur.aligned && e0 + MOQ_MT_CHUNK <= cur.sg.n)
      acc = chunk_absmax<DT, true, NT>(cur.sg.x, e0, cur.sg.n, acc);
    else
      acc = chunk_absmax<DT, false>(cur.sg.x, e0, cur.sg.n, acc);
  }
  acc = block_max_u32(acc, smem);
  if (threadIdx.x == 0) atomicMax(reinterpret_cast<uint32_t*>(cur.sg.amax), acc);
}

// Two-stage form of the per-tensor abs-max: stage 1 sweeps memory as one dense window (one workgroup per chunk, read_grid
// in moq_chunk.h: 0.86 of 8 TB/s; the strided 8-chunks-per-workgroup order of rounds 1-3 reached 0.83 on the same data)
// and stores ONE value per chunk with a plain store; stage 2 folds the per-chunk values of every tensor (one workgroup
// per tensor, 3.4 MB in total for Llama-3-8B).  No same-address atomics at all.
template <int DT>
__global__ __launch_bounds__(kBlock) void mt_amax_chunks_kernel(const moq_seg* __restrict__ segs,
                                                                const int64_t* __restrict__ blk_start,
                                                                int n_seg, int64_t n_chunks,
                                                                uint32_t* __restrict__ chunk_max) {
  __shared__ uint32_t smem[kBlock / 64];
  if ((int64_t)blockIdx.x >= n_chunks) return;
  SegCursor cur;
  cur.init(segs, blk_start, n_seg, blockIdx.x);
  for (int64_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    cur.seek(c);
    const int64_t e0 = (c - cur.c_begin) * MOQ_MT_CHUNK;
    uint32_t acc;
    if (cur.aligned && e0 + MOQ_MT_CHUNK <= cur.sg.n)
      acc = chunk_absmax<DT, true, true>(cur.sg.x, e0, cur.sg.n, 0u);
    else
      acc = chunk_absmax<DT, false>(cur.sg.x, e0, cur.sg.n, 0u);
    acc = block_max_u32(acc, smem);
    if (threadIdx.x == 0) chunk_max[c] = acc;
    __syncthreads();  // smem is reused by the next chunk
  }
}
__global__ __launch_bounds__(kBlock) void mt_amax_fold_kernel(const moq_seg* __restrict__ segs,
                                                              const int64_t* __restrict__ blk_start,
                                                              const uint32_t* __restrict__ chunk_max) {
  __shared__ uint32_t smem[kBlock / 64];
  const int s = blockIdx.x;
  uint32_t acc = 0;
  for (int64_t c = blk_start[s] + threadIdx.x; c < blk_start[s + 1]; c += kBlock) {
    const uint32_t v = chunk_max[c];
    acc = v > acc ? v : acc;
  }
  acc = block_max_u32(acc, smem);
  if (threadIdx.x == 0) segs[s].amax[0] = __uint_as_float(acc);
}

// Stage 2 for RUNNING maxima with several readers per tensor (moq_mt_amax_running): fold f takes the maximum of segment
// folds[f].seg's chunk values and merges it into *folds[f].dst -- the calibrator's running abs-max, compared as bit
// patterns like every abs-max of this library (NaN > inf > finite: a NaN sticks).  A destination may appear in several
// folds of one launch (an input quantizer called twice in a layer), hence the atomic.
// One 1024-thread workgroup per fold, every thread's loads in flight together: a 117 MB down_proj input is 7168 chunk
// values, which ONE wave walking them 64 at a time (the first form) took 15.9 us over -- 40 % of the 38 us sweep it follows
// and the reason the per-layer launch saved only 29 of 137 ms on the FP8 calibration loop of Llama-3-8B
// (profiles/r05g_fp8_flow_kernels.md).
constexpr int kFoldBlock = 1024;
__global__ __launch_bounds__(kFoldBlock) void mt_amax_fold_running_kernel(const int64_t* __restrict__ blk_start,
                                                                          const uint32_t* __restrict__ chunk_max,
                                                                          const moq_amax_fold* __restrict__ folds) {
  __shared__ uint32_t s_max[kFoldBlock / 64];
  const moq_amax_fold f = folds[blockIdx.x];
  const int64_t c0 = blk_start[f.seg], c1 = blk_start[f.seg + 1];
  uint32_t acc = 0;
  for (int64_t c = c0 + threadIdx.x; c < c1; c += 8 * kFoldBlock) {
    uint32_t v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = c + u * kFoldBlock < c1 ? chunk_max[c + u * kFoldBlock] : 0u;
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = v[u] > acc ? v[u] : acc;
  }
  acc = group_max_u32<64>(acc);
  if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t m = s_max[0];
#pragma unroll
    for (int w = 1; w < kFoldBlock / 64; ++w) m = s_max[w] > m ? s_max[w] : m;
    if (m != 0) atomicMax(reinterpret_cast<uint32_t*>(f.dst), m);
  }
}

__global__ void mt_zero_amax_kernel(const moq_seg* __restrict__ segs, int n_seg) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_seg) segs[i].amax[0] = 0.0f;
}

// Copy-shaped passes (read + write) run fastest when the workgroups sweep memory as ONE dense window: the grid is one
// workgroup per chunk (copy_grid, moq_chunk.h -- why), chunk = blockIdx + k * gridDim covers grids capped below the chunk
// count.  Chunks visited by a workgroup grow monotonically, so the segment cursor only ever advances.
template <int DT, class Op>
__global__ __launch_bounds__(kBlock) void mt_map_kernel(const moq_seg* __restrict__ segs,
                                                        const int64_t* __restrict__ blk_start,
                                                        int n_seg, int64_t n_chunks, int num_bits,
                                                        int is_unsigned, int narrow) {
  if ((int64_t)blockIdx.x >= n_chunks) return;
  SegCursor cur;
  cur.init(segs, blk_start, n_seg, blockIdx.x);
  Op op;
  bool fresh = true;
  for (int64_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    fresh |= cur.seek(c);
    if (fresh) {
      fresh = false;
      if constexpr (__is_same(Op, OpIntQdq)) {
        op.q = make_intq(num_bits, is_unsigned, narrow);
        op.set(cur.sg.amax[0]);
      } else {
        op.sc = fp8_scale(cur.sg.amax[0]);
      }
    }
    const int64_t e0 = (c - cur.c_begin) * MOQ_MT_CHUNK;
    if (cur.aligned && e0 + MOQ_MT_CHUNK <= cur.sg.n)
      chunk_apply<DT, true>(cur.sg.x, cur.sg.y, e0, cur.sg.n, op);
    else
      chunk_apply<DT, false>(cur.sg.x, cur.sg.y, e0, cur.sg.n, op);
  }
}

template <int DT, int LPG>
__global__ __launch_bounds__(kBlock) void mt_group_kernel(const moq_seg* __restrict__ segs,
                                                          const int64_t* __restrict__ blk_start,
                                                          int n_seg, int64_t n_chunks, int num_bits,
                                                          int is_unsigned, int narrow) {
  constexpr int G = LPG * Elem<DT>::kVec;
  const IntQ q = make_intq(num_bits, is_unsigned, narrow);
  if ((int64_t)blockIdx.x >= n_chunks) return;
  SegCursor cur;
  cur.init(segs, blk_start, n_seg, blockIdx.x);
  for (int64_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    cur.seek(c);
    const int64_t e0 = (c - cur.c_begin) * MOQ_MT_CHUNK;
    if (cur.aligned && e0 + MOQ_MT_CHUNK <= cur.sg.n) {
      group_chunk<DT, LPG, true, false>(cur.sg.x, cur.sg.y, cur.sg.amax, e0, q, nullptr, 0);
    } else {
      const int64_t g0 = e0 / G;
      const int64_t g1 = (e0 + MOQ_MT_CHUNK < cur.sg.n ? e0 + MOQ_MT_CHUNK : cur.sg.n) / G;
      for (int64_t grp = g0 + threadIdx.x; grp < g1; grp += kBlock)
        group_scalar<DT, true, false>(cur.sg.x, cur.sg.y, cur.sg.amax, grp, G, q, nullptr, 0);
    }
  }
}

}  // namespace moq

// ================================================================================================
// C-ABI
// ================================================================================================
using namespace moq;

static inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }

static int mt_grid(int64_t n_chunks) { return (int)(n_chunks < 2048 ? n_chunks : 2048); }  // reductions

extern "C" int moq_amax(const void* x, int64_t n, int dt, float* out, int accumulate, void* stream) {
  if (out == nullptr || n < 0 || (n > 0 && x == nullptr)) {
    set_error("moq_amax: null pointer or negative size");
    return MOQ_ERR_INVALID;
  }
  if (!accumulate) {
    if (hipMemsetAsync(out, 0, sizeof(float), S(stream)) != hipSuccess) return check_launch("moq_amax memset");
  }
  if (n == 0) return MOQ_OK;
  const int64_t chunks = (n + MOQ_MT_CHUNK - 1) / MOQ_MT_CHUNK;
  // <= 1024 workgroups (four per CU: 64 KiB of loads in flight per CU and chunk round); at most one same-address atomic
  // per workgroup at the end, none when the running maximum already covers the workgroup's result
  const int grid = (int)(chunks < 1024 ? chunks : 1024);
  MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((amax_kernel<DT>), dim3(grid), dim3(kBlock), 0, S(stream),
                                            x, n, reinterpret_cast<uint32_t*>(out)));
  return check_launch("moq_amax");
}

template <bool FP8>
static int launch_map(const void* x, void* y, int64_t n, int dt, const float* amax, int amax_mode,
                      int64_t axis_size, int64_t inner, int num_bits, int is_unsigned, int narrow,
                      void* stream, const char* who) {
  if (n < 0 || (n > 0 && (x == nullptr || y == nullptr))) {
    set_error("%s: null pointer or negative size", who);
    return MOQ_ERR_INVALID;
  }
  if (!FP8 && (num_bits < 2 || num_bits > 16)) {
    set_error("%s: num_bits=%d out of range [2,16]", who, num_bits);
    return MOQ_ERR_INVALID;
  }
  if (n == 0) return MOQ_OK;
  const int grid = copy_grid((n + MOQ_MT_CHUNK - 1) / MOQ_MT_CHUNK);
  if (amax_mode == MOQ_AMAX_SCALAR) {
    if (amax == nullptr) {
      if (!FP8) {
        set_error("%s: amax must not be NULL", who);
        return MOQ_ERR_INVALID;
      }
      MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((map_scalar_amax_kernel<DT, OpFp8Cast>), dim3(grid),
                                                dim3(kBlock), copy_lds_1t(), S(stream), x, y, n, amax, 0, 0, 0));
    } else if (FP8) {
      MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((map_scalar_amax_kernel<DT, OpFp8Qdq>), dim3(grid),
                                                dim3(kBlock), copy_lds_1t(), S(stream), x, y, n, amax, 0, 0, 0));
    } else {
      MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((map_scalar_amax_kernel<DT, OpIntQdq>), dim3(grid),
                                                dim3(kBlock), copy_lds_1t(), S(stream), x, y, n, amax, num_bits,
                                                is_unsigned, narrow));
    }
  } else if (amax_mode == MOQ_AMAX_AXIS) {
    if (amax == nullptr || axis_size <= 0 || inner <= 0) {
      set_error("%s: axis mode needs amax, axis_size > 0, inner > 0", who);
      return MOQ_ERR_INVALID;
    }
    const int wrap = n > axis_size * inner ? 1 : 0;  // outer > 1: the amax index wraps around
    MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((map_axis_amax_kernel<DT, FP8>), dim3(grid), dim3(kBlock),
                                              0, S(stream), x, y, n, amax, axis_size, inner,
                                              log2_or_neg(inner), wrap, num_bits, is_unsigned, narrow));
  } else {
    set_error("%s: unknown amax_mode %d", who, amax_mode);
    return MOQ_ERR_INVALID;
  }
  return check_launch(who);
}

extern "C" int moq_fake_quant_int(const void* x, void* y, int64_t n, int dt, const float* amax,
                                  int amax_mode, int64_t axis_size, int64_t inner, int num_bits,
                                  int is_unsigned, int narrow_range, void* stream) {
  return launch_map<false>(x, y, n, dt, amax, amax_mode, axis_size, inner, num_bits, is_unsigned,
                           narrow_range, stream, "moq_fake_quant_int");
}

extern "C" int moq_fake_quant_e4m3(const void* x, void* y, int64_t n, int dt, const float* amax,
                                   int amax_mode, int64_t axis_size, int64_t inner, void* stream) {
  return launch_map<true>(x, y, n, dt, amax, amax_mode, axis_size, inner, 8, 0, 0, stream,
                          "moq_fake_quant_e4m3");
}

// LPG dispatch for the group kernels
#define MOQ_DISPATCH_LPG(lpg, ...)                                   \
  switch (lpg) {                                                     \
    case 1: { constexpr int LPG = 1; __VA_ARGS__; } break;           \
    case 2: { constexpr int LPG = 2; __VA_ARGS__; } break;           \
    case 4: { constexpr int LPG = 4; __VA_ARGS__; } break;           \
    case 8: { constexpr int LPG = 8; __VA_ARGS__; } break;           \
    case 16: { constexpr int LPG = 16; __VA_ARGS__; } break;         \
    case 32: { constexpr int LPG = 32; __VA_ARGS__; } break;         \
    case 64: { constexpr int LPG = 64; __VA_ARGS__; } break;         \
    default: set_error("group size %d not supported (g / elements-per-16B must be a power of two <= 64)", g); \
             return MOQ_ERR_UNSUPPORTED;                             \
  }

namespace moq {
int launch_group(const void* x, void* y, float* amax_out, int64_t n_groups, int g, int dt, int num_bits,
                 int is_unsigned, int narrow, bool qdq, const void* s, int64_t cols, void* stream,
                 const char* who) {
  if (n_groups < 0 || g <= 0 || (n_groups > 0 && (x == nullptr || (qdq && y == nullptr)))) {
    set_error("%s: null pointer or bad sizes", who);
    return MOQ_ERR_INVALID;
  }
  if (qdq && (num_bits < 2 || num_bits > 16)) {
    set_error("%s: num_bits=%d out of range [2,16]", who, num_bits);
    return MOQ_ERR_INVALID;
  }
  if (n_groups == 0) return MOQ_OK;
  const int vec = dt == MOQ_F32 ? 4 : 8;
  if (g % vec != 0) {
    set_error("%s: group size %d is not a multiple of %d", who, g, vec);
    return MOQ_ERR_UNSUPPORTED;
  }
  if (s != nullptr && (cols <= 0 || cols % g != 0)) {
    set_error("%s: cols must be a positive multiple of g", who);
    return MOQ_ERR_INVALID;
  }
  const int lpg = g / vec;
  const int grid = copy_grid((n_groups * (int64_t)g + MOQ_MT_CHUNK - 1) / MOQ_MT_CHUNK);
#define MOQ_LAUNCH_GROUP(QDQ, PRE)                                                                   \
  MOQ_DISPATCH_DTYPE(dt, MOQ_DISPATCH_LPG(lpg, hipLaunchKernelGGL((group_kernel<DT, LPG, QDQ, PRE>),  \
                                                                  dim3(grid), dim3(kBlock),                        \
                                                                  (QDQ) ? copy_lds_1t(32 * 1024) : 0,                \
                                                                  S(stream), x, y, amax_out, n_groups, \
                                                                  num_bits, is_unsigned, narrow, s,   \
                                                                  cols)))
  if (qdq && s != nullptr) { MOQ_LAUNCH_GROUP(true, true); }
  else if (qdq) { MOQ_LAUNCH_GROUP(true, false); }
  else { MOQ_LAUNCH_GROUP(false, false); }
#undef MOQ_LAUNCH_GROUP
  return check_launch(who);
}
}  // namespace moq

extern "C" int moq_amax_qdq_int_group(const void* x, void* y, float* amax_out, int64_t n_groups, int g,
                                      int dt, int num_bits, int is_unsigned, int narrow_range,
                                      void* stream) {
  return launch_group(x, y, amax_out, n_groups, g, dt, num_bits, is_unsigned, narrow_range, true,
                      nullptr, 0, stream, "moq_amax_qdq_int_group");
}

extern "C" int moq_awq_scale_qdq(const void* w, const void* s, void* y, int64_t rows, int64_t cols,
                                 int g, int dt, int num_bits, void* stream) {
  if (s == nullptr) {
    set_error("moq_awq_scale_qdq: scale vector is NULL");
    return MOQ_ERR_INVALID;
  }
  if (rows < 0 || cols <= 0 || g <= 0 || cols % g != 0) {
    set_error("moq_awq_scale_qdq: cols must be a positive multiple of g");
    return MOQ_ERR_INVALID;
  }
  return launch_group(w, y, nullptr, rows * (cols / g), g, dt, num_bits, 0, 0, true, s, cols, stream,
                      "moq_awq_scale_qdq");
}

// ---------------------------------------------------------------- multi-tensor
extern "C" int64_t moq_mt_plan(const int64_t* n_host, int n_seg, int64_t* blk_start_host) {
  if (n_host == nullptr || blk_start_host == nullptr || n_seg < 0) {
    set_error("moq_mt_plan: null pointer");
    return MOQ_ERR_INVALID;
  }
  int64_t acc = 0;
  for (int i = 0; i < n_seg; ++i) {
    if (n_host[i] < 0) {
      set_error("moq_mt_plan: negative element count in segment %d", i);
      return MOQ_ERR_INVALID;
    }
    blk_start_host[i] = acc;
    acc += (n_host[i] + MOQ_MT_CHUNK - 1) / MOQ_MT_CHUNK;
  }
  blk_start_host[n_seg] = acc;
  return acc;
}

static int mt_check(const moq_seg* segs, const int64_t* blk_start, int n_seg, int64_t n_chunks,
                    const char* who) {
  if (n_seg < 0 || n_chunks < 0 || (n_seg > 0 && (segs == nullptr || blk_start == nullptr))) {
    set_error("%s: null pointer or negative size", who);
    return MOQ_ERR_INVALID;
  }
  return MOQ_OK;
}
extern "C" int moq_mt_amax(const moq_seg* segs, const int64_t* blk_start, int n_seg, int64_t n_chunks,
                           int dt, void* stream) {
  int rc = mt_check(segs, blk_start, n_seg, n_chunks, "moq_mt_amax");
  if (rc != MOQ_OK || n_seg == 0) return rc;
  hipLaunchKernelGGL(mt_zero_amax_kernel, dim3((n_seg + 255) / 256), dim3(256), 0, S(stream), segs, n_seg);
  if (n_chunks > 0) {
    // (experiment build) MOQ_TUNE_AMAX_KEEP=1: plain (cache-allocating) loads, for a QDQ pass that follows while the
    // tensors are still in the 256 MB Infinity Cache
    const bool keep = moq_tune("MOQ_TUNE_AMAX_KEEP", 0) != 0;
    if (keep) {
      MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((mt_amax_kernel<DT, false>), dim3(mt_grid(n_chunks)), dim3(kBlock),
                                                0, S(stream), segs, blk_start, n_seg, n_chunks));
    } else {
      MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((mt_amax_kernel<DT, true>), dim3(mt_grid(n_chunks)), dim3(kBlock),
                                                0, S(stream), segs, blk_start, n_seg, n_chunks));
    }
  }
  return check_launch("moq_mt_amax");
}

extern "C" int moq_mt_amax_ws(const moq_seg* segs, const int64_t* blk_start, int n_seg, int64_t n_chunks, int dt,
                             float* chunk_scratch, void* stream) {
  int rc = mt_check(segs, blk_start, n_seg, n_chunks, "moq_mt_amax_ws");
  if (rc != MOQ_OK || n_seg == 0) return rc;
  if (chunk_scratch == nullptr) {
    set_error("moq_mt_amax_ws: chunk_scratch (n_chunks floats) must not be NULL");
    return MOQ_ERR_INVALID;
  }
  if (n_chunks > 0) {
    MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((mt_amax_chunks_kernel<DT>), dim3(read_grid(n_chunks)), dim3(kBlock), 0,
                                              S(stream), segs, blk_start, n_seg, n_chunks,
                                              reinterpret_cast<uint32_t*>(chunk_scratch)));
  }
  hipLaunchKernelGGL(mt_amax_fold_kernel, dim3((unsigned)n_seg), dim3(kBlock), 0, S(stream), segs, blk_start,
                     reinterpret_cast<const uint32_t*>(chunk_scratch));
  return check_launch("moq_mt_amax_ws");
}

extern "C" int moq_mt_amax_running(const moq_seg* segs, const int64_t* blk_start, int n_seg, int64_t n_chunks, int dt,
                                  float* chunk_scratch, const moq_amax_fold* folds, int n_folds, void* stream) {
  int rc = mt_check(segs, blk_start, n_seg, n_chunks, "moq_mt_amax_running");
  if (rc != MOQ_OK || n_seg == 0) return rc;
  if (chunk_scratch == nullptr || n_folds < 0 || (n_folds > 0 && folds == nullptr)) {
    set_error("moq_mt_amax_running: chunk_scratch (n_chunks floats) and folds must not be NULL");
    return MOQ_ERR_INVALID;
  }
  if (n_chunks > 0) {
    MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((mt_amax_chunks_kernel<DT>), dim3(read_grid(n_chunks)), dim3(kBlock), 0,
                                              S(stream), segs, blk_start, n_seg, n_chunks,
                                              reinterpret_cast<uint32_t*>(chunk_scratch)));
  }
  if (n_folds > 0) {
    hipLaunchKernelGGL(mt_amax_fold_running_kernel, dim3((unsigned)n_folds), dim3(kFoldBlock), 0, S(stream), blk_start,
                       reinterpret_cast<const uint32_t*>(chunk_scratch), folds);
  }
  return check_launch("moq_mt_amax_running");
}

// internal (moq_formats.hip's mask + apply pass): stage 2 of the two-stage abs-max over per-chunk maxima somebody else wrote
int moq_mt_amax_fold_launch(const moq_seg* segs, const int64_t* blk_start, int n_seg, const void* chunk_scratch,
                            void* stream) {
  hipLaunchKernelGGL(mt_amax_fold_kernel, dim3((unsigned)n_seg), dim3(kBlock), 0, S(stream), segs, blk_start,
                     reinterpret_cast<const uint32_t*>(chunk_scratch));
  return check_launch("moq_mt_amax_fold");
}

extern "C" int moq_mt_fake_quant_e4m3(const moq_seg* segs, const int64_t* blk_start, int n_seg,
                                      int64_t n_chunks, int dt, void* stream) {
  int rc = mt_check(segs, blk_start, n_seg, n_chunks, "moq_mt_fake_quant_e4m3");
  if (rc != MOQ_OK || n_seg == 0 || n_chunks == 0) return rc;
  MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((mt_map_kernel<DT, OpFp8Qdq>), dim3(copy_grid(n_chunks)),
                                            dim3(kBlock), copy_lds(), S(stream), segs, blk_start, n_seg, n_chunks,
                                            0, 0, 0));
  return check_launch("moq_mt_fake_quant_e4m3");
}

extern "C" int moq_mt_fake_quant_int(const moq_seg* segs, const int64_t* blk_start, int n_seg,
                                     int64_t n_chunks, int dt, int num_bits, int is_unsigned,
                                     int narrow_range, void* stream) {
  int rc = mt_check(segs, blk_start, n_seg, n_chunks, "moq_mt_fake_quant_int");
  if (rc != MOQ_OK || n_seg == 0 || n_chunks == 0) return rc;
  if (num_bits < 2 || num_bits > 16) {
    set_error("moq_mt_fake_quant_int: num_bits=%d out of range", num_bits);
    return MOQ_ERR_INVALID;
  }
  MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((mt_map_kernel<DT, OpIntQdq>), dim3(copy_grid(n_chunks)),
                                            dim3(kBlock), copy_lds(), S(stream), segs, blk_start, n_seg, n_chunks,
                                            num_bits, is_unsigned, narrow_range));
  return check_launch("moq_mt_fake_quant_int");
}

extern "C" int moq_mt_amax_qdq_int_group(const moq_seg* segs, const int64_t* blk_start, int n_seg,
                                         int64_t n_chunks, int g, int dt, int num_bits, int is_unsigned,
                                         int narrow_range, void* stream) {
  int rc = mt_check(segs, blk_start, n_seg, n_chunks, "moq_mt_amax_qdq_int_group");
  if (rc != MOQ_OK || n_seg == 0 || n_chunks == 0) return rc;
  const int vec = dt == MOQ_F32 ? 4 : 8;
  if (g <= 0 || g % vec != 0 || MOQ_MT_CHUNK % g != 0) {
    set_error("moq_mt_amax_qdq_int_group: unsupported group size %d", g);
    return MOQ_ERR_UNSUPPORTED;
  }
  const int lpg = g / vec;
  MOQ_DISPATCH_DTYPE(dt, MOQ_DISPATCH_LPG(lpg, hipLaunchKernelGGL((mt_group_kernel<DT, LPG>),
                                                                  dim3(copy_grid(n_chunks)), dim3(kBlock),
                                                                  copy_lds(), S(stream), segs, blk_start, n_seg,
                                                                  n_chunks, num_bits, is_unsigned,
                                                                  narrow_range)));
  return check_launch("moq_mt_amax_qdq_int_group");
}

extern "C" int moq_awq_err_weight(const void* w, const void* s, const float* r, float* e_out, void* a_out, int64_t rows,
                                  int64_t cols, int g, int dt, int num_bits, int planes, void* stream) {
  if (w == nullptr || s == nullptr || r == nullptr || e_out == nullptr || a_out == nullptr || rows <= 0 || cols <= 0 ||
      g <= 0 || num_bits < 2 || num_bits > 8 || planes < 1 || planes > 3) {
    set_error("moq_awq_err_weight: bad arguments");
    return MOQ_ERR_INVALID;
  }
  const int vec = dt == MOQ_F32 ? 4 : 8;
  const int lpg = g / vec;
  if (cols % g != 0 || g % vec != 0 || lpg > 64 || (lpg & (lpg - 1)) != 0 || MOQ_MT_CHUNK % g != 0 ||
      cols >= (1LL << 31) ||
      ((reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(r) |
        reinterpret_cast<uintptr_t>(e_out) | reinterpret_cast<uintptr_t>(a_out)) & 15u) != 0) {
    set_error("moq_awq_err_weight: needs cols %% g == 0, g / %d a power of two <= 64, 16-byte aligned pointers", vec);
    return MOQ_ERR_UNSUPPORTED;
  }
  const int64_t n = rows * cols;
  const int grid = copy_grid((n + MOQ_MT_CHUNK - 1) / MOQ_MT_CHUNK);
  const int cs = log2_or_neg(cols);
  MOQ_DISPATCH_DTYPE(dt, MOQ_DISPATCH_LPG(lpg, hipLaunchKernelGGL((awq_err_weight_kernel<DT, LPG>), dim3(grid),
                                                                  dim3(kBlock), copy_lds_1t(), S(stream), w, s, r, e_out,
                                                                  reinterpret_cast<uint16_t*>(a_out), n, cols, cs,
                                                                  num_bits, planes)));
  return check_launch("moq_awq_err_weight");
}
