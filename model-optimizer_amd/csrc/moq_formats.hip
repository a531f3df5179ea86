// moq_formats.hip -- MX dynamic-block QDQ, histogram, 2:4 mask, real INT4 pack/unpack, export packer and
// the column-scale fold.  All HBM-bound streaming kernels with 16-byte lane accesses.
#include <type_traits>

#include "moq_common.h"
#include "moq_chunk.h"
#include "moq_hist.h"
#include "moq_mx.h"

namespace moq {

__device__ __forceinline__ bool al16(const void* p) {
  return (reinterpret_cast<uintptr_t>(p) & 15u) == 0;
}

// fast path: cols % block == 0, block % kVec == 0 -> an MX block is LPG adjacent lanes of one packet.
// Chunk skeleton: all packets of a chunk in flight before the first use, non-temporal loads and stores, dense
// strided grid.  FMT >= 0 fixes the element format at compile time (E2M1 / E4M3: the MXFP4 / MXFP8 presets) so
// that the rounding constants fold; FMT < 0 reads it from `fmt`.
// block scale of the E8M0 formats / of the two-level (NVFP4-style) formats, as a functor of the block's abs-max
struct MxScaleE8M0 {
  float fmt_max;
  __device__ __forceinline__ void operator()(float am, float& sc, float& un) const { mx_scale_e8m0(am, fmt_max, sc, un); }
};
struct MxScaleGeneral {
  float fmt_max;
  MxFmt sf;
  const float* global;
  __device__ __forceinline__ void operator()(float am, float& sc, float& un) const {
    mx_scale_general(am, fmt_max, sf, global, sc, un);
  }
};
// The two-level scales with a tensor-wide amax, without the two fp64 divisions per block of mx_scale_general (they, not
// the memory stream, bounded the NVFP4-style kernel: 0.47 of 8 TB/s in round 4).  The block scale is
//     q = (double)r / T,   r = round_to_scale_format(arg),   T = (double)sf.maxv * (double)(emax / global)   (one value per tensor)
// and r is a scale-format value: r = M * 2^k with a 4-bit M in 8..15 whenever the scale format keeps at most three
// mantissa bits (E4M3, E5M2, ...; k comes from r's fp32 exponent, subnormals of the scale format included).  Rounding
// commutes with scaling by a power of two while nothing leaves the normal range, so
//     unscale = (float)q       = (float)fl64(M / T)        * 2^k
//     scale   = (float)(1 / q) = (float)fl64(1 / fl64(M / T)) * 2^-k
// with the eight (M) pairs tabulated once per workgroup in LDS by the very fp64 divisions the reference performs.  A result
// whose exponent would leave [-125, 126] (a block abs-max ~38 decades from the tensor's) takes mx_scale_general; r == 0
// gives unscale 0 and scale +inf as there.
struct MxScaleTwoLevelTable {
  float fmt_max;
  MxFmt sf;
  const float* global;
  double two_level;    // T
  bool global_bad;
  const float2* table;  // LDS: [M - 8] -> {(float)fl64(M / T), (float)fl64(1 / fl64(M / T))}
  __device__ __forceinline__ void operator()(float am, float& sc, float& un) const {
    sc = 1.0f;
    un = 1.0f;
    if (mx_bad_amax(am) || global_bad) return;
    const float local_unscale = am / fmt_max;
    const float arg = (float)((double)local_unscale * two_level);
    const float r = mx_round_abs(arg, sf);
    if (r == 0.0f) {
      un = 0.0f;
      sc = __uint_as_float(0x7F800000u);
      return;
    }
    const uint32_t rb = __float_as_uint(r);
    const int k = (int)((rb >> 23) & 0xFFu) - 130;  // r = (8 + top three mantissa bits) * 2^k
    const float2 t = table[(rb >> 20) & 7u];
    const int eu = (int)((__float_as_uint(t.x) >> 23) & 0xFFu) - 127 + k, es = (int)((__float_as_uint(t.y) >> 23) & 0xFFu) - 127 - k;
    if (eu < -125 || eu > 126 || es < -125 || es > 126 || k < -126 || k > 126) {
      mx_scale_general(am, fmt_max, sf, global, sc, un);
      return;
    }
    un = t.x * __uint_as_float((uint32_t)(k + 127) << 23);
    sc = t.y * __uint_as_float((uint32_t)(127 - k) << 23);
  }
};
// the packet's block abs-max -> scale -> element rounding by integer ops: every format, every scale functor
template <int DT, int LPG, class ScaleFn>
__device__ __forceinline__ void mx_packet_general(float (&v)[8], const MxFmt f, const ScaleFn& scale_of) {
  constexpr int V = Elem<DT>::kVec;
  float am = 0.0f;
#pragma unroll
  for (int i = 0; i < V; ++i) am = __builtin_fmaxf(am, mx_abs_clamped(v[i]));
  // non-negative floats order like their bit patterns
  am = __uint_as_float(group_max_u32<LPG>(__float_as_uint(am)));
  float sc, un;
  scale_of(am, sc, un);
#pragma unroll
  for (int i = 0; i < V; ++i) v[i] = mx_qdq(v[i], sc, un, f);
}
// HW4: E2M1 elements with E8M0 scales (MXFP4) -- the chip's scaled FP4 converters (mx_e2m1_hw, moq_mx.h); a block they do
// not take (a NaN inside, an exponent at the edge of the range) runs the general packet code: the branch is uniform over the
// block's LPG lanes, so the DPP butterfly inside it sees whole blocks
template <int DT, int LPG, bool HW4 = false, class ScaleFn>
__device__ __forceinline__ void mx_chunk(const char* xb, char* yb, int64_t e0, int64_t n, const MxFmt f,
                                         const ScaleFn& scale_of) {
  constexpr int V = Elem<DT>::kVec;
  constexpr int P = Chunk<DT>::kPackets;
  constexpr int ES = 16 / V;
  Pack16 in[P];
  // n is a multiple of the block (= LPG * V elements), so a group is entirely live or entirely past the end
#pragma unroll
  for (int u = 0; u < P; ++u) {
    const int64_t e = e0 + packet_off<DT>(u);
    if (e < n) in[u] = load16_nt(xb + e * ES);
    else in[u].w[0] = in[u].w[1] = in[u].w[2] = in[u].w[3] = 0u;
  }
#pragma unroll
  for (int u = 0; u < P; ++u) {
    const int64_t e = e0 + packet_off<DT>(u);
    float v[8];
    unpack<DT>(in[u], v);
    if constexpr (HW4) {
      if (!mx_e2m1_hw<V>(v, group_max_u32<LPG>(pack_absmax<DT>(in[u])))) mx_packet_general<DT, LPG>(v, f, scale_of);
    } else {
      mx_packet_general<DT, LPG>(v, f, scale_of);
    }
    if (e < n) store16_nt(yb + e * ES, pack<DT>(v));
  }
}
template <int DT, int LPG, int FMT>
__global__ __launch_bounds__(kBlock) void mx_kernel(const void* __restrict__ x, void* __restrict__ y,
                                                    int64_t n, int fmt) {
  const MxFmt f = mx_fmt(FMT >= 0 ? FMT : fmt);
  const int64_t n_chunks = (n + MOQ_MT_CHUNK - 1) / MOQ_MT_CHUNK;
  const MxScaleE8M0 scale_of{f.maxv};
  for (int64_t c = blockIdx.x; c < n_chunks; c += gridDim.x)
    mx_chunk<DT, LPG, FMT == MOQ_E2M1>(reinterpret_cast<const char*>(x), reinterpret_cast<char*>(y), c * MOQ_MT_CHUNK, n, f, scale_of);
}
// two-level block scales (an element format as the scale format, optional tensor-wide amax) on the same skeleton; the
// one-thread-per-block generic kernel moved 2 bytes per lane and load (0.21 of the HBM roofline at g = 16)
// FMT / SFMT >= 0 fix the element / scale format at compile time (E2M1 elements with E4M3 scales, the NVFP4-style preset),
// < 0 read them from the arguments.
template <int DT, int LPG, int FMT, int SFMT>
__global__ __launch_bounds__(kBlock) void mx_two_level_kernel(const void* __restrict__ x, void* __restrict__ y, int64_t n,
                                                              int fmt, int scale_fmt, const float* __restrict__ global_amax) {
  __shared__ float2 s_table[8];
  const MxFmt f = mx_fmt(FMT >= 0 ? FMT : fmt);
  const MxFmt sf = mx_fmt(SFMT >= 0 ? SFMT : scale_fmt);
  const int64_t n_chunks = (n + MOQ_MT_CHUNK - 1) / MOQ_MT_CHUNK;
  if (global_amax != nullptr && sf.kind != 2 && sf.m <= 3) {  // (workgroup-uniform)
    const float g = *global_amax;
    const bool g_bad = mx_bad_amax(g);
    const double two_level = (double)sf.maxv * (double)(f.maxv / g);
    if (threadIdx.x < 8 && !g_bad) {
      const double q = (double)(float)(8 + threadIdx.x) / two_level;
      s_table[threadIdx.x] = make_float2((float)q, (float)(1.0 / q));
    }
    __syncthreads();
    const MxScaleTwoLevelTable scale_of{f.maxv, sf, global_amax, two_level, g_bad, s_table};
    for (int64_t c = blockIdx.x; c < n_chunks; c += gridDim.x)
      mx_chunk<DT, LPG>(reinterpret_cast<const char*>(x), reinterpret_cast<char*>(y), c * MOQ_MT_CHUNK, n, f, scale_of);
    return;
  }
  const MxScaleGeneral scale_of{f.maxv, sf, global_amax};
  for (int64_t c = blockIdx.x; c < n_chunks; c += gridDim.x)
    mx_chunk<DT, LPG>(reinterpret_cast<const char*>(x), reinterpret_cast<char*>(y), c * MOQ_MT_CHUNK, n, f, scale_of);
}
// the same over a segment table: all weights of a layer / model in ONE launch (every segment 16-byte aligned with
// n % block == 0, checked when the table is built)
template <int DT, int LPG, int FMT>
__global__ __launch_bounds__(kBlock) void mt_mx_kernel(const moq_seg* __restrict__ segs,
                                                       const int64_t* __restrict__ blk_start, int n_seg,
                                                       int64_t n_chunks, int fmt) {
  if ((int64_t)blockIdx.x >= n_chunks) return;
  const MxFmt f = mx_fmt(FMT >= 0 ? FMT : fmt);
  SegCursor cur;
  cur.init(segs, blk_start, n_seg, blockIdx.x);
  const MxScaleE8M0 scale_of{f.maxv};
  for (int64_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    cur.seek(c);
    mx_chunk<DT, LPG, FMT == MOQ_E2M1>(reinterpret_cast<const char*>(cur.sg.x), reinterpret_cast<char*>(cur.sg.y),
                                       (c - cur.c_begin) * MOQ_MT_CHUNK, cur.sg.n, f, scale_of);
  }
}
// SmoothQuant fold + MX QDQ in one pass over a segment table (moq_mt_fold_mx_fused): the element is multiplied by its column's
// fp32 scale and rounded to the storage dtype -- what the separate fold writes back to memory -- before the block abs-max is
// taken.
//
// TILED chunks.  A bf16 packet is 16 B of weights but needs 32 B of fp32 column scales; walked linearly (a chunk = 8192
// consecutive elements, every packet its own columns) the scale reads are twice the weight reads -- L2 hits, but three load
// instructions per packet instead of one: 0.68 of the HBM roofline on all Llama-3-70B weights against the plain MX kernel's
// 0.76.  So a chunk of a foldable tensor is a TILE instead: kPackets ROWS x (kBlock * kVec) COLUMNS (4 x 2048 for 16-bit
// types, 8 x 1024 for fp32) -- still exactly MOQ_MT_CHUNK elements, so the segment table's chunk plan holds -- in which
// packet u of thread t is row u, columns t * kVec ...: all of a thread's packets sit in the SAME columns, one scale load (two
// 16-byte reads) serves the tile, and a wave instruction still moves one contiguous KiB of a row.  Needs cols % (kBlock * kVec)
// == 0 and rows % kPackets == 0 (every Llama / Mixtral weight); other tensors take the linear walk below.
template <int DT>
__device__ __forceinline__ uint32_t fold_col(uint32_t c0, int off, uint32_t cols) {
  uint32_t t = c0 + (uint32_t)off;
  if (cols >= (uint32_t)MOQ_MT_CHUNK) return t >= cols ? t - cols : t;
  return t % cols;
}
template <int DT>
__device__ __forceinline__ void fold_load(const float* __restrict__ scale, uint32_t col, float (&sf)[8]) {
  constexpr int V = Elem<DT>::kVec;
  // (the pointer comes out of a table in memory: without the address-space cast the compiler emits FLAT loads for it)
  const u32x4_t a = *((gptr_c16)(uintptr_t)(scale + col));
  sf[0] = __uint_as_float(a.x); sf[1] = __uint_as_float(a.y); sf[2] = __uint_as_float(a.z); sf[3] = __uint_as_float(a.w);
  if constexpr (V == 8) {
    const u32x4_t b = *((gptr_c16)(uintptr_t)(scale + col + 4));
    sf[4] = __uint_as_float(b.x); sf[5] = __uint_as_float(b.y); sf[6] = __uint_as_float(b.z); sf[7] = __uint_as_float(b.w);
  }
}
// where packet u of this thread lives: element offset inside the tensor, and (TILED only) its columns are the thread's own
template <int DT>
struct FoldWalk {
  bool tiled;
  uint32_t cols, wpb;  // row length; tiles per band of kPackets rows
  __device__ __forceinline__ void set(int64_t n, int64_t cols_) {
    constexpr int W = kBlock * Elem<DT>::kVec;
    cols = (uint32_t)cols_;
    tiled = cols % W == 0 && (n / cols_) % Chunk<DT>::kPackets == 0;
    wpb = cols / W;
  }
};
template <int DT, int LPG, bool TILED, bool HW4>
__device__ __forceinline__ void mx_fold_chunk(const char* xb, char* yb, int64_t j, int64_t n, const MxFmt f,
                                              const MxScaleE8M0& scale_of, const float* __restrict__ scale,
                                              const FoldWalk<DT>& wk) {
  constexpr int V = Elem<DT>::kVec;
  constexpr int P = Chunk<DT>::kPackets;
  constexpr int ES = 16 / V;
  Pack16 in[P];
  int64_t e[P];
  float sf[TILED ? 1 : P][8];
  if constexpr (TILED) {
    const int64_t band = j / wk.wpb;
    const uint32_t col = (uint32_t)(j - band * wk.wpb) * (kBlock * V) + threadIdx.x * V;
#pragma unroll
    for (int u = 0; u < P; ++u) e[u] = (band * P + u) * (int64_t)wk.cols + col;
#pragma unroll
    for (int u = 0; u < P; ++u) in[u] = load16_nt(xb + e[u] * ES);  // (rows % P == 0: every packet of a tile is live)
    fold_load<DT>(scale, col, sf[0]);
  } else {
    const int64_t e0 = j * MOQ_MT_CHUNK;
    const uint32_t c0 = (uint32_t)(e0 % (int64_t)wk.cols);
#pragma unroll
    for (int u = 0; u < P; ++u) {
      e[u] = e0 + packet_off<DT>(u);
      if (e[u] < n) in[u] = load16_nt(xb + e[u] * ES);
      else in[u].w[0] = in[u].w[1] = in[u].w[2] = in[u].w[3] = 0u;
    }
#pragma unroll
    for (int u = 0; u < P; ++u) fold_load<DT>(scale, fold_col<DT>(c0, packet_off<DT>(u), wk.cols), sf[u]);
  }
#pragma unroll
  for (int u = 0; u < P; ++u) {
    float v[8];
    unpack<DT>(in[u], v);
#pragma unroll
    for (int i = 0; i < V; ++i) v[i] = v[i] * sf[TILED ? 0 : u][i];
    // the fold's result AS STORED: rounded to the dtype (pack), and read back (unpack) -- the packed form also gives the
    // block's abs-max pattern with packed 16-bit ops
    const Pack16 folded = pack<DT>(v);
    unpack<DT>(folded, v);
    if constexpr (HW4) {
      if (!mx_e2m1_hw<V>(v, group_max_u32<LPG>(pack_absmax<DT>(folded)))) mx_packet_general<DT, LPG>(v, f, scale_of);
    } else {
      mx_packet_general<DT, LPG>(v, f, scale_of);
    }
    if (TILED || e[u] < n) store16_nt(yb + e[u] * ES, pack<DT>(v));
  }
}
template <int DT, int LPG, int FMT>
__global__ __launch_bounds__(kBlock) void mt_fold_mx_kernel(const moq_seg* __restrict__ segs,
                                                            const int64_t* __restrict__ blk_start,
                                                            const moq_fold_seg* __restrict__ side, int n_seg,
                                                            int64_t n_chunks, int fmt) {
  if ((int64_t)blockIdx.x >= n_chunks) return;
  const MxFmt f = mx_fmt(FMT >= 0 ? FMT : fmt);
  SegCursor cur;
  cur.init(segs, blk_start, n_seg, blockIdx.x);
  moq_fold_seg sd = side[cur.s];
  FoldWalk<DT> wk;
  wk.set(cur.sg.n, sd.cols);
  const MxScaleE8M0 scale_of{f.maxv};
  for (int64_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    if (cur.seek(c)) {
      sd = side[cur.s];
      wk.set(cur.sg.n, sd.cols);
    }
    const int64_t j = c - cur.c_begin;
    const char* xb = reinterpret_cast<const char*>(cur.sg.x);
    char* yb = reinterpret_cast<char*>(cur.sg.y);
    if (sd.scale == nullptr)
      mx_chunk<DT, LPG, FMT == MOQ_E2M1>(xb, yb, j * MOQ_MT_CHUNK, cur.sg.n, f, scale_of);
    else if (wk.tiled)
      mx_fold_chunk<DT, LPG, true, FMT == MOQ_E2M1>(xb, yb, j, cur.sg.n, f, scale_of, sd.scale, wk);
    else
      mx_fold_chunk<DT, LPG, false, FMT == MOQ_E2M1>(xb, yb, j, cur.sg.n, f, scale_of, sd.scale, wk);
  }
}
// generic path: one thread per MX block, handles ragged last blocks (virtual zero padding), any alignment and
// every block-scale format (scale_fmt == MOQ_E8M0: the exponent/mantissa test; else the general path above)
template <int DT>
__global__ void mx_generic_kernel(const void* __restrict__ x, void* __restrict__ y, int64_t rows,
                                  int64_t cols, int block, int fmt, int scale_fmt,
                                  const float* __restrict__ global_amax) {
  const MxFmt f = mx_fmt(fmt);
  const MxFmt sf = mx_fmt(scale_fmt);
  const int64_t bpr = (cols + block - 1) / block;
  for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < rows * bpr;
       b += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = b / bpr, c0 = (b % bpr) * block;
    const int64_t c1 = c0 + block < cols ? c0 + block : cols;
    float am = 0.0f;
    for (int64_t c = c0; c < c1; ++c) am = __builtin_fmaxf(am, mx_abs_clamped(load1<DT>(x, r * cols + c)));
    float sc, un;
    if (scale_fmt == MOQ_E8M0) mx_scale_e8m0(am, f.maxv, sc, un);
    else mx_scale_general(am, f.maxv, sf, global_amax, sc, un);
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

// The workgroup is kHistBlock = 1024 threads (16 waves) around ONE LDS histogram: the histogram (64 KiB with 8 copies
// of 2048 bins) limits a CU to two workgroups, and with 256-thread workgroups that meant two waves per SIMD -- too few
// to hide the load -> bin -> ds_add dependency chain.  Each 256-thread quarter walks its own chunks.
constexpr int kHistBlock = 1024;
template <int DT, bool FAST, bool SHARED>
__device__ __forceinline__ void hist_chunk(const void* x, int64_t e0, int64_t n, uint32_t* lds_hist, int bins,
                                           float max_edge, const SharedDiv& sd, int skip_zeros, int rshift, int copy) {
  constexpr int V = Elem<DT>::kVec;
  constexpr int P = Chunk<DT>::kPackets;
  const int tid = threadIdx.x & (kBlock - 1);
  Pack16 in[P];
#pragma unroll
  for (int u = 0; u < P; ++u) in[u] = ld_packet<DT, FAST>(x, e0 + (u * kBlock + tid) * V, n);
#pragma unroll
  for (int u = 0; u < P; ++u) {
    const int64_t e = e0 + (u * kBlock + tid) * V;
    float v[8];
    unpack<DT>(in[u], v);
#pragma unroll
    for (int i = 0; i < V; ++i) {
      int b = hist_bin<SHARED>(__builtin_fabsf(v[i]), bins, max_edge, sd, skip_zeros);
      if constexpr (!FAST) b = e + i < n ? b : bins;
      atomicAdd(&lds_hist[(b << rshift) + copy], 1u);
    }
  }
}
template <int DT, bool SHARED>
__global__ __launch_bounds__(kHistBlock) void hist_kernel(const void* __restrict__ x, int64_t n,
                                                          unsigned long long* __restrict__ counts, int bins,
                                                          float max_edge, int skip_zeros, int rshift) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_hist[];
  const int slots = (bins + 1) << rshift;
  const int copy = (int)(threadIdx.x & ((1u << rshift) - 1u));
  for (int b = threadIdx.x; b < slots; b += kHistBlock) lds_hist[b] = 0;
  __syncthreads();
  const bool al = al16(x);
  const SharedDiv sd = make_shared_div(max_edge);
  const int64_t n_chunks = (n + MOQ_MT_CHUNK - 1) / MOQ_MT_CHUNK;
  constexpr int Q = kHistBlock / kBlock;
  for (int64_t c = (int64_t)blockIdx.x * Q + (threadIdx.x / kBlock); c < n_chunks; c += (int64_t)gridDim.x * Q) {
    const int64_t e0 = c * MOQ_MT_CHUNK;
    if (al && e0 + MOQ_MT_CHUNK <= n)
      hist_chunk<DT, true, SHARED>(x, e0, n, lds_hist, bins, max_edge, sd, skip_zeros, rshift, copy);
    else
      hist_chunk<DT, false, SHARED>(x, e0, n, lds_hist, bins, max_edge, sd, skip_zeros, rshift, copy);
  }
  __syncthreads();
  for (int b = threadIdx.x; b < bins; b += kHistBlock) {
    uint32_t cnt = 0;
    for (int r = 0; r < (1 << rshift); ++r) cnt += lds_hist[(b << rshift) + r];
    if (cnt) atomicAdd(&counts[b], (unsigned long long)cnt);
  }
}
// bin counts beyond the LDS budget (after many growth steps of the calibrator): straight to L2 atomics
template <int DT>
__global__ __launch_bounds__(kBlock) void hist_global_kernel(const void* __restrict__ x, int64_t n,
                                                             unsigned long long* __restrict__ counts, int bins,
                                                             float max_edge, int skip_zeros) {
  const SharedDiv sd = make_shared_div(max_edge);
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n; e += (int64_t)gridDim.x * kBlock) {
    const int b = hist_bin<false>(__builtin_fabsf(load1<DT>(x, e)), bins, max_edge, sd, skip_zeros);
    if (b < bins) atomicAdd(&counts[b], 1ull);
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
  // mask bytes for elements 0..3 packed little-endian; the winner's word travels with the running maximum as a
  // select between literals (indexing a table by the winner made it a constant-memory load per group of four)
  constexpr uint32_t tbl[6] = {0x01000100u, 0x00000101u, 0x00010100u, 0x00010001u, 0x01000001u, 0x01010000u};
  uint32_t bm = tbl[0];
  float bv = s[0];
#pragma unroll
  for (int p = 1; p < 6; ++p) {
    // argmax: NaN is max, first wins (`&` / `|`: no short-circuit, hence no divergent branch per pattern)
    const bool better = (bv == bv) & ((s[p] != s[p]) | (s[p] > bv));
    bm = better ? tbl[p] : bm;
    bv = better ? s[p] : bv;
  }
  return bm;
}
// (global address space stated: a pointer read from the segment table would otherwise get flat_store)
__device__ __forceinline__ void store8_nt(void* p, uint32_t a, uint32_t b) {
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  typedef __attribute__((address_space(1))) u32x2* gptr_8;
  u32x2 v = {a, b};
  __builtin_nontemporal_store(v, (gptr_8)(uintptr_t)p);
}
__device__ __forceinline__ void store4_nt(void* p, uint32_t a) {
  typedef __attribute__((address_space(1))) uint32_t* gptr_4;
  __builtin_nontemporal_store(a, (gptr_4)(uintptr_t)p);
}
// FAST is a template parameter and the chunk loop branches on it ONCE per chunk: with `fast ? a : b` per packet hipcc
// kept two loads in flight per lane instead of the chunk's four
template <int DT, bool fast>
__device__ __forceinline__ void mask_chunk(const void* w, uint8_t* mask, int64_t e0, int64_t n) {
  constexpr int V = Elem<DT>::kVec;  // 8 (two groups of 4) or 4 (one group)
  constexpr int P = Chunk<DT>::kPackets;
  Pack16 in[P];
#pragma unroll
  for (int u = 0; u < P; ++u) in[u] = ld_packet<DT, fast>(w, e0 + packet_off<DT>(u), n);
#pragma unroll
  for (int u = 0; u < P; ++u) {
    const int64_t e = e0 + packet_off<DT>(u);
    float v[8];
    unpack<DT>(in[u], v);
#pragma unroll
    for (int i = 0; i < V; ++i) v[i] = __builtin_fabsf(v[i]);
    const uint32_t m0 = mask4(v[0], v[1], v[2], v[3]);
    uint32_t m1 = 0;
    if constexpr (V == 8) m1 = mask4(v[4], v[5], v[6], v[7]);
    if constexpr (fast) {
      if constexpr (V == 8) store8_nt(mask + e, m0, m1);
      else store4_nt(mask + e, m0);
    } else {  // n % 4 == 0: groups of four are all-in or all-out
      if (e + 4 <= n)
        for (int i = 0; i < 4; ++i) mask[e + i] = (m0 >> (8 * i)) & 1;
      if (V == 8 && e + 8 <= n)
        for (int i = 0; i < 4; ++i) mask[e + 4 + i] = (m1 >> (8 * i)) & 1;
    }
  }
}
// mask + APPLY in one pass (sparsify: `weight.mul_(mask)`, and what every later pass over a SparseModule's weight sees --
// weight * mask, sparsity/weight_sparsity/module.py:97-101): the chunk's mask bytes go to `mask`, the weight is rewritten in
// place as dtype(w * m) with m in {0, 1} -- an IEEE product, so a pruned negative weight becomes -0.0 and inf / NaN times 0
// is NaN exactly like the tensor multiply -- and the thread's abs-max pattern of the KEPT values is returned (the per-tensor
// amax of the masked weight for a calibration that follows: one read of the model instead of three).
template <int DT, bool fast>
__device__ __forceinline__ uint32_t mask_apply_chunk(void* w, uint8_t* mask, int64_t e0, int64_t n) {
  constexpr int V = Elem<DT>::kVec;
  constexpr int P = Chunk<DT>::kPackets;
  Pack16 in[P];
#pragma unroll
  for (int u = 0; u < P; ++u) in[u] = ld_packet<DT, fast>(w, e0 + packet_off<DT>(u), n);
  uint32_t top = 0;
#pragma unroll
  for (int u = 0; u < P; ++u) {
    const int64_t e = e0 + packet_off<DT>(u);
    float v[8], a[8];
    unpack<DT>(in[u], v);
#pragma unroll
    for (int i = 0; i < V; ++i) a[i] = __builtin_fabsf(v[i]);
    const uint32_t m0 = mask4(a[0], a[1], a[2], a[3]);
    uint32_t m1 = 0;
    if constexpr (V == 8) m1 = mask4(a[4], a[5], a[6], a[7]);
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const uint32_t bit = ((i < 4 ? m0 : m1) >> (8 * (i & 3))) & 1u;
      v[i] = v[i] * (bit ? 1.0f : 0.0f);  // exact in fp32; the storage rounding of pack() is the identity on w and on +-0
    }
    const Pack16 outp = pack<DT>(v);
    if (fast || e + V <= n) {
      const uint32_t m = pack_absmax<DT>(outp);
      top = m > top ? m : top;
    }
    if constexpr (fast) {
      if constexpr (V == 8) store8_nt(mask + e, m0, m1);
      else store4_nt(mask + e, m0);
      st_packet<DT, true>(w, e, n, outp);
    } else {  // n % 4 == 0: groups of four are all-in or all-out
      if (e + 4 <= n)
        for (int i = 0; i < 4; ++i) mask[e + i] = (m0 >> (8 * i)) & 1;
      if (V == 8 && e + 8 <= n)
        for (int i = 0; i < 4; ++i) mask[e + 4 + i] = (m1 >> (8 * i)) & 1;
      if (!(e + V <= n)) {  // a ragged last packet: its in-range elements one by one (their abs-max too)
        for (int i = 0; i < V && e + i < n; ++i) {
          const uint32_t bits = __float_as_uint(round_to_dtype<DT>(v[i])) & 0x7FFFFFFFu;
          top = bits > top ? bits : top;
        }
      }
      st_packet<DT, false>(w, e, n, outp);
    }
  }
  return top;
}
template <int DT>
__global__ __launch_bounds__(kBlock) void mt_mask24_apply_kernel(const moq_seg* __restrict__ segs,
                                                                 const int64_t* __restrict__ blk_start, int n_seg,
                                                                 int64_t n_chunks, uint32_t* __restrict__ chunk_max) {
  __shared__ uint32_t smem[kBlock / 64];
  if ((int64_t)blockIdx.x >= n_chunks) return;
  SegCursor cur;
  cur.init(segs, blk_start, n_seg, blockIdx.x);
  for (int64_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    cur.seek(c);
    const int64_t e0 = (c - cur.c_begin) * MOQ_MT_CHUNK;
    const bool al = al16(cur.sg.x) && (reinterpret_cast<uintptr_t>(cur.sg.y) & 7u) == 0;
    uint32_t top;
    if (al && e0 + MOQ_MT_CHUNK <= cur.sg.n)
      top = mask_apply_chunk<DT, true>(const_cast<void*>(cur.sg.x), reinterpret_cast<uint8_t*>(cur.sg.y), e0, cur.sg.n);
    else
      top = mask_apply_chunk<DT, false>(const_cast<void*>(cur.sg.x), reinterpret_cast<uint8_t*>(cur.sg.y), e0, cur.sg.n);
    if (chunk_max != nullptr) {  // (workgroup-uniform)
      top = block_max_u32(top, smem);
      if (threadIdx.x == 0) chunk_max[c] = top;
      __syncthreads();
    }
  }
}

template <int DT>
__global__ __launch_bounds__(kBlock) void mask24_kernel(const void* __restrict__ w,
                                                        uint8_t* __restrict__ mask, int64_t n) {
  const bool al = al16(w) && (reinterpret_cast<uintptr_t>(mask) & 7u) == 0;
  const int64_t n_chunks = (n + MOQ_MT_CHUNK - 1) / MOQ_MT_CHUNK;
  for (int64_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    const int64_t e0 = c * MOQ_MT_CHUNK;
    if (al && e0 + MOQ_MT_CHUNK <= n) mask_chunk<DT, true>(w, mask, e0, n);
    else mask_chunk<DT, false>(w, mask, e0, n);
  }
}
// the same over a segment table (segs[s].y = the uint8 / bool mask of segs[s].x): one launch per layer / model
template <int DT>
__global__ __launch_bounds__(kBlock) void mt_mask24_kernel(const moq_seg* __restrict__ segs,
                                                           const int64_t* __restrict__ blk_start, int n_seg,
                                                           int64_t n_chunks) {
  if ((int64_t)blockIdx.x >= n_chunks) return;
  SegCursor cur;
  cur.init(segs, blk_start, n_seg, blockIdx.x);
  for (int64_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    cur.seek(c);
    const int64_t e0 = (c - cur.c_begin) * MOQ_MT_CHUNK;
    const bool al = al16(cur.sg.x) && (reinterpret_cast<uintptr_t>(cur.sg.y) & 7u) == 0;
    if (al && e0 + MOQ_MT_CHUNK <= cur.sg.n) mask_chunk<DT, true>(cur.sg.x, reinterpret_cast<uint8_t*>(cur.sg.y), e0, cur.sg.n);
    else mask_chunk<DT, false>(cur.sg.x, reinterpret_cast<uint8_t*>(cur.sg.y), e0, cur.sg.n);
  }
}

// ================================================================================================
// real INT4 (a15) and the export packer
// ================================================================================================
// quantise one element to its biased nibble (q + 8); arithmetic in the storage dtype like the reference
template <int DT>
__device__ __forceinline__ uint32_t int4_nibble(float v, float s, int rounding) {
  float t = round_to_dtype<DT>(v * s);
  if (rounding == MOQ_ROUND_HALF_EVEN) {
    // qtensor/int4_tensor.py:72-76: round() (half-even), clamp [-8, 7], + 8
    float r = __builtin_rintf(t);
    r = __builtin_fminf(__builtin_fmaxf(r, -8.0f), 7.0f);
    return (uint32_t)(int)(r + 8.0f) & 0xFu;
  }
  // tensor_quant_gpu.cu:322-333: clamp first, roundf(v + 8) (half away), in the storage dtype
  t = __builtin_fminf(__builtin_fmaxf(t, -8.0f), 7.0f);
  t = round_to_dtype<DT>(t + 8.0f);
  return (uint32_t)(int)__builtin_roundf(t) & 0xFu;
}
// FAST layout (host-checked): x 16-byte aligned, g % kVec == 0, n % kVec == 0, out 4-byte aligned: a packet has ONE
// scale and becomes one 4-byte (2-byte for f32) store; chunk skeleton with all loads in flight.
template <int DT, bool FASTL>
__global__ __launch_bounds__(kBlock) void int4_pack_kernel(const void* __restrict__ x,
                                                           const void* __restrict__ scales,
                                                           uint8_t* __restrict__ out, int64_t n, int g,
                                                           int g_shift, int rounding) {
  constexpr int V = Elem<DT>::kVec;
  constexpr int P = Chunk<DT>::kPackets;
  const int64_t n_chunks = (n + MOQ_MT_CHUNK - 1) / MOQ_MT_CHUNK;
  GroupIndex gi;
  gi.g = (uint32_t)g;
  gi.shift = g_shift;
  for (int64_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    const int64_t e0 = c * MOQ_MT_CHUNK;
    if constexpr (FASTL) {
      gi.seek(e0);
      // (the branch-free full-chunk body of int4_unpack_kernel measured 4 % SLOWER here: 0.716 -> 0.685 of the roofline)
      Pack16 in[P];
      float sc[P];
#pragma unroll
      for (int u = 0; u < P; ++u) {
        const int64_t e = e0 + packet_off<DT>(u);
        if (e < n) {
          in[u] = load16_nt(reinterpret_cast<const char*>(x) + e * (16 / V));
          sc[u] = load1<DT>(scales, gi.at((uint32_t)packet_off<DT>(u)));
        }
      }
#pragma unroll
      for (int u = 0; u < P; ++u) {
        const int64_t e = e0 + packet_off<DT>(u);
        if (e >= n) continue;
        float v[8];
        unpack<DT>(in[u], v);
        uint32_t q[8];
        if (rounding == MOQ_ROUND_HALF_EVEN) {
          // round().clamp(-8, 7) + 8 without a float -> int conversion: rint(clamp(t)) == clamp(rint(t)) for integer bounds,
          // and adding 1.5 * 2^23 + 8 rounds to nearest-even at integer spacing with the biased nibble in the low mantissa
          // bits (a NaN product leaves v_med3_f32 as -8, the value the fmin / fmax chain gave it: nibble 0).  Only the
          // low four bits of q[] are meaningful below.
#pragma unroll
          for (int i = 0; i < V; ++i)
            q[i] = __float_as_uint(__builtin_amdgcn_fmed3f(round_to_dtype<DT>(v[i] * sc[u]), -8.0f, 7.0f) + 12582920.0f);
        } else {
#pragma unroll
          for (int i = 0; i < V; ++i) q[i] = int4_nibble<DT>(v[i], sc[u], rounding);
        }
        // byte k = (q[2k] << 4) | q[2k + 1]: one v_and + one v_lshl_or per pair, the four low bytes gathered by v_perm_b32
        uint32_t by[4] = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < V; i += 2) by[i / 2] = (q[i] << 4) | (q[i + 1] & 0xFu);
        const uint32_t lo2 = __builtin_amdgcn_perm(by[1], by[0], 0x0c0c0400u);
        if constexpr (V == 8) {
          store4_nt(out + e / 2, lo2 | __builtin_amdgcn_perm(by[3], by[2], 0x04000c0cu));
        } else {
          *reinterpret_cast<uint16_t*>(out + e / 2) = (uint16_t)lo2;
        }
      }
    } else {
      for (int u = 0; u < P; ++u) {
        const int64_t e = e0 + packet_off<DT>(u);
        uint32_t q[8];
        for (int i = 0; i < V; ++i)
          q[i] = e + i < n ? int4_nibble<DT>(load1<DT>(x, e + i), load1<DT>(scales, (e + i) / g), rounding) : 8u;
        for (int i = 0; i + 1 < V; i += 2)
          if (e + i + 1 < n) out[(e + i) / 2] = (uint8_t)((q[i] << 4) | q[i + 1]);
      }
    }
  }
}
// 4 bytes -> 8 elements per lane.  FAST layout (host-checked): q 4-byte aligned, n_bytes % 4 == 0, g % 8 == 0,
// out 16-byte aligned, 16-bit dtype: one scale and one shared exact division per packet.
template <int DT, bool FASTL>
__global__ __launch_bounds__(kBlock) void int4_unpack_kernel(const uint8_t* __restrict__ q,
                                                             const void* __restrict__ scales,
                                                             void* __restrict__ out, int64_t n_bytes,
                                                             int g, int g_shift) {
  if constexpr (FASTL && DT != MOQ_F32) {
    constexpr int P = 4;
    const int64_t n = 2 * n_bytes;  // elements
    const int64_t n_chunks = (n + MOQ_MT_CHUNK - 1) / MOQ_MT_CHUNK;
    GroupIndex gi;
    gi.g = (uint32_t)g;
    gi.shift = g_shift;
    for (int64_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
      const int64_t e0 = c * MOQ_MT_CHUNK;
      gi.seek(e0);
      // a chunk inside the tensor runs a copy of the body without per-packet bounds branches (with them hipcc waits
      // for every load on its own: one packet in flight per lane instead of the chunk's P)
      auto body = [&](auto FULL) {
      constexpr bool full = decltype(FULL)::value;
      uint32_t wv[P];
      float sc[P];
#pragma unroll
      for (int u = 0; u < P; ++u) {
        const int64_t e = e0 + packet_off<DT>(u);
        if (full || e < n) {
          wv[u] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(q + e / 2));
          sc[u] = load1<DT>(scales, gi.at((uint32_t)packet_off<DT>(u)));
        }
      }
      if constexpr (full) __builtin_amdgcn_sched_barrier(0);  // every load of the chunk is issued before the first use
#pragma unroll
      for (int u = 0; u < P; ++u) {
        const int64_t e = e0 + packet_off<DT>(u);
        if (!full && e >= n) continue;
        float v[8];
        // tensor_quant_gpu.cu:275-281: (nibble - 8) / scale in the scale dtype.  The eight integer numerators of a packet
        // share the scale: one refined reciprocal, five FMAs per quotient (SharedDiv -- exact for integers up to 2^16 and
        // a scale inside its window; the kernel spent half its issue slots on eight IEEE division sequences per packet)
        const SharedDiv sd = make_shared_div(sc[u]);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const uint32_t byte = (wv[u] >> (8 * b)) & 0xFFu;
          v[2 * b] = (float)((int)(byte >> 4) - 8);
          v[2 * b + 1] = (float)((int)(byte & 0xFu) - 8);
        }
        if (sd.fast) {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = shared_div_in_window(v[i], sd);
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = v[i] / sc[u];
        }
        store16_nt(reinterpret_cast<char*>(out) + e * 2, pack<DT>(v));
      }
      };
      if (e0 + MOQ_MT_CHUNK <= n) body(std::true_type{});
      else body(std::false_type{});
    }
  } else {
    const int64_t n_words = (n_bytes + 3) / 4;
    for (int64_t wi = (int64_t)blockIdx.x * kBlock + threadIdx.x; wi < n_words;
         wi += (int64_t)gridDim.x * kBlock) {
      for (int b = 0; b < 4; ++b) {
        const int64_t bi = wi * 4 + b;
        if (bi >= n_bytes) break;
        const uint32_t byte = q[bi];
        const float s = load1<DT>(scales, (2 * bi) / g);
        store1<DT>(out, 2 * bi, (float)((int)(byte >> 4) - 8) / s);
        store1<DT>(out, 2 * bi + 1, (float)((int)(byte & 0xFu) - 8) / s);
      }
    }
  }
}
// export packer: a lane owns 8 (4 for f32) adjacent columns of a row PAIR (2i, 2i+1): two 16-byte loads,
// one 8-byte (4-byte) store; rows of the pair are cols*elem bytes apart so both loads are fully coalesced.
// grid.x = row-pair groups (kExportPairs pairs per workgroup: 2 * kExportPairs loads in flight per lane),
// grid.y = column tiles of kBlock packets.  FAST layout (host-checked): 16-byte aligned rows, g % kVec == 0.
constexpr int kExportPairs = 2;
template <int DT, bool FASTL>
__global__ __launch_bounds__(kBlock) void int4_export_kernel(const void* __restrict__ w,
                                                             const float* __restrict__ wsf,
                                                             uint8_t* __restrict__ out, int64_t rows,
                                                             int64_t cols, int g, int g_shift) {
  constexpr int V = Elem<DT>::kVec;
  const int64_t spr = cols / g;
  if constexpr (FASTL) {
    const int64_t c0 = ((int64_t)blockIdx.y * kBlock + threadIdx.x) * V;
    if (c0 >= cols) return;
    const int64_t sidx = g_shift >= 0 ? (c0 >> g_shift) : (c0 / g);
    Pack16 pa[kExportPairs], pb[kExportPairs];
    float sa[kExportPairs], sb[kExportPairs];
#pragma unroll
    for (int k = 0; k < kExportPairs; ++k) {
      const int64_t r2 = (int64_t)blockIdx.x * kExportPairs + k;
      if (2 * r2 + 1 < rows) {
        pa[k] = load16_nt(reinterpret_cast<const char*>(w) + ((2 * r2) * cols + c0) * (16 / V));
        pb[k] = load16_nt(reinterpret_cast<const char*>(w) + ((2 * r2 + 1) * cols + c0) * (16 / V));
        sa[k] = wsf[(2 * r2) * spr + sidx];
        sb[k] = wsf[(2 * r2 + 1) * spr + sidx];
      }
    }
#pragma unroll
    for (int k = 0; k < kExportPairs; ++k) {
      const int64_t r2 = (int64_t)blockIdx.x * kExportPairs + k;
      if (2 * r2 + 1 >= rows) continue;
      float a[8], b[8];
      unpack<DT>(pa[k], a);
      unpack<DT>(pb[k], b);
      // quant_utils.py:800-805: (w / wsf).round().clamp(-8, 7) with fp32 division; the 8 quotients of a row share
      // their denominator (exact shared division, moq_common.h)
      const SharedDiv da = make_shared_div(sa[k]), db = make_shared_div(sb[k]);
      // the shared division is exact for |numerator| <= 2^16 and a scale inside its window; anything else (inf / NaN among
      // them) takes the IEEE divide
      const uint32_t ma = pack_absmax<DT>(pa[k]), mb = pack_absmax<DT>(pb[k]);
      uint32_t byte[8];
      if (da.fast && db.fast && (ma > mb ? ma : mb) <= 0x47800000u) {
        // rint + clamp + two's-complement nibble through the low mantissa bits of q + 1.5 * 2^23 (see int4_pack_kernel);
        // only the low byte of byte[] is meaningful
#pragma unroll
        for (int i = 0; i < V; i += 2) {
          typedef float f32x2 __attribute__((ext_vector_type(2)));
          const f32x2 na = {a[i], a[i + 1]}, nb = {b[i], b[i + 1]};
          const f32x2 ya = {da.y, da.y}, yb = {db.y, db.y}, nda = {-da.d, -da.d}, ndb = {-db.d, -db.d};
          const f32x2 a0 = na * ya, b0 = nb * yb;
          const f32x2 a1 = __builtin_elementwise_fma(__builtin_elementwise_fma(nda, a0, na), ya, a0);
          const f32x2 b1 = __builtin_elementwise_fma(__builtin_elementwise_fma(ndb, b0, nb), yb, b0);
          const f32x2 qa = __builtin_elementwise_fma(__builtin_elementwise_fma(nda, a1, na), ya, a1);
          const f32x2 qb = __builtin_elementwise_fma(__builtin_elementwise_fma(ndb, b1, nb), yb, b1);
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const uint32_t la = __float_as_uint(__builtin_amdgcn_fmed3f(j ? qa.y : qa.x, -8.0f, 7.0f) + 12582912.0f);
            const uint32_t lb = __float_as_uint(__builtin_amdgcn_fmed3f(j ? qb.y : qb.x, -8.0f, 7.0f) + 12582912.0f);
            byte[i + j] = (lb << 4) | (la & 0xFu);
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < V; ++i) {
          float qa = __builtin_rintf(a[i] / sa[k]);
          float qb = __builtin_rintf(b[i] / sb[k]);
          qa = __builtin_fminf(__builtin_fmaxf(qa, -8.0f), 7.0f);
          qb = __builtin_fminf(__builtin_fmaxf(qb, -8.0f), 7.0f);
          byte[i] = ((uint32_t)(int)qa & 0xFu) | (((uint32_t)(int)qb & 0xFu) << 4);
        }
      }
      const uint32_t w0 = __builtin_amdgcn_perm(byte[1], byte[0], 0x0c0c0400u) | __builtin_amdgcn_perm(byte[3], byte[2], 0x04000c0cu);
      uint8_t* dst = out + r2 * cols + c0;
      if constexpr (V == 8) {
        store8_nt(dst, w0, __builtin_amdgcn_perm(byte[5], byte[4], 0x0c0c0400u) | __builtin_amdgcn_perm(byte[7], byte[6], 0x04000c0cu));
      } else {
        store4_nt(dst, w0);
      }
    }
  } else {
    const int64_t n_items = (rows / 2) * cols;
    for (int64_t it = (int64_t)blockIdx.x * kBlock + threadIdx.x; it < n_items; it += (int64_t)gridDim.x * kBlock) {
      const int64_t r2 = it / cols, c = it % cols;
      const float a = load1<DT>(w, (2 * r2) * cols + c), b = load1<DT>(w, (2 * r2 + 1) * cols + c);
      const float sa = wsf[(2 * r2) * spr + c / g], sb = wsf[(2 * r2 + 1) * spr + c / g];
      float qa = __builtin_rintf(a / sa), qb = __builtin_rintf(b / sb);
      qa = __builtin_fminf(__builtin_fmaxf(qa, -8.0f), 7.0f);
      qb = __builtin_fminf(__builtin_fmaxf(qb, -8.0f), 7.0f);
      out[r2 * cols + c] = (uint8_t)(((uint32_t)(int)qa & 0xFu) | (((uint32_t)(int)qb & 0xFu) << 4));
    }
  }
}

// ================================================================================================
// column scale fold (a11/a12 postprocess)
// ================================================================================================
// MODE 0: y = dtype(w * mul[c]);  MODE 1: y = dtype((w * mul[c]) / div[c])  (fp32 multiply, fp32 IEEE divide, one
// rounding to the storage dtype: _apply_weight_pre_quant_scale, model_calib.py:1208-1216, and _update_pre_quant_scale
// of the export resmooth step, export/quant_utils.py:1285-1296).  Chunk skeleton; the column of a packet comes from
// one 64-bit division per chunk.  FAST layout (host-checked): 16-byte aligned w / y / mul / div, cols % kVec == 0.
template <int DT, int MODE, bool FASTL>
__global__ __launch_bounds__(kBlock) void scale_cols_kernel(const void* __restrict__ w,
                                                            const float* __restrict__ mul,
                                                            const float* __restrict__ div,
                                                            void* __restrict__ y, int64_t rows,
                                                            int64_t cols) {
  constexpr int V = Elem<DT>::kVec;
  constexpr int P = Chunk<DT>::kPackets;
  const int64_t n = rows * cols;
  if constexpr (FASTL) {
    const int64_t n_chunks = (n + MOQ_MT_CHUNK - 1) / MOQ_MT_CHUNK;
    // column of packet u = (col0 + toff[u]) mod cols with col0 = e0 mod cols (uniform, once per chunk) and
    // toff[u] = packet_off(u) mod cols (once per thread): both below cols, so one conditional subtraction wraps
    int64_t toff[P];
#pragma unroll
    for (int u = 0; u < P; ++u) toff[u] = (int64_t)packet_off<DT>(u) % cols;
    for (int64_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
      const int64_t e0 = c * MOQ_MT_CHUNK;
      const int64_t col0 = e0 % cols;  // uniform
      // a chunk inside the tensor runs a copy of the body without per-packet bounds branches (with them hipcc waits for
      // every load on its own); the scale vectors of all packets are requested together with the data
      auto body = [&](auto FULL) {
        constexpr bool full = decltype(FULL)::value;
        Pack16 in[P];
        float4 m[P][V / 4], d[P][V / 4];
#pragma unroll
        for (int u = 0; u < P; ++u) {
          const int64_t e = e0 + packet_off<DT>(u);
          if (full || e < n) {
            in[u] = load16_nt(reinterpret_cast<const char*>(w) + e * (16 / V));
            int64_t col = col0 + toff[u];
            col = col >= cols ? col - cols : col;
#pragma unroll
            for (int i = 0; i < V / 4; ++i) {
              m[u][i] = *reinterpret_cast<const float4*>(mul + col + 4 * i);
              if constexpr (MODE == 1) d[u][i] = *reinterpret_cast<const float4*>(div + col + 4 * i);
            }
          }
        }
        if constexpr (full) __builtin_amdgcn_sched_barrier(0);  // every load of the chunk is issued before the first use
#pragma unroll
        for (int u = 0; u < P; ++u) {
          const int64_t e = e0 + packet_off<DT>(u);
          if (!full && e >= n) continue;
          float v[8];
          unpack<DT>(in[u], v);
#pragma unroll
          for (int i = 0; i < V; i += 4) {
            const float4 mm = m[u][i / 4];
            v[i] *= mm.x; v[i + 1] *= mm.y; v[i + 2] *= mm.z; v[i + 3] *= mm.w;
            if constexpr (MODE == 1) {
              const float4 dd = d[u][i / 4];
              v[i] /= dd.x; v[i + 1] /= dd.y; v[i + 2] /= dd.z; v[i + 3] /= dd.w;
            }
          }
          store16_nt(reinterpret_cast<char*>(y) + e * (16 / V), pack<DT>(v));
        }
      };
      if (e0 + MOQ_MT_CHUNK <= n) body(std::true_type{});
      else body(std::false_type{});
    }
  } else {
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n; e += (int64_t)gridDim.x * kBlock) {
      float v = load1<DT>(w, e) * mul[e % cols];
      if constexpr (MODE == 1) v = v / div[e % cols];
      store1<DT>(y, e, v);
    }
  }
}

// y[a, r, c] = dtype(w[r, c] * s[a, c]) for a < n_scales: ONE read of w, n_scales writes (the pre-scaled activation
// copies of an AWQ search step: x / s_alpha for every contender of a linear, model_calib.py:1489-1495).  Requires the fast
// layout (checked by the host entry: 16-byte aligned pointers, cols % kVec == 0).
// Round 5: on the chunk skeleton (round 1's form was a grid-stride loop with one packet and one 64-bit modulo per
// iteration, ordinary stores: 92.7 us per call in the HF-topology AWQ flow = 2.2 TB/s, 10 240 calls = 3.7 % of its kernel
// time).  One chunk per workgroup, the chunk's packets requested before the first use, a candidate's scale vectors for
// all packets requested together, non-temporal stores (every copy is written once and read once by the error GEMM): 4.0 TB/s
// on 67 MB x (1 + 5 candidates) -- six address windows per launch, one read and five written; prefetching the next
// candidate's scales under the current one's stores changed nothing (measured), the windows are what is left.
template <int DT>
__global__ __launch_bounds__(kBlock) void scale_cols_multi_kernel(const void* __restrict__ w,
                                                                  const float* __restrict__ s,
                                                                  void* __restrict__ y, int64_t rows,
                                                                  int64_t cols, int n_scales) {
  constexpr int V = Elem<DT>::kVec;
  constexpr int P = Chunk<DT>::kPackets;
  const int64_t n = rows * cols;
  const int64_t n_chunks = (n + MOQ_MT_CHUNK - 1) / MOQ_MT_CHUNK;
  int64_t toff[P];
#pragma unroll
  for (int u = 0; u < P; ++u) toff[u] = (int64_t)packet_off<DT>(u) % cols;
  for (int64_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    const int64_t e0 = c * MOQ_MT_CHUNK;
    const int64_t col0 = e0 % cols;  // uniform: one 64-bit division per chunk
    auto body = [&](auto FULL) {
      constexpr bool full = decltype(FULL)::value;
      Pack16 in[P];
      int64_t col[P];
#pragma unroll
      for (int u = 0; u < P; ++u) {
        const int64_t e = e0 + packet_off<DT>(u);
        col[u] = col0 + toff[u];
        col[u] = col[u] >= cols ? col[u] - cols : col[u];
        if (full || e < n) in[u] = load16_nt(reinterpret_cast<const char*>(w) + e * (16 / V));
      }
      float v[P][8];
#pragma unroll
      for (int u = 0; u < P; ++u) unpack<DT>(in[u], v[u]);
      for (int a = 0; a < n_scales; ++a) {
        const float* sa = s + (int64_t)a * cols;
        float4 m[P][V / 4];
#pragma unroll
        for (int u = 0; u < P; ++u)
#pragma unroll
          for (int i = 0; i < V / 4; ++i) m[u][i] = *reinterpret_cast<const float4*>(sa + col[u] + 4 * i);
        char* ya = reinterpret_cast<char*>(y) + (int64_t)a * n * (16 / V);
#pragma unroll
        for (int u = 0; u < P; ++u) {
          const int64_t e = e0 + packet_off<DT>(u);
          if (!full && e >= n) continue;
          float o[8];
#pragma unroll
          for (int i = 0; i < V; i += 4) {
            const float4 mm = m[u][i / 4];
            o[i] = v[u][i] * mm.x; o[i + 1] = v[u][i + 1] * mm.y; o[i + 2] = v[u][i + 2] * mm.z; o[i + 3] = v[u][i + 3] * mm.w;
          }
          store16_nt(ya + e * (16 / V), pack<DT>(o));
        }
      }
    };
    if (e0 + MOQ_MT_CHUNK <= n) body(std::true_type{});
    else body(std::false_type{});
  }
}


// convert_to_exmy (tensor_quant_mx.cu:398, tensor_quant_mx.h:163-186): one fp32 value -> the nearest value of the
// element format, no scale.  The pybind surface exposes it for scalars; here it is elementwise over an array.
__global__ __launch_bounds__(kBlock) void mx_convert_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n,
                                                            int fmt) {
  const MxFmt f = mx_fmt(fmt);
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const float v = x[i];
    const float r = mx_round_abs(__builtin_fabsf(v), f);
    // table formats take the sign from `x < 0` (h:104-160: -0 and NaN count as positive), the fp8 / int8
    // conversions keep the sign bit
    const bool neg = f.kind == 0 ? v < 0.0f : (__float_as_uint(v) >> 31) != 0;
    y[i] = neg ? -r : r;
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
  if (scale_fmt == MOQ_E8M0 ? global_amax != nullptr : mx_fmt(scale_fmt).kind < 0) {
    set_error("moq_mx_fused_amax_convert: block scales are E8M0 (no global amax) or an element format "
              "(E4M3, ...; optional global amax)");
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
  if (scale_fmt == MOQ_E8M0 && aligned && cols % block == 0 && block % vec == 0 && lpg <= 64 && (lpg & (lpg - 1)) == 0) {
    const int grid = copy_grid((n + MOQ_MT_CHUNK - 1) / MOQ_MT_CHUNK);
#define MOQ_MX_LAUNCH(L, F) \
  MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((mx_kernel<DT, L, F>), dim3(grid), dim3(kBlock), copy_lds_1t(24 * 1024), S(stream), x, y, n, fmt))
#define MOQ_MX_CASE(L)                                                  \
  case L:                                                               \
    if (L <= 8 && fmt == MOQ_E2M1) { MOQ_MX_LAUNCH(L, (L <= 8 ? MOQ_E2M1 : -1)); }       \
    else if (L <= 8 && fmt == MOQ_E4M3) { MOQ_MX_LAUNCH(L, (L <= 8 ? MOQ_E4M3 : -1)); }  \
    else { MOQ_MX_LAUNCH(L, -1); }                                      \
    break;
    switch (lpg) {
      MOQ_MX_CASE(1) MOQ_MX_CASE(2) MOQ_MX_CASE(4) MOQ_MX_CASE(8) MOQ_MX_CASE(16) MOQ_MX_CASE(32) MOQ_MX_CASE(64)
      default: set_error("unreachable"); return MOQ_ERR_INVALID;
    }
#undef MOQ_MX_CASE
#undef MOQ_MX_LAUNCH
  } else if (scale_fmt != MOQ_E8M0 && aligned && cols % block == 0 && block % vec == 0 && lpg <= 8 && (lpg & (lpg - 1)) == 0) {
    const int grid = copy_grid((n + MOQ_MT_CHUNK - 1) / MOQ_MT_CHUNK);
#define MOQ_MX2_LAUNCH(L, F, SF)                                                                                       \
  MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((mx_two_level_kernel<DT, L, F, SF>), dim3(grid), dim3(kBlock), copy_lds_1t(),  \
                                            S(stream), x, y, n, fmt, scale_fmt, global_amax))
#define MOQ_MX2_CASE(L)                                                                  \
  case L:                                                                                \
    if (fmt == MOQ_E2M1 && scale_fmt == MOQ_E4M3) { MOQ_MX2_LAUNCH(L, MOQ_E2M1, MOQ_E4M3); } \
    else { MOQ_MX2_LAUNCH(L, -1, -1); }                                                   \
    break;
    switch (lpg) {
      MOQ_MX2_CASE(1) MOQ_MX2_CASE(2) MOQ_MX2_CASE(4) MOQ_MX2_CASE(8)
      default: set_error("unreachable"); return MOQ_ERR_INVALID;
    }
#undef MOQ_MX2_CASE
#undef MOQ_MX2_LAUNCH
  } else {
    const int64_t nb = rows * ((cols + block - 1) / block);
    const int grid = stream_grid(kBlock, nb);
    MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((mx_generic_kernel<DT>), dim3(grid), dim3(kBlock), 0, S(stream),
                                              x, y, rows, cols, block, fmt, scale_fmt, global_amax));
  }
  return check_launch("moq_mx_fused_amax_convert");
}

extern "C" int moq_mt_mx_fused_amax_convert(const moq_seg* segs, const int64_t* blk_start, int n_seg,
                                            int64_t n_chunks, int block, int dt, int fmt, void* stream) {
  if (n_seg < 0 || n_chunks < 0 || (n_seg > 0 && (segs == nullptr || blk_start == nullptr))) {
    set_error("moq_mt_mx_fused_amax_convert: null pointer or negative size");
    return MOQ_ERR_INVALID;
  }
  if (mx_fmt(fmt).kind < 0) {
    set_error("moq_mt_mx_fused_amax_convert: unknown element format %d", fmt);
    return MOQ_ERR_INVALID;
  }
  const int vec = dt == MOQ_F32 ? 4 : 8;
  const int lpg = block / vec;
  if (block <= 0 || block % vec != 0 || lpg > 8 || (lpg & (lpg - 1)) != 0) {
    set_error("moq_mt_mx_fused_amax_convert: block sizes %d..%d (powers of two) are supported, got %d", vec, 8 * vec, block);
    return MOQ_ERR_UNSUPPORTED;
  }
  if (n_seg == 0 || n_chunks == 0) return MOQ_OK;
  const int grid = copy_grid(n_chunks);
#define MOQ_MTMX_LAUNCH(L, F) \
  MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((mt_mx_kernel<DT, L, F>), dim3(grid), dim3(kBlock), copy_lds(24 * 1024), S(stream), segs, blk_start, n_seg, n_chunks, fmt))
#define MOQ_MTMX_CASE(L)                                     \
  case L:                                                    \
    if (fmt == MOQ_E2M1) { MOQ_MTMX_LAUNCH(L, MOQ_E2M1); }   \
    else if (fmt == MOQ_E4M3) { MOQ_MTMX_LAUNCH(L, MOQ_E4M3); } \
    else { MOQ_MTMX_LAUNCH(L, -1); }                         \
    break;
  switch (lpg) {
    MOQ_MTMX_CASE(1) MOQ_MTMX_CASE(2) MOQ_MTMX_CASE(4) MOQ_MTMX_CASE(8)
    default: set_error("unreachable"); return MOQ_ERR_INVALID;
  }
#undef MOQ_MTMX_CASE
#undef MOQ_MTMX_LAUNCH
  return check_launch("moq_mt_mx_fused_amax_convert");
}

extern "C" int moq_mt_fold_mx_fused(const moq_seg* segs, const int64_t* blk_start, const moq_fold_seg* side, int n_seg,
                                    int64_t n_chunks, int block, int dt, int fmt, void* stream) {
  if (n_seg < 0 || n_chunks < 0 || (n_seg > 0 && (segs == nullptr || blk_start == nullptr || side == nullptr))) {
    set_error("moq_mt_fold_mx_fused: null pointer or negative size");
    return MOQ_ERR_INVALID;
  }
  if (mx_fmt(fmt).kind < 0) {
    set_error("moq_mt_fold_mx_fused: unknown element format %d", fmt);
    return MOQ_ERR_INVALID;
  }
  const int vec = dt == MOQ_F32 ? 4 : 8;
  const int lpg = block / vec;
  if (block <= 0 || block % vec != 0 || lpg > 8 || (lpg & (lpg - 1)) != 0) {
    set_error("moq_mt_fold_mx_fused: block sizes %d..%d (powers of two) are supported, got %d", vec, 8 * vec, block);
    return MOQ_ERR_UNSUPPORTED;
  }
  if (n_seg == 0 || n_chunks == 0) return MOQ_OK;
  const int grid = copy_grid(n_chunks);
#define MOQ_MTFM_LAUNCH(L, F) \
  MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((mt_fold_mx_kernel<DT, L, F>), dim3(grid), dim3(kBlock), copy_lds(24 * 1024), S(stream), segs, blk_start, side, n_seg, n_chunks, fmt))
#define MOQ_MTFM_CASE(L)                                     \
  case L:                                                    \
    if (fmt == MOQ_E2M1) { MOQ_MTFM_LAUNCH(L, MOQ_E2M1); }   \
    else { MOQ_MTFM_LAUNCH(L, -1); }                         \
    break;
  switch (lpg) {
    MOQ_MTFM_CASE(1) MOQ_MTFM_CASE(2) MOQ_MTFM_CASE(4) MOQ_MTFM_CASE(8)
    default: set_error("unreachable"); return MOQ_ERR_INVALID;
  }
#undef MOQ_MTFM_CASE
#undef MOQ_MTFM_LAUNCH
  return check_launch("moq_mt_fold_mx_fused");
}

extern "C" int moq_hist_abs(const void* x, int64_t n, int dt, unsigned long long* counts, int bins,
                            float max_edge, int skip_zeros, void* stream) {
  if (n < 0 || bins <= 0 || counts == nullptr || (n > 0 && x == nullptr)) {
    set_error("moq_hist_abs: bad arguments");
    return MOQ_ERR_INVALID;
  }
  if (n == 0) return MOQ_OK;
  // 16-bit inputs: the histogram stage of the fused input-quantizer kernel (moq_inputq.hip), which tabulates the
  // binning rule per |x| pattern once per workgroup instead of evaluating it per element.  (Experiment build:
  // MOQ_TUNE_HIST=0 selects the arithmetic kernel below, the fp32 path, for A/B measurements.)
  const bool lut_hist = moq_tune("MOQ_TUNE_HIST", 1) != 0;
  // Below ~12 M elements the table does not pay: every workgroup tabulates 32 768 patterns before its first element, which
  // is as much arithmetic as binning 32 K elements directly (kernel-only, 2048 bins: 8.4 MB 15.2 us with the table, 10.0 us
  // without; 33.5 MB 21.1 / 21.1 us; 67 MB 26.9 / 31.9 us -- profiles/r03_amax_flow_sizes.md).  Same hist_bin() either way.
  const int64_t kLutHistMinElems = moq_tune("MOQ_TUNE_HIST_MIN_ELEMS", 12ll << 20);
  if (lut_hist && dt != MOQ_F32 && bins < kHistMaxLdsBins && n >= kLutHistMinElems)
    return moq_input_quant(x, nullptr, nullptr, 1, n, dt, nullptr, nullptr, 0, 0, 0, 0, counts, bins, max_edge,
                           skip_zeros, stream);
  // <= 512 workgroups of 1024 threads: the flush costs `bins` 64-bit global atomics per workgroup
  int64_t blocks = ((n + MOQ_MT_CHUNK - 1) / MOQ_MT_CHUNK + 3) / 4;
  if (blocks > 512) blocks = 512;
  if (bins < kHistMaxLdsBins) {
    int rshift = 0;
    while (rshift < 3 && ((int64_t)(bins + 1) << (rshift + 1)) <= kHistMaxLdsBins + 8) ++rshift;
    const size_t lds = ((size_t)(bins + 1) << rshift) * 4;
    // the shared-denominator division is exact for numerators up to 2^16 and denominators in [2^-60, 2^60]
    const bool shared = max_edge >= 0x1p-60f && max_edge <= 0x1p60f && max_edge * (float)bins <= 65536.0f;
    if (shared) {
      MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((hist_kernel<DT, true>), dim3((int)blocks), dim3(kHistBlock), lds, S(stream),
                                                x, n, counts, bins, max_edge, skip_zeros, rshift));
    } else {
      MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((hist_kernel<DT, false>), dim3((int)blocks), dim3(kHistBlock), lds, S(stream),
                                                x, n, counts, bins, max_edge, skip_zeros, rshift));
    }
  } else {
    MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((hist_global_kernel<DT>), dim3(stream_grid(kBlock, n)), dim3(kBlock), 0,
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
  const int grid = copy_grid((n + MOQ_MT_CHUNK - 1) / MOQ_MT_CHUNK);
  MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((mask24_kernel<DT>), dim3(grid), dim3(kBlock), copy_lds_1t(32 * 1024), S(stream), w,
                                            mask, n));
  return check_launch("moq_mask_2to4");
}

extern "C" int moq_mt_mask_2to4(const moq_seg* segs, const int64_t* blk_start, int n_seg, int64_t n_chunks, int dt,
                                void* stream) {
  if (n_seg < 0 || n_chunks < 0 || (n_seg > 0 && (segs == nullptr || blk_start == nullptr))) {
    set_error("moq_mt_mask_2to4: null pointer or negative size");
    return MOQ_ERR_INVALID;
  }
  if (n_seg == 0 || n_chunks == 0) return MOQ_OK;
  MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((mt_mask24_kernel<DT>), dim3(copy_grid(n_chunks)), dim3(kBlock), copy_lds(),
                                            S(stream), segs, blk_start, n_seg, n_chunks));
  return check_launch("moq_mt_mask_2to4");
}

// stage 2 of the two-stage per-tensor abs-max (moq_stream.hip): C++ linkage, not part of the C-ABI
int moq_mt_amax_fold_launch(const moq_seg* segs, const int64_t* blk_start, int n_seg, const void* chunk_scratch, void* stream);
extern "C" int moq_mt_mask_2to4_apply(const moq_seg* segs, const int64_t* blk_start, int n_seg, int64_t n_chunks, int dt,
                                      void* chunk_scratch, void* stream) {
  if (n_seg < 0 || n_chunks < 0 || (n_seg > 0 && (segs == nullptr || blk_start == nullptr))) {
    set_error("moq_mt_mask_2to4_apply: null pointer or negative size");
    return MOQ_ERR_INVALID;
  }
  if (n_seg == 0 || n_chunks == 0) return MOQ_OK;
  MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((mt_mask24_apply_kernel<DT>), dim3(copy_grid(n_chunks)), dim3(kBlock), copy_lds(),
                                            S(stream), segs, blk_start, n_seg, n_chunks,
                                            reinterpret_cast<uint32_t*>(chunk_scratch)));
  int rc = check_launch("moq_mt_mask_2to4_apply");
  if (rc != MOQ_OK || chunk_scratch == nullptr) return rc;
  return moq_mt_amax_fold_launch(segs, blk_start, n_seg, chunk_scratch, stream);  // per-chunk maxima -> segs[s].amax[0]
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
  const int grid = copy_grid((n + MOQ_MT_CHUNK - 1) / MOQ_MT_CHUNK);
  const bool fastl = (reinterpret_cast<uintptr_t>(x) & 15u) == 0 && (g % vec) == 0 && (n % vec) == 0 &&
                     (reinterpret_cast<uintptr_t>(out) & 3u) == 0;
  const int gs = log2_or_neg(g);
  if (fastl) {
    MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((int4_pack_kernel<DT, true>), dim3(grid), dim3(kBlock), copy_lds_1t(), S(stream),
                                              x, scales, out, n, g, gs, rounding));
  } else {
    MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((int4_pack_kernel<DT, false>), dim3(grid), dim3(kBlock), copy_lds_1t(), S(stream),
                                              x, scales, out, n, g, gs, rounding));
  }
  return check_launch("moq_int4_pack");
}

extern "C" int moq_int4_unpack(const uint8_t* q, const void* scales, void* out, int64_t n_bytes, int g,
                               int dt, void* stream) {
  if (n_bytes < 0 || g <= 0 || g % 2 != 0 || (n_bytes > 0 && (q == nullptr || scales == nullptr || out == nullptr))) {
    set_error("moq_int4_unpack: bad arguments");
    return MOQ_ERR_INVALID;
  }
  if (n_bytes == 0) return MOQ_OK;
  const bool fastl = dt != MOQ_F32 && (reinterpret_cast<uintptr_t>(q) & 3u) == 0 && (g % 8) == 0 &&
                     (n_bytes % 4) == 0 && (reinterpret_cast<uintptr_t>(out) & 15u) == 0;
  const int gs = log2_or_neg(g);
  if (fastl) {
    const int grid = copy_grid((2 * n_bytes + MOQ_MT_CHUNK - 1) / MOQ_MT_CHUNK);
    MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((int4_unpack_kernel<DT, true>), dim3(grid), dim3(kBlock), copy_lds_1t(), S(stream),
                                              q, scales, out, n_bytes, g, gs));
  } else {
    const int grid = stream_grid(kBlock, (n_bytes + 3) / 4);
    MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((int4_unpack_kernel<DT, false>), dim3(grid), dim3(kBlock), 0, S(stream),
                                              q, scales, out, n_bytes, g, gs));
  }
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
  const bool fastl = (reinterpret_cast<uintptr_t>(w) & 15u) == 0 && (cols % vec) == 0 && (g % vec) == 0 &&
                     (reinterpret_cast<uintptr_t>(out) & 7u) == 0;
  const int gs = log2_or_neg(g);
  const int64_t col_tiles = (cols / vec + kBlock - 1) / kBlock;
  if (fastl && col_tiles <= 65535) {
    dim3 grid((unsigned)((rows / 2 + kExportPairs - 1) / kExportPairs), (unsigned)col_tiles);
    MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((int4_export_kernel<DT, true>), grid, dim3(kBlock), 0, S(stream),
                                              w, wsf, out, rows, cols, g, gs));
  } else {
    const int grid = stream_grid(kBlock, (rows / 2) * cols);
    MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((int4_export_kernel<DT, false>), dim3(grid), dim3(kBlock), 0, S(stream),
                                              w, wsf, out, rows, cols, g, gs));
  }
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
  const bool fastl = ((reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(s)) & 15u) == 0 &&
                     (cols % vec) == 0;
  if (fastl) {
    MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((scale_cols_kernel<DT, 0, true>), dim3(copy_grid((n + MOQ_MT_CHUNK - 1) / MOQ_MT_CHUNK)),
                                              dim3(kBlock), copy_lds_1t(32 * 1024), S(stream), w, s, (const float*)nullptr, y, rows, cols));
  } else {
    MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((scale_cols_kernel<DT, 0, false>), dim3(stream_grid(kBlock, n)), dim3(kBlock), 0,
                                              S(stream), w, s, (const float*)nullptr, y, rows, cols));
  }
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
  MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((scale_cols_multi_kernel<DT>), dim3(copy_grid((n + MOQ_MT_CHUNK - 1) / MOQ_MT_CHUNK)),
                                            dim3(kBlock), 0, S(stream), w, s, y, rows, cols, n_scales));
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
  const bool fastl = ((reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(mul) |
                       reinterpret_cast<uintptr_t>(div)) & 15u) == 0 && (cols % vec) == 0;
  if (fastl) {
    MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((scale_cols_kernel<DT, 1, true>), dim3(copy_grid((n + MOQ_MT_CHUNK - 1) / MOQ_MT_CHUNK)),
                                              dim3(kBlock), copy_lds_1t(), S(stream), w, mul, div, y, rows, cols));
  } else {
    MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((scale_cols_kernel<DT, 1, false>), dim3(stream_grid(kBlock, n)), dim3(kBlock), 0,
                                              S(stream), w, mul, div, y, rows, cols));
  }
  return check_launch("moq_rescale_cols");
}

extern "C" int moq_mx_convert(const float* x, float* y, int64_t n, int fmt, void* stream) {
  if (n < 0 || (n > 0 && (x == nullptr || y == nullptr))) {
    set_error("moq_mx_convert: null pointer or negative size");
    return MOQ_ERR_INVALID;
  }
  if (mx_fmt(fmt).kind < 0) {
    set_error("moq_mx_convert: unknown element format %d", fmt);
    return MOQ_ERR_INVALID;
  }
  if (n == 0) return MOQ_OK;
  hipLaunchKernelGGL(mx_convert_kernel, dim3(stream_grid(kBlock, n)), dim3(kBlock), 0, S(stream), x, y, n, fmt);
  return check_launch("moq_mx_convert");
}
