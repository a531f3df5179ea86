// moq_reduce.hip -- abs-max over all dimensions but one (per-channel / per-group / per-column amax) and
// the fused column statistics of an activation batch (AWQ act-scale numerator + SmoothQuant channel amax).
//
// Three layouts of the contiguous [outer, axis_size, inner] view, each with its own access pattern:
//   rows    (inner >= one 16-byte packet): every (o, a) pair is a contiguous run of `inner` elements.
//           A wave streams a row segment with 1-KiB instructions and finishes with a 64-lane butterfly.
//   columns (inner == 1): out[a] reduces a strided column of the [outer, axis_size] matrix.  A lane owns
//           8 (bf16) adjacent columns, walks down the rows with 16-byte loads and keeps 4 packed
//           v_pk_max_u16 accumulators; one atomicMax per column per row-block.
//   generic (anything else / unaligned): one element per lane, correctness path.
// All of them are HBM-read bound: 2 B/element (bf16) in, ~0 out.
#include "moq_common.h"

namespace moq {

constexpr int kColRows = 16;  // rows per workgroup in the column kernels

__device__ __forceinline__ bool al16(const void* p) {
  return (reinterpret_cast<uintptr_t>(p) & 15u) == 0;
}

// ---------------------------------------------------------------- rows
// grid: blockIdx.x enumerates (row, segment) work items four per block (one per wave)
template <int DT>
__global__ __launch_bounds__(kBlock) void amax_rows_kernel(const void* __restrict__ x, int64_t n_rows,
                                                           int64_t axis_size, int64_t inner,
                                                           int64_t seg_packets, int64_t segs_per_row,
                                                           uint32_t* __restrict__ out) {
  constexpr int V = Elem<DT>::kVec;
  const int lane = threadIdx.x & 63;
  const int64_t wave_id = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const int64_t n_items = n_rows * segs_per_row;
  const int64_t n_waves = (int64_t)gridDim.x * (kBlock / 64);
  const int64_t row_packets = inner / V;
  const char* base = reinterpret_cast<const char*>(x);
  for (int64_t item = wave_id; item < n_items; item += n_waves) {
    const int64_t row = item / segs_per_row, seg = item % segs_per_row;
    const int64_t p0 = seg * seg_packets;
    const int64_t p1 = p0 + seg_packets < row_packets ? p0 + seg_packets : row_packets;
    const char* rp = base + row * inner * (16 / V);
    uint32_t acc = 0;
    int64_t p = p0 + lane;
    // 8 independent 16-byte loads in flight per lane, read-once stream (non-temporal)
    for (; p + 7 * 64 < p1; p += 8 * 64) {
      Pack16 q[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) q[u] = load16_nt(rp + (p + u * 64) * 16);
      uint32_t m[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) m[u] = pack_absmax<DT>(q[u]);
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = acc > m[u] ? acc : m[u];
    }
    for (; p < p1; p += 64) {
      uint32_t m = pack_absmax<DT>(load16_nt(rp + p * 16));
      acc = acc > m ? acc : m;
    }
    acc = group_max_u32<64>(acc);
    if (lane == 0) atomicMax(&out[row % axis_size], acc);
  }
}

// ---------------------------------------------------------------- columns
// A workgroup owns a tile of kColTile = 64 lanes x kVec adjacent columns and kColRowsWG = 64 rows: wave w walks rows
// r0 + 16 w .. +16 with sixteen 16-byte loads per lane in flight (a wave instruction = 1 KiB of one row), the four
// waves are combined through LDS in wave order, and ONE result per column leaves the workgroup: a plain store of
// the partial |x| sum (deterministic two-stage sum) and one atomicMax of the abs-max pattern.
// grid.x = column tiles, grid.y = row blocks.  partial: [n_rowblk, cols] fp32.
constexpr int kColRowsWG = 64;
template <int DT, bool SUM, bool AMAX>
__global__ __launch_bounds__(kBlock) void col_stats_kernel(const void* __restrict__ x, int64_t rows,
                                                           int64_t cols, uint32_t* __restrict__ amax_out,
                                                           float* __restrict__ partial) {
  constexpr int V = Elem<DT>::kVec;
  constexpr int kTileCols = 64 * V;
  __shared__ float s_sum[SUM ? 4 : 1][SUM ? kTileCols : 1];
  __shared__ uint32_t s_max[AMAX ? 4 : 1][AMAX ? kTileCols : 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t c0 = ((int64_t)blockIdx.x * 64 + lane) * V;
  const int64_t r0 = (int64_t)blockIdx.y * kColRowsWG + wave * kColRows;
  const char* base = reinterpret_cast<const char*>(x);
  const int64_t row_bytes = cols * (16 / V);
  uint32_t am[V];
  float sm[V];
#pragma unroll
  for (int i = 0; i < V; ++i) { am[i] = 0; sm[i] = 0.0f; }
  if (c0 < cols) {
    Pack16 pk[kColRows];
#pragma unroll
    for (int r = 0; r < kColRows; ++r)
      if (r0 + r < rows) pk[r] = load16_nt(base + (r0 + r) * row_bytes + c0 * (16 / V));
#pragma unroll
    for (int r = 0; r < kColRows; ++r) {
      if (r0 + r < rows) {
        float f[8];
        unpack<DT>(pk[r], f);
#pragma unroll
        for (int i = 0; i < V; ++i) {
          const uint32_t a = absbits(f[i]);
          if (AMAX) am[i] = a > am[i] ? a : am[i];
          if (SUM) sm[i] += __uint_as_float(a);  // rows are added in row order: deterministic
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < V; ++i) {
    if (SUM) s_sum[wave][lane * V + i] = sm[i];
    if (AMAX) s_max[wave][lane * V + i] = am[i];
  }
  __syncthreads();
  for (int j = threadIdx.x; j < kTileCols; j += kBlock) {
    const int64_t c = (int64_t)blockIdx.x * kTileCols + j;
    if (c >= cols) continue;
    if (SUM) partial[(int64_t)blockIdx.y * cols + c] = ((s_sum[0][j] + s_sum[1][j]) + s_sum[2][j]) + s_sum[3][j];
    if (AMAX) {
      uint32_t m = s_max[0][j];
#pragma unroll
      for (int w = 1; w < 4; ++w) m = s_max[w][j] > m ? s_max[w][j] : m;
      atomicMax(&amax_out[c], m);
    }
  }
}

// sum of partial[b * cols + c] over the row blocks b = 0 .. n_blk - 1, added strictly in block order (the results of the
// column statistics and of the AWQ weight scale are defined by that order).  The chain of adds is serial by definition; what
// a thread can do is keep MANY loads in flight: 128 per round trip for long chains, then 32, then 8 (round 4: 8 -- a
// 28672-row weight's 1792 partials cost 224 dependent L2 round trips per column, more than the sweep that produced them).
template <int D>
__device__ __forceinline__ void ordered_rounds(const float* __restrict__ partial, int64_t n_blk, int64_t cols, int64_t c,
                                               int64_t& b, float& s) {
  for (; b + D <= n_blk; b += D) {
    float v[D];
#pragma unroll
    for (int u = 0; u < D; ++u) v[u] = partial[(b + u) * cols + c];
#pragma unroll
    for (int u = 0; u < D; ++u) s += v[u];
  }
}
__device__ __forceinline__ float ordered_block_sum(const float* __restrict__ partial, int64_t n_blk, int64_t cols, int64_t c) {
  float s = 0.0f;
  int64_t b = 0;
  ordered_rounds<128>(partial, n_blk, cols, c, b, s);
  ordered_rounds<32>(partial, n_blk, cols, c, b, s);
  ordered_rounds<8>(partial, n_blk, cols, c, b, s);
  for (; b < n_blk; ++b) s += partial[b * cols + c];
  return s;
}
constexpr int kFinBlock = 64;  // threads per finalize workgroup: 8192 columns = 128 workgroups (256-thread ones filled 32 CUs)

// sum_out[c] (+)= sum over row blocks, in row-block order (deterministic)
__global__ void col_sum_finalize_kernel(const float* __restrict__ partial, int64_t n_blk, int64_t cols,
                                        float* __restrict__ sum_out, int accumulate) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  const float s = ordered_block_sum(partial, n_blk, cols, c);
  sum_out[c] = accumulate ? sum_out[c] + s : s;
}

// acc[c] += float(dtype(sum over row blocks / rows)): get_act_scale of awq_lite (model_calib.py:1471-1472,
// x.abs().mean(0) in the activation dtype -- fp32 accumulation, IEEE division, ONE rounding to the dtype -- then
// .to(float32)), added to the running sum of the per-batch means
template <int DT>
__global__ void col_mean_accum_kernel(const float* __restrict__ partial, int64_t n_blk, int64_t rows, int64_t cols,
                                      float* __restrict__ acc) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  const float s = ordered_block_sum(partial, n_blk, cols, c);
  acc[c] = acc[c] + round_to_dtype<DT>(s / (float)rows);
}

// ---------------------------------------------------------------- AWQ weight scale (a12)
// get_weight_scale (model_calib.py:1453-1469): scale[r,c] = dt(|w[r,c]| / dt(gamax[r, c/g] + tiny_dt)),
// w_scale[c] = dt(mean_r scale[r,c]) widened to fp32.  Same grid shape as the column statistics: a lane owns
// V adjacent columns of kColRows rows; the g columns of a group are LPG adjacent lanes, so the group amax is
// a DPP butterfly and the weight is read exactly once.
template <int DT, int LPG>
__global__ __launch_bounds__(kBlock) void awq_wscale_kernel(const void* __restrict__ w, int64_t rows,
                                                            int64_t cols, float tiny,
                                                            float* __restrict__ partial) {
  constexpr int V = Elem<DT>::kVec;
  const int64_t c0 = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * V;
  if (c0 >= cols) return;
  const int64_t r0 = (int64_t)blockIdx.y * kColRows;
  const int64_t r1 = r0 + kColRows < rows ? r0 + kColRows : rows;
  const char* base = reinterpret_cast<const char*>(w);
  const int64_t row_bytes = cols * (16 / V);
  float sm[V];
#pragma unroll
  for (int i = 0; i < V; ++i) sm[i] = 0.0f;
  Pack16 pk[kColRows];
#pragma unroll
  for (int r = 0; r < kColRows; ++r)
    if (r0 + r < r1) pk[r] = load16(base + (r0 + r) * row_bytes + c0 * (16 / V));
#pragma unroll
  for (int r = 0; r < kColRows; ++r) {
    if (r0 + r < r1) {
      const float gmax = __uint_as_float(group_max_u32<LPG>(pack_absmax<DT>(pk[r])));
      const float den = round_to_dtype<DT>(gmax + tiny);
      float f[8];
      unpack<DT>(pk[r], f);
      // the V quotients of a packet share `den`: one refined reciprocal and five FMAs each (SharedDiv, moq_common.h) instead
      // of V IEEE divisions -- the sweep was VALU-bound on them (0.26 of 8 TB/s in round 4).  Exact when neither a residual
      // nor the quotient can leave the normal range: den in [2^-60, 2^20] and every numerator zero or >= 2^-100 (then
      // |x| / den >= 2^-120); any other packet divides the IEEE way.
      const SharedDiv sd = make_shared_div(den);
      if (sd.fast && den <= 0x1p20f && !pack_has_tiny_nonzero<DT>(pk[r])) {
#pragma unroll
        for (int i = 0; i < V; ++i) {
          const float n0 = __builtin_fabsf(f[i]);
          const float q0 = n0 * sd.y;
          const float q1 = __builtin_fmaf(__builtin_fmaf(-sd.d, q0, n0), sd.y, q0);
          sm[i] += round_to_dtype<DT>(__builtin_fmaf(__builtin_fmaf(-sd.d, q1, n0), sd.y, q1));
        }
      } else {
#pragma unroll
        for (int i = 0; i < V; ++i) sm[i] += round_to_dtype<DT>(__builtin_fabsf(f[i]) / den);
      }
    }
  }
  float* dst = partial + (int64_t)blockIdx.y * cols + c0;
#pragma unroll
  for (int i = 0; i < V; ++i) dst[i] = sm[i];
}
template <int DT>
__global__ void awq_wscale_finalize_kernel(const float* __restrict__ partial, int64_t n_blk, int64_t rows,
                                           int64_t cols, float* __restrict__ out) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  const float s = ordered_block_sum(partial, n_blk, cols, c);
  out[c] = round_to_dtype<DT>(s / (float)rows);  // torch.mean: fp32 accumulate, one rounding to dtype
}

// ---------------------------------------------------------------- generic
template <int DT>
__global__ void amax_generic_kernel(const void* __restrict__ x, int64_t n, int64_t axis_size,
                                    int64_t inner, uint32_t* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    atomicMax(&out[(i / inner) % axis_size], absbits(load1<DT>(x, i)));
}
template <int DT>
__global__ void col_sum_generic_kernel(const void* __restrict__ x, int64_t rows, int64_t cols,
                                       float* __restrict__ sum_out, int accumulate) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  float s = 0.0f;
  for (int64_t r = 0; r < rows; ++r) s += __builtin_fabsf(load1<DT>(x, r * cols + c));
  sum_out[c] = accumulate ? sum_out[c] + s : s;
}

// out[o, i] = max_b |x[o, b, i]|: lanes run along `inner` (coalesced), each walks the `mid` entries of its column.
// One step of reduce_block_amax (core_utils.py:43-90) for a blocked dim that is not the last: the first such step
// reads the tensor once, the later ones work on data already reduced by a block size.
template <int DT>
__global__ __launch_bounds__(kBlock) void amax_mid_kernel(const void* __restrict__ x, int64_t outer, int64_t mid,
                                                          int64_t inner, float* __restrict__ out) {
  const int64_t n_out = outer * inner;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < n_out; t += (int64_t)gridDim.x * kBlock) {
    const int64_t o = t / inner, i = t - o * inner;
    const int64_t base = o * mid * inner + i;
    uint32_t acc = 0;
    for (int64_t b = 0; b < mid; ++b) {
      const uint32_t v = absbits(load1<DT>(x, base + b * inner));
      acc = v > acc ? v : acc;
    }
    out[t] = __uint_as_float(acc);
  }
}
// inner % V == 0, 16-byte aligned base: one 16-byte packet of `inner` per lane and step of `mid` (the scalar form above
// moves 2 bytes per lane and load: 2 TB/s), V running maxima in registers, four steps of `mid` in flight
template <int DT>
__global__ __launch_bounds__(kBlock) void amax_mid_vec_kernel(const void* __restrict__ x, int64_t outer, int64_t mid,
                                                              int64_t inner, float* __restrict__ out) {
  constexpr int V = Elem<DT>::kVec;
  const int64_t ipk = inner / V, n_pk = outer * ipk;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < n_pk; t += (int64_t)gridDim.x * kBlock) {
    const int64_t o = t / ipk, i = (t - o * ipk) * V;
    const char* p = reinterpret_cast<const char*>(x) + (o * mid * inner + i) * (16 / V);
    const int64_t step = inner * (16 / V);
    uint32_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto fold = [&](const Pack16& pk) {
      float f[8];
      unpack<DT>(pk, f);
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const uint32_t a = absbits(f[e]);
        acc[e] = a > acc[e] ? a : acc[e];
      }
    };
    int64_t b = 0;
    for (; b + 4 <= mid; b += 4) {
      const Pack16 p0 = load16_nt(p + b * step), p1 = load16_nt(p + (b + 1) * step);
      const Pack16 p2 = load16_nt(p + (b + 2) * step), p3 = load16_nt(p + (b + 3) * step);
      fold(p0); fold(p1); fold(p2); fold(p3);
    }
    for (; b < mid; ++b) fold(load16_nt(p + b * step));
    float* op = out + o * inner + i;
#pragma unroll
    for (int e = 0; e < V; ++e) op[e] = __uint_as_float(acc[e]);
  }
}

int launch_group(const void* x, void* y, float* amax_out, int64_t n_groups, int g, int dt, int num_bits,
                 int is_unsigned, int narrow, bool qdq, const void* s, int64_t cols, void* stream,
                 const char* who);

}  // namespace moq

using namespace moq;
static inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }

extern "C" int moq_amax_axis(const void* x, int64_t outer, int64_t axis_size, int64_t inner, int dt,
                             float* out, int accumulate, void* stream) {
  if (outer < 0 || axis_size < 0 || inner < 0 || out == nullptr) {
    set_error("moq_amax_axis: bad sizes or NULL out");
    return MOQ_ERR_INVALID;
  }
  const int64_t n = outer * axis_size * inner;
  if (n > 0 && x == nullptr) {
    set_error("moq_amax_axis: NULL input");
    return MOQ_ERR_INVALID;
  }
  if (axis_size == 0) return MOQ_OK;
  const int vec = dt == MOQ_F32 ? 4 : 8;
  const bool aligned = (reinterpret_cast<uintptr_t>(x) & 15u) == 0;
  // per-group fast path: short power-of-two rows -> sub-wave butterfly kernel (plain stores, no atomics)
  if (!accumulate && outer == 1 && aligned && inner % vec == 0) {
    const int64_t lpg = inner / vec;
    if (lpg >= 1 && lpg <= 64 && (lpg & (lpg - 1)) == 0 && n > 0)
      return launch_group(x, nullptr, out, axis_size, (int)inner, dt, 8, 0, 0, false, nullptr, 0, stream,
                          "moq_amax_axis(group)");
  }
  if (!accumulate) {
    if (hipMemsetAsync(out, 0, sizeof(float) * axis_size, S(stream)) != hipSuccess)
      return check_launch("moq_amax_axis memset");
  }
  if (n == 0) return MOQ_OK;
  uint32_t* ob = reinterpret_cast<uint32_t*>(out);
  if (aligned && inner >= vec && inner % vec == 0) {
    const int64_t n_rows = outer * axis_size;
    const int64_t row_packets = inner / vec;
    // split long rows so that at least ~8192 wave work items exist
    int64_t segs = 1;
    if (n_rows < 8192) segs = (8192 + n_rows - 1) / n_rows;
    int64_t seg_packets = (row_packets + segs - 1) / segs;
    if (seg_packets < 256) seg_packets = 256 < row_packets ? 256 : row_packets;  // >= 4 KiB per item
    segs = (row_packets + seg_packets - 1) / seg_packets;
    const int64_t items = n_rows * segs;
    int64_t blocks = (items + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((amax_rows_kernel<DT>), dim3((int)blocks), dim3(kBlock), 0,
                                              S(stream), x, n_rows, axis_size, inner, seg_packets, segs,
                                              ob));
  } else if (aligned && inner == 1 && axis_size % vec == 0) {
    dim3 grid((unsigned)((axis_size / vec + 63) / 64), (unsigned)((outer + kColRowsWG - 1) / kColRowsWG));
    MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((col_stats_kernel<DT, false, true>), grid, dim3(kBlock), 0,
                                              S(stream), x, outer, axis_size, ob, (float*)nullptr));
  } else {
    const int grid = stream_grid(kBlock, n);
    MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((amax_generic_kernel<DT>), dim3(grid), dim3(kBlock), 0,
                                              S(stream), x, n, axis_size, inner, ob));
  }
  return check_launch("moq_amax_axis");
}

extern "C" int64_t moq_col_stats_workspace(int64_t tokens, int64_t cols) {
  if (tokens < 0 || cols < 0) return MOQ_ERR_INVALID;
  return ((tokens + kColRows - 1) / kColRows) * cols;
}

extern "C" int moq_col_abs_stats(const void* x, int64_t tokens, int64_t cols, int dt, float* sum_out,
                                 float* amax_out, float* partial, int accumulate, void* stream) {
  if (tokens < 0 || cols <= 0 || (tokens > 0 && x == nullptr) || (sum_out == nullptr && amax_out == nullptr)) {
    set_error("moq_col_abs_stats: bad arguments");
    return MOQ_ERR_INVALID;
  }
  if (!accumulate && amax_out != nullptr) {
    if (hipMemsetAsync(amax_out, 0, sizeof(float) * cols, S(stream)) != hipSuccess)
      return check_launch("moq_col_abs_stats memset");
  }
  if (tokens == 0) {
    if (!accumulate && sum_out != nullptr) (void)hipMemsetAsync(sum_out, 0, sizeof(float) * cols, S(stream));
    return check_launch("moq_col_abs_stats");
  }
  const int vec = dt == MOQ_F32 ? 4 : 8;
  const bool fast = (reinterpret_cast<uintptr_t>(x) & 15u) == 0 && cols % vec == 0;
  uint32_t* ab = reinterpret_cast<uint32_t*>(amax_out);
  if (fast && (sum_out == nullptr || partial != nullptr)) {
    const int64_t n_blk = (tokens + kColRowsWG - 1) / kColRowsWG;
    dim3 grid((unsigned)((cols / vec + 63) / 64), (unsigned)n_blk);
    if (sum_out != nullptr && amax_out != nullptr) {
      MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((col_stats_kernel<DT, true, true>), grid, dim3(kBlock), 0,
                                                S(stream), x, tokens, cols, ab, partial));
    } else if (sum_out != nullptr) {
      MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((col_stats_kernel<DT, true, false>), grid, dim3(kBlock), 0,
                                                S(stream), x, tokens, cols, ab, partial));
    } else {
      MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((col_stats_kernel<DT, false, true>), grid, dim3(kBlock), 0,
                                                S(stream), x, tokens, cols, ab, partial));
    }
    if (sum_out != nullptr)
      hipLaunchKernelGGL(col_sum_finalize_kernel, dim3((unsigned)((cols + kFinBlock - 1) / kFinBlock)), dim3(kFinBlock), 0,
                         S(stream), partial, n_blk, cols, sum_out, accumulate);
  } else {
    if (sum_out != nullptr && partial == nullptr && fast) {
      set_error("moq_col_abs_stats: workspace `partial` is required for the column sum");
      return MOQ_ERR_INVALID;
    }
    if (amax_out != nullptr) {
      const int grid = stream_grid(kBlock, tokens * cols);
      MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((amax_generic_kernel<DT>), dim3(grid), dim3(kBlock), 0,
                                                S(stream), x, tokens * cols, cols, (int64_t)1, ab));
    }
    if (sum_out != nullptr) {
      MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((col_sum_generic_kernel<DT>),
                                                dim3((unsigned)((cols + 255) / 256)), dim3(256), 0,
                                                S(stream), x, tokens, cols, sum_out, accumulate));
    }
  }
  return check_launch("moq_col_abs_stats");
}

extern "C" int moq_col_abs_mean_accum(const void* x, int64_t tokens, int64_t cols, int dt, float* acc,
                                      float* partial, void* stream) {
  if (tokens <= 0 || cols <= 0 || x == nullptr || acc == nullptr || partial == nullptr) {
    set_error("moq_col_abs_mean_accum: bad arguments");
    return MOQ_ERR_INVALID;
  }
  const int vec = dt == MOQ_F32 ? 4 : 8;
  if ((reinterpret_cast<uintptr_t>(x) & 15u) != 0 || cols % vec != 0) {
    set_error("moq_col_abs_mean_accum: needs a 16-byte aligned batch and cols %% %d == 0", vec);
    return MOQ_ERR_UNSUPPORTED;
  }
  const int64_t n_blk = (tokens + kColRowsWG - 1) / kColRowsWG;
  dim3 grid((unsigned)((cols / vec + 63) / 64), (unsigned)n_blk);
  MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((col_stats_kernel<DT, true, false>), grid, dim3(kBlock), 0, S(stream), x,
                                            tokens, cols, (uint32_t*)nullptr, partial));
  MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((col_mean_accum_kernel<DT>), dim3((unsigned)((cols + kFinBlock - 1) / kFinBlock)),
                                            dim3(kFinBlock), 0, S(stream), partial, n_blk, tokens, cols, acc));
  return check_launch("moq_col_abs_mean_accum");
}

extern "C" int moq_awq_weight_scale(const void* w, int64_t rows, int64_t cols, int g, int dt, float* out,
                                    float* partial, void* stream) {
  if (rows <= 0 || cols <= 0 || g <= 0 || w == nullptr || out == nullptr || partial == nullptr) {
    set_error("moq_awq_weight_scale: bad arguments");
    return MOQ_ERR_INVALID;
  }
  const int vec = dt == MOQ_F32 ? 4 : 8;
  const int lpg = g / vec;
  if (cols % g != 0 || g % vec != 0 || lpg > 64 || (lpg & (lpg - 1)) != 0 ||
      (reinterpret_cast<uintptr_t>(w) & 15u) != 0) {
    set_error("moq_awq_weight_scale: needs cols %% g == 0, g/%d a power of two <= 64, 16-byte aligned weight", vec);
    return MOQ_ERR_UNSUPPORTED;
  }
  // torch.finfo(dtype).tiny: smallest positive normal of the storage dtype (model_calib.py:1465)
  const float tiny = dt == MOQ_F16 ? 6.103515625e-05f : 1.17549435e-38f;
  const int64_t n_blk = (rows + kColRows - 1) / kColRows;
  dim3 grid((unsigned)((cols / vec + kBlock - 1) / kBlock), (unsigned)n_blk);
#define MOQ_WS_CASE(L) \
  case L: MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((awq_wscale_kernel<DT, L>), grid, dim3(kBlock), 0, S(stream), w, rows, cols, tiny, partial)); break;
  switch (lpg) {
    MOQ_WS_CASE(1) MOQ_WS_CASE(2) MOQ_WS_CASE(4) MOQ_WS_CASE(8) MOQ_WS_CASE(16) MOQ_WS_CASE(32) MOQ_WS_CASE(64)
    default: set_error("unreachable"); return MOQ_ERR_INVALID;
  }
#undef MOQ_WS_CASE
  MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((awq_wscale_finalize_kernel<DT>), dim3((unsigned)((cols + kFinBlock - 1) / kFinBlock)),
                                            dim3(kFinBlock), 0, S(stream), partial, n_blk, rows, cols, out));
  return check_launch("moq_awq_weight_scale");
}

extern "C" int moq_amax_mid(const void* x, int64_t outer, int64_t mid, int64_t inner, int dt, float* out,
                            void* stream) {
  if (outer < 0 || mid <= 0 || inner < 0 || (outer * inner > 0 && (x == nullptr || out == nullptr))) {
    set_error("moq_amax_mid: bad arguments");
    return MOQ_ERR_INVALID;
  }
  if (outer * inner == 0) return MOQ_OK;
  const int vec = dt == MOQ_F32 ? 4 : 8;
  if (inner % vec == 0 && (reinterpret_cast<uintptr_t>(x) & 15u) == 0) {
    MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((amax_mid_vec_kernel<DT>), dim3(stream_grid(kBlock, outer * inner / vec)),
                                              dim3(kBlock), 0, S(stream), x, outer, mid, inner, out));
    return check_launch("moq_amax_mid");
  }
  MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((amax_mid_kernel<DT>), dim3(stream_grid(kBlock, outer * inner)), dim3(kBlock), 0,
                                            S(stream), x, outer, mid, inner, out));
  return check_launch("moq_amax_mid");
}
