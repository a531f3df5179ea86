// moq_sparse.hip -- SparseGPT mask search: create_sgpt_mask's column sweep
// (sparsity/weight_sparsity/sparsegpt.py:72-133) and the 16-bit transpose that feeds the MFMA Hessian
// accumulation (moq_hessian_accum, moq_gemm.hip).
//
// create_sgpt_mask walks the columns of a [rows, col_bs] weight block ONE AT A TIME in Python: for column j it
// (at j % m == 0) ranks the next m columns by w^2 / diag(Hinv)^2 and prunes the n smallest, then spreads the
// pruning error of column j over the columns to its right with a rank-1 update -- ~6 torch ops per column, 128
// columns per block, cols / 128 blocks per weight.  Rows never interact inside a block, so here ONE wave owns a
// row: lane l keeps columns l and l + 64 of the block in registers, the pivot value is broadcast with a lane
// read, and the whole 128-column sweep is one kernel.  Every arithmetic step is the reference's fp32 step in the
// reference's order (no FMA contraction), so q, the per-column errors and the pruning decisions are bit-identical
// to the oracle; the trailing-block update (delta @ Hinv[i1:i2, i2:], an fp32 GEMM) stays on the library.
#include "moq_common.h"

namespace moq {

constexpr int kSgptMaxBlock = 128;

// n:m selection inside m consecutive lanes: rank of this lane's error among the group (ties: lower column first)
template <int M>
__device__ __forceinline__ int group_rank(float e, int lane) {
  int rank = 0;
  const int base = lane & ~(M - 1), me = lane & (M - 1);
#pragma unroll
  for (int o = 0; o < M; ++o) {
    const float other = __shfl(e, base + o, 64);
    rank += (o != me) && (other < e || (other == e && o < me));
  }
  return rank;
}

// w: [rows, ld] fp32 working weights; the block is columns [i1, i1 + bs).  hinv: [ld, ld] fp32 upper Cholesky
// factor of H^-1.  On return w[:, i1:i1+bs] holds q_blk and delta[rows, bs] the per-column errors
// err_j = (w_j - q_j) / d_j.  prune_n of every prune_m consecutive columns are zeroed (prune_m in {2, 4, 8}).
template <int M>
__global__ __launch_bounds__(kBlock) void sgpt_sweep_kernel(float* __restrict__ w, int64_t rows, int64_t ld,
                                                            int64_t i1, int bs, const float* __restrict__ hinv,
                                                            float* __restrict__ delta, int prune_n) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  float* wr = w + row * ld + i1;
  const bool in0 = lane < bs, in1 = lane + 64 < bs;
  float w0 = in0 ? wr[lane] : 0.0f, w1 = in1 ? wr[lane + 64] : 0.0f;
  // diag(Hinv) of the block: d_k = hinv[i1 + k, i1 + k]
  const float d0 = in0 ? hinv[(i1 + lane) * ld + i1 + lane] : 1.0f;
  const float d1 = in1 ? hinv[(i1 + lane + 64) * ld + i1 + lane + 64] : 1.0f;
  bool m0 = false, m1 = false;  // pruned?
  float q0 = 0.0f, q1 = 0.0f, e0 = 0.0f, e1 = 0.0f;
  for (int j = 0; j < bs; ++j) {
    const int src = j & 63;
    const bool hi = j >= 64;
    if ((j % M) == 0) {
      // errors of the next M columns from the CURRENT weights (sparsegpt.py:111-115); groups never straddle the
      // 64-column halves (M divides 64)
      const float wv = hi ? w1 : w0, dv = hi ? d1 : d0;
      const float err = (wv * wv) / (dv * dv + 1e-9f);
      const int rank = group_rank<M>(err, lane);
      const bool in_group = (lane & ~(M - 1)) == (src & ~(M - 1));
      if (in_group && rank < prune_n) {
        if (hi) m1 = true; else m0 = true;
      }
    }
    // pivot column j: q = masked ? 0 : w;  err = (w - q) / d
    const float wj = __shfl(hi ? w1 : w0, src, 64);
    const float dj = __shfl(hi ? d1 : d0, src, 64);
    const bool mj = __shfl((int)(hi ? m1 : m0), src, 64) != 0;
    const float qj = mj ? 0.0f : wj;
    const float err = (wj - qj) / dj;
    if (lane == src) {
      if (hi) { q1 = qj; e1 = err; } else { q0 = qj; e0 = err; }
    }
    // w_blk[:, j:] -= err (x) hinv_blk[j, j:]   (product rounded, then subtracted: two roundings like the matmul)
    const float* hrow = hinv + (i1 + j) * ld + i1;
    if (in0 && lane >= j) w0 = w0 - err * hrow[lane];
    if (in1 && lane + 64 >= j) w1 = w1 - err * hrow[lane + 64];
  }
  if (in0) { wr[lane] = q0; delta[row * bs + lane] = e0; }
  if (in1) { wr[lane + 64] = q1; delta[row * bs + lane + 64] = e1; }
}

// y[c, r] = x[r, c] for 16-bit elements through a 64 x 64 LDS tile (+1 column of padding: conflict-free columns)
// y_ld: leading dimension of y (>= rows): y may be a column block of a wider [cols, y_ld] staging buffer
__global__ __launch_bounds__(kBlock) void transpose16_kernel(const uint16_t* __restrict__ x,
                                                             uint16_t* __restrict__ y, int64_t rows,
                                                             int64_t cols, int64_t y_ld) {
  __shared__ uint16_t tile[64][66];
  const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
  for (int i = ty; i < 64; i += 4) {
    const int64_t r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < rows && c < cols) ? x[r * cols + c] : (uint16_t)0;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int64_t c = c0 + i, r = r0 + tx;
    if (c < cols && r < rows) y[c * y_ld + r] = tile[tx][i];
  }
}

}  // namespace moq

using namespace moq;
static inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }

extern "C" int moq_sgpt_block_sweep(float* w, int64_t rows, int64_t ld, int64_t i1, int bs, const float* hinv,
                                    float* delta, int prune_n, int prune_m, void* stream) {
  if (w == nullptr || hinv == nullptr || delta == nullptr || rows < 0 || ld <= 0 || i1 < 0 || bs <= 0 ||
      i1 + bs > ld) {
    set_error("moq_sgpt_block_sweep: bad arguments");
    return MOQ_ERR_INVALID;
  }
  if (bs > kSgptMaxBlock || (prune_m != 2 && prune_m != 4 && prune_m != 8) || prune_n < 0 || prune_n > prune_m ||
      bs % prune_m != 0) {
    set_error("moq_sgpt_block_sweep: needs col block <= %d, m in {2,4,8}, block %% m == 0", kSgptMaxBlock);
    return MOQ_ERR_UNSUPPORTED;
  }
  if (rows == 0) return MOQ_OK;
  const dim3 grid((unsigned)((rows + 3) / 4)), block(kBlock);
  switch (prune_m) {
    case 2: hipLaunchKernelGGL((sgpt_sweep_kernel<2>), grid, block, 0, S(stream), w, rows, ld, i1, bs, hinv, delta, prune_n); break;
    case 4: hipLaunchKernelGGL((sgpt_sweep_kernel<4>), grid, block, 0, S(stream), w, rows, ld, i1, bs, hinv, delta, prune_n); break;
    default: hipLaunchKernelGGL((sgpt_sweep_kernel<8>), grid, block, 0, S(stream), w, rows, ld, i1, bs, hinv, delta, prune_n); break;
  }
  return check_launch("moq_sgpt_block_sweep");
}

extern "C" int moq_transpose16_ld(const void* x, void* y, int64_t rows, int64_t cols, int64_t y_ld, void* stream);
extern "C" int moq_transpose16(const void* x, void* y, int64_t rows, int64_t cols, void* stream) {
  return moq_transpose16_ld(x, y, rows, cols, rows, stream);
}
extern "C" int moq_transpose16_ld(const void* x, void* y, int64_t rows, int64_t cols, int64_t y_ld, void* stream) {
  if (rows < 0 || cols < 0 || y_ld < rows || (rows * cols > 0 && (x == nullptr || y == nullptr))) {
    set_error("moq_transpose16: bad arguments");
    return MOQ_ERR_INVALID;
  }
  if (rows * cols == 0) return MOQ_OK;
  const int64_t gy = (rows + 63) / 64, gx = (cols + 63) / 64;
  if (gy > 65535) {
    set_error("moq_transpose16: too many rows");
    return MOQ_ERR_UNSUPPORTED;
  }
  hipLaunchKernelGGL(transpose16_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(kBlock), 0, S(stream),
                     reinterpret_cast<const uint16_t*>(x), reinterpret_cast<uint16_t*>(y), rows, cols, y_ld);
  return check_launch("moq_transpose16");
}
