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
// to the oracle.
//
// Between two column blocks the reference spreads the block's errors over all columns to the right with an fp32 GEMM,
// w_rows[:, i2:] -= delta_blk.matmul(hessian_inv[i1:i2, i2:]) (sparsegpt.py:124) -- in the BLAS library's summation
// order, whatever that is on the machine at hand, and the 2:4 decisions downstream inherit it.  Here that update is the
// kernel `sgpt_trailing_kernel` with a DEFINED order: every output is the fp32 fused-multiply-add chain over the block's
// columns k = 0, 1, ... in ascending order, started at +0, subtracted from the weight once.  It runs on the matrix
// cores: v_mfma_f32_32x32x2_f32 takes fp32 inputs and is bit-for-bit an fmaf chain (MI355X_MICROARCH.md, "F32 (f32 in)":
// exact f32, 155 TFLOP/s = the fp32 vector rate), so the oracle restates it with fmaf and masks are reproducible bit
// for bit from a given inverse factor (tests/test_gpu_sparsegpt.py).
#include <atomic>

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

// ---- trailing update: w[r, i2 + c] -= chain_{k < bs} fma(delta[r, k], hinv[i1 + k, i2 + c]), i2 = i1 + bs.
//
// One workgroup (4 waves) owns a 128 x 128 output tile; K is the whole column block (<= 128), so the operands are staged
// ONCE: A = delta tile [128 r][128 k] (row pitch 129 words: the MFMA's A fragment reads 32 consecutive rows at one k --
// an odd pitch spreads them over the banks), B = hinv tile [128 k][128 c] (lanes run along c).  Rows past `rows`, columns
// past the matrix and k >= bs are filled with zeros: fma(0, 0, acc) = acc for every acc this chain can hold (it starts at
// +0 and round-to-nearest never produces -0 from a sum that is not (-0) + (-0)).  Each wave multiplies a 64 x 64 quarter
// as 2 x 2 tiles of 32 x 32: per k pair two A and two B words per lane from LDS, four MFMAs.  MFMA operand layout
// (32x32x2): a = A[m = lane & 31][k = lane >> 5], b = B[k = lane >> 5][n = lane & 31]; accumulator register e of a lane
// holds row 8 (e >> 2) + (e & 3) + 4 (lane >> 5), column lane & 31 -- lanes run along the weight's columns, so the
// read-modify-write of w is 128-byte runs.
typedef float f32x16v __attribute__((ext_vector_type(16)));
constexpr int kTuTile = 128, kTuPitchA = 129;
constexpr size_t kTuLds = ((size_t)kTuTile * kTuPitchA + (size_t)kTuTile * kTuTile) * sizeof(float);

// one operand quad: a 16-byte load when the layout allows it (VEC), four words otherwise; elements that are not part of
// the matrix (`n_valid` < 4, counted from the quad's first element) come back as zeros.  No branch: an invalid quad reads
// a safe address and is masked, so the sixteen loads of a thread can all be in flight together.
template <bool VEC>
__device__ __forceinline__ float4 tu_load_quad(const float* __restrict__ base, int64_t off, int n_valid) {
  const bool any = n_valid > 0;
  const float* src = base + (any ? off : 0);
  float4 v;
  if constexpr (VEC) {
    v = *reinterpret_cast<const float4*>(src);
  } else {
    v.x = src[0];
    v.y = src[n_valid > 1 ? 1 : 0];
    v.z = src[n_valid > 2 ? 2 : 0];
    v.w = src[n_valid > 3 ? 3 : 0];
  }
  v.x = n_valid > 0 ? v.x : 0.0f;
  v.y = n_valid > 1 ? v.y : 0.0f;
  v.z = n_valid > 2 ? v.z : 0.0f;
  v.w = n_valid > 3 ? v.w : 0.0f;
  return v;
}

// VA: delta quads are 16-byte aligned and whole (bs % 4 == 0); VB: hinv quads are (ld % 4 == 0, i2 % 4 == 0)
template <bool VA, bool VB>
__global__ __launch_bounds__(256) void sgpt_trailing_kernel(float* __restrict__ w, int64_t rows, int64_t ld, int64_t i1,
                                                            int bs, const float* __restrict__ delta,
                                                            const float* __restrict__ hinv) {
  extern __shared__ __attribute__((aligned(16))) float tu_lds[];
  float* sa = tu_lds;                        // [128 r][129]
  float* sb = tu_lds + kTuTile * kTuPitchA;  // [128 k][128 c]
  const int64_t i2 = i1 + bs, ncols = ld - i2;
  const int64_t r0 = (int64_t)blockIdx.y * kTuTile, c0 = (int64_t)blockIdx.x * kTuTile;
  const int tid = threadIdx.x;
  {
    // thread t takes quad q = t & 31 of rows / k rows (t >> 5) + 8 i.  All 32 loads of a thread are issued before the
    // first LDS write (a load per iteration, waited for at once, would put sixteen memory latencies in a row in front
    // of a tile's 7 us of MFMAs)
    const int q = tid & 31;
    float4 va[kTuTile / 8], vb[kTuTile / 8];
#pragma unroll
    for (int i = 0; i < kTuTile / 8; ++i) {
      const int r = (tid >> 5) + 8 * i;
      const int nv = r0 + r < rows ? bs - 4 * q : 0;  // valid words of the quad
      va[i] = tu_load_quad<VA>(delta, (r0 + r) * bs + 4 * q, nv);
    }
#pragma unroll
    for (int i = 0; i < kTuTile / 8; ++i) {
      const int k = (tid >> 5) + 8 * i;
      const int64_t left = ncols - (c0 + 4 * q);
      const int nv = k < bs ? (int)(left > 4 ? 4 : left) : 0;
      vb[i] = tu_load_quad<VB>(hinv, (i1 + k) * ld + i2 + c0 + 4 * q, nv);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < kTuTile / 8; ++i) {
      float* dst = sa + ((tid >> 5) + 8 * i) * kTuPitchA + 4 * q;
      dst[0] = va[i].x; dst[1] = va[i].y; dst[2] = va[i].z; dst[3] = va[i].w;
    }
#pragma unroll
    for (int i = 0; i < kTuTile / 8; ++i)
      *reinterpret_cast<float4*>(sb + ((tid >> 5) + 8 * i) * kTuTile + 4 * q) = vb[i];
  }
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;  // the wave's 64 x 64 quarter
  const int m = lane & 31, h = lane >> 5;
  f32x16v acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
  // The read half of the weight tile's read-modify-write is requested HERE, before the MFMA loop: 64 loads per lane
  // that do not depend on the contraction ride under its ~7 us instead of standing behind it.
  float cur[2][2][16];
  float* ptr[2][2][16];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t c = c0 + wc * 64 + j * 32 + m;
      const bool c_ok = c < ncols;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t r = r0 + wr * 64 + i * 32 + 8 * (e >> 2) + (e & 3) + 4 * h;
        ptr[i][j][e] = (c_ok && r < rows) ? w + r * ld + i2 + c : nullptr;
        cur[i][j][e] = *(ptr[i][j][e] ? ptr[i][j][e] : w);
      }
    }
  __builtin_amdgcn_sched_barrier(0);
  const float* pa = sa + (wr * 64 + m) * kTuPitchA + h;
  const float* pb = sb + h * kTuTile + wc * 64 + m;
  const int kpairs = (bs + 1) / 2;
  // the operands of k pair j + 1 are read while the four MFMAs of pair j issue (64 cycles each: an LDS round trip
  // hides under one pair); the pad rows / columns behind the last pair are zeros, so reading one pair too far is safe
  float a0 = pa[0], a1 = pa[32 * kTuPitchA], b0 = pb[0], b1 = pb[32];
  for (int j = 0; j < kpairs; ++j) {
    const int jn = j + 1 < kTuTile / 2 ? j + 1 : j;
    const float a0n = pa[2 * jn], a1n = pa[32 * kTuPitchA + 2 * jn];
    const float b0n = pb[2 * jn * kTuTile], b1n = pb[2 * jn * kTuTile + 32];
    __builtin_amdgcn_sched_barrier(0);
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    a0 = a0n; a1 = a1n; b0 = b0n; b1 = b1n;
  }
  // the weight tile comes back: w -= acc (the loads went out before the MFMA loop)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e)
        if (ptr[i][j][e]) *ptr[i][j][e] = cur[i][j][e] - acc[i][j][e];
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

extern "C" int moq_sgpt_trailing_update(float* w, int64_t rows, int64_t ld, int64_t i1, int bs, const float* delta,
                                        const float* hinv, void* stream) {
  if (w == nullptr || hinv == nullptr || delta == nullptr || rows < 0 || ld <= 0 || i1 < 0 || bs <= 0 || i1 + bs > ld) {
    set_error("moq_sgpt_trailing_update: bad arguments");
    return MOQ_ERR_INVALID;
  }
  if (bs > kTuTile) {
    set_error("moq_sgpt_trailing_update: column block of %d > %d", bs, kTuTile);
    return MOQ_ERR_UNSUPPORTED;
  }
  const int64_t ncols = ld - (i1 + bs);
  if (rows == 0 || ncols == 0) return MOQ_OK;
  const int64_t gx = (ncols + kTuTile - 1) / kTuTile, gy = (rows + kTuTile - 1) / kTuTile;
  if (gy > 65535 || gx > 0x7FFFFFFF) {
    set_error("moq_sgpt_trailing_update: matrix too large");
    return MOQ_ERR_UNSUPPORTED;
  }
  static std::atomic<uint64_t> attr_set{0};  // > 64 KiB of dynamic LDS: opt-in attribute, once per device
  int device = 0;
  (void)hipGetDevice(&device);
  const uint64_t bit = 1ull << (device & 63);
  if (!(attr_set.load(std::memory_order_acquire) & bit)) {
    (void)hipFuncSetAttribute((const void*)sgpt_trailing_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kTuLds);
    (void)hipFuncSetAttribute((const void*)sgpt_trailing_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kTuLds);
    (void)hipFuncSetAttribute((const void*)sgpt_trailing_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kTuLds);
    (void)hipFuncSetAttribute((const void*)sgpt_trailing_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kTuLds);
    attr_set.fetch_or(bit, std::memory_order_release);
  }
  const bool va = bs % 4 == 0 && (reinterpret_cast<uintptr_t>(delta) & 15u) == 0;
  const bool vb = ld % 4 == 0 && (i1 + bs) % 4 == 0 && (reinterpret_cast<uintptr_t>(hinv) & 15u) == 0;
  const dim3 grid((unsigned)gx, (unsigned)gy), block(256);
  if (va && vb) hipLaunchKernelGGL((sgpt_trailing_kernel<true, true>), grid, block, kTuLds, S(stream), w, rows, ld, i1, bs, delta, hinv);
  else if (va) hipLaunchKernelGGL((sgpt_trailing_kernel<true, false>), grid, block, kTuLds, S(stream), w, rows, ld, i1, bs, delta, hinv);
  else if (vb) hipLaunchKernelGGL((sgpt_trailing_kernel<false, true>), grid, block, kTuLds, S(stream), w, rows, ld, i1, bs, delta, hinv);
  else hipLaunchKernelGGL((sgpt_trailing_kernel<false, false>), grid, block, kTuLds, S(stream), w, rows, ld, i1, bs, delta, hinv);
  return check_launch("moq_sgpt_trailing_update");
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
