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
#include "moq_mx.h"

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

// ---- GPTQ column sweep (quantization/utils/calib_utils.py:241-276, gptq_blockwise_update) --------------------------
// The reference walks the columns of a block one at a time and fake-quantizes the WHOLE weight matrix for every column
// (`qdq = quantize_fn(wblk)`, one column of it kept): ~8 torch ops and a full-matrix QDQ per column.  With a calibrated
// (static) amax the quantizer is elementwise, so only the pivot column's QDQ matters, rows never interact inside a block,
// and the sweep has the shape of the SparseGPT one: one wave per row, lane l holds columns l and l + 64 of the block,
// the pivot is broadcast with a lane read.  q_j = QDQ(w_j) with the amax entry of (row, column): the library's own
// fp32 INT-k / FP8-E4M3 quantize-dequantize (qdq_int, fp8_scale + e4m3 round trip: tensor_quant.py:607-645, :46-59);
// err_j = (w_j - q_j) / hinv_jj; every column k >= j of the block gets w_k -= fl(err_j * hinv_jk) -- product rounded,
// then subtracted, the order the oracle restates (orc_gptq_block_sweep).  amax entry of element (r, c):
// amax[r * amax_row_stride + c / g]  (per tensor: stride 0, g >= ld; per output channel: stride 1, g >= ld; static
// blocks of g columns: stride ld / g).
// FMT 3: MX dynamic blocks (E8M0 block scale from the block's CURRENT abs-max, fused_amax_convert: tensor_quant_mx.cu:239-294).
// Here the reference's full-matrix call matters: the scale of the pivot's block moves with the weights, columns already
// swept included (a swept column holds w_k - err_k * hinv_kk, its quantized value up to rounding).  A block of g <= 64
// columns is g adjacent lanes of one half of the wave's 128 columns, so the block abs-max is a butterfly over lanes per
// pivot step; `iq.hi` carries the element format (moq_mx_type), no amax table is read.
// FMT 4: the same with the block scale in an element format (E4M3: the NVFP4-style two-level scale) relative to the calibrated
// tensor-wide amax amax[0] (compute_scale_with_global, cu:139-183; the fp64 steps of mx_scale_general once per pivot step);
// `iq.lo` carries the scale format.
template <int FMT>  // 1: INT-k, 2: FP8-E4M3, 3: MX element format with E8M0 block scales, 4: element-format block scales
__global__ __launch_bounds__(kBlock) void gptq_sweep_kernel(float* __restrict__ w, int64_t rows, int64_t ld, int64_t i1,
                                                            int bs, const float* __restrict__ hinv,
                                                            float* __restrict__ delta, const float* __restrict__ amax,
                                                            int64_t amax_row_stride, int64_t g, IntQ iq) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  float* wr = w + row * ld + i1;
  const bool in0 = lane < bs, in1 = lane + 64 < bs;
  float w0 = in0 ? wr[lane] : 0.0f, w1 = in1 ? wr[lane + 64] : 0.0f;
  const float d0 = in0 ? hinv[(i1 + lane) * ld + i1 + lane] : 1.0f;
  const float d1 = in1 ? hinv[(i1 + lane + 64) * ld + i1 + lane + 64] : 1.0f;
  float a0 = 1.0f, a1 = 1.0f;
  if constexpr (FMT == 1 || FMT == 2) {
    a0 = in0 ? amax[row * amax_row_stride + (i1 + lane) / g] : 1.0f;
    a1 = in1 ? amax[row * amax_row_stride + (i1 + lane + 64) / g] : 1.0f;
  }
  const MxFmt mxf = mx_fmt(FMT >= 3 ? (int)iq.hi : MOQ_E2M1);
  const MxFmt mxsf = mx_fmt(FMT == 4 ? (int)iq.lo : MOQ_E4M3);
  float q0 = 0.0f, q1 = 0.0f, e0 = 0.0f, e1 = 0.0f;
  for (int j = 0; j < bs; ++j) {
    const int src = j & 63;
    const bool hi = j >= 64;
    const float wj = __shfl(hi ? w1 : w0, src, 64);
    const float dj = __shfl(hi ? d1 : d0, src, 64);
    float aj;
    if constexpr (FMT >= 3) {
      // abs-max of every block of g lanes in the pivot's half, from the current weights (|x| clamped to FLT_MAX, NaN dropped:
      // compute_max_warp / block, cu:185-226); then the pivot's own block
      float am = mx_abs_clamped(hi ? w1 : w0);
      for (int o = 1; o < (int)g; o <<= 1) am = __builtin_fmaxf(am, __shfl_xor(am, o, 64));
      aj = __shfl(am, src, 64);
    } else {
      aj = __shfl(hi ? a1 : a0, src, 64);
    }
    float qj;
    if constexpr (FMT == 1) {
      qj = qdq_int(wj, int_scale(aj, iq.hi), iq);
    } else if constexpr (FMT == 3) {
      float sc, un;
      mx_scale_e8m0(aj, mxf.maxv, sc, un);
      qj = mx_qdq(wj, sc, un, mxf);
    } else if constexpr (FMT == 4) {
      float sc, un;
      mx_scale_general(aj, mxf.maxv, mxsf, amax, sc, un);
      qj = mx_qdq(wj, sc, un, mxf);
    } else {
      const Fp8Scale sc = fp8_scale(aj);
      const float t = wj * sc.s;
      float c = __builtin_fminf(__builtin_fmaxf(t, -448.0f), 448.0f);
      c = (t != t) ? t : c;  // torch.clamp keeps NaN
      float ra, rb;
      e4m3_roundtrip2(c, 0.0f, ra, rb);
      qj = ra * sc.inv;
    }
    const float err = (wj - qj) / dj;
    if (lane == src) {
      if (hi) { q1 = qj; e1 = err; } else { q0 = qj; e0 = err; }
    }
    const float* hrow = hinv + (i1 + j) * ld + i1;
    if (in0 && lane >= j) w0 = w0 - err * hrow[lane];
    if (in1 && lane + 64 >= j) w1 = w1 - err * hrow[lane + 64];
  }
  if (in0) { wr[lane] = q0; delta[row * bs + lane] = e0; }
  if (in1) { wr[lane + 64] = q1; delta[row * bs + lane + 64] = e1; }
}

// ---- trailing update: w[r, i2 + c] -= chain_{k < bs} fma(delta[r, k], hinv[i1 + k, i2 + c]), i2 = i1 + bs.
//
// One workgroup (4 waves) owns 128 rows and walks a strip of 64-column tiles.  K is the whole column block (<= 128), so
// the delta tile [128 r][128 k] is staged ONCE per workgroup (row pitch 129 words: the MFMA's A fragment reads 32
// consecutive rows at one k -- an odd pitch spreads them over the banks); the Hinv tiles [128 k][64 c] stream through two
// LDS stages: while the 128 MFMAs of a column tile issue, the next tile's Hinv quads AND the next tile's weights (the read
// half of w -= acc) are already on their way into registers -- requested before the loop, consumed after it.  Rows past
// `rows`, columns past the matrix and k >= bs are zeros: fma(0, 0, acc) = acc for every acc this chain can hold (it starts
// at +0 and round-to-nearest never produces -0 from a sum that is not (-0) + (-0)).  Wave v multiplies rows 32 v .. + 31 of
// the tile against its 64 columns as two 32 x 32 tiles: per k pair one A and two B words per lane from LDS, two MFMAs.
// MFMA operand layout (32x32x2): a = A[m = lane & 31][k = lane >> 5], b = B[k = lane >> 5][n = lane & 31]; accumulator
// register e of a lane holds row 8 (e >> 2) + (e & 3) + 4 (lane >> 5), column lane & 31 -- lanes run along the weight's
// columns, so the read-modify-write of w is 128-byte runs.
// (First version, round 3: one 128 x 128 tile per workgroup, everything staged and waited for in turn -- 27 TFLOP/s; with
// the loads batched 48, the library's fp32 matmul on the same updates: 47-50.)
typedef float f32x16v __attribute__((ext_vector_type(16)));
constexpr int kTuRows = 128, kTuCols = 64, kTuK = 128, kTuPitchA = 129;
constexpr size_t kTuLds = ((size_t)kTuRows * kTuPitchA + 2 * (size_t)kTuK * kTuCols) * sizeof(float);

// one operand quad: a 16-byte load when the layout allows it (VEC), four words otherwise; elements that are not part of
// the matrix (`n_valid` < 4, counted from the quad's first element) come back as zeros.  No branch: an invalid quad reads
// a safe address and is masked, so all loads of a thread can be in flight together.
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
                                                            const float* __restrict__ hinv, int tiles_per_wg) {
  extern __shared__ __attribute__((aligned(16))) float tu_lds[];
  float* sa = tu_lds;                        // [128 r][129]
  float* sb = tu_lds + kTuRows * kTuPitchA;  // 2 x [128 k][64 c]
  const int64_t i2 = i1 + bs, ncols = ld - i2;
  const int64_t r0 = (int64_t)blockIdx.y * kTuRows;
  const int64_t n_ctiles = (ncols + kTuCols - 1) / kTuCols;
  const int64_t ct0 = (int64_t)blockIdx.x * tiles_per_wg;
  const int steps = (int)(n_ctiles - ct0 < tiles_per_wg ? n_ctiles - ct0 : tiles_per_wg);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 31, h = lane >> 5;

  // Hinv tile of column tile `ct` -> registers: thread t takes quad q = t & 15 of k rows (t >> 4) + 16 i
  float4 breg[kTuK / 16];
  auto load_b = [&](int64_t ct) {
    const int q = tid & 15;
    const int64_t c = ct * kTuCols + 4 * q;
    const int64_t left = ncols - c;
#pragma unroll
    for (int i = 0; i < kTuK / 16; ++i) {
      const int k = (tid >> 4) + 16 * i;
      const int nv = k < bs ? (int)(left > 4 ? 4 : left) : 0;
      breg[i] = tu_load_quad<VB>(hinv, (i1 + k) * ld + i2 + c, nv);
    }
  };
  auto store_b = [&](int stage) {
    float* dst = sb + stage * (kTuK * kTuCols) + 4 * (tid & 15);
#pragma unroll
    for (int i = 0; i < kTuK / 16; ++i)
      *reinterpret_cast<float4*>(dst + ((tid >> 4) + 16 * i) * kTuCols) = breg[i];
  };
  // the weights this lane will update in column tile `ct` (two accumulator tiles of 16).  Interior tiles take the
  // unpredicated form -- a wave-uniform row base plus one 32-bit lane offset, 32 loads back to back; tiles on the
  // matrix edge predicate every element.
  float cur[2][16], nxt[2][16];
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int64_t lane_off = (int64_t)(4 * h) * ld + m;
  const bool rows_full = r0 + kTuRows <= rows;
  auto load_w = [&](int64_t ct, float (&dst)[2][16]) {
    const float* base = w + (r0 + wave_u * 32) * ld + i2 + ct * kTuCols;  // wave-uniform
    if (rows_full && (ct + 1) * kTuCols <= ncols) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) dst[j][e] = base[(int64_t)(8 * (e >> 2) + (e & 3)) * ld + j * 32 + lane_off];
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int64_t c = ct * kTuCols + j * 32 + m;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int64_t r = r0 + wave_u * 32 + 8 * (e >> 2) + (e & 3) + 4 * h;
          dst[j][e] = (c < ncols && r < rows) ? w[r * ld + i2 + c] : 0.0f;
        }
      }
    }
  };
  load_b(ct0);
  load_w(ct0, cur);
  {
    // delta tile: thread t takes k quad q = t & 31 of rows (t >> 5) + 8 i; all sixteen loads before the first LDS write
    const int q = tid & 31;
    float4 va[kTuRows / 8];
#pragma unroll
    for (int i = 0; i < kTuRows / 8; ++i) {
      const int r = (tid >> 5) + 8 * i;
      const int nv = r0 + r < rows ? bs - 4 * q : 0;
      va[i] = tu_load_quad<VA>(delta, (r0 + r) * bs + 4 * q, nv);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < kTuRows / 8; ++i) {
      float* dst = sa + ((tid >> 5) + 8 * i) * kTuPitchA + 4 * q;
      dst[0] = va[i].x; dst[1] = va[i].y; dst[2] = va[i].z; dst[3] = va[i].w;
    }
  }
  store_b(0);
  __syncthreads();

  const float* pa = sa + (wave_u * 32 + m) * kTuPitchA + h;
  const int kpairs = (bs + 1) / 2;
  for (int s = 0; s < steps; ++s) {
    const int64_t ct = ct0 + s;
    const bool more = s + 1 < steps;
    if (more) {  // (wave-uniform) the next column tile's operands and weights: requested now, used after the MFMAs
      load_b(ct + 1);
      load_w(ct + 1, nxt);
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x16v acc0, acc1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc0[e] = 0.0f; acc1[e] = 0.0f; }
    const float* pb = sb + (s & 1) * (kTuK * kTuCols) + h * kTuCols + m;
    // The operands of k pair j + 2 are requested while the MFMAs of pair j issue: an LDS round trip (~130 cycles) is
    // longer than one pair's two MFMAs (128), so a distance of one pair still exposed it every iteration.  Four register
    // slots, the loop body unrolled four times; pairs past the column block read zero rows (the tile is padded to 128 k).
    float ra[4], rb0[4], rb1[4];
    auto rd = [&](int slot, int jj) {
      const int j2 = jj < kTuK / 2 ? jj : kTuK / 2 - 1;
      ra[slot] = pa[2 * j2];
      rb0[slot] = pb[2 * j2 * kTuCols];
      rb1[slot] = pb[2 * j2 * kTuCols + 32];
    };
    rd(0, 0);
    rd(1, 1);
    for (int j = 0; j < kpairs; j += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        rd((u + 2) & 3, j + u + 2);
        __builtin_amdgcn_sched_barrier(0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[u], rb0[u], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[u], rb1[u], acc1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // w -= acc for this column tile
    {
      float* base = w + (r0 + wave_u * 32) * ld + i2 + ct * kTuCols;
      if (rows_full && (ct + 1) * kTuCols <= ncols) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          base[(int64_t)(8 * (e >> 2) + (e & 3)) * ld + lane_off] = cur[0][e] - acc0[e];
          base[(int64_t)(8 * (e >> 2) + (e & 3)) * ld + 32 + lane_off] = cur[1][e] - acc1[e];
        }
      } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int64_t c = ct * kTuCols + j * 32 + m;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int64_t r = r0 + wave_u * 32 + 8 * (e >> 2) + (e & 3) + 4 * h;
            if (c < ncols && r < rows) w[r * ld + i2 + c] = cur[j][e] - (j == 0 ? acc0[e] : acc1[e]);
          }
        }
      }
    }
    if (more) {
      store_b((s + 1) & 1);  // the stage the tile before this one was read from (everyone passed the last barrier)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) cur[j][e] = nxt[j][e];
    }
    // workgroup barrier on LDS traffic ONLY: __syncthreads() would also wait (vmcnt(0)) for this tile's 32 weight stores
    // per lane to reach memory -- microseconds per column tile, with nothing depending on them
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  }
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

// The same for whole tiles of 16-byte aligned matrices (rows % 64 == 0, cols % 64 == 0, y_ld % 8 == 0): 16-byte global
// loads and stores instead of 2-byte ones, no bounds branch.  A thread loads chunk t % 8 (eight elements) of rows t / 8
// and t / 8 + 32, parks them in the tile as dwords (pitch 66 elements = 33 dwords: a column walk touches 32 different
// banks), gathers eight consecutive r of one c back from the tile and stores 16 bytes of y's row c.
__global__ __launch_bounds__(kBlock) void transpose16_tiles_kernel(const uint16_t* __restrict__ x,
                                                                   uint16_t* __restrict__ y, int64_t cols,
                                                                   int64_t y_ld) {
  __shared__ uint32_t tile[64][33];
  const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
  const int t = threadIdx.x, lr = t >> 3, ch = t & 7;
  Pack16 v[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) v[h] = load16_nt(x + (r0 + lr + 32 * h) * cols + c0 + ch * 8);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    uint32_t* d = &tile[lr + 32 * h][ch * 4];
    d[0] = v[h].w[0]; d[1] = v[h].w[1]; d[2] = v[h].w[2]; d[3] = v[h].w[3];
  }
  __syncthreads();
  const uint16_t* t16 = reinterpret_cast<const uint16_t*>(&tile[0][0]);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int c = lr + 32 * h;  // column of x = row of y; ch = which eight consecutive r
    uint32_t w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t lo = t16[(ch * 8 + 2 * j) * 66 + c], hi = t16[(ch * 8 + 2 * j + 1) * 66 + c];
      w[j] = lo | (hi << 16);
    }
    Pack16 o;
    o.w[0] = w[0]; o.w[1] = w[1]; o.w[2] = w[2]; o.w[3] = w[3];
    store16_nt(y + (c0 + c) * y_ld + r0 + ch * 8, o);
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

extern "C" int moq_gptq_block_sweep(float* w, int64_t rows, int64_t ld, int64_t i1, int bs, const float* hinv,
                                    float* delta, const float* amax, int64_t amax_row_stride, int64_t g, int fmt,
                                    int num_bits, int is_unsigned, int narrow, void* stream) {
  if (w == nullptr || hinv == nullptr || delta == nullptr || (amax == nullptr && fmt != 3) || rows < 0 || ld <= 0 || i1 < 0 ||
      bs <= 0 || i1 + bs > ld || amax_row_stride < 0 || g <= 0) {
    set_error("moq_gptq_block_sweep: bad arguments");
    return MOQ_ERR_INVALID;
  }
  if (bs > kSgptMaxBlock || fmt < 1 || fmt > 4 || (fmt == 1 && (num_bits < 2 || num_bits > 16))) {
    set_error("moq_gptq_block_sweep: needs col block <= %d, fmt 1 (INT-k, 2 <= k <= 16), 2 (FP8-E4M3), 3 (MX) or 4 (two-level)",
              kSgptMaxBlock);
    return MOQ_ERR_UNSUPPORTED;
  }
  if (fmt >= 3 && (g > 64 || (g & (g - 1)) != 0 || i1 % g != 0 || bs % g != 0 || mx_fmt(num_bits).kind < 0 ||
                   (fmt == 4 && mx_fmt(is_unsigned).kind < 0))) {
    set_error("moq_gptq_block_sweep: dynamic blocks need a block size that is a power of two <= 64 dividing i1 and bs, and "
              "known element / scale formats");
    return MOQ_ERR_UNSUPPORTED;
  }
  if (rows == 0) return MOQ_OK;
  const dim3 grid((unsigned)((rows + 3) / 4)), block(kBlock);
  IntQ iq = make_intq(fmt == 1 ? num_bits : 8, is_unsigned, narrow);
  if (fmt >= 3) {
    iq.hi = (float)num_bits;    // the element format code rides in the clamp slots (unused by these formats),
    iq.lo = (float)is_unsigned;  // the scale format of fmt 4 in the other one
    if (fmt == 3)
      hipLaunchKernelGGL((gptq_sweep_kernel<3>), grid, block, 0, S(stream), w, rows, ld, i1, bs, hinv, delta, amax,
                         amax_row_stride, g, iq);
    else
      hipLaunchKernelGGL((gptq_sweep_kernel<4>), grid, block, 0, S(stream), w, rows, ld, i1, bs, hinv, delta, amax,
                         amax_row_stride, g, iq);
    return check_launch("moq_gptq_block_sweep");
  }
  if (fmt == 1)
    hipLaunchKernelGGL((gptq_sweep_kernel<1>), grid, block, 0, S(stream), w, rows, ld, i1, bs, hinv, delta, amax,
                       amax_row_stride, g, iq);
  else
    hipLaunchKernelGGL((gptq_sweep_kernel<2>), grid, block, 0, S(stream), w, rows, ld, i1, bs, hinv, delta, amax,
                       amax_row_stride, g, iq);
  return check_launch("moq_gptq_block_sweep");
}

extern "C" int moq_sgpt_trailing_update(float* w, int64_t rows, int64_t ld, int64_t i1, int bs, const float* delta,
                                        const float* hinv, void* stream) {
  if (w == nullptr || hinv == nullptr || delta == nullptr || rows < 0 || ld <= 0 || i1 < 0 || bs <= 0 || i1 + bs > ld) {
    set_error("moq_sgpt_trailing_update: bad arguments");
    return MOQ_ERR_INVALID;
  }
  if (bs > kTuK) {
    set_error("moq_sgpt_trailing_update: column block of %d > %d", bs, kTuK);
    return MOQ_ERR_UNSUPPORTED;
  }
  const int64_t ncols = ld - (i1 + bs);
  if (rows == 0 || ncols == 0) return MOQ_OK;
  const int64_t n_ctiles = (ncols + kTuCols - 1) / kTuCols, gy = (rows + kTuRows - 1) / kTuRows;
  // column tiles per workgroup: long strips amortise the delta tile and keep the pipeline full, but the grid should
  // still cover the chip about twice
  // (rounded UP: 576 workgroups on 256 CUs run as three rounds with the last one a quarter full -- 512 as two)
  int64_t per_wg = (n_ctiles * gy + 511) / 512;
  per_wg = per_wg < 1 ? 1 : (per_wg > 32 ? 32 : per_wg);
  const int64_t gx = (n_ctiles + per_wg - 1) / per_wg;
  if (gy > 65535 || gx > 0x7FFFFFFF) {
    set_error("moq_sgpt_trailing_update: matrix too large");
    return MOQ_ERR_UNSUPPORTED;
  }
  static std::atomic<uint64_t> attr_set{0};  // > 64 KiB of dynamic LDS: opt-in attribute, once per device
  int device = 0;
  (void)hipGetDevice(&device);
  const uint64_t bit = 1ull << (device & 63);
  if (!(attr_set.load(std::memory_order_acquire) & bit)) {
    lds_opt_in((const void*)sgpt_trailing_kernel<true, true>, (int)kTuLds, "sgpt_trailing_kernel");
    lds_opt_in((const void*)sgpt_trailing_kernel<true, false>, (int)kTuLds, "sgpt_trailing_kernel");
    lds_opt_in((const void*)sgpt_trailing_kernel<false, true>, (int)kTuLds, "sgpt_trailing_kernel");
    lds_opt_in((const void*)sgpt_trailing_kernel<false, false>, (int)kTuLds, "sgpt_trailing_kernel");
    attr_set.fetch_or(bit, std::memory_order_release);
  }
  const bool va = bs % 4 == 0 && (reinterpret_cast<uintptr_t>(delta) & 15u) == 0;
  const bool vb = ld % 4 == 0 && (i1 + bs) % 4 == 0 && (reinterpret_cast<uintptr_t>(hinv) & 15u) == 0;
  const dim3 grid((unsigned)gx, (unsigned)gy), block(256);
  const int pw = (int)per_wg;
  if (va && vb) hipLaunchKernelGGL((sgpt_trailing_kernel<true, true>), grid, block, kTuLds, S(stream), w, rows, ld, i1, bs, delta, hinv, pw);
  else if (va) hipLaunchKernelGGL((sgpt_trailing_kernel<true, false>), grid, block, kTuLds, S(stream), w, rows, ld, i1, bs, delta, hinv, pw);
  else if (vb) hipLaunchKernelGGL((sgpt_trailing_kernel<false, true>), grid, block, kTuLds, S(stream), w, rows, ld, i1, bs, delta, hinv, pw);
  else hipLaunchKernelGGL((sgpt_trailing_kernel<false, false>), grid, block, kTuLds, S(stream), w, rows, ld, i1, bs, delta, hinv, pw);
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
  const bool tiles = rows % 64 == 0 && cols % 64 == 0 && y_ld % 8 == 0 &&
                     ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15u) == 0;
  if (tiles) {
    hipLaunchKernelGGL(transpose16_tiles_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(kBlock), 0, S(stream),
                       reinterpret_cast<const uint16_t*>(x), reinterpret_cast<uint16_t*>(y), cols, y_ld);
  } else {
    hipLaunchKernelGGL(transpose16_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(kBlock), 0, S(stream),
                       reinterpret_cast<const uint16_t*>(x), reinterpret_cast<uint16_t*>(y), rows, cols, y_ld);
  }
  return check_launch("moq_transpose16");
}
