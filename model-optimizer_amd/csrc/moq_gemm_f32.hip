// moq_gemm_f32.hip -- the AWQ-lite search contractions for fp32 MODELS on the fp32 matrix cores.
//
// For 16-bit models the error GEMM, the Gram scoring contraction and the plain NT GEMM run on the bf16 / f16 MFMA loop
// of moq_gemm.hip.  An fp32 model's search (quantization/model_calib.py:1489-1495, :1535-1556 with fp32 activations and
// weights) used to fall back to the library (torch.matmul / F.linear); this file is the same three epilogues over
// v_mfma_f32_32x32x2_f32 -- fp32 inputs, fp32 accumulate, bit-for-bit an fmaf chain in ascending k (MI355X_MICROARCH.md:
// 155 TFLOP/s, the fp32 vector rate) -- so no library GEMM is left inside the search.
//
//     out[t, n] = sum_k x[t, k] * w[n, k]  (+ bias[n])         x: [T, K], w: [N, K], both K-contiguous ("NT")
//     MODE 0: partial[wg] = sum (out - ref)^2                   ref fp32 [T, N]   (update_loss, fp32 arithmetic throughout)
//     MODE 1: store out
//     MODE 3: partial[wg] = sum acc * ref                       (<E G, E> of the Gram search; G symmetric, so G^T = G)
//
// Tiling: 128(t) x 128(n) per 256-thread workgroup, K-step 32; each of the four waves owns 64 x 64 as 2 x 2 MFMA tiles
// (64 accumulator registers).  Operand tiles [128][32] fp32 live in LDS with a row pitch of 33 words: the MFMA's A / B
// fragment reads 32 consecutive rows at one k, an odd pitch spreads them over the banks.  Two LDS stages; the next
// K-step's eight 16-byte loads per thread are issued before the current step's 64 MFMAs and written to the other stage
// after them.  Rows past the matrix edge and k past K are zeros (fma(0, 0, acc) = acc).
// Roofline: fp32 MFMA, 2 T N K flop against 155 TFLOP/s.  A cold path (BASELINE's configs are bf16): what matters is that
// it is ours, deterministic and tested; it is not tuned beyond keeping the loads off the MFMAs' critical path.
#include <atomic>

#include "moq_common.h"

namespace moq {

typedef float f32x16w __attribute__((ext_vector_type(16)));
constexpr int kF32Tile = 128, kF32BK = 32, kF32Pitch = 33;
constexpr int kF32Stage = 2 * kF32Tile * kF32Pitch;  // words: A tile + B tile
constexpr size_t kF32Lds = 2 * (size_t)kF32Stage * sizeof(float);

template <int MODE>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ ref, const float* __restrict__ bias,
                                                          float* __restrict__ out, float* __restrict__ partial, int T,
                                                          int N, int K, int tiles_n, int64_t x_stride, int64_t w_stride) {
  extern __shared__ __attribute__((aligned(16))) float f32_lds[];
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tt = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  const int t0 = tt * kF32Tile, n0 = tn * kF32Tile;
  x += (int64_t)blockIdx.y * x_stride;
  w += (int64_t)blockIdx.y * w_stride;
  if constexpr (MODE == 1) out += (int64_t)blockIdx.y * (int64_t)T * N;
  // loader: thread -> k quad q = tid & 7, rows (tid >> 3) + 32 i of both operand tiles
  const int q = tid & 7, lr = tid >> 3;
  float4 ra[4], rb[4];
  auto gload = [&](int k0) {
    const int k = k0 + 4 * q;
    const bool k_ok = k < K;  // K % 4 == 0 (host-checked): a quad is all in or all out
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = lr + 32 * i;
      const bool a_ok = k_ok && t0 + r < T, b_ok = k_ok && n0 + r < N;
      const float4 va = *reinterpret_cast<const float4*>(a_ok ? x + (int64_t)(t0 + r) * K + k : x);
      const float4 vb = *reinterpret_cast<const float4*>(b_ok ? w + (int64_t)(n0 + r) * K + k : w);
      ra[i] = a_ok ? va : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      rb[i] = b_ok ? vb : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
  };
  auto lstore = [&](int stage) {
    float* sa = f32_lds + stage * kF32Stage;
    float* sb = sa + kF32Tile * kF32Pitch;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float* da = sa + (lr + 32 * i) * kF32Pitch + 4 * q;
      float* db = sb + (lr + 32 * i) * kF32Pitch + 4 * q;
      da[0] = ra[i].x; da[1] = ra[i].y; da[2] = ra[i].z; da[3] = ra[i].w;
      db[0] = rb[i].x; db[1] = rb[i].y; db[2] = rb[i].z; db[3] = rb[i].w;
    }
  };
  const int wr = wave >> 1, wc = wave & 1;
  const int m = lane & 31, h = lane >> 5;
  f32x16w acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  const int nk = (K + kF32BK - 1) / kF32BK;
  gload(0);
  lstore(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) gload((kt + 1) * kF32BK);
    __builtin_amdgcn_sched_barrier(0);
    const float* pa = f32_lds + (kt & 1) * kF32Stage + (wr * 64 + m) * kF32Pitch + h;
    const float* pb = f32_lds + (kt & 1) * kF32Stage + kF32Tile * kF32Pitch + (wc * 64 + m) * kF32Pitch + h;
    float a0 = pa[0], a1 = pa[32 * kF32Pitch], b0 = pb[0], b1 = pb[32 * kF32Pitch];
#pragma unroll
    for (int j = 0; j < kF32BK / 2; ++j) {
      const int jn = j + 1 < kF32BK / 2 ? j + 1 : j;
      const float a0n = pa[2 * jn], a1n = pa[32 * kF32Pitch + 2 * jn];
      const float b0n = pb[2 * jn], b1n = pb[32 * kF32Pitch + 2 * jn];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      a0 = a0n; a1 = a1n; b0 = b0n; b1 = b1n;
    }
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 1 < nk) lstore((kt + 1) & 1);  // the other stage: everyone finished reading it before the last barrier
    __syncthreads();
  }
  // accumulator e of a lane: row (t) 8 (e >> 2) + (e & 3) + 4 h, column (n) m
  float sq = 0.0f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wc * 64 + j * 32 + m;
      if (n >= N) continue;
      const float bv = (MODE != 3 && bias != nullptr) ? bias[n] : 0.0f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int t = t0 + wr * 64 + i * 32 + 8 * (e >> 2) + (e & 3) + 4 * h;
        if (t >= T) continue;
        const int64_t off = (int64_t)t * N + n;
        if constexpr (MODE == 0) {
          const float d = (acc[i][j][e] + bv) - ref[off];
          sq += d * d;
        } else if constexpr (MODE == 1) {
          out[off] = acc[i][j][e] + bv;
        } else {
          sq += acc[i][j][e] * ref[off];
        }
      }
    }
  if constexpr (MODE != 1) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sq += __shfl_xor(sq, off, 64);
    if (lane == 0) red[wave] = sq;
    __syncthreads();
    if (tid == 0) partial[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
  }
}

}  // namespace moq

using namespace moq;

// tiles per candidate (= partial sums per candidate for MODE 0 / 3), or a negative status
template <int MODE>
static int64_t launch_f32(const void* x, const void* w, const void* ref, const void* bias, void* out, float* partial,
                          int64_t tokens, int64_t cout, int64_t cin, int n_cand, int64_t x_stride, int64_t w_stride,
                          void* stream) {
  const int64_t tiles_t = (tokens + kF32Tile - 1) / kF32Tile, tiles_n = (cout + kF32Tile - 1) / kF32Tile;
  const int64_t nblk = tiles_t * tiles_n;
  if (nblk > 0x7FFFFFFF || cin % 4 != 0) {
    set_error("gemm (fp32): too many tiles or Cin %% 4 != 0");
    return MOQ_ERR_UNSUPPORTED;
  }
  static std::atomic<uint64_t> attr_set{0};
  int device = 0;
  (void)hipGetDevice(&device);
  const uint64_t bit = 1ull << (device & 63);
  if (!(attr_set.load(std::memory_order_acquire) & bit)) {
    lds_opt_in((const void*)gemm_f32_kernel<MODE>, (int)kF32Lds, "gemm_f32_kernel");
    attr_set.fetch_or(bit, std::memory_order_release);
  }
  hipLaunchKernelGGL((gemm_f32_kernel<MODE>), dim3((unsigned)nblk, (unsigned)n_cand), dim3(256), kF32Lds,
                     reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const float*>(x),
                     reinterpret_cast<const float*>(w), reinterpret_cast<const float*>(ref),
                     reinterpret_cast<const float*>(bias), reinterpret_cast<float*>(out), partial, (int)tokens, (int)cout,
                     (int)cin, (int)tiles_n, x_stride, w_stride);
  return nblk;
}

// entry points used by moq_gemm.hip's C-ABI functions when dt == MOQ_F32
int64_t moq_f32_launch_err(const void* x, const void* w, const void* ref, const void* bias, float* partial, int64_t tokens,
                           int64_t cout, int64_t cin, int n_cand, int64_t x_stride, int64_t w_stride, void* stream) {
  return launch_f32<0>(x, w, ref, bias, nullptr, partial, tokens, cout, cin, n_cand, x_stride, w_stride, stream);
}
int64_t moq_f32_launch_store(const void* x, const void* w, const void* bias, void* out, int64_t tokens, int64_t cout,
                             int64_t cin, void* stream) {
  return launch_f32<1>(x, w, nullptr, bias, out, nullptr, tokens, cout, cin, 1, 0, 0, stream);
}
int64_t moq_f32_launch_dot(const void* a, const void* b, const void* ref, float* partial, int64_t rows, int64_t cols,
                           int64_t k, void* stream) {
  return launch_f32<3>(a, b, ref, nullptr, nullptr, partial, rows, cols, k, 1, 0, 0, stream);
}
