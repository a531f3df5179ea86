// moq_block2d.hip -- 2-D block quantization (a2): abs-max and quantize-dequantize over br x bc tiles of a
// [rows, cols] tensor (the FP8 2-D blockwise weight preset, block_sizes {-1: 128, -2: 128}).
//
// The reference reshapes to [rows/br, br, cols/bc, bc], reduces twice (reduce_block_amax,
// quantization/utils/core_utils.py:43-90; TensorQuantizer._setup_for_blockquant's general path,
// nn/modules/tensor_quantizer.py:1018-1043) and, because the amax then has two non-singleton dims, falls back to the
// eager fake-quant (tensor_quant.py:80) -- ~10 elementwise passes.  Here one workgroup owns one tile: its
// br x bc elements (<= 16 packets per thread) stay in registers between the abs-max reduction and the QDQ, so the
// tensor is read once and written once (4 B/element in bf16, HBM-bound).
#include "moq_common.h"

namespace moq {

constexpr int kB2dMaxPackets = 16;

// MODE 0: amax only (ACC: running max with the stored value); MODE 1: QDQ with the given amax; MODE 2: amax + QDQ.
template <int DT, bool FP8, int MODE>
__global__ __launch_bounds__(kBlock) void block2d_kernel(const void* __restrict__ x, void* __restrict__ y,
                                                         float* __restrict__ amax, int64_t cols, int br, int bc,
                                                         int ppr_shift, int accumulate, int num_bits,
                                                         int is_unsigned, int narrow) {
  constexpr int V = Elem<DT>::kVec;
  constexpr int ES = 16 / V;
  __shared__ uint32_t smem[kBlock / 64];
  __shared__ float s_amax;
  const int ppr = bc / V;          // packets per tile row
  const int n_items = br * ppr;    // packets per tile
  const int64_t base = ((int64_t)blockIdx.y * br) * cols + (int64_t)blockIdx.x * bc;
  const char* xb = reinterpret_cast<const char*>(x);
  Pack16 in[kB2dMaxPackets];
  uint32_t acc = 0;
#pragma unroll
  for (int u = 0; u < kB2dMaxPackets; ++u) {
    const int item = u * kBlock + (int)threadIdx.x;
    if (item < n_items) {
      const int r = ppr_shift >= 0 ? (item >> ppr_shift) : (item / ppr);
      const int p = item - r * ppr;
      in[u] = load16_nt(xb + (base + (int64_t)r * cols + p * V) * ES);
      if constexpr (MODE != 1) {
        const uint32_t m = pack_absmax<DT>(in[u]);
        acc = m > acc ? m : acc;
      }
    }
  }
  const int64_t a_idx = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
  float am;
  if constexpr (MODE != 1) {
    acc = block_max_u32(acc, smem);
    if (threadIdx.x == 0) {
      if (accumulate) {
        const uint32_t old = __float_as_uint(amax[a_idx]);
        acc = old > acc ? old : acc;
      }
      if (amax != nullptr) amax[a_idx] = __uint_as_float(acc);
      s_amax = __uint_as_float(acc);
    }
    if constexpr (MODE == 0) return;
    __syncthreads();
    am = s_amax;
  } else {
    am = amax[a_idx];
  }
  char* yb = reinterpret_cast<char*>(y);
  const IntQ q = make_intq(num_bits, is_unsigned, narrow);
  const float iscale = int_scale(am, q.hi);
  const SharedDiv sd = make_shared_div(iscale);
  const Fp8Scale fs = fp8_scale(am);
#pragma unroll
  for (int u = 0; u < kB2dMaxPackets; ++u) {
    const int item = u * kBlock + (int)threadIdx.x;
    if (item < n_items) {
      const int r = ppr_shift >= 0 ? (item >> ppr_shift) : (item / ppr);
      const int p = item - r * ppr;
      float f[8];
      unpack<DT>(in[u], f);
#pragma unroll
      for (int i = 0; i < V; i += 2) {
        if constexpr (FP8) {
          const float a = f[i] * fs.s, b = f[i + 1] * fs.s;
          float ca = __builtin_fminf(__builtin_fmaxf(a, -448.0f), 448.0f);
          float cb = __builtin_fminf(__builtin_fmaxf(b, -448.0f), 448.0f);
          ca = (a != a) ? a : ca;
          cb = (b != b) ? b : cb;
          float ra, rb;
          e4m3_roundtrip2(ca, cb, ra, rb);
          f[i] = ra * fs.inv;
          f[i + 1] = rb * fs.inv;
        } else {
          f[i] = qdq_int_shared(f[i], iscale, sd, q);
          f[i + 1] = qdq_int_shared(f[i + 1], iscale, sd, q);
        }
      }
      store16_nt(yb + (base + (int64_t)r * cols + p * V) * ES, pack<DT>(f));
    }
  }
}

// Abs-max of tiles the packet kernel does not take (a tile row that is not whole 16-byte packets: the 2 ... 8 wide blocks of
// small test shapes and of odd layouts, or an unaligned base).  One workgroup per tile, one element per thread and step,
// 16-bit rows read as single elements; the statistic is the same fp32 pattern (an unsigned max of |x|'s bits, NaN on top).
template <int DT>
__global__ __launch_bounds__(kBlock) void block2d_amax_generic_kernel(const void* __restrict__ x, float* __restrict__ amax,
                                                                      int64_t cols, int br, int bc, int accumulate) {
  __shared__ uint32_t smem[kBlock / 64];
  using T = typename Elem<DT>::storage;
  const T* xe = reinterpret_cast<const T*>(x) + ((int64_t)blockIdx.y * br) * cols + (int64_t)blockIdx.x * bc;
  uint32_t acc = 0;
  const int n = br * bc;
  for (int i = (int)threadIdx.x; i < n; i += kBlock) {
    const int r = i / bc, c = i - r * bc;
    uint32_t a;
    if constexpr (DT == MOQ_F32) a = __float_as_uint(xe[(int64_t)r * cols + c]) & 0x7FFFFFFFu;
    else a = widen_abs16<DT>((uint32_t)xe[(int64_t)r * cols + c] & 0x7FFFu);
    acc = a > acc ? a : acc;
  }
  acc = block_max_u32(acc, smem);
  if (threadIdx.x == 0) {
    const int64_t a_idx = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
    if (accumulate) {
      const uint32_t old = __float_as_uint(amax[a_idx]);
      acc = old > acc ? old : acc;
    }
    amax[a_idx] = __uint_as_float(acc);
  }
}

}  // namespace moq

using namespace moq;
static inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }

extern "C" int moq_block2d(const void* x, void* y, float* amax, int64_t rows, int64_t cols, int br, int bc, int dt,
                           int mode, int accumulate, int fp8, int num_bits, int is_unsigned, int narrow_range,
                           void* stream) {
  if (rows < 0 || cols < 0 || br <= 0 || bc <= 0 || mode < 0 || mode > 2 || (rows * cols > 0 && x == nullptr) ||
      (mode != 0 && rows * cols > 0 && y == nullptr) || (mode != 2 && amax == nullptr)) {
    set_error("moq_block2d: bad arguments");
    return MOQ_ERR_INVALID;
  }
  if (!fp8 && mode != 0 && (num_bits < 2 || num_bits > 16)) {
    set_error("moq_block2d: num_bits=%d out of range [2,16]", num_bits);
    return MOQ_ERR_INVALID;
  }
  if (rows * cols == 0) return MOQ_OK;
  const int vec = dt == MOQ_F32 ? 4 : 8;
  const int64_t packets = (int64_t)br * (bc / vec);
  if (mode == 0 && rows % br == 0 && cols % bc == 0 && cols / bc <= 0x7FFFFFFF && rows / br <= 65535 &&
      (int64_t)br * bc <= 0x7FFFFFFF &&
      (bc % vec != 0 || packets > (int64_t)kB2dMaxPackets * kBlock || (reinterpret_cast<uintptr_t>(x) & 15u) != 0)) {
    const dim3 ggrid((unsigned)(cols / bc), (unsigned)(rows / br));
    MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((block2d_amax_generic_kernel<DT>), ggrid, dim3(kBlock), 0, S(stream), x, amax, cols,
                                              br, bc, accumulate))
    return check_launch("moq_block2d");
  }
  if (rows % br != 0 || cols % bc != 0 || bc % vec != 0 || packets > (int64_t)kB2dMaxPackets * kBlock ||
      ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15u) != 0 || cols / bc > 0x7FFFFFFF ||
      rows / br > 65535) {
    set_error("moq_block2d: needs rows %% br == 0, cols %% bc == 0, bc %% %d == 0, br * bc <= %d elements, 16-byte "
              "aligned tensors (pad on the host like reduce_block_padding)", vec, kB2dMaxPackets * kBlock * vec);
    return MOQ_ERR_UNSUPPORTED;
  }
  const int ppr = bc / vec;
  int shift = -1;
  if ((ppr & (ppr - 1)) == 0) { shift = 0; while ((1 << shift) < ppr) ++shift; }
  const dim3 grid((unsigned)(cols / bc), (unsigned)(rows / br));
#define MOQ_B2D(FP8V, MODEV) \
  MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((block2d_kernel<DT, FP8V, MODEV>), grid, dim3(kBlock), 0, S(stream), x, y, amax, cols, br, bc, shift, accumulate, num_bits, is_unsigned, narrow_range))
  if (mode == 0) { MOQ_B2D(false, 0); }
  else if (fp8) { if (mode == 1) { MOQ_B2D(true, 1); } else { MOQ_B2D(true, 2); } }
  else { if (mode == 1) { MOQ_B2D(false, 1); } else { MOQ_B2D(false, 2); } }
#undef MOQ_B2D
  return check_launch("moq_block2d");
}
