// moq_inputq.hip -- the input-quantizer pass of TensorQuantizer.forward in ONE read of the activation
// (nn/modules/tensor_quantizer.py:1119-1221):
//
//     v = x                                             [rows, cols] in the model dtype
//     v = dtype(x * pre_quant_scale[col])               :1143-1144  (AWQ / SmoothQuant fold on the input side)
//     collect(v): running abs-max                       :1186-1196 -> calib/max.py:63-85
//                 |v| histogram, known range            :            calib/histogram.py:95-130
//     y = QDQ(v, amax)   INT-k or FP8-E4M3, per tensor  :1198-1212 -> tensor_quant.py:607-645, :46-59
//
// The reference runs these as separate eager passes (multiply, amax + amin, histc, ~8 elementwise kernels for the
// fake quantization); round 1 had one kernel per stage (scale_cols, amax, hist_abs, fake_quant_*).  Here every
// enabled stage works on the registers of the same 16-byte packets: 2 B/elem read, 2 B/elem written when there is
// an output -- the stage list is a template, so a disabled stage costs nothing.
//
// Histogram stage: LDS atomics retire about two lanes per clock per CU when the lanes of a wave pile onto a few bins
// (round 1: 2.3 TB/s), and activations do exactly that -- the range is set by a handful of outliers, the bulk sits in
// the lowest bins.  The lowest kHotBins bins are therefore counted WITHOUT atomics, in packed per-lane registers
// (hot_add below) that reach the LDS histogram once per wave at the end.  Only elements beyond the hot bins take the
// (guarded) LDS atomic; with a
// flat distribution that is every element, but then the lanes of a wave spread over the whole histogram and the
// atomics do not serialise.  Counts are exact integers either way (torch.histc semantics, moq_hist.h).
//
// Layout: 1024-thread workgroups (four 256-thread quarters walking their own 8192-element chunks, moq_chunk.h) so
// that the one LDS histogram of a workgroup (<= 64 KiB) is shared by 16 waves; <= 512 workgroups.
// Roofline: HBM.  Algorithmic bytes per element (bf16): 2 (statistics only) or 4 (with an output).
#include <stdlib.h>

#include "moq_common.h"
#include "moq_chunk.h"
#include "moq_hist.h"
#include "moq_ops.h"

namespace moq {

constexpr int kIqBlock = 1024;
constexpr int kHotBins = 8;

// Hot bins without atomics: every lane keeps the counts of the kHotBins lowest bins for ITS elements in two packed
// registers (8-bit fields: bins 0-3 in `lo`, 4-7 in `hi`) -- nine full-rate VALU ops per element (shift, shift, and
// twice compare / select / add), no scalar-unit work (the CU's one scalar ALU would cap a ballot + s_bcnt1 scheme at
// ~3 TB/s) and no cross-lane traffic.  A lane sees 32 elements per chunk, so the fields are emptied into 32-bit
// per-lane counters every kHotFlush = 7 chunks (224 <= 255); the counters are summed over the wave once, at the end.
static_assert(kHotBins == 8, "two packed accumulators of four 8-bit fields");
constexpr int kHotFlush = 7;
struct HotAcc {
  uint32_t lo, hi;
};
__device__ __forceinline__ void hot_add(HotAcc& a, int b) {
  const uint32_t m = 1u << (((uint32_t)b << 3) & 31u);  // field of bin (b & 3); the hardware masks the shift anyway
  a.lo += (uint32_t)b < 4u ? m : 0u;
  a.hi += (uint32_t)(b - 4) < 4u ? m : 0u;
}
__device__ __forceinline__ void hot_flush(HotAcc& a, uint32_t (&cnt)[kHotBins]) {
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    cnt[h] += (a.lo >> (8 * h)) & 0xFFu;
    cnt[4 + h] += (a.hi >> (8 * h)) & 0xFFu;
  }
  a.lo = a.hi = 0;
}

struct IqParams {
  const void* x;
  const float* pqs;        // [cols] fp32 holding model-dtype values, or null
  void* y;                 // output (may alias x), or null: statistics only
  int64_t n, cols;
  uint32_t* amax_bits;     // running abs-max as an fp32 bit pattern (atomicMax), or null
  const float* qdq_amax;   // fp32 [1] for the QDQ stage
  int num_bits, is_unsigned, narrow;
  unsigned long long* counts;  // histogram stage
  int bins;
  float max_edge;
  int skip_zeros, rshift;
};

// FMT: 0 no quantization, 1 INT-k, 2 FP8-E4M3
template <int DT, int FMT, bool PQS, bool AMAX, bool HIST, bool SHARED>
__global__ __launch_bounds__(kIqBlock) void input_quant_kernel(const IqParams p) {
  constexpr int V = Elem<DT>::kVec;
  constexpr int P = Chunk<DT>::kPackets;
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_hist[];
  __shared__ uint32_t s_max[kIqBlock / 64];
  const int tid = threadIdx.x & (kBlock - 1);
  const int lane = threadIdx.x & 63;
  const int copy = (int)(threadIdx.x & ((1u << p.rshift) - 1u));
  if constexpr (HIST) {
    const int slots = (p.bins + 1) << p.rshift;
    for (int b = threadIdx.x; b < slots; b += kIqBlock) lds_hist[b] = 0;
    __syncthreads();
  }
  OpIntQdq opi;
  OpFp8Qdq opf;
  if constexpr (FMT == 1) {
    opi.q = make_intq(p.num_bits, p.is_unsigned, p.narrow);
    opi.set(p.qdq_amax[0]);
  } else if constexpr (FMT == 2) {
    opf.sc = fp8_scale(p.qdq_amax[0]);
  }
  const SharedDiv sd = make_shared_div(p.max_edge);
  HotAcc hot_acc = {0u, 0u};
  uint32_t hot[kHotBins];
#pragma unroll
  for (int h = 0; h < kHotBins; ++h) hot[h] = 0;
  int hot_age = 0;
  uint32_t amax_acc = 0;
  const bool al = aligned16(p.x) && (p.y == nullptr || aligned16(p.y));
  const int64_t n_chunks = (p.n + MOQ_MT_CHUNK - 1) / MOQ_MT_CHUNK;
  constexpr int Q = kIqBlock / kBlock;
  const int quarter = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kBlock));
  for (int64_t c = (int64_t)blockIdx.x * Q + quarter; c < n_chunks; c += (int64_t)gridDim.x * Q) {
    const int64_t e0 = c * MOQ_MT_CHUNK;
    const bool fast = al && e0 + MOQ_MT_CHUNK <= p.n;
    if constexpr (HIST) {
      if (++hot_age > kHotFlush) {
        hot_flush(hot_acc, hot);
        hot_age = 1;
      }
    }
    int64_t col0 = 0;
    if constexpr (PQS) col0 = e0 % p.cols;  // wave-uniform; cols % V == 0 (host-checked): a packet stays in one row
    Pack16 in[P];
#pragma unroll
    for (int u = 0; u < P; ++u) {
      const int64_t e = e0 + (u * kBlock + tid) * V;
      in[u] = fast ? ld_packet<DT, true>(p.x, e, p.n) : ld_packet<DT, false>(p.x, e, p.n);
    }
#pragma unroll
    for (int u = 0; u < P; ++u) {
      const int64_t e = e0 + (u * kBlock + tid) * V;
      float f[8];
      unpack<DT>(in[u], f);
      if constexpr (PQS) {
        int64_t col = col0 + (u * kBlock + tid) * V;
        if (col >= p.cols) col %= p.cols;
        if (e < p.n) {
#pragma unroll
          for (int i = 0; i < V; i += 4) {
            const float4 s = *reinterpret_cast<const float4*>(p.pqs + col + i);
            f[i] = round_to_dtype<DT>(f[i] * s.x);
            f[i + 1] = round_to_dtype<DT>(f[i + 1] * s.y);
            f[i + 2] = round_to_dtype<DT>(f[i + 2] * s.z);
            f[i + 3] = round_to_dtype<DT>(f[i + 3] * s.w);
          }
        }
      }
      if constexpr (AMAX) {
        // past-the-end elements were loaded as zeros (ld_packet), the identity of abs-max
#pragma unroll
        for (int i = 0; i < V; ++i) {
          const uint32_t a = absbits(f[i]);
          amax_acc = a > amax_acc ? a : amax_acc;
        }
      }
      if constexpr (HIST) {
#pragma unroll
        for (int i = 0; i < V; ++i) {
          int b = hist_bin<SHARED>(__builtin_fabsf(f[i]), p.bins, p.max_edge, sd, p.skip_zeros);
          if (!fast) b = e + i < p.n ? b : p.bins;
          hot_add(hot_acc, b);
          if (b >= kHotBins && b < p.bins) atomicAdd(&lds_hist[(b << p.rshift) + copy], 1u);
        }
      }
      if constexpr (FMT == 1) opi(f, V);
      if constexpr (FMT == 2) opf(f, V);
      if constexpr (FMT != 0 || PQS) {
        if (p.y != nullptr) {
          if (fast) st_packet<DT, true>(p.y, e, p.n, pack<DT>(f));
          else st_packet<DT, false>(p.y, e, p.n, pack<DT>(f));
        }
      }
    }
  }
  if constexpr (AMAX) {
    amax_acc = group_max_u32<64>(amax_acc);
    if (lane == 0) s_max[threadIdx.x >> 6] = amax_acc;
  }
  if constexpr (HIST) {
    // the lanes' hot-bin counters are summed over the wave (butterfly), lane h adds bin h to the workgroup histogram
    hot_flush(hot_acc, hot);
    uint32_t mine = 0;
#pragma unroll
    for (int h = 0; h < kHotBins; ++h) {
      uint32_t v = hot[h];
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) v += (uint32_t)__shfl_xor((int)v, off, 64);
      mine = lane == h ? v : mine;
    }
    if (lane < kHotBins && mine != 0) atomicAdd(&lds_hist[(lane << p.rshift) + copy], mine);
  }
  __syncthreads();
  if constexpr (AMAX) {
    if (threadIdx.x == 0) {
      uint32_t m = s_max[0];
      for (int w = 1; w < kIqBlock / 64; ++w) m = s_max[w] > m ? s_max[w] : m;
      atomicMax(p.amax_bits, m);  // non-negative float patterns order like uints; NaN patterns sit above inf
    }
  }
  if constexpr (HIST) {
    for (int b = threadIdx.x; b < p.bins; b += kIqBlock) {
      uint32_t cnt = 0;
      for (int r = 0; r < (1 << p.rshift); ++r) cnt += lds_hist[(b << p.rshift) + r];
      if (cnt) atomicAdd(&p.counts[b], (unsigned long long)cnt);
    }
  }
}

}  // namespace moq

using namespace moq;
static inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }

constexpr int kIqMaxLdsBins = 16384;  // 64 KiB of LDS, as moq_hist_abs

template <int DT, int FMT, bool PQS>
static void launch_iq(const IqParams& p, bool amax, bool hist, bool shared, int blocks, size_t lds, void* stream) {
#define MOQ_IQ_GO(A, H, SH) \
  hipLaunchKernelGGL((input_quant_kernel<DT, FMT, PQS, A, H, SH>), dim3(blocks), dim3(kIqBlock), lds, S(stream), p)
  if (amax && hist) { if (shared) MOQ_IQ_GO(true, true, true); else MOQ_IQ_GO(true, true, false); }
  else if (hist) { if (shared) MOQ_IQ_GO(false, true, true); else MOQ_IQ_GO(false, true, false); }
  else if (amax) MOQ_IQ_GO(true, false, false);
  else MOQ_IQ_GO(false, false, false);
#undef MOQ_IQ_GO
}

extern "C" int moq_input_quant(const void* x, const float* pre_quant_scale, void* y, int64_t rows, int64_t cols, int dt,
                               float* amax_running, const float* qdq_amax, int fmt, int num_bits, int is_unsigned,
                               int narrow_range, unsigned long long* hist_counts, int hist_bins, float hist_max_edge,
                               int hist_skip_zeros, void* stream) {
  if (rows < 0 || cols <= 0 || (rows > 0 && x == nullptr)) {
    set_error("moq_input_quant: bad arguments");
    return MOQ_ERR_INVALID;
  }
  if (fmt < 0 || fmt > 2 || (fmt != 0 && (qdq_amax == nullptr || y == nullptr))) {
    set_error("moq_input_quant: fmt must be 0 (none), 1 (INT-k) or 2 (FP8-E4M3); a quantizing call needs qdq_amax and y");
    return MOQ_ERR_INVALID;
  }
  if (fmt == 1 && (num_bits < 2 || num_bits > 16)) {
    set_error("moq_input_quant: num_bits=%d out of range", num_bits);
    return MOQ_ERR_INVALID;
  }
  if (hist_counts != nullptr && (hist_bins < kHotBins || hist_bins >= kIqMaxLdsBins)) {
    set_error("moq_input_quant: histogram stage needs %d <= bins < %d (use moq_hist_abs beyond)", kHotBins, kIqMaxLdsBins);
    return MOQ_ERR_UNSUPPORTED;
  }
  const int vec = dt == MOQ_F32 ? 4 : 8;
  if (pre_quant_scale != nullptr && (cols % vec != 0 || (reinterpret_cast<uintptr_t>(pre_quant_scale) & 15u) != 0)) {
    set_error("moq_input_quant: pre_quant_scale needs cols %% %d == 0 and a 16-byte aligned scale vector", vec);
    return MOQ_ERR_UNSUPPORTED;
  }
  if (pre_quant_scale != nullptr && fmt == 0 && y == nullptr && amax_running == nullptr && hist_counts == nullptr) {
    set_error("moq_input_quant: nothing to do");
    return MOQ_ERR_INVALID;
  }
  const int64_t n = rows * cols;
  if (n == 0) return MOQ_OK;
  IqParams p;
  p.x = x; p.pqs = pre_quant_scale; p.y = y; p.n = n; p.cols = cols;
  p.amax_bits = reinterpret_cast<uint32_t*>(amax_running);
  p.qdq_amax = qdq_amax; p.num_bits = num_bits; p.is_unsigned = is_unsigned; p.narrow = narrow_range;
  p.counts = hist_counts; p.bins = hist_bins > 0 ? hist_bins : 1; p.max_edge = hist_max_edge;
  p.skip_zeros = hist_skip_zeros; p.rshift = 0;
  size_t lds = 0;
  bool shared = false;
  if (hist_counts != nullptr) {
    while (p.rshift < 3 && ((int64_t)(hist_bins + 1) << (p.rshift + 1)) <= kIqMaxLdsBins + 8) ++p.rshift;
    lds = ((size_t)(hist_bins + 1) << p.rshift) * 4;
    shared = hist_max_edge >= 0x1p-60f && hist_max_edge <= 0x1p60f && hist_max_edge * (float)hist_bins <= 65536.0f;
  }
  int64_t blocks = ((n + MOQ_MT_CHUNK - 1) / MOQ_MT_CHUNK + 3) / 4;
  if (blocks > 512) blocks = 512;
  const bool amax = amax_running != nullptr, hist = hist_counts != nullptr;
#define MOQ_IQ_FMT(F)                                                                                         \
  if (pre_quant_scale != nullptr) {                                                                           \
    MOQ_DISPATCH_DTYPE(dt, (launch_iq<DT, F, true>(p, amax, hist, shared, (int)blocks, lds, stream)));        \
  } else {                                                                                                    \
    MOQ_DISPATCH_DTYPE(dt, (launch_iq<DT, F, false>(p, amax, hist, shared, (int)blocks, lds, stream)));       \
  }
  if (fmt == 0) { MOQ_IQ_FMT(0) } else if (fmt == 1) { MOQ_IQ_FMT(1) } else { MOQ_IQ_FMT(2) }
#undef MOQ_IQ_FMT
  return check_launch("moq_input_quant");
}
