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
// Histogram stage.  LDS atomics are not what bounds a histogram on this chip -- a probe (tools/exp/lds_atomic_probe.hip,
// profiles/r02_lds_atomic_probe.txt) retires 10 random / 6.5 hot-binned ds_add_u32 lanes per clock per CU, four times
// what round 1's kernel used -- the ~15 VALU instructions per element of torch.histc's binning rule are (multiply, exact
// division, convert, clamp, range tests).  A 16-bit input has only 2^15 distinct |x| patterns, so each workgroup first
// tabulates the rule once -- lut[pattern] = histogram slot, built with the very same hist_bin() (64 KiB of LDS) -- and
// an element then costs a field extract, one ds_read_u16 and the ds_add_u32: 3 VALU + 2 LDS instructions.  Results are
// identical by construction.  fp32 inputs keep the arithmetic rule.  Counts are exact integers either way.
//
// Round 6, PATTERN COUNTS (PAT): the table read is itself one of the element's two LDS instructions and sits in front of
// the atomic (a round trip per packet), and the table build is ~5 us per launch.  So when the bins leave room (<= ~7000),
// the workgroup counts the 2^15 |x| PATTERNS instead -- 128 KiB of u32 counters, one fire-and-forget ds_add_u32 per
// element, nothing to build -- and applies the binning rule once per pattern THAT OCCURRED when it flushes: hist_bin() of
// the pattern's value, its count added to the workgroup's bins, the bins to global memory as before.  Patterns spread
// over the counters by themselves (an octave is 128 / 1024 patterns wide), except |x| == 0 (ReLU outputs, padding): zeros
// take one counter per lane.  Same hist_bin(), same counts.
//
// Layout: 1024-thread workgroups (four 256-thread quarters walking their own 8192-element chunks, moq_chunk.h) so
// that the one LDS histogram (+ table) of a workgroup is shared by 16 waves; <= 512 workgroups (one per CU with the
// table: 64 KiB table + <= 64 KiB of interleaved histogram copies).
// Roofline: HBM.  Algorithmic bytes per element (bf16): 2 (statistics only) or 4 (with an output).
#include <stdlib.h>

#include "moq_common.h"
#include "moq_chunk.h"
#include "moq_hist.h"
#include "moq_ops.h"

namespace moq {

constexpr int kIqBlock = 1024;
constexpr int kLutEntries = 32768;  // |x| patterns of a 16-bit float
// Same-address LDS atomics serialise (the probe: 1.6 lanes per clock on 4 addresses, 6.5 on 4 bins x 8 copies, 14.6
// when every lane has its own dword), and activations whose range is set by a few outliers put most elements into
// the very lowest bins.  The kHotBins lowest bins therefore get one copy PER LANE (conflict-free whatever the data);
// the other bins share R = 2^rshift interleaved copies.  Slot layout: [hot bins: bin * 64 + lane][bins: bin * R + copy].
// PAT layout: [pattern counters: 32768][past-the-end trash: 64][|x| == 0, one per lane: 64][bins + 1]
constexpr int kPatTrash = kLutEntries;
constexpr int kPatZero = kLutEntries + 64;
constexpr int kPatSlots = kLutEntries + 128;
constexpr int kHotBins = 32;
constexpr int kHotSlots = kHotBins * 64;
__device__ __forceinline__ uint32_t slot_base(int bin, int rshift) {
  return bin < kHotBins ? (uint32_t)bin * 64u : (uint32_t)kHotSlots + ((uint32_t)bin << rshift);
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
  int dbg;  // experiment build only: timing diagnostics that skip parts of the histogram stage (results are wrong)
};

// FMT: 0 no quantization, 1 INT-k, 2 FP8-E4M3
#ifdef MOQ_EXPERIMENTS
#define MOQ_IQ_DBG(bit) ((p.dbg >> (bit)) & 1)
#else
#define MOQ_IQ_DBG(bit) 0
#endif
template <int DT, int FMT, bool PQS, bool AMAX, bool HIST, bool SHARED, bool PAT>
__global__ __launch_bounds__(kIqBlock) void input_quant_kernel(const IqParams p) {
  constexpr int V = Elem<DT>::kVec;
  constexpr int P = Chunk<DT>::kPackets;
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_hist[];
  __shared__ uint32_t s_max[kIqBlock / 64];
  const int tid = threadIdx.x & (kBlock - 1);
  const int lane = threadIdx.x & 63;
  const int copy = (int)(threadIdx.x & ((1u << p.rshift) - 1u));
  const SharedDiv sd = make_shared_div(p.max_edge);
  constexpr bool PATM = HIST && PAT && DT != MOQ_F32;
  constexpr bool LUT = HIST && DT != MOQ_F32 && !PATM;
  const int slots = PATM ? kPatSlots + p.bins + 1 : kHotSlots + ((p.bins + 1) << p.rshift);
  // LDS: [histogram: slots x u32][table: 32768 x u16 (16-bit inputs)]
  uint16_t* lut = reinterpret_cast<uint16_t*>(lds_hist + slots);
  // The first two chunks of every quarter are requested BEFORE the histogram is zeroed and the table is built: at the
  // sizes the calibration flow presents (one linear input per batch: 33-117 MB, 2-8 chunks per quarter) the ~1.5 us of
  // table arithmetic would otherwise sit in front of the first byte of the stream.
  const bool al = aligned16(p.x) && (p.y == nullptr || aligned16(p.y));
  const int64_t n_chunks = (p.n + MOQ_MT_CHUNK - 1) / MOQ_MT_CHUNK;
  constexpr int Q = kIqBlock / kBlock;
  const int quarter = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kBlock));
  const int64_t stride = (int64_t)gridDim.x * Q;
  auto load_chunk = [&](int64_t c, Pack16 (&in)[P]) {
    const int64_t e0 = c * MOQ_MT_CHUNK;
    const bool fast = al && e0 + MOQ_MT_CHUNK <= p.n;
    if (fast) {  // one branch per chunk, not per packet: the P loads go out back to back
#pragma unroll
      for (int u = 0; u < P; ++u) in[u] = ld_packet<DT, true>(p.x, e0 + (u * kBlock + tid) * V, p.n);
    } else {
#pragma unroll
      for (int u = 0; u < P; ++u) in[u] = ld_packet<DT, false>(p.x, e0 + (u * kBlock + tid) * V, p.n);
    }
  };
  Pack16 buf_a[P], buf_b[P];
  const int64_t c_first = (int64_t)blockIdx.x * Q + quarter;
  if (c_first < n_chunks) load_chunk(c_first, buf_a);
  if (c_first + stride < n_chunks) load_chunk(c_first + stride, buf_b);
  if constexpr (HIST) {
    if constexpr (PATM) {  // (the allocation is rounded up to whole 16-byte packets)
      const u32x4_t zero = {0u, 0u, 0u, 0u};
      for (int b = threadIdx.x; b < (slots + 3) / 4; b += kIqBlock) reinterpret_cast<u32x4_t*>(lds_hist)[b] = zero;
    } else {
      for (int b = threadIdx.x; b < slots; b += kIqBlock) lds_hist[b] = 0;
    }
    if constexpr (LUT) {
      // lut[|x| pattern] = first slot of the pattern's bin (slot_base; invalid -> the trash bin `bins`)
      for (int v = threadIdx.x; v < kLutEntries; v += kIqBlock) {
        float a;
        if constexpr (DT == MOQ_BF16) a = __uint_as_float((uint32_t)v << 16);
        else { uint16_t h = (uint16_t)v; a = (float)*reinterpret_cast<_Float16*>(&h); }
        lut[v] = (uint16_t)slot_base(hist_bin<SHARED>(a, p.bins, p.max_edge, sd, p.skip_zeros), p.rshift);
      }
    }
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
  uint32_t amax_acc = 0;
  uint32_t amax_seen = 0;
  if constexpr (AMAX) {
    if (threadIdx.x == 0) amax_seen = __builtin_nontemporal_load(p.amax_bits);
  }
  int64_t toff[PQS ? P : 1];  // (packet offset inside a chunk) mod cols, once per thread
  if constexpr (PQS) {
#pragma unroll
    for (int u = 0; u < P; ++u) toff[u] = (int64_t)((u * kBlock + tid) * V) % p.cols;
  }
  // Two chunks in flight per quarter: the loads of the next chunk are issued before the current one is worked on, so
  // a wave's HBM latency hides under its own LDS / VALU phase (with the 64 KiB table there is one workgroup per CU).
  auto work_chunk = [&](int64_t c, const Pack16 (&in)[P]) {
    const int64_t e0 = c * MOQ_MT_CHUNK;
    const bool fast = al && e0 + MOQ_MT_CHUNK <= p.n;
    int64_t col0 = 0;
    if constexpr (PQS) col0 = e0 % p.cols;  // wave-uniform; cols % V == 0 (host-checked): a packet stays in one row
    // the pre_quant_scale vectors of all packets are requested before the first is used (column = col0 + toff[u], both
    // below cols: one conditional subtraction wraps; the first form took a 64-bit modulo per packet in a divergent
    // branch and loaded the vectors one packet at a time)
    float4 pq[PQS ? P : 1][V / 4];
    if constexpr (PQS) {
#pragma unroll
      for (int u = 0; u < P; ++u) {
        int64_t col = col0 + toff[u];
        col = col >= p.cols ? col - p.cols : col;
#pragma unroll
        for (int i = 0; i < V / 4; ++i) pq[u][i] = *reinterpret_cast<const float4*>(p.pqs + col + 4 * i);
      }
    }
#pragma unroll
    for (int u = 0; u < P; ++u) {
      const int64_t e = e0 + (u * kBlock + tid) * V;
      float f[8];
      unpack<DT>(in[u], f);
      if constexpr (PQS) {
        if (fast || e < p.n) {
#pragma unroll
          for (int i = 0; i < V; i += 4) {
            const float4 s = pq[u][i / 4];
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
        if constexpr (PATM) {
          const Pack16 pv = PQS ? pack<DT>(f) : in[u];
          const uint32_t zslot = (uint32_t)(kPatZero + lane);
          if (MOQ_IQ_DBG(3)) {
          } else if (fast) {  // (uniform; two bodies so that the live one carries no range tests)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const uint32_t lo = pv.w[i] & 0x7FFFu, hi = (pv.w[i] >> 16) & 0x7FFFu;
              atomicAdd(&lds_hist[lo ? lo : zslot], 1u);
              atomicAdd(&lds_hist[hi ? hi : zslot], 1u);
            }
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const uint32_t lo = pv.w[i] & 0x7FFFu, hi = (pv.w[i] >> 16) & 0x7FFFu;
              atomicAdd(&lds_hist[e + 2 * i < p.n ? (lo ? lo : zslot) : (uint32_t)kPatTrash], 1u);
              atomicAdd(&lds_hist[e + 2 * i + 1 < p.n ? (hi ? hi : zslot) : (uint32_t)kPatTrash], 1u);
            }
          }
        } else if constexpr (LUT) {
          // the 16-bit patterns of v: straight from the packet, or re-packed when a pre_quant_scale changed them
          const Pack16 pv = PQS ? pack<DT>(f) : in[u];
          uint32_t slot[8];  // all table reads of the packet first, then its atomics: one LDS round trip per packet
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            slot[2 * i] = lut[pv.w[i] & 0x7FFFu];
            slot[2 * i + 1] = lut[(pv.w[i] >> 16) & 0x7FFFu];
          }
          if (!fast) {
#pragma unroll
            for (int i = 0; i < 8; ++i) slot[i] = e + i < p.n ? slot[i] : slot_base(p.bins, p.rshift);
          }
#pragma unroll
          for (int i = 0; i < 8; ++i) atomicAdd(&lds_hist[slot[i] + (slot[i] < (uint32_t)kHotSlots ? lane : copy)], 1u);
        } else {
#pragma unroll
          for (int i = 0; i < V; ++i) {
            int b = hist_bin<SHARED>(__builtin_fabsf(f[i]), p.bins, p.max_edge, sd, p.skip_zeros);
            if (!fast) b = e + i < p.n ? b : p.bins;
            atomicAdd(&lds_hist[slot_base(b, p.rshift) + (b < kHotBins ? lane : copy)], 1u);
          }
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
  };
  {
    int64_t c = c_first;  // buf_a / buf_b already hold chunks c and c + stride (requested above)
    while (c < n_chunks) {
      const int64_t c1 = c + stride;
      work_chunk(c, buf_a);
      if (c1 >= n_chunks) break;
      const int64_t c2 = c1 + stride;
      if (c2 < n_chunks) load_chunk(c2, buf_a);
      work_chunk(c1, buf_b);
      if (c2 >= n_chunks) break;
      const int64_t c3 = c2 + stride;
      if (c3 < n_chunks) load_chunk(c3, buf_b);
      c = c2;
    }
  }
  if constexpr (AMAX) {
    amax_acc = group_max_u32<64>(amax_acc);
    if (lane == 0) s_max[threadIdx.x >> 6] = amax_acc;
  }
  __syncthreads();
  if constexpr (AMAX) {
    if (threadIdx.x == 0) {
      uint32_t m = s_max[0];
      for (int w = 1; w < kIqBlock / 64; ++w) m = s_max[w] > m ? s_max[w] : m;
      // non-negative float patterns order like uints; NaN patterns sit above inf.  Skipped when the running maximum
      // already covered this workgroup's result at its start (amax_kernel's note)
      if (m > amax_seen) atomicMax(p.amax_bits, m);
    }
  }
  if constexpr (PATM) {
    // the binning rule once per pattern that occurred; step i covers patterns [1024 i, 1024 i + 1024): the occupied
    // octaves are a few steps in which every wave has work, and a wave skips the steps its lanes saw nothing in
    uint32_t* wg_bins = lds_hist + kPatSlots;
    auto bin_of = [&](int v) {
      float a;
      if constexpr (DT == MOQ_BF16) a = __uint_as_float((uint32_t)v << 16);
      else { uint16_t h = (uint16_t)v; a = (float)*reinterpret_cast<_Float16*>(&h); }
      return hist_bin<SHARED>(a, p.bins, p.max_edge, sd, p.skip_zeros);
    };
    // Bins grow with the pattern, so the lanes of a wave (consecutive patterns) that share a bin are neighbours -- and whole
    // waves share ONE where many octaves fall into the lowest bin or outside the range: those add their sum once (64
    // same-address LDS atomics of one instruction serialise)
    auto add_counts = [&](int v, uint32_t cnt) {  // (called by whole waves)
      const unsigned long long live = __builtin_amdgcn_ballot_w64(cnt != 0);
      if (live == 0) return;
      const int b = bin_of(v);
      const int b0 = __builtin_amdgcn_readlane(b, __builtin_ctzll(live));
      if (__builtin_amdgcn_ballot_w64(cnt != 0 && b != b0) == 0) {
        uint32_t tot = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) tot += (uint32_t)__shfl_xor((int)tot, off, 64);
        if (lane == 0) atomicAdd(&wg_bins[b0], tot);
      } else if (cnt) {
        atomicAdd(&wg_bins[b], cnt);
      }
    };
    if (!MOQ_IQ_DBG(0)) {
      constexpr int kSteps = kLutEntries / kIqBlock;
      uint32_t cnt[kSteps];  // all counters of the thread are requested before the first is looked at
#pragma unroll
      for (int i = 0; i < kSteps; ++i) cnt[i] = lds_hist[i * kIqBlock + (int)threadIdx.x];
#pragma unroll
      for (int i = 0; i < kSteps; ++i) add_counts(i * kIqBlock + (int)threadIdx.x, cnt[i]);
      if (threadIdx.x < 64) add_counts(0, lds_hist[kPatZero + threadIdx.x]);
    }
    __syncthreads();
    if (!MOQ_IQ_DBG(1)) {
      // every workgroup starts its flush at another bin: 256 workgroups arriving at the same addresses in the same order
      // queue up behind each other at the memory side
      const int rot = (int)(((int64_t)blockIdx.x * p.bins) / gridDim.x);
      for (int k = threadIdx.x; k < p.bins; k += kIqBlock) {
        const int b = k + rot < p.bins ? k + rot : k + rot - p.bins;
        const uint32_t c = wg_bins[b];
        if (c) atomicAdd(&p.counts[b], (unsigned long long)c);
      }
    }
  } else if constexpr (HIST) {
    for (int b = threadIdx.x; b < p.bins; b += kIqBlock) {
      uint32_t cnt = 0;
      const int copies = b < kHotBins ? 64 : (1 << p.rshift);
      const uint32_t base = slot_base(b, p.rshift);
      for (int r = 0; r < copies; ++r) cnt += lds_hist[base + r];
      if (cnt) atomicAdd(&p.counts[b], (unsigned long long)cnt);
    }
  }
}

}  // namespace moq

using namespace moq;
static inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }

constexpr int kIqMaxLdsBins = 16384;   // bin counts beyond this go to moq_hist_abs' global-atomic kernel
constexpr size_t kIqLdsBudget = 156 * 1024;  // of the CU's 160 KiB (a few words of static LDS come on top)

template <int DT, int FMT, bool PQS>
static void launch_iq(const IqParams& p, bool amax, bool hist, bool shared, bool pat, int blocks, size_t lds, void* stream) {
#define MOQ_IQ_GO(A, H, SH, PT)                                                                                       \
  do {                                                                                                              \
    if (lds > 64 * 1024) {                                                                                          \
      /* > 64 KiB of dynamic LDS needs the opt-in attribute, per device; setting it again is harmless */           \
      lds_opt_in((const void*)input_quant_kernel<DT, FMT, PQS, A, H, SH, PT>, (int)kIqLdsBudget, "input_quant_kernel");  \
    }                                                                                                               \
    hipLaunchKernelGGL((input_quant_kernel<DT, FMT, PQS, A, H, SH, PT>), dim3(blocks), dim3(kIqBlock), lds, S(stream), p); \
  } while (0)
  if constexpr (DT != MOQ_F32) {
    if (hist && pat) {
      if (amax) { if (shared) MOQ_IQ_GO(true, true, true, true); else MOQ_IQ_GO(true, true, false, true); }
      else { if (shared) MOQ_IQ_GO(false, true, true, true); else MOQ_IQ_GO(false, true, false, true); }
      return;
    }
  }
  if (amax && hist) { if (shared) MOQ_IQ_GO(true, true, true, false); else MOQ_IQ_GO(true, true, false, false); }
  else if (hist) { if (shared) MOQ_IQ_GO(false, true, true, false); else MOQ_IQ_GO(false, true, false, false); }
  else if (amax) MOQ_IQ_GO(true, false, false, false);
  else MOQ_IQ_GO(false, false, false, false);
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
  if (hist_counts != nullptr && (hist_bins < 1 || hist_bins >= kIqMaxLdsBins)) {
    set_error("moq_input_quant: histogram stage needs 1 <= bins < %d (use moq_hist_abs beyond)", kIqMaxLdsBins);
    return MOQ_ERR_UNSUPPORTED;
  }
  const int vec = dt == MOQ_F32 ? 4 : 8;
  if (pre_quant_scale != nullptr && (cols % vec != 0 || (reinterpret_cast<uintptr_t>(pre_quant_scale) & 15u) != 0)) {
    set_error("moq_input_quant: pre_quant_scale needs cols %% %d == 0 and a 16-byte aligned scale vector", vec);
    return MOQ_ERR_UNSUPPORTED;
  }
  if (fmt == 0 && amax_running == nullptr && hist_counts == nullptr && (pre_quant_scale == nullptr || y == nullptr)) {
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
  p.dbg = (int)moq_tune("MOQ_TUNE_IQ_DBG", 0);
  size_t lds = 0;
  bool shared = false, pat = false;
  int64_t blocks = ((n + MOQ_MT_CHUNK - 1) / MOQ_MT_CHUNK + 3) / 4;
  if (blocks > 512) blocks = 512;
  if (hist_counts != nullptr) {
    // interleaved copies of the histogram (hot bins spread over banks) as far as LDS allows; the table of a 16-bit
    // input takes 64 KiB, which leaves room for one workgroup per CU
    // 16-bit inputs whose bins fit beside 128 KiB of pattern counters: count patterns, bin them at the flush (PAT)
    const size_t pat_lds = (((size_t)kPatSlots + (size_t)hist_bins + 1 + 3) / 4) * 16;
    pat = dt != MOQ_F32 && pat_lds <= kIqLdsBudget && moq_tune("MOQ_TUNE_HIST_PAT", 1) != 0;
    const size_t table = dt == MOQ_F32 ? 0 : (size_t)kLutEntries * 2;
    const size_t room = (table ? kIqLdsBudget : (size_t)72 * 1024 + 32) - table - (size_t)kHotSlots * 4;
    while (p.rshift < 3 && ((size_t)(hist_bins + 1) << (p.rshift + 1)) * 4 <= room) ++p.rshift;
    lds = pat ? pat_lds : ((size_t)kHotSlots + ((size_t)(hist_bins + 1) << p.rshift)) * 4 + table;
    if (lds > kIqLdsBudget) {
      set_error("moq_input_quant: %d bins do not fit the LDS histogram", hist_bins);
      return MOQ_ERR_UNSUPPORTED;
    }
    if (table && blocks > 256) blocks = 256;
    shared = hist_max_edge >= 0x1p-60f && hist_max_edge <= 0x1p60f && hist_max_edge * (float)hist_bins <= 65536.0f;
  }
  const bool amax = amax_running != nullptr, hist = hist_counts != nullptr;
#define MOQ_IQ_FMT(F)                                                                                         \
  if (pre_quant_scale != nullptr) {                                                                           \
    MOQ_DISPATCH_DTYPE(dt, (launch_iq<DT, F, true>(p, amax, hist, shared, pat, (int)blocks, lds, stream)));        \
  } else {                                                                                                    \
    MOQ_DISPATCH_DTYPE(dt, (launch_iq<DT, F, false>(p, amax, hist, shared, pat, (int)blocks, lds, stream)));       \
  }
  if (fmt == 0) { MOQ_IQ_FMT(0) } else if (fmt == 1) { MOQ_IQ_FMT(1) } else { MOQ_IQ_FMT(2) }
#undef MOQ_IQ_FMT
  return check_launch("moq_input_quant");
}
