// moq_calib.hip -- fused MSE amax-multiplier sweep (a5): MseCalibrator.collect (quantization/calib/mse.py:83-121)
// with the quantizer's fake-quant as quant_func (_mse_quant_func, quantization/model_calib.py:639-662).
//
// The reference loops over the K = ceil((stop - start) / step) + 1 (39 by default) candidate amax values and, per
// candidate, runs a QDQ pass, an elementwise squared error and a reduction -- K x ~5 passes over the tensor.
// Here the tensor is read ONCE: a lane keeps its 16-byte packets in registers and evaluates all K candidates on
// them.  ~12 lane-ops per element per candidate make this the one VALU-bound kernel of the path
// (K * 12 ~ 470 lane-ops per element against ~50 per element at the HBM rate), so the design goal is op count:
// one exact shared-denominator division per (row, candidate), no per-element reciprocal, errors accumulated in
// registers, one DPP / shuffle reduction per candidate at the end of a row segment.
//
// Layout: the tensor is [rows, inner] with one amax per row group: cand[k, row % axis_size].  Rows are what the
// quantizer's calibrator reduces over -- per-channel weights (inner = Cin), static blocks (inner = g, the
// (-1, g) view) or the whole tensor (rows = 1).  x is upcast to fp32 like the reference (mse.py:92), the
// QDQ result is NOT rounded to the storage dtype, the error sum is fp32.
#include "moq_common.h"

namespace moq {

constexpr int kMaxCand = 64;

template <bool FP8>
struct CandQ {
  float scale, inv;  // INT: scale = bound / amax (0 = tiny amax), FP8: s and 1/s
  SharedDiv sd;
};

template <bool FP8>
__device__ __forceinline__ CandQ<FP8> make_cand(float amax, const IntQ& q) {
  CandQ<FP8> c;
  if constexpr (FP8) {
    const Fp8Scale s = fp8_scale(amax);
    c.scale = s.s;
    c.inv = s.inv;
  } else {
    c.scale = int_scale(amax, q.hi);
    c.inv = 0.0f;
    c.sd = make_shared_div(c.scale);
  }
  return c;
}

// squared error of one element under one candidate: the general form (any scale, NaN / zero-sign patches of the QDQ cores
// kept) -- used for INT candidates whose scale is 0 or outside the shared division's exact window
template <bool FP8>
__device__ __forceinline__ float sq_err(float x, const CandQ<FP8>& c, const IntQ& q) {
  float y;
  if constexpr (FP8) {
    const float a = x * c.scale;
    float ca = __builtin_fminf(__builtin_fmaxf(a, -448.0f), 448.0f);
    ca = (a != a) ? a : ca;
    float r0, r1;
    e4m3_roundtrip2(ca, 0.0f, r0, r1);
    y = r0 * c.inv;
  } else {
    y = qdq_int_shared(x, c.scale, c.sd, q);
  }
  const float d = x - y;
  return d * d;
}

// Squared errors of TWO elements under one candidate, the hot form (round 5; the ISA census of round 4's loop showed 13.6
// (INT) / 11.4 (FP8) VALU instructions per element and candidate, 9 / 11 of them unpacked).  What the error does not need
// from the QDQ cores is dropped:
//   * the NaN re-injection (2 ops): a NaN can only come from a NaN x (the caller has checked that the scale is an ordinary
//     number), and then d = x - y is NaN whatever y is;
//   * the sign of a zero quotient (2 ops) and torch.clamp's -0-preserving lower bound (cmp + select + canonicalising
//     min = 4 ops -> one v_med3_f32): x - (+0) == x - (-0);
//   * FP8: the converter packs two elements per v_cvt_pk_fp8_f32 (was one element and a zero: a v_mov + half a convert).
// Everything else is the same arithmetic in the same order -- y is the QDQ value bit for bit -- written on float2 so that
// the multiplies, the Markstein steps of the shared division, the difference, the square and the running sum are
// v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32: 6.3 (INT) / 5.5 (FP8) instruction slots per element and candidate.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
template <bool FP8>
__device__ __forceinline__ f32x2_t sq_err2(f32x2_t x, const CandQ<FP8>& c, const IntQ& q) {
  f32x2_t y;
  if constexpr (FP8) {
    const f32x2_t a = x * c.scale;
    const float c0 = __builtin_amdgcn_fmed3f(a.x, -448.0f, 448.0f), c1 = __builtin_amdgcn_fmed3f(a.y, -448.0f, 448.0f);
    float r0, r1;
    e4m3_roundtrip2(c0, c1, r0, r1);
    const f32x2_t r = {r0, r1};
    y = r * c.inv;
  } else {
    const f32x2_t p = x * c.scale;
    f32x2_t t;
    t.x = __builtin_amdgcn_fmed3f(__builtin_rintf(p.x), q.lo, q.hi);
    t.y = __builtin_amdgcn_fmed3f(__builtin_rintf(p.y), q.lo, q.hi);
    const f32x2_t yy = {c.sd.y, c.sd.y}, nd = {-c.sd.d, -c.sd.d};
    const f32x2_t q0 = t * yy;  // shared_div_in_window, two lanes of it
    const f32x2_t r0 = __builtin_elementwise_fma(nd, q0, t);
    const f32x2_t q1 = __builtin_elementwise_fma(r0, yy, q0);
    const f32x2_t r1 = __builtin_elementwise_fma(nd, q1, t);
    y = __builtin_elementwise_fma(r1, yy, q1);
  }
  const f32x2_t d = x - y;
  return d * d;
}
// may the hot form be used for this candidate?  (wave-uniform per row and candidate)
template <bool FP8>
__device__ __forceinline__ bool cand_hot(const CandQ<FP8>& c) {
  if constexpr (FP8) return c.scale == c.scale && c.inv == c.inv;  // a NaN amax takes the patched form
  else return c.scale != 0.0f && c.sd.fast;
}

// One wave per (row, segment) work item; a segment is kSeg = 4096 consecutive elements of a row, held as 64
// fp32 values per lane across all candidates (zero padding past the row end: QDQ(0) = 0 adds no error).
// cand: [n_cand, axis_size] fp32.  partial: [n_items, n_cand] fp32.
constexpr int kSeg = 4096;
template <int DT, bool FP8>
__global__ __launch_bounds__(kBlock) void mse_rows_kernel(const void* __restrict__ x, int64_t n_rows,
                                                          int64_t axis_size, int64_t inner,
                                                          int64_t segs_per_row,
                                                          const float* __restrict__ cand, int n_cand,
                                                          float* __restrict__ partial, int num_bits,
                                                          int is_unsigned, int narrow) {
  constexpr int V = Elem<DT>::kVec;
  constexpr int P = kSeg / (64 * V);  // packets per lane: 8 (16-bit) or 16 (f32)
  const IntQ q = make_intq(num_bits, is_unsigned, narrow);
  const int lane = threadIdx.x & 63;
  const int64_t wave_id = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const int64_t n_items = n_rows * segs_per_row;
  const int64_t n_waves = (int64_t)gridDim.x * (kBlock / 64);
  const bool vec_ok = (inner % V) == 0 && (reinterpret_cast<uintptr_t>(x) & 15u) == 0;
  for (int64_t item = wave_id; item < n_items; item += n_waves) {
    const int64_t row = item / segs_per_row, seg = item % segs_per_row;
    const int64_t e0 = seg * kSeg;
    float f[P * V];
#pragma unroll
    for (int u = 0; u < P; ++u) {
      const int64_t e = e0 + ((int64_t)u * 64 + lane) * V;
      if (vec_ok && e + V <= inner) {
        unpack<DT>(load16_nt(reinterpret_cast<const char*>(x) + (row * inner + e) * (16 / V)), &f[u * V]);
      } else {
#pragma unroll
        for (int i = 0; i < V; ++i) f[u * V + i] = e + i < inner ? load1<DT>(x, row * inner + e + i) : 0.0f;
      }
    }
    const float* crow = cand + (row % axis_size);
    for (int k = 0; k < n_cand; ++k) {
      const CandQ<FP8> c = make_cand<FP8>(crow[(int64_t)k * axis_size], q);
      float a0 = 0.0f, a1 = 0.0f;  // two chains (even / odd elements): one packed accumulator in the hot form
      if (cand_hot<FP8>(c)) {
        f32x2_t a2 = {0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < P * V; i += 2) {
          const f32x2_t xv = {f[i], f[i + 1]};
          a2 += sq_err2<FP8>(xv, c, q);
        }
        a0 = a2.x;
        a1 = a2.y;
      } else {
#pragma unroll
        for (int i = 0; i < P * V; i += 2) {
          a0 += sq_err<FP8>(f[i], c, q);
          a1 += sq_err<FP8>(f[i + 1], c, q);
        }
      }
      float acc = a0 + a1;
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
      if (lane == 0) partial[item * n_cand + k] = acc;
    }
  }
}

// Short power-of-two rows (static blocks, inner = LPG * V <= 64 * V; LPG = 16-byte packets per row).  A candidate costs
// ~25 VALU slots of setup per lane (the IEEE division of int_scale, the refined reciprocal) before any element is touched:
// with one packet per lane (round 4) that was half of the hot loop's 8 x 6.5 slots of real work -- the g = 128 sweep ran at
// 0.45 of the vector issue rate where the long-row kernel reaches 0.7-0.8.  A lane therefore owns PPL = min(4, LPG) packets
// of its row (packets sub, sub + LN, ... : the LN = LPG / PPL lanes of a row still read one contiguous run), the setup is
// paid once per 8 * PPL elements, the per-candidate sum is an LN-wide butterfly, and lane 0 of every row stores -- 16 rows
// of a wave write 64 adjacent bytes per candidate instead of 4 x 4.
template <int DT, int LPG, bool FP8>
__global__ __launch_bounds__(kBlock) void mse_group_kernel(const void* __restrict__ x, int64_t n_rows,
                                                           int64_t axis_size,
                                                           const float* __restrict__ cand, int n_cand,
                                                           float* __restrict__ loss, int accumulate,
                                                           int num_bits, int is_unsigned, int narrow) {
  constexpr int V = Elem<DT>::kVec;
  constexpr int PPL = LPG < 4 ? LPG : 4, LN = LPG / PPL;
  const IntQ q = make_intq(num_bits, is_unsigned, narrow);
  const int64_t n_items = n_rows * LN;
  for (int64_t it = (int64_t)blockIdx.x * kBlock + threadIdx.x; it < ((n_items + 63) & ~(int64_t)63);
       it += (int64_t)gridDim.x * kBlock) {
    const bool live = it < n_items;
    const int64_t row = (live ? it : n_items - 1) / LN;
    const int sub = (int)(it - (it / LN) * LN);
    float f[PPL * V];
    if (live) {
      Pack16 pk[PPL];
#pragma unroll
      for (int j = 0; j < PPL; ++j)
        pk[j] = load16_nt(reinterpret_cast<const char*>(x) + (row * LPG + j * LN + sub) * 16);
#pragma unroll
      for (int j = 0; j < PPL; ++j) {
        float t[8];
        unpack<DT>(pk[j], t);
#pragma unroll
        for (int i = 0; i < V; ++i) f[j * V + i] = t[i];
      }
    } else {
#pragma unroll
      for (int i = 0; i < PPL * V; ++i) f[i] = 0.0f;
    }
    const int64_t a = row % axis_size;
    for (int k = 0; k < n_cand; ++k) {
      const CandQ<FP8> c = make_cand<FP8>(cand[(int64_t)k * axis_size + a], q);
      float acc = 0.0f;
      if (cand_hot<FP8>(c)) {
        f32x2_t a2 = {0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < PPL * V; i += 2) {
          const f32x2_t xv = {f[i], f[i + 1]};
          a2 += sq_err2<FP8>(xv, c, q);
        }
        acc = a2.x + a2.y;
      } else {
#pragma unroll
        for (int i = 0; i < PPL * V; ++i) acc += sq_err<FP8>(f[i], c, q);
      }
#pragma unroll
      for (int off = LN / 2; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
      if (live && sub == 0) {
        float* dst = loss + (int64_t)k * axis_size + a;
        // outer == 1 for block layouts, so every (k, a) has exactly one writer: plain read-modify-write
        *dst = accumulate ? *dst + acc : acc;
      }
    }
  }
}

// loss[k, a] (+)= sum over the work items of rows with row % axis_size == a, in item order (deterministic)
__global__ void mse_finalize_kernel(const float* __restrict__ partial, int64_t n_rows, int64_t axis_size,
                                    int64_t segs_per_row, int n_cand, float* __restrict__ loss,
                                    int accumulate) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)n_cand * axis_size) return;
  const int k = (int)(idx / axis_size);
  const int64_t a = idx % axis_size;
  float s = 0.0f;
  for (int64_t row = a; row < n_rows; row += axis_size)
    for (int64_t seg = 0; seg < segs_per_row; ++seg) s += partial[(row * segs_per_row + seg) * n_cand + k];
  loss[idx] = accumulate ? loss[idx] + s : s;
}
// The same for outputs with MANY work items each (per-tensor: one output per candidate, 57 K segments of a 470 MB weight):
// one workgroup per output -- thread t adds items t, t + 256, ... in order, the 256 partial sums are folded in a fixed
// tree.  Deterministic like the serial form, which spent 13 of the sweep's 17.6 ms on 39 threads walking 57 K items each.
__global__ __launch_bounds__(256) void mse_finalize_wide_kernel(const float* __restrict__ partial, int64_t n_rows,
                                                                int64_t axis_size, int64_t segs_per_row, int n_cand,
                                                                float* __restrict__ loss, int accumulate) {
  __shared__ float sm[256];
  const int64_t idx = blockIdx.x;  // (k, a)
  const int k = (int)(idx / axis_size);
  const int64_t a = idx % axis_size;
  const int64_t rows_a = (n_rows - a + axis_size - 1) / axis_size, n_items = rows_a * segs_per_row;
  float s = 0.0f;
  for (int64_t it = threadIdx.x; it < n_items; it += 256) {
    const int64_t row = a + (it / segs_per_row) * axis_size, seg = it % segs_per_row;
    s += partial[(row * segs_per_row + seg) * n_cand + k];
  }
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) sm[threadIdx.x] += sm[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[idx] = accumulate ? loss[idx] + sm[0] : sm[0];
}


// ---------------------------------------------------------------------------------------------------------------
// calibrate_weights (quantization/calib/histogram.py:346-433): one |w| histogram PER OUTPUT CHANNEL with numpy's
// np.histogram(a, bins, range=(0, a.max())) semantics.  numpy builds float32 edges e_k = fp32(fp32(k * step) + first),
// step = (last - first) / bins, e_bins = last, estimates the bin as trunc(((v - first) / (last - first)) * bins) and
// then corrects the estimate against the edges (decrement if v < e_idx, then increment if v >= e_{idx+1}); the last
// bin is closed.  The same steps in the same fp32 arithmetic here; grid = (column splits, rows), LDS histogram per
// workgroup, int32 global atomics on flush.  first / last per row come from the host (row abs-max, or -0.5 / +0.5
// for an all-zero row exactly like numpy's _get_outer_edges).
template <int DT>
__global__ __launch_bounds__(kBlock) void row_hist_np_kernel(const void* __restrict__ x, int64_t cols, int bins,
                                                             const float* __restrict__ first,
                                                             const float* __restrict__ last,
                                                             int* __restrict__ counts) {
  extern __shared__ int lds_hist[];
  const int64_t row = blockIdx.y;
  for (int b = threadIdx.x; b <= bins; b += kBlock) lds_hist[b] = 0;  // (+ the trash slot)
  __syncthreads();
  const float lo = first[row], hi = last[row];
  const float width = hi - lo;
  const float fbins = (float)bins;
  const float step = width / fbins;
  const int64_t per = (cols + gridDim.x - 1) / gridDim.x;
  const int64_t c0 = blockIdx.x * per, c1 = c0 + per < cols ? c0 + per : cols;
  const int64_t base = row * cols;
  // numpy ESTIMATES the bin -- trunc(((v - first) / (last - first)) * bins) -- and then CORRECTS the estimate against the
  // edges: one step down if v < e[idx], else one step up if v >= e[idx + 1] (last bin closed).  Whatever estimate lies
  // within one bin of the edge-defined bin therefore ends in the same place, and numpy's own always does (its error is
  // ~bins x 1e-7 of a bin).  So the estimate needs no IEEE division per element (10 of round 4's ~35 vector instructions
  // per element in a kernel that ran at the vector issue rate): one multiply by bins / (last - first); the edges keep
  // numpy's own two roundings (fp32(k * step) + first; -ffp-contract=off).  Elements outside [first, last] (NaN: numpy
  // raises on non-finite ranges) go to a trash slot instead of around a branch.  Two elements per instruction where the
  // ISA has a packed form.
  const float scale = fbins / width;  // (width > 0: the host maps an all-zero row to [-0.5, 0.5])
  const f32x2 lo2 = {lo, lo}, sc2 = {scale, scale}, st2 = {step, step};
  const int last_bin = bins - 1;
  auto count2 = [&](float xa, float xb) {
    const f32x2 v = {__builtin_fabsf(xa), __builtin_fabsf(xb)};
    const f32x2 q = (v - lo2) * sc2;
    int ia = (int)q.x, ib = (int)q.y;
    ia = ia < last_bin ? ia : last_bin;
    ib = ib < last_bin ? ib : last_bin;
    const f32x2 k0 = {(float)ia, (float)ib}, k1 = {(float)(ia + 1), (float)(ib + 1)};
    const f32x2 e0 = k0 * st2 + lo2;
    f32x2 e1 = k1 * st2 + lo2;
    e1.x = ia == last_bin ? hi : e1.x;  // e[bins] = last, exactly
    e1.y = ib == last_bin ? hi : e1.y;
    // (bitwise, not short-circuit: no exec-mask branches)
    ia += (int)((v.x >= e1.x) & (ia != last_bin)) - (int)(v.x < e0.x);
    ib += (int)((v.y >= e1.y) & (ib != last_bin)) - (int)(v.y < e0.y);
    ia = ((v.x >= lo) & (v.x <= hi)) ? ia : bins;
    ib = ((v.y >= lo) & (v.y <= hi)) ? ib : bins;
    atomicAdd(&lds_hist[ia], 1);
    atomicAdd(&lds_hist[ib], 1);
  };
  auto count = [&](float xv) { count2(xv, __builtin_nanf("")); };  // (tail elements: the partner lands in the trash slot)
  constexpr int V = Elem<DT>::kVec;
  const char* xb = reinterpret_cast<const char*>(x);
  if (((base + c0) % V) == 0 && (reinterpret_cast<uintptr_t>(x) & 15u) == 0) {
    // 16-byte packets per lane (the scalar form moved 2 bytes per lane and load), the row's tail element by element
    const int64_t n_pk = (c1 - c0) / V;
    for (int64_t p = threadIdx.x; p < n_pk; p += kBlock) {
      const Pack16 pk = load16_nt(xb + (base + c0 + p * V) * (16 / V));
      float f[8];
      unpack<DT>(pk, f);
#pragma unroll
      for (int e = 0; e < V; e += 2) count2(f[e], f[e + 1]);
    }
    for (int64_t c = c0 + n_pk * V + threadIdx.x; c < c1; c += kBlock) count(load1<DT>(x, base + c));
  } else {
    for (int64_t c = c0 + threadIdx.x; c < c1; c += kBlock) count(load1<DT>(x, base + c));
  }
  __syncthreads();
  if (gridDim.x == 1) {
    // the row is this workgroup's alone: plain coalesced stores of every bin instead of one global atomic per
    // non-empty bin (58 M atomics for a 28672 x 8192 weight)
    for (int b = threadIdx.x; b < bins; b += kBlock) counts[row * bins + b] = lds_hist[b];
  } else {
    for (int b = threadIdx.x; b < bins; b += kBlock) {
      const int n = lds_hist[b];
      if (n) atomicAdd(&counts[row * bins + b], n);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Histogram threshold searches on the device (calib/histogram.py:210-283 entropy, :326-343 percentile).
// ------------------------------------------------------------------------------------------------
// Entropy: candidate i (one workgroup) folds source bins [0, i) of the histogram into `nq` buckets, spreads every bucket's
// count evenly over its NON-EMPTY source bins (new_density) and takes KL(reference_density || new_density) where the
// reference density is bins[:i] with the clipped tail added to its last bin -- the reference's loop body, candidate by
// candidate, in fp64.  bins[0] is replaced by bins[1] as there.  The divergences go back to the caller, who takes the LAST
// minimum; sums are formed in this kernel's own order (integer bucket sums exact, the two fp64 reductions as a tree), so
// a divergence can differ from numpy's in the last bits: the host re-scores candidates that tie with the minimum to 1e-9
// with the reference's own arithmetic (calib.py) -- the returned amax is the reference's bit for bit.
constexpr int kEntMaxBuckets = 4096;
__global__ __launch_bounds__(256) void hist_entropy_kernel(const int64_t* __restrict__ hist, int n_bins, int nq,
                                                           int start_bin, int stride, double* __restrict__ div_out) {
  __shared__ unsigned long long sums[kEntMaxBuckets];
  __shared__ int members[kEntMaxBuckets];
  __shared__ double red[256];
  __shared__ long long red_i[256];
  const int i = start_bin + (int)blockIdx.x * stride;  // candidate: clip after source bin i - 1
  const int tid = threadIdx.x;
  auto bin = [&](int j) -> long long { return hist[(j == 0 && n_bins > 1) ? 1 : j]; };
  for (int b = tid; b < nq; b += 256) {
    sums[b] = 0ull;
    members[b] = 0;
  }
  // tail = sum(bins[i:]), total = sum(bins): exact integers
  long long tail = 0, head = 0;
  for (int j = tid; j < n_bins; j += 256) {
    const long long v = bin(j);
    if (j >= i) tail += v; else head += v;
  }
  red_i[tid] = tail;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if (tid < off) red_i[tid] += red_i[tid + off];
    __syncthreads();
  }
  tail = red_i[0];
  __syncthreads();
  red_i[tid] = head;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if (tid < off) red_i[tid] += red_i[tid + off];
    __syncthreads();
  }
  const long long total = red_i[0] + tail;
  __syncthreads();
  // bucket of source bin j: floor(j * nq / i)  (np.digitize against linspace(0, i, nq + 1), exact for nq a power of two)
  for (int j = tid; j < i; j += 256) {
    const long long v = bin(j);
    if (v != 0) {
      const int b = (int)(((long long)j * nq) / i);
      atomicAdd(&sums[b], (unsigned long long)v);
      atomicAdd(&members[b], 1);
    }
  }
  __syncthreads();
  // new_sum = sum of new_density over the candidate's bins (the normaliser scipy.stats.entropy applies to qk)
  double part = 0.0;
  for (int j = tid; j < i; j += 256) {
    if (bin(j) != 0) {
      const int b = (int)(((long long)j * nq) / i);
      part += (double)sums[b] / (double)members[b];
    }
  }
  red[tid] = part;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if (tid < off) red[tid] += red[tid + off];
    __syncthreads();
  }
  const double new_sum = red[0];
  __syncthreads();
  const double old_sum = (double)total;  // sum(reference_density): bins[:i] plus the tail = every count
  part = 0.0;
  for (int j = tid; j < i; j += 256) {
    const long long v = bin(j);
    const double pj = (double)(v + (j == i - 1 ? tail : 0)) / old_sum;
    double qj = 0.0;
    if (v != 0) {
      const int b = (int)(((long long)j * nq) / i);
      qj = ((double)sums[b] / (double)members[b]) / new_sum;
    }
    // scipy.special.rel_entr: x log(x / y) for x, y > 0; 0 for x == 0, y >= 0; inf otherwise
    if (pj > 0.0 && qj > 0.0) part += pj * log(pj / qj);
    else if (!(pj == 0.0 && qj >= 0.0)) part += __builtin_inf();
  }
  red[tid] = part;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if (tid < off) red[tid] += red[tid + off];
    __syncthreads();
  }
  if (tid == 0) div_out[blockIdx.x] = red[0];
}

// Percentile: idx[r] = np.searchsorted(np.cumsum(hist[r] / hist[r].sum()), q) -- the FIRST bin whose running fraction
// reaches q, the running sum formed sequentially in fp64 exactly like np.cumsum (one thread walks a row; 64 rows per
// workgroup, the counts staged through the LDS in 64 x 64 tiles so that the global reads stay coalesced).  idx = bins
// when the sum never reaches q (the caller's edges have bins + 1 entries).
template <typename CT>
__global__ __launch_bounds__(64) void hist_percentile_kernel(const CT* __restrict__ hist, int64_t rows, int bins, double q,
                                                             int64_t* __restrict__ idx_out) {
  __shared__ long long tile[64][65];
  const int64_t r0 = (int64_t)blockIdx.x * 64;
  const int t = threadIdx.x;
  // pass 1: the row totals (exact integers)
  long long total = 0;
  for (int c0 = 0; c0 < bins; c0 += 64) {
    for (int rr = 0; rr < 64; ++rr) {
      const int64_t r = r0 + rr;
      tile[rr][t] = (r < rows && c0 + t < bins) ? (long long)hist[r * bins + c0 + t] : 0;
    }
    __syncthreads();
#pragma unroll 8
    for (int c = 0; c < 64; ++c) total += tile[t][c];
    __syncthreads();
  }
  // pass 2: sequential fp64 running sum of hist / total
  const double tot = (double)total;
  double acc = 0.0;
  int found = -1;
  for (int c0 = 0; c0 < bins; c0 += 64) {
    for (int rr = 0; rr < 64; ++rr) {
      const int64_t r = r0 + rr;
      tile[rr][t] = (r < rows && c0 + t < bins) ? (long long)hist[r * bins + c0 + t] : 0;
    }
    __syncthreads();
    if (found < 0) {
      const int lim = bins - c0 < 64 ? bins - c0 : 64;
      for (int c = 0; c < lim; ++c) {
        acc += (double)tile[t][c] / tot;
        if (acc >= q) {
          found = c0 + c;
          break;
        }
      }
    }
    __syncthreads();
  }
  if (total == 0) found = 0;  // an empty row: numpy's cdf is all NaN and searchsorted answers 0
  if (r0 + t < rows) idx_out[r0 + t] = found < 0 ? bins : found;
}

}  // namespace moq

using namespace moq;
static inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }

extern "C" int moq_hist_entropy(const int64_t* hist, int64_t n_bins, int num_quant_bins, int start_bin, int stride,
                                double* divergences, void* stream) {
  if (hist == nullptr || divergences == nullptr || n_bins < 1 || stride < 1 || start_bin < 1) {
    set_error("moq_hist_entropy: bad arguments");
    return MOQ_ERR_INVALID;
  }
  if (num_quant_bins < 1 || num_quant_bins > kEntMaxBuckets || (num_quant_bins & (num_quant_bins - 1)) != 0 ||
      n_bins > (1 << 24)) {
    set_error("moq_hist_entropy: needs a power-of-two bucket count <= %d and at most 2^24 bins", kEntMaxBuckets);
    return MOQ_ERR_UNSUPPORTED;
  }
  if (start_bin > n_bins) return MOQ_OK;  // no candidate (range(start_bin, n_bins + 1, stride) is empty)
  const int64_t n_cand = (n_bins - start_bin) / stride + 1;
  hipLaunchKernelGGL(hist_entropy_kernel, dim3((unsigned)n_cand), dim3(256), 0, S(stream), hist, (int)n_bins,
                     num_quant_bins, start_bin, stride, divergences);
  return check_launch("moq_hist_entropy");
}

extern "C" int moq_hist_percentile(const void* hist, int elem_bytes, int64_t rows, int64_t bins, double q, int64_t* idx,
                                   void* stream) {
  if (rows < 0 || bins < 1 || (rows > 0 && (hist == nullptr || idx == nullptr)) || !(q >= 0.0 && q <= 1.0)) {
    set_error("moq_hist_percentile: bad arguments");
    return MOQ_ERR_INVALID;
  }
  if ((elem_bytes != 4 && elem_bytes != 8) || bins > (1 << 24) || rows > ((int64_t)1 << 36)) {
    set_error("moq_hist_percentile: counts must be int32 or int64, bins <= 2^24");
    return MOQ_ERR_UNSUPPORTED;
  }
  if (rows == 0) return MOQ_OK;
  const unsigned grid = (unsigned)((rows + 63) / 64);
  if (elem_bytes == 4)
    hipLaunchKernelGGL((hist_percentile_kernel<int>), dim3(grid), dim3(64), 0, S(stream),
                       reinterpret_cast<const int*>(hist), rows, (int)bins, q, idx);
  else
    hipLaunchKernelGGL((hist_percentile_kernel<long long>), dim3(grid), dim3(64), 0, S(stream),
                       reinterpret_cast<const long long*>(hist), rows, (int)bins, q, idx);
  return check_launch("moq_hist_percentile");
}

static void mse_plan(int64_t n_rows, int64_t inner, int64_t* seg_elems, int64_t* segs) {
  (void)n_rows;
  *seg_elems = kSeg;
  *segs = (inner + kSeg - 1) / kSeg;
}

extern "C" int64_t moq_mse_sweep_workspace(int64_t outer, int64_t axis_size, int64_t inner, int n_cand) {
  if (outer < 0 || axis_size <= 0 || inner <= 0 || n_cand <= 0) return MOQ_ERR_INVALID;
  int64_t se, segs;
  mse_plan(outer * axis_size, inner, &se, &segs);
  return outer * axis_size * segs * n_cand;
}

extern "C" int moq_mse_sweep(const void* x, int64_t outer, int64_t axis_size, int64_t inner, int dt,
                             const float* cand_amax, int n_cand, float* loss, float* partial, int accumulate,
                             int fp8, int num_bits, int is_unsigned, int narrow_range, void* stream) {
  if (x == nullptr || cand_amax == nullptr || loss == nullptr || outer <= 0 || axis_size <= 0 || inner <= 0) {
    set_error("moq_mse_sweep: null pointer or bad sizes");
    return MOQ_ERR_INVALID;
  }
  if (n_cand < 1 || n_cand > kMaxCand) {
    set_error("moq_mse_sweep: n_cand must be in [1, %d]", kMaxCand);
    return MOQ_ERR_UNSUPPORTED;
  }
  if (!fp8 && (num_bits < 2 || num_bits > 16)) {
    set_error("moq_mse_sweep: num_bits=%d out of range [2,16]", num_bits);
    return MOQ_ERR_INVALID;
  }
  const int vec = dt == MOQ_F32 ? 4 : 8;
  const int64_t n_rows = outer * axis_size;
  const bool aligned = (reinterpret_cast<uintptr_t>(x) & 15u) == 0;
  const int64_t lpg = inner % vec == 0 ? inner / vec : 0;
  if (outer == 1 && aligned && lpg >= 1 && lpg <= 64 && (lpg & (lpg - 1)) == 0) {
    const int grid = stream_grid(kBlock, n_rows * (lpg < 4 ? 1 : lpg / 4));  // work items: rows x lanes per row
#define MOQ_MSE_G(L)                                                                                            \
  case L:                                                                                                       \
    if (fp8) { MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((mse_group_kernel<DT, L, true>), dim3(grid), dim3(kBlock), 0, S(stream), x, n_rows, axis_size, cand_amax, n_cand, loss, accumulate, num_bits, is_unsigned, narrow_range)); } \
    else { MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((mse_group_kernel<DT, L, false>), dim3(grid), dim3(kBlock), 0, S(stream), x, n_rows, axis_size, cand_amax, n_cand, loss, accumulate, num_bits, is_unsigned, narrow_range)); } \
    break;
    switch ((int)lpg) {
      MOQ_MSE_G(1) MOQ_MSE_G(2) MOQ_MSE_G(4) MOQ_MSE_G(8) MOQ_MSE_G(16) MOQ_MSE_G(32) MOQ_MSE_G(64)
      default: set_error("unreachable"); return MOQ_ERR_INVALID;
    }
#undef MOQ_MSE_G
    return check_launch("moq_mse_sweep(group)");
  }
  if (partial == nullptr) {
    set_error("moq_mse_sweep: workspace `partial` is required for this layout");
    return MOQ_ERR_INVALID;
  }
  int64_t seg_elems, segs;
  mse_plan(n_rows, inner, &seg_elems, &segs);
  const int64_t items = n_rows * segs;
  int64_t blocks = (items + 3) / 4;
  if (blocks > 4096) blocks = 4096;
  if (fp8) {
    MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((mse_rows_kernel<DT, true>), dim3((unsigned)blocks), dim3(kBlock), 0,
                                              S(stream), x, n_rows, axis_size, inner, segs, cand_amax,
                                              n_cand, partial, num_bits, is_unsigned, narrow_range));
  } else {
    MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((mse_rows_kernel<DT, false>), dim3((unsigned)blocks), dim3(kBlock), 0,
                                              S(stream), x, n_rows, axis_size, inner, segs, cand_amax,
                                              n_cand, partial, num_bits, is_unsigned, narrow_range));
  }
  const int64_t n_out = (int64_t)n_cand * axis_size;
  const int64_t items_per_out = ((n_rows + axis_size - 1) / axis_size) * segs;
  if (items_per_out >= 1024 && n_out <= 65535) {
    hipLaunchKernelGGL(mse_finalize_wide_kernel, dim3((unsigned)n_out), dim3(256), 0, S(stream), partial, n_rows,
                       axis_size, segs, n_cand, loss, accumulate);
  } else {
    hipLaunchKernelGGL(mse_finalize_kernel, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, S(stream), partial,
                       n_rows, axis_size, segs, n_cand, loss, accumulate);
  }
  return check_launch("moq_mse_sweep");
}

extern "C" int moq_row_hist_np(const void* x, int64_t rows, int64_t cols, int dt, int bins, const float* first,
                               const float* last, int* counts, void* stream) {
  if (rows < 0 || cols < 0 || bins <= 0 || (rows * cols > 0 && (x == nullptr || first == nullptr || last == nullptr ||
                                                               counts == nullptr))) {
    set_error("moq_row_hist_np: bad arguments");
    return MOQ_ERR_INVALID;
  }
  if (bins > 16384 || rows > 65535 || cols >= ((int64_t)1 << 31)) {
    set_error("moq_row_hist_np: needs bins <= 16384, rows <= 65535, cols < 2^31");
    return MOQ_ERR_UNSUPPORTED;
  }
  if (rows * cols == 0) return MOQ_OK;
  // enough workgroups to fill the chip, but at least ~8 elements per thread per workgroup
  int64_t splits = (2048 + rows - 1) / rows;
  const int64_t max_splits = (cols + 8 * kBlock - 1) / (8 * kBlock);
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  // one workgroup per row writes every bin of its row with plain stores; only rows split over several workgroups (few,
  // long rows) accumulate with atomics and need their counts cleared first -- the caller hands over UNINITIALISED memory
  // (a 28672-row weight has 235 MB of counts: zero-filling them cost as much as the histogram)
  if (splits > 1 && hipMemsetAsync(counts, 0, (size_t)rows * (size_t)bins * sizeof(int), S(stream)) != hipSuccess) {
    set_error("moq_row_hist_np: clearing the counts failed");
    return MOQ_ERR_LAUNCH;
  }
  MOQ_DISPATCH_DTYPE(dt, hipLaunchKernelGGL((row_hist_np_kernel<DT>), dim3((unsigned)splits, (unsigned)rows), dim3(kBlock),
                                            (size_t)(bins + 1) * 4, reinterpret_cast<hipStream_t>(stream), x, cols, bins,
                                            first, last, counts));
  return check_launch("moq_row_hist_np");
}
