// moq_gemm.hip -- the one dense contraction of the PTQ hot path: the AWQ-lite search error GEMM (a12).
//
// For every candidate alpha the reference runs the patched linear forward
//     out = F.linear(x * (1/s), QDQ(W * s), bias);  loss[alpha] += (out - out_actual).float().pow(2).mean()
// (quantization/model_calib.py:1489-1495, :1552-1556).  Here the contraction and the loss are ONE kernel:
// MFMA tiles of  out^T[n, t] = sum_k What[n, k] * xs[t, k]  stay in registers, are rounded to the model dtype
// exactly where the reference materialises `out`, subtracted from out_actual in the model dtype, squared in
// fp32 and reduced -- `out` (T x Cout) is never written to HBM.
//
// Tiling (gfx950, wave64): 128(n) x 128(t) x 64(k) per 256-thread workgroup, 2 x 2 waves, each wave owns
// 64 x 64 as 2 x 2 v_mfma_f32_32x32x16 tiles (64 accumulator VGPRs).  Both operands are K-contiguous
// ([Cout, Cin] weights, [tokens, Cin] activations), so an MFMA fragment is one 16-byte run of k per lane.
// HBM -> LDS goes through `buffer_load_dwordx4 ... lds` (no staging VGPRs, no ds_write pass); the buffer
// descriptor's bounds check returns zeros for rows past the matrix edge and for the K tail, so ragged
// shapes need no second kernel.  LDS rows are 128 B (64 k); the 16-byte chunk c of row r is stored at chunk
// position c ^ ((r >> 1) & 7): a ds_read_b128 fragment read (32 rows x one chunk column) then touches 16
// distinct 16-byte slots per 16-lane service group = conflict-free (MI355X_MICROARCH.md, LDS table).  The
// swizzle is applied on the *source* address because the LDS side of the DMA is lane-linear.
// Two LDS buffers (64 KiB per workgroup -> 2 workgroups per CU); tile k+1 streams in while tile k is in the
// matrix cores; one barrier per K-step.
//
// Roofline: MFMA-bound.  2 * T * Cout * Cin flop per launch against ~2.5 PFLOP/s dense bf16.
#include <stdlib.h>

#include "moq_common.h"

namespace moq {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

constexpr int kTile = 128;                          // rows per operand tile
constexpr int kBK = 64;                             // k per stage
constexpr int kRowBytes = kBK * 2;                  // 128 B
constexpr int kTileBytes = kTile * kRowBytes;       // 16 KiB per operand per stage
constexpr int kStageBytes = 2 * kTileBytes;         // A + B
constexpr int kGemmLds = 2 * kStageBytes;           // double buffered: 64 KiB

template <int DT>
__device__ __forceinline__ f32x16_t mfma32(const Pack16& a, const Pack16& b, f32x16_t c) {
  if constexpr (DT == MOQ_BF16) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(&a),
                                                   *reinterpret_cast<const bf16x8_t*>(&b), c, 0, 0, 0);
  } else {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8_t*>(&a),
                                                  *reinterpret_cast<const f16x8_t*>(&b), c, 0, 0, 0);
  }
}

// One operand tile (128 rows x 64 k) HBM -> LDS.  `rsrc` covers the tile's valid rows only (base = first
// row of the tile, num_records = valid_rows * ld * 2), so out-of-range rows read as zero.  Each wave issues
// 4 instructions of 8 rows x 128 B.
__device__ __forceinline__ void stage_tile(__amdgpu_buffer_rsrc_t rsrc, uint8_t* lds_tile, int64_t ld_bytes,
                                           int k0, int K, int wave, int lane) {
  const int r_local = lane >> 3, pos = lane & 7;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int rbase = (wave * 4 + j) * 8;
    const int r = rbase + r_local;
    const int c = pos ^ ((r >> 1) & 7);
    const int k = k0 + c * 8;
    // K tail (K % 8 == 0 guaranteed): chunks at or past K must read as zero -> force an out-of-range offset
    const int voff = k < K ? (int)(r * ld_bytes + k * 2) : 0x7FFFFFF0;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(lds_tile + rbase * kRowBytes), 16, voff, 0, 0, 0);
  }
}

// fragment read: row r of the tile, 16-byte chunk c (k = c*8 .. c*8+7)
__device__ __forceinline__ Pack16 read_frag(const uint8_t* lds_tile, int r, int c) {
  return *reinterpret_cast<const Pack16*>(lds_tile + r * kRowBytes + ((c ^ ((r >> 1) & 7)) << 4));
}

// MODE 0: accumulate the squared error against `ref` into partial[block]; MODE 1: store out[t, n].
template <int DT, int MODE, bool DBUF>
__global__ __launch_bounds__(256, DBUF ? 2 : 4) void err_gemm_kernel(const void* __restrict__ x,    // [T, K]
                                                          const void* __restrict__ w,    // [N, K]
                                                          const void* __restrict__ ref,  // [T, N] (MODE 0)
                                                          const void* __restrict__ bias, // [N] or null
                                                          void* __restrict__ out,        // [T, N] (MODE 1)
                                                          float* __restrict__ partial, int T, int N, int K,
                                                          int tiles_t, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // XCD-aware tile order: workgroup b runs on XCD b % 8; give each XCD a contiguous run of tiles so that
  // the W tile it is streaming is shared through that XCD's L2 (bijective for any grid size).
  const int nblk = tiles_t * tiles_n;
  int bid = blockIdx.x;
  {
    const int xcd = bid & 7, idx = bid >> 3, q = nblk >> 3, r = nblk & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tn = bid / tiles_t, tt = bid % tiles_t;  // consecutive workgroups share the W tile
  const int n0 = tn * kTile, t0 = tt * kTile;
  const int rows_w = N - n0 < kTile ? N - n0 : kTile;
  const int rows_x = T - t0 < kTile ? T - t0 : kTile;
  const int64_t ld_bytes = (int64_t)K * 2;
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint8_t*>(reinterpret_cast<const uint8_t*>(w)) + (int64_t)n0 * ld_bytes, 0,
      (int)(rows_w * ld_bytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint8_t*>(reinterpret_cast<const uint8_t*>(x)) + (int64_t)t0 * ld_bytes, 0,
      (int)(rows_x * ld_bytes), 0x00020000);

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  const int wn = wave >> 1, wt = wave & 1;  // wave's 64 x 64 quadrant: n rows wn*64.., t cols wt*64..
  const int fr = lane & 31, fh = lane >> 5;
  const int nk = (K + kBK - 1) / kBK;

  if constexpr (DBUF) {
    // two LDS stages: all 16 fragments of tile kt go to registers first, then the DMA of tile kt+1 is issued
    // and runs under the 16 MFMAs (the compiler orders an LDS-DMA before any later ds_read of the same
    // array with vmcnt(0), so reads must precede the issue for the overlap to exist)
    stage_tile(rs_w, smem, ld_bytes, 0, K, wave, lane);
    stage_tile(rs_x, smem + kTileBytes, ld_bytes, 0, K, wave, lane);
    for (int kt = 0; kt < nk; ++kt) {
      __builtin_amdgcn_s_waitcnt(0);
      __syncthreads();  // tile kt landed for everyone; everyone finished reading the other stage
      const uint8_t* cur = smem + (kt & 1) * kStageBytes;
      const uint8_t* la = cur + (wn * 64) * kRowBytes;
      const uint8_t* lb = cur + kTileBytes + (wt * 64) * kRowBytes;
      Pack16 a0[4], a1[4], b0[4], b1[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int c = ks * 2 + fh;
        a0[ks] = read_frag(la, fr, c); a1[ks] = read_frag(la, 32 + fr, c);
        b0[ks] = read_frag(lb, fr, c); b1[ks] = read_frag(lb, 32 + fr, c);
      }
      if (kt + 1 < nk) {
        uint8_t* nxt = smem + ((kt + 1) & 1) * kStageBytes;
        stage_tile(rs_w, nxt, ld_bytes, (kt + 1) * kBK, K, wave, lane);
        stage_tile(rs_x, nxt + kTileBytes, ld_bytes, (kt + 1) * kBK, K, wave, lane);
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        acc[0][0] = mfma32<DT>(a0[ks], b0[ks], acc[0][0]);
        acc[0][1] = mfma32<DT>(a0[ks], b1[ks], acc[0][1]);
        acc[1][0] = mfma32<DT>(a1[ks], b0[ks], acc[1][0]);
        acc[1][1] = mfma32<DT>(a1[ks], b1[ks], acc[1][1]);
      }
    }
  } else {
    // one LDS stage (32 KiB): up to 4 workgroups per CU overlap each other's DMA waits
    for (int kt = 0; kt < nk; ++kt) {
      stage_tile(rs_w, smem, ld_bytes, kt * kBK, K, wave, lane);
      stage_tile(rs_x, smem + kTileBytes, ld_bytes, kt * kBK, K, wave, lane);
      __builtin_amdgcn_s_waitcnt(0);
      __syncthreads();
      const uint8_t* la = smem + (wn * 64) * kRowBytes;
      const uint8_t* lb = smem + kTileBytes + (wt * 64) * kRowBytes;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int c = ks * 2 + fh;
        const Pack16 a0 = read_frag(la, fr, c), a1 = read_frag(la, 32 + fr, c);
        const Pack16 b0 = read_frag(lb, fr, c), b1 = read_frag(lb, 32 + fr, c);
        acc[0][0] = mfma32<DT>(a0, b0, acc[0][0]);
        acc[0][1] = mfma32<DT>(a0, b1, acc[0][1]);
        acc[1][0] = mfma32<DT>(a1, b0, acc[1][0]);
        acc[1][1] = mfma32<DT>(a1, b1, acc[1][1]);
      }
      __syncthreads();  // all fragment reads done before the next tile overwrites the stage
    }
  }

  // epilogue.  C layout of 32x32: col (t) = lane & 31, row (n) = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5):
  // a lane holds runs of 4 consecutive n for one t -> 8-byte accesses of out / out_actual rows.
  float sq = 0.0f;
  const bool has_bias = bias != nullptr;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = n0 + wn * 64 + i * 32 + 8 * q + 4 * fh;
      if (n >= N) continue;  // N % 4 == 0 is required by the host, so a run of 4 is all-in or all-out
      float bv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      if (has_bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[e] = load1<DT>(bias, n + e);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int t = t0 + wt * 64 + j * 32 + fr;
        if (t >= T) continue;
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = round_to_dtype<DT>(acc[i][j][q * 4 + e] + bv[e]);
        const int64_t off = (int64_t)t * N + n;
        if constexpr (MODE == 0) {
          const uint2 rv = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(ref) + off);
          float rf[4];
          if constexpr (DT == MOQ_BF16) {
            rf[0] = __uint_as_float(rv.x << 16); rf[1] = __uint_as_float(rv.x & 0xFFFF0000u);
            rf[2] = __uint_as_float(rv.y << 16); rf[3] = __uint_as_float(rv.y & 0xFFFF0000u);
          } else {
            const f16x2 h0 = *reinterpret_cast<const f16x2*>(&rv.x), h1 = *reinterpret_cast<const f16x2*>(&rv.y);
            rf[0] = (float)h0.x; rf[1] = (float)h0.y; rf[2] = (float)h1.x; rf[3] = (float)h1.y;
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float d = round_to_dtype<DT>(o[e] - rf[e]);  // (out - out_actual) in the model dtype
            sq += d * d;                                       // .float().pow(2)
          }
        } else {
          float f4[8] = {o[0], o[1], o[2], o[3], 0, 0, 0, 0};
          const Pack16 p = pack<DT>(f4);
          uint2 st;
          st.x = p.w[0]; st.y = p.w[1];
          *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(out) + off) = st;
        }
      }
    }
  }
  if constexpr (MODE == 0) {
    // deterministic workgroup sum: butterfly inside the wave, fixed order across the 4 waves
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sq += __shfl_xor(sq, off, 64);
    __syncthreads();  // all LDS tile reads are done; reuse the first words
    float* red = reinterpret_cast<float*>(smem);
    if (lane == 0) red[wave] = sq;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
  }
}

// loss_acc[0] += (float)(sum(partial) / count): partial sums are added in index order in double
__global__ void err_finalize_kernel(const float* __restrict__ partial, int n, double inv_count,
                                    float* __restrict__ loss_acc) {
  __shared__ double sm[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += (double)partial[i];
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) sm[threadIdx.x] += sm[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss_acc[0] += (float)(sm[0] * inv_count);
}

}  // namespace moq

using namespace moq;
static inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }

static int gemm_check(const void* x, const void* w, int64_t tokens, int64_t cout, int64_t cin, int dt,
                      const char* who) {
  if (x == nullptr || w == nullptr || tokens < 0 || cout <= 0 || cin <= 0) {
    set_error("%s: null pointer or bad sizes", who);
    return MOQ_ERR_INVALID;
  }
  if (dt != MOQ_BF16 && dt != MOQ_F16) {
    set_error("%s: only bf16 / f16 operands run on the MFMA path (fp32 models use the library GEMM)", who);
    return MOQ_ERR_UNSUPPORTED;
  }
  if (cin % 8 != 0 || cout % 4 != 0) {
    set_error("%s: needs Cin %% 8 == 0 and Cout %% 4 == 0 (got Cin=%lld, Cout=%lld)", who, (long long)cin,
              (long long)cout);
    return MOQ_ERR_UNSUPPORTED;
  }
  if (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) & 15u) != 0) {
    set_error("%s: operands must be 16-byte aligned", who);
    return MOQ_ERR_UNSUPPORTED;
  }
  if (cin > (1 << 22) || tokens > (1 << 30) || cout > (1 << 30)) {
    set_error("%s: dimension too large", who);
    return MOQ_ERR_UNSUPPORTED;
  }
  return MOQ_OK;
}

static int64_t n_tiles(int64_t tokens, int64_t cout) {
  return ((tokens + kTile - 1) / kTile) * ((cout + kTile - 1) / kTile);
}

template <int MODE>
static int launch_gemm(const void* x, const void* w, const void* ref, const void* bias, void* out,
                       float* partial, int64_t tokens, int64_t cout, int64_t cin, int dt, void* stream) {
  const int tiles_t = (int)((tokens + kTile - 1) / kTile), tiles_n = (int)((cout + kTile - 1) / kTile);
  const int64_t nblk = (int64_t)tiles_t * tiles_n;
  if (nblk > 0x7FFFFFFF) {
    set_error("gemm: too many tiles");
    return MOQ_ERR_UNSUPPORTED;
  }
  // MOQ_TUNE_GEMM_DBUF=1 selects the two-stage variant (A/B knob; default is the single-stage kernel)
  static const bool dbuf = [] { const char* e = getenv("MOQ_TUNE_GEMM_DBUF"); return e && atoi(e) != 0; }();
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)err_gemm_kernel<MOQ_BF16, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kGemmLds);
    (void)hipFuncSetAttribute((const void*)err_gemm_kernel<MOQ_BF16, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kGemmLds);
    (void)hipFuncSetAttribute((const void*)err_gemm_kernel<MOQ_F16, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kGemmLds);
    (void)hipFuncSetAttribute((const void*)err_gemm_kernel<MOQ_F16, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kGemmLds);
    attr_set = true;
  }
#define MOQ_LAUNCH_GEMM(DTV, DB)                                                                              \
  hipLaunchKernelGGL((err_gemm_kernel<DTV, MODE, DB>), dim3((unsigned)nblk), dim3(256),                       \
                     DB ? kGemmLds : kStageBytes, S(stream), x, w, ref, bias, out, partial, (int)tokens,      \
                     (int)cout, (int)cin, tiles_t, tiles_n)
  if (dt == MOQ_BF16) {
    if (dbuf) MOQ_LAUNCH_GEMM(MOQ_BF16, true); else MOQ_LAUNCH_GEMM(MOQ_BF16, false);
  } else {
    if (dbuf) MOQ_LAUNCH_GEMM(MOQ_F16, true); else MOQ_LAUNCH_GEMM(MOQ_F16, false);
  }
#undef MOQ_LAUNCH_GEMM
  return MOQ_OK;
}

extern "C" int64_t moq_awq_err_gemm_workspace(int64_t tokens, int64_t cout) {
  if (tokens < 0 || cout < 0) return MOQ_ERR_INVALID;
  const int64_t n = n_tiles(tokens, cout);
  return n < 1 ? 1 : n;
}

extern "C" int moq_awq_err_gemm(const void* x, const void* w, const void* out_actual, const void* bias,
                                int64_t tokens, int64_t cout, int64_t cin, int dt, float* partial,
                                float* loss_acc, void* stream) {
  int rc = gemm_check(x, w, tokens, cout, cin, dt, "moq_awq_err_gemm");
  if (rc != MOQ_OK) return rc;
  if (out_actual == nullptr || partial == nullptr || loss_acc == nullptr) {
    set_error("moq_awq_err_gemm: out_actual / partial / loss_acc must not be NULL");
    return MOQ_ERR_INVALID;
  }
  if ((reinterpret_cast<uintptr_t>(out_actual) & 7u) != 0) {
    set_error("moq_awq_err_gemm: out_actual must be 8-byte aligned");
    return MOQ_ERR_UNSUPPORTED;
  }
  if (tokens == 0) return MOQ_OK;
  rc = launch_gemm<0>(x, w, out_actual, bias, nullptr, partial, tokens, cout, cin, dt, stream);
  if (rc != MOQ_OK) return rc;
  hipLaunchKernelGGL(err_finalize_kernel, dim3(1), dim3(256), 0, S(stream), partial, (int)n_tiles(tokens, cout),
                     1.0 / ((double)tokens * (double)cout), loss_acc);
  return check_launch("moq_awq_err_gemm");
}

extern "C" int moq_gemm_nt(const void* x, const void* w, const void* bias, void* out, int64_t tokens,
                           int64_t cout, int64_t cin, int dt, void* stream) {
  int rc = gemm_check(x, w, tokens, cout, cin, dt, "moq_gemm_nt");
  if (rc != MOQ_OK) return rc;
  if (out == nullptr || (reinterpret_cast<uintptr_t>(out) & 7u) != 0) {
    set_error("moq_gemm_nt: out must be a non-NULL 8-byte aligned pointer");
    return MOQ_ERR_INVALID;
  }
  if (tokens == 0) return MOQ_OK;
  rc = launch_gemm<1>(x, w, nullptr, bias, out, nullptr, tokens, cout, cin, dt, stream);
  if (rc != MOQ_OK) return rc;
  return check_launch("moq_gemm_nt");
}
