// moq_gemm.hip -- the one dense contraction of the PTQ hot path: the AWQ-lite search error GEMM (a12).
//
// For every candidate alpha the reference runs the patched linear forward
//     out = F.linear(x * (1/s), QDQ(W * s), bias);  loss[alpha] += (out - out_actual).float().pow(2).mean()
// (quantization/model_calib.py:1489-1495, :1552-1556).  Here the contraction and the loss are ONE kernel:
// MFMA tiles of  out^T[n, t] = sum_k What[n, k] * xs[t, k]  stay in registers, are rounded to the model dtype
// exactly where the reference materialises `out`, subtracted from out_actual in the model dtype, squared in
// fp32 and reduced -- `out` (T x Cout) is never written to HBM.
//
// Tiling (gfx950, wave64): 256(n) x 256(t) x 64(k) per 512-thread workgroup, 2 x 4 waves, each wave owns 128 x 64 as
// 4 x 2 v_mfma_f32_32x32x16 tiles (128 accumulator VGPRs).  Both operands are K-contiguous ([Cout, Cin] weights,
// [tokens, Cin] activations), so an MFMA fragment is one 16-byte run of k per lane.
// HBM -> LDS goes through `buffer_load_dwordx4 ... lds` (no staging VGPRs, no ds_write pass); the buffer
// descriptor's bounds check returns zeros for rows past the matrix edge and for the K tail, so ragged
// shapes need no second kernel.  LDS rows are 128 B (64 k); the 16-byte chunk c of row r is stored at chunk
// position c ^ ((r >> 1) & 7): a ds_read_b128 fragment read (32 rows x one chunk column) then touches 16
// distinct 16-byte slots per 16-lane service group = conflict-free (MI355X_MICROARCH.md, LDS table).  The
// swizzle is applied on the *source* address because the LDS side of the DMA is lane-linear.
// Two LDS stages of an A and a B tile (128 KiB per workgroup, one workgroup per CU); tile k+1 streams in while tile k is
// in the matrix cores; one barrier per K-step.
//
// Roofline: MFMA-bound.  2 * T * Cout * Cin flop per launch against ~2.5 PFLOP/s dense bf16.
//
// This file is the RELEASE contraction: the two loop structures that ship (GEO 10, the default, and GEO 4, its
// known-good predecessor -- same tile, same k order per accumulator, bit-identical results).  Every other structure that was
// built and measured on the way (128 x 128 tiles, four-stage K = 32, the ping-pong groups, direct-to-register token
// operands, and timing-only diagnostics) was removed from the tree in round 4 -- rounds 1-3 of the history hold
// csrc/exp/moq_gemm_exp.hip; their measurements are profiles/r01_gemm_table.md, r02_gemm_table.md, r03_gemm_geo_regprefetch.md.
#include <atomic>
#include <type_traits>
#include <stdlib.h>

#include "moq_common.h"

namespace moq {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

constexpr int kBK = 64;            // k per stage
constexpr int kRowBytes = kBK * 2;  // 128 B LDS rows

// Waves form a WN x WT grid; each wave owns NI x NJ MFMA tiles of 32 x 32.
//   GEO 4:  256 x 256, 8 waves, two 64 KiB stages, ONE workgroup per CU; the next tile's LDS-DMA is issued from inline asm
//           right after the barrier; the K-loop is rotated by one sub-step: the MFMAs of the LAST sub-step of tile k run
//           after the barrier that opens tile k + 1, underneath that tile's first fragment reads
//   GEO 10 (default): the GEO 4 loop as ONE pinned stream in which every MFMA is followed by one memory instruction --
//           a fragment read of the NEXT sub-step or an LDS-DMA piece of the next tile (32 MFMAs : 24 reads + 8 pieces per
//           wave and K-tile = 1 : 1, the recipe of the hand-scheduled kernels); the pieces go out in the first two
//           sub-steps so that they have two sub-steps of lead before the tile-boundary wait
template <int GEO> struct Geo { static constexpr int TILE = 256, WAVES = 8, WN = 2, NI = 4, NJ = 2, STAGES = 2; };
template <int GEO> constexpr int tile_bytes() { return Geo<GEO>::TILE * kRowBytes; }
template <int GEO> constexpr int stage_bytes() { return 2 * tile_bytes<GEO>(); }
template <int GEO> constexpr int lds_bytes() { return Geo<GEO>::STAGES * stage_bytes<GEO>(); }
// dynamic LDS of a launch: the operand stages, or (MODE 0) the staged out_actual tile of the epilogue: 128 pieces x 1040 bytes
template <int MODE, int GEO> constexpr int launch_lds_bytes() {
  return MODE == 0 && lds_bytes<GEO>() < 128 * 1040 ? 128 * 1040 : lds_bytes<GEO>();
}

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

typedef int i32x4_t __attribute__((ext_vector_type(4)));
template <int V> using IC = std::integral_constant<int, V>;
typedef __attribute__((address_space(3))) uint8_t* lds_u8_t;

// One operand tile (TILE rows x 64 k) HBM -> LDS.  `rsrc` covers the tile's valid rows only (base = first row
// of the tile, num_records = valid_rows * ld * 2), so out-of-range rows read as zero.  Each wave-instruction
// moves 8 rows x 128 B; lane l fills LDS slot (row l >> 3, chunk position l & 7) with the source chunk
// (l & 7) ^ ((row >> 1) & 7).
// The descriptor travels as 4 dwords (V# layout: base[47:0], stride = 0, num_records, flags 0x00020000 = raw dword
// format) so that the inline-asm form can name it as an SGPR quad.
struct TileDesc {
  i32x4_t words;  // wave-uniform
  __amdgpu_buffer_rsrc_t rsrc;
};
__device__ __forceinline__ TileDesc make_tile_desc(const void* base, int num_bytes) {
  TileDesc d;
  const uint64_t a = reinterpret_cast<uint64_t>(base);
  d.words.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
  d.words.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)((a >> 32) & 0xFFFFu));
  d.words.z = __builtin_amdgcn_readfirstlane(num_bytes);
  d.words.w = 0x00020000;
  d.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, num_bytes, 0x00020000);
  return d;
}

template <int GEO, bool ASM>
__device__ __forceinline__ void stage_tile(const TileDesc& desc, uint8_t* lds_tile, int64_t ld_bytes,
                                           int k0, int K, int wave, int lane) {
  constexpr int PER_WAVE = Geo<GEO>::TILE / 8 / Geo<GEO>::WAVES;
  const int r_local = lane >> 3, pos = lane & 7;
  const i32x4_t rs = desc.words;
#pragma unroll
  for (int j = 0; j < PER_WAVE; ++j) {
    const int rbase = (wave * PER_WAVE + j) * 8;
    const int r = rbase + r_local;
    const int c = pos ^ ((r >> 1) & 7);
    const int k = k0 + c * 8;
    // K tail (K % 8 == 0 guaranteed): chunks at or past K must read as zero -> force an out-of-range offset
    const int voff = k < K ? (int)(r * ld_bytes + k * 2) : 0x7FFFFFF0;
    uint8_t* dst = lds_tile + rbase * kRowBytes;
    if constexpr (ASM) {
      const uint32_t m0v = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_u8_t)dst);
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                   :: "s"(m0v), "v"(voff), "s"(rs) : "memory");
    } else {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(desc.rsrc, (lds_ptr_t)dst, 16, voff, 0, 0, 0);
    }
  }
}

// fragment read: row r of the tile, 16-byte chunk c (k = c*8 .. c*8+7)
__device__ __forceinline__ Pack16 read_frag(const uint8_t* lds_tile, int r, int c) {
  return *reinterpret_cast<const Pack16*>(lds_tile + r * kRowBytes + ((c ^ ((r >> 1) & 7)) << 4));
}


// Workgroup -> output tile.  (1) XCD-aware: workgroup b runs on XCD b % 8; each XCD gets a contiguous run of tile ids
// (bijective for any grid size).  (2) Grouped order inside the run: consecutive ids -- the 32 workgroups an XCD runs at a
// time -- cover kTileGroup weight tiles x (32 / kTileGroup) token tiles instead of 2 x 16, so a K-step of the XCD touches
// 8 + 4 operand slices instead of 2 + 16: a third less traffic behind the XCD's L2 (hit rate 72 % -> 81 % by count).
constexpr int kTileGroup = 8;
__device__ __forceinline__ void tile_of_block(int bid, int tiles_t, int tiles_n, int group, int& tn, int& tt) {
  const int nblk = tiles_t * tiles_n;
  {
    const int xcd = bid & 7, idx = bid >> 3, q = nblk >> 3, r = nblk & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  if (group <= 1) {
    tn = bid / tiles_t;  // consecutive workgroups share the W tile
    tt = bid % tiles_t;
    return;
  }
  const int per_group = group * tiles_t;
  const int gid = bid / per_group, first_n = gid * group;
  const int gsz = tiles_n - first_n < group ? tiles_n - first_n : group;
  const int in_group = bid - gid * per_group;
  tn = first_n + in_group % gsz;
  tt = in_group / gsz;
}

// Gram mode (MODE 2): only tiles on and above the diagonal are contracted.  Dealing the n x n tile grid out in runs per XCD
// left XCD 7 with 13x the live tiles of XCD 0 (rows 49..55 of 56 against rows 0..6) -- the launch took as long as that
// XCD.  The triangle is folded into a rectangle instead: row r is paired with row n - 1 - r (r + 1 and n - r live tiles:
// n + 1 together), so the grid is ceil(n / 2) x (n + 1) workgroups, every one of them live (but the second half of an
// odd side's middle row), and tile_of_block's XCD runs and 8 x 4 groups apply to that rectangle.  Returns false for a
// dead cell.
__device__ __forceinline__ bool gram_tile(int r, int c, int n, int& tn, int& tt) {
  if (c <= r) {
    tn = r;
    tt = c;
    return true;
  }
  if (2 * r == n - 1) return false;  // odd side: the middle row has no partner
  tn = n - 1 - r;
  tt = c - (r + 1);
  return true;
}

// The epilogue shared by every loop structure: the wave's NI x NJ accumulator tiles against `ref` / into `out`.
// C layout of 32x32: col (t) = lane & 31, row (n) = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5): a lane holds runs of 4
// consecutive n for one t -> 8-byte accesses of out / out_actual rows.
template <int DT, int MODE, int NI, int NJ, int WAVES>
__device__ __forceinline__ void gemm_epilogue(f32x16_t (&acc)[NI][NJ], const void* __restrict__ ref,
                                              const void* __restrict__ bias, void* __restrict__ out,
                                              float* __restrict__ partial, uint8_t* smem, int T, int N, int n0, int t0,
                                              int tn, int tt, int wn, int wt, int fr, int fh, int lane, int wave,
                                              float decay, float scale, int upper_only) {
  float sq = 0.0f;
  const bool has_bias = bias != nullptr;
  // MODE 0: the workgroup's 256 x 256 tile of out_actual is STAGED through the LDS (free once the K loop is over) with
  // full-line LDS-DMA requests.  Read straight from memory, the C layout makes a lane fetch 8 bytes of ONE row -- a wave
  // instruction touches 32 rows, 16 useful bytes of every 128-byte line, and a 4096 x 14336 launch adds 7 M L2 requests
  // to the 29 M of its operand stream (PMC, profiles/r03_gemm_geo_regprefetch.md): 5-10 % of the launch at Cin = 4096.
  // Layout: piece p = rows 2p, 2p + 1 of the tile (512 bytes each, lanes 0-31 / 32-63), 1040 bytes apart (16 bytes of
  // padding: consecutive rows land on different banks).  Rows past T read as zero through the descriptor's range check;
  // columns past N read the next row's bytes -- both belong to cells the loop below skips.
  constexpr int kRefPitch = 1040;
  bool staged = false;
  if constexpr (MODE == 0) {
    staged = ((N & 7) == 0) && ((reinterpret_cast<uintptr_t>(ref) & 15u) == 0);  // 16-byte rows (workgroup-uniform)
    if (staged) {
      __syncthreads();  // every wave is done with the operand stages
      const int64_t origin = ((int64_t)t0 * N + n0) * 2, left = (int64_t)T * N * 2 - origin;
      const TileDesc rd = make_tile_desc(reinterpret_cast<const uint8_t*>(ref) + origin,
                                         (int)(left < 0x7FFFFFFF ? left : 0x7FFFFFFF));
      const i32x4_t rr = rd.words;
      const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_u8_t)smem);
      const int vrow = (lane >> 5) * N * 2 + (lane & 31) * 16;
#pragma unroll
      for (int k = 0; k < 128 / WAVES; ++k) {
        const int p = wave + WAVES * k;
        const uint32_t m0v = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(p * kRefPitch));  // (p holds the wave index)
        const int voff = p * 2 * N * 2 + vrow;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                     :: "s"(m0v), "v"(voff), "s"(rr) : "memory");
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = n0 + (wn * NI + i) * 32 + 8 * q + 4 * fh;
      if (n >= N) continue;  // N % 4 == 0 is required by the host, so a run of 4 is all-in or all-out
      float bv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      if (has_bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[e] = load1<DT>(bias, n + e);
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int t = t0 + (wt * NJ + j) * 32 + fr;
        if (t >= T) continue;
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = round_to_dtype<DT>(acc[i][j][q * 4 + e] + bv[e]);
        const int64_t off = (int64_t)t * N + n;
        if constexpr (MODE == 2) {
          float4* hp = reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + off);
          float4 h = *hp;
          h.x = h.x * decay + scale * acc[i][j][q * 4 + 0];
          h.y = h.y * decay + scale * acc[i][j][q * 4 + 1];
          h.z = h.z * decay + scale * acc[i][j][q * 4 + 2];
          h.w = h.w * decay + scale * acc[i][j][q * 4 + 3];
          *hp = h;
          if (tn != tt && !(upper_only & 1)) {  // mirrored block (MODE 2: bit 0 = upper_only, the rest = tile group): out[n + e, t]; lanes run along t -> 128-byte runs
            float* hm = reinterpret_cast<float*>(out) + (int64_t)n * N + t;
#pragma unroll
            for (int e = 0; e < 4; ++e) hm[(int64_t)e * N] = hm[(int64_t)e * N] * decay + scale * acc[i][j][q * 4 + e];
          }
        } else if constexpr (MODE == 3) {
          const float4 rv = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(ref) + off);
          sq += acc[i][j][q * 4 + 0] * rv.x;
          sq += acc[i][j][q * 4 + 1] * rv.y;
          sq += acc[i][j][q * 4 + 2] * rv.z;
          sq += acc[i][j][q * 4 + 3] * rv.w;
        } else if constexpr (MODE == 0) {
          uint2 rv;
          if (staged) {
            const int tl = t - t0, nl = n - n0;
            rv = *reinterpret_cast<const uint2*>(smem + (tl >> 1) * kRefPitch + (tl & 1) * 512 + nl * 2);
          } else {
            rv = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(ref) + off);
          }
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
  if constexpr (MODE == 0 || MODE == 3) {
    // deterministic workgroup sum: butterfly inside the wave, fixed order across the waves
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sq += __shfl_xor(sq, off, 64);
    __syncthreads();  // all LDS tile reads are done; reuse the first words
    float* red = reinterpret_cast<float*>(smem);
    if (lane == 0) red[wave] = sq;
    __syncthreads();
    if (threadIdx.x == 0) {
      float s = red[0];
      for (int v = 1; v < WAVES; ++v) s += red[v];
      partial[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = s;
    }
  }
}

// MODE 0: accumulate the squared error against `ref` into partial[block]; MODE 1: store out[t, n];
// MODE 2: out is fp32 [T, N], T == N, x == w: out = out * decay + scale * acc (running Gram / Hessian X^T X of
//         SparseGPT and of the AWQ Gram search); only tiles with n-tile >= t-tile are contracted.  upper_only = 0:
//         the mirror images are written from the transposed accumulators (4-byte read-modify-writes: slow, but the
//         matrix is complete after every call); upper_only = 1: they are left alone and the caller mirrors the matrix
//         ONCE when the accumulation is over (moq_symmetrize);
// MODE 3: `ref` is fp32 [T, N]: partial[block] = sum acc * ref (the dot product <x w^T, ref> of the AWQ Gram search).
template <int DT, int MODE, int GEO>
__global__ __launch_bounds__(Geo<GEO>::WAVES * 64, 2)
void err_gemm_kernel(const void* __restrict__ x,     // [T, K]
                     const void* __restrict__ w,     // [N, K]
                     const void* __restrict__ ref,   // [T, N] (MODE 0)
                     const void* __restrict__ bias,  // [N] or null
                     void* __restrict__ out,         // [T, N] (MODE 1)
                     float* __restrict__ partial, int T, int N, int K, int tiles_t, int tiles_n,
                     int64_t x_stride, int64_t w_stride, float decay, float scale, int upper_only) {
  constexpr int TILE = Geo<GEO>::TILE, NI = Geo<GEO>::NI, NJ = Geo<GEO>::NJ;
  constexpr int WTC = Geo<GEO>::WAVES / Geo<GEO>::WN;  // waves along t
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int tn, tt;
  tile_of_block(blockIdx.x, tiles_t, tiles_n, MODE == 2 ? (upper_only >> 1) : upper_only, tn, tt);  // group size travels in upper_only
  if constexpr (MODE == 2) {
    // tiles_t / tiles_n describe the folded triangle here: (n + 1) columns x ceil(n / 2) row pairs (gram_tile)
    if (!gram_tile(tn, tt, tiles_t - 1, tn, tt)) return;
  }
  const int n0 = tn * TILE, t0 = tt * TILE;
  // blockIdx.y = candidate index of a batched launch (all alphas of one linear in one grid): every candidate has
  // its own operands x[a] / w[a] and its own partial-sum plane; `ref` / `bias` are shared
  x = reinterpret_cast<const uint8_t*>(x) + (int64_t)blockIdx.y * x_stride * 2;
  w = reinterpret_cast<const uint8_t*>(w) + (int64_t)blockIdx.y * w_stride * 2;
  if constexpr (MODE == 1) out = reinterpret_cast<uint8_t*>(out) + (int64_t)blockIdx.y * (int64_t)T * N * 2;
  const int rows_w = N - n0 < TILE ? N - n0 : TILE;
  const int rows_x = T - t0 < TILE ? T - t0 : TILE;
  const int64_t ld_bytes = (int64_t)K * 2;
  const TileDesc rs_w = make_tile_desc(reinterpret_cast<const uint8_t*>(w) + (int64_t)n0 * ld_bytes,
                                       (int)(rows_w * ld_bytes));
  const TileDesc rs_x = make_tile_desc(reinterpret_cast<const uint8_t*>(x) + (int64_t)t0 * ld_bytes,
                                       (int)(rows_x * ld_bytes));

  f32x16_t acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  const int wn = wave / WTC, wt = wave % WTC;  // wave's sub-tile: n rows wn*NI*32.., t cols wt*NJ*32..
  const int fr = lane & 31, fh = lane >> 5;
  const int nk = (K + kBK - 1) / kBK;
  constexpr int TB = tile_bytes<GEO>(), SB = stage_bytes<GEO>();

  {
    const uint8_t* la0 = smem + (wn * NI * 32) * kRowBytes;
    const uint8_t* lb0 = smem + TB + (wt * NJ * 32) * kRowBytes;
    Pack16 a[2][NI], b[2][NJ];
    auto read_sub = [&](int buf, int stage_off, int ks) {
      const int c = ks * 2 + fh;
#pragma unroll
      for (int i = 0; i < NI; ++i) a[buf][i] = read_frag(la0 + stage_off, i * 32 + fr, c);
#pragma unroll
      for (int j = 0; j < NJ; ++j) b[buf][j] = read_frag(lb0 + stage_off, j * 32 + fr, c);
    };
    auto mma_sub = [&](int buf) {
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = mfma32<DT>(a[buf][i], b[buf][j], acc[i][j]);
    };
    // LDS-DMA piece p (0..7) of this wave for the tile in stage `sn`: 8 rows of W (p < 4) or of x (p >= 4).  Everything
    // that does not change from tile to tile is computed once: the LDS base of the wave's rows as a 32-bit LDS address
    // (no generic -> local cast, whose null check costs ~8 instructions per piece) and the lane's byte offset inside the
    // operand tile for k0 = 0 (W and x have the same row pitch, so one set serves both).  The tile's k offset travels in
    // the instruction's SCALAR offset, which the descriptor's range check ignores: rows past the matrix edge still read
    // as zero.  The ragged last tile (K % 64 != 0) swaps in a second offset set whose chunks at or past K point out of
    // range (one v_cndmask on a wave-uniform condition).  `live` = false (no next tile): the piece still issues --
    // against an empty descriptor, filling its rows of the idle stage with zeros -- instead of branching around every
    // piece of the pinned streams.  Three to four instructions per piece, no branch, one VALU.
    const uint32_t lds_wave = __builtin_amdgcn_readfirstlane(
        (uint32_t)(uintptr_t)(lds_u8_t)smem + (uint32_t)(wave * 4 * 8 * kRowBytes));
    const bool k_ragged = (K & (kBK - 1)) != 0;
    int voff[4], voff_tail[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = (wave * 4 + j) * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((r >> 1) & 7);
      voff[j] = (int)(r * ld_bytes + c * 16);
      voff_tail[j] = (nk - 1) * kBK + c * 8 < K ? voff[j] : 0x7FFFFFF0;
    }
    const i32x4_t rsw = rs_w.words, rsx = rs_x.words;
    auto piece = [&](int sn, int k0, auto P, bool live = true) {  // P: compile-time piece index (register selection)
      constexpr int p = decltype(P)::value, j = p & 3;
      const uint32_t m0v = lds_wave + (uint32_t)(sn * SB + (p < 4 ? 0 : TB) + j * 8 * kRowBytes);
      const int koff = k0 * 2;
      // (operands copied to locals first: clang rejects captured variables named directly in an asm statement of a
      // generic lambda)
      const int vfull = voff[j], vtail = voff_tail[j];
      const int vo = (k_ragged && k0 + kBK > K) ? vtail : vfull;
      i32x4_t rr = p < 4 ? rsw : rsx;
      rr.z = live ? rr.z : 0;
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                   :: "s"(m0v), "v"(vo), "s"(rr), "s"(koff) : "memory");
    };
    stage_tile<GEO, true>(rs_w, smem, ld_bytes, 0, K, wave, lane);
    stage_tile<GEO, true>(rs_x, smem + TB, ld_bytes, 0, K, wave, lane);
    for (int kt = 0; kt < nk; ++kt) {
      const int so = (kt & 1) * SB;
      // tile kt landed (own part; the barrier makes it everyone's) and every fragment read of tile kt - 1 has
      // completed -- its last sub-step sits in a[1] / b[1], not yet multiplied
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __syncthreads();
      read_sub(0, so, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, NI + NJ, 0);
      if constexpr (GEO == 10) {
        const int sn = (kt + 1) & 1;
        const bool more = kt + 1 < nk;
        const int k0 = (kt + 1) * kBK;
        {
          // eight MFMAs of register buffer BUF, each followed by one read of sub-step KS into the other buffer (six)
          // and, when `on`, by one LDS-DMA piece after every second MFMA (pieces p0 .. p0 + 3)
          auto group = [&](auto BUF, auto KS, bool on, auto P0) {
            constexpr int buf = decltype(BUF)::value, ks = decltype(KS)::value, nb = buf ^ 1, p0 = decltype(P0)::value;
            const int c = ks * 2 + fh;
            auto one = [&](auto NC) {
              constexpr int n = decltype(NC)::value;
              acc[n >> 1][n & 1] = mfma32<DT>(a[buf][n >> 1], b[buf][n & 1], acc[n >> 1][n & 1]);
              __builtin_amdgcn_sched_barrier(0);
              if constexpr (n < 4) a[nb][n] = read_frag(la0 + so, n * 32 + fr, c);
              else if constexpr (n < 6) b[nb][n - 4] = read_frag(lb0 + so, (n - 4) * 32 + fr, c);
              if constexpr ((n & 1) != 0 && p0 >= 0) piece(sn, k0, IC<p0 + (n >> 1)>{}, on);
              __builtin_amdgcn_sched_barrier(0);
            };
            one(std::integral_constant<int, 0>{}); one(std::integral_constant<int, 1>{});
            one(std::integral_constant<int, 2>{}); one(std::integral_constant<int, 3>{});
            one(std::integral_constant<int, 4>{}); one(std::integral_constant<int, 5>{});
            one(std::integral_constant<int, 6>{}); one(std::integral_constant<int, 7>{});
          };
          using I0 = std::integral_constant<int, 0>;
          using I1 = std::integral_constant<int, 1>;
          if (kt > 0) {
            group(I1{}, I0{}, more, I0{});  // last sub-step of tile kt - 1 under the first reads of tile kt
          } else {
            read_sub(0, so, 0);
            if (more) {
              piece(sn, k0, IC<0>{}); piece(sn, k0, IC<1>{}); piece(sn, k0, IC<2>{}); piece(sn, k0, IC<3>{});
            }
          }
          group(I0{}, I1{}, more, std::integral_constant<int, 4>{});
          group(I1{}, std::integral_constant<int, 2>{}, false, std::integral_constant<int, -1>{});
          group(I0{}, std::integral_constant<int, 3>{}, false, std::integral_constant<int, -1>{});
        }
        continue;
      }
      if (kt > 0) {
        mma_sub(1);  // last sub-step of tile kt - 1, under the reads above
        __builtin_amdgcn_sched_group_barrier(0x008, NI * NJ, 0);
      }
      if (kt + 1 < nk) {  // next tile's DMA, issued in the gaps of the MFMAs above
        const int sn = (kt + 1) & 1, k0 = (kt + 1) * kBK;
        piece(sn, k0, IC<0>{}); piece(sn, k0, IC<1>{}); piece(sn, k0, IC<2>{}); piece(sn, k0, IC<3>{});
        piece(sn, k0, IC<4>{}); piece(sn, k0, IC<5>{}); piece(sn, k0, IC<6>{}); piece(sn, k0, IC<7>{});
      }
      read_sub(1, so, 1);
      __builtin_amdgcn_sched_group_barrier(0x100, NI + NJ, 0);
      mma_sub(0);
      __builtin_amdgcn_sched_group_barrier(0x008, NI * NJ, 0);
      read_sub(0, so, 2);
      __builtin_amdgcn_sched_group_barrier(0x100, NI + NJ, 0);
      mma_sub(1);
      __builtin_amdgcn_sched_group_barrier(0x008, NI * NJ, 0);
      read_sub(1, so, 3);
      __builtin_amdgcn_sched_group_barrier(0x100, NI + NJ, 0);
      mma_sub(0);
      __builtin_amdgcn_sched_group_barrier(0x008, NI * NJ, 0);
    }
    mma_sub(1);  // last sub-step of the last tile
    // the last tile's "dead" pieces (zero fills of the idle stage) must have landed before the epilogue reuses the LDS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }

  gemm_epilogue<DT, MODE, NI, NJ, Geo<GEO>::WAVES>(acc, ref, bias, out, partial, smem, T, N, n0, t0, tn, tt, wn, wt, fr, fh,
                                                   lane, wave, decay, scale, upper_only);
}

// loss_acc[0] += (float)(sum(partial) / count): partial sums are added in index order in double
__global__ void err_finalize_kernel(const float* __restrict__ partial, int n, double inv_count,
                                    float* __restrict__ loss_acc) {
  __shared__ double sm[256];
  partial += (int64_t)blockIdx.x * n;  // one workgroup per candidate
  loss_acc += blockIdx.x;
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

// fp32 operands: the fp32 matrix-core kernel of moq_gemm_f32.hip (same epilogues, 128 x 128 tiles)
int64_t moq_f32_launch_err(const void* x, const void* w, const void* ref, const void* bias, float* partial, int64_t tokens,
                           int64_t cout, int64_t cin, int n_cand, int64_t x_stride, int64_t w_stride, void* stream);
int64_t moq_f32_launch_store(const void* x, const void* w, const void* bias, void* out, int64_t tokens, int64_t cout,
                             int64_t cin, void* stream);
int64_t moq_f32_launch_dot(const void* a, const void* b, const void* ref, float* partial, int64_t rows, int64_t cols,
                           int64_t k, void* stream);

static int gemm_check(const void* x, const void* w, int64_t tokens, int64_t cout, int64_t cin, int dt,
                      const char* who) {
  if (x == nullptr || w == nullptr || tokens < 0 || cout <= 0 || cin <= 0) {
    set_error("%s: null pointer or bad sizes", who);
    return MOQ_ERR_INVALID;
  }
  if (dt != MOQ_BF16 && dt != MOQ_F16 && dt != MOQ_F32) {
    set_error("%s: operands must be bf16, f16 or f32", who);
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

static int gemm_geo() {
  // MOQ_TUNE_GEMM_GEO = 4 selects the block-issue loop, anything else the default GEO 10 stream (read once).  Both are
  // release kernels with bit-identical results (tests/test_gpu_gemm.py); the knob exists for A/B timing
  // (profiles/r02_gemm_table.md: GEO 10 runs 3-5 % ahead).
  static const int geo = [] {
    const char* e = getenv("MOQ_TUNE_GEMM_GEO");
    return e && atoi(e) == 4 ? 4 : 10;
  }();
  return geo;
}
static int64_t n_tiles_for(int64_t tokens, int64_t cout, int tile) {
  return ((tokens + tile - 1) / tile) * ((cout + tile - 1) / tile);
}

template <int MODE, int GEO>
static void launch_geo(const void* x, const void* w, const void* ref, const void* bias, void* out, float* partial,
                       int64_t tokens, int64_t cout, int64_t cin, int dt, int n_cand, int64_t x_stride,
                       int64_t w_stride, void* stream, float decay = 0.0f, float scale = 0.0f, int upper_only = 0) {
  constexpr int TILE = Geo<GEO>::TILE;
  constexpr int kLds = launch_lds_bytes<MODE, GEO>();
  int tiles_t = (int)((tokens + TILE - 1) / TILE), tiles_n = (int)((cout + TILE - 1) / TILE);
  if (MODE == 2) {  // the folded upper triangle (gram_tile): (n + 1) x ceil(n / 2) workgroups
    const int n = tiles_n;
    tiles_t = n + 1;
    tiles_n = (n + 1) / 2;
  }
  const unsigned nblk = (unsigned)(tiles_t * tiles_n);
  // > 64 KiB of dynamic LDS needs the opt-in attribute -- per DEVICE (a process may drive several GPUs), set by
  // whichever thread gets there first (setting it twice is harmless, so a relaxed bit mask is enough)
  static std::atomic<uint64_t> attr_set{0};
  int device = 0;
  (void)hipGetDevice(&device);
  const uint64_t bit = 1ull << (device & 63);
  if (!(attr_set.load(std::memory_order_acquire) & bit)) {
    lds_opt_in((const void*)err_gemm_kernel<MOQ_BF16, MODE, GEO>, (int)kLds, "err_gemm_kernel");
    lds_opt_in((const void*)err_gemm_kernel<MOQ_F16, MODE, GEO>, (int)kLds, "err_gemm_kernel");
    attr_set.fetch_or(bit, std::memory_order_release);
  }
  const dim3 grid(nblk, (unsigned)n_cand), block(Geo<GEO>::WAVES * 64);
  if (dt == MOQ_BF16) {
    hipLaunchKernelGGL((err_gemm_kernel<MOQ_BF16, MODE, GEO>), grid, block, kLds, S(stream), x, w, ref,
                       bias, out, partial, (int)tokens, (int)cout, (int)cin, tiles_t, tiles_n, x_stride, w_stride,
                       decay, scale, upper_only);
  } else {
    hipLaunchKernelGGL((err_gemm_kernel<MOQ_F16, MODE, GEO>), grid, block, kLds, S(stream), x, w, ref,
                       bias, out, partial, (int)tokens, (int)cout, (int)cin, tiles_t, tiles_n, x_stride, w_stride,
                       decay, scale, upper_only);
  }
}

// returns the number of per-tile partial sums each candidate produced (MODE 0), or a negative status
template <int MODE>
static int64_t launch_gemm(const void* x, const void* w, const void* ref, const void* bias, void* out,
                           float* partial, int64_t tokens, int64_t cout, int64_t cin, int dt, int n_cand,
                           int64_t x_stride, int64_t w_stride, void* stream, float decay = 0.0f,
                           float scale = 0.0f, int upper_only = 0) {
  const int geo = gemm_geo();
  if constexpr (MODE != 2) {
    // the kernels' last argument is the Gram mode's upper_only flag; the other modes carry the tile-group size of the
    // workgroup -> tile order in it (tile_of_block)
    upper_only = kTileGroup;
  } else {
    // Gram mode: bit 0 stays the upper_only flag, the tile-group size rides above it
    upper_only = (upper_only ? 1 : 0) | (kTileGroup << 1);
  }
  const int tile = 256;
  const int64_t nblk = n_tiles_for(tokens, cout, tile);
  if (nblk > 0x7FFFFFFF) {
    set_error("gemm: too many tiles");
    return MOQ_ERR_UNSUPPORTED;
  }
  if (geo == 4)
    launch_geo<MODE, 4>(x, w, ref, bias, out, partial, tokens, cout, cin, dt, n_cand, x_stride, w_stride, stream, decay, scale, upper_only);
  else
    launch_geo<MODE, 10>(x, w, ref, bias, out, partial, tokens, cout, cin, dt, n_cand, x_stride, w_stride, stream, decay, scale, upper_only);
  return nblk;
}

extern "C" int64_t moq_awq_err_gemm_workspace(int64_t tokens, int64_t cout) {
  if (tokens < 0 || cout < 0) return MOQ_ERR_INVALID;
  const int64_t n = n_tiles_for(tokens, cout, 128);  // upper bound over all tile geometries
  return n < 1 ? 1 : n;
}

static int err_gemm_common(const void* x, const void* w, const void* out_actual, const void* bias,
                           int64_t tokens, int64_t cout, int64_t cin, int dt, int n_cand, int64_t x_stride,
                           int64_t w_stride, float* partial, float* loss_acc, void* stream, const char* who) {
  int rc = gemm_check(x, w, tokens, cout, cin, dt, who);
  if (rc != MOQ_OK) return rc;
  if (out_actual == nullptr || partial == nullptr || loss_acc == nullptr) {
    set_error("%s: out_actual / partial / loss_acc must not be NULL", who);
    return MOQ_ERR_INVALID;
  }
  if ((reinterpret_cast<uintptr_t>(out_actual) & 7u) != 0) {
    set_error("%s: out_actual must be 8-byte aligned", who);
    return MOQ_ERR_UNSUPPORTED;
  }
  if (n_cand < 1 || n_cand > 65535 || x_stride < 0 || w_stride < 0 || ((x_stride | w_stride) & 7) != 0) {
    set_error("%s: n_cand must be in [1, 65535] and candidate strides multiples of 8 elements", who);
    return MOQ_ERR_INVALID;
  }
  if (tokens == 0) return MOQ_OK;
  const int64_t nblk = dt == MOQ_F32
      ? moq_f32_launch_err(x, w, out_actual, bias, partial, tokens, cout, cin, n_cand, x_stride, w_stride, stream)
      : launch_gemm<0>(x, w, out_actual, bias, nullptr, partial, tokens, cout, cin, dt, n_cand, x_stride, w_stride, stream);
  if (nblk < 0) return (int)nblk;
  hipLaunchKernelGGL(err_finalize_kernel, dim3((unsigned)n_cand), dim3(256), 0, S(stream), partial, (int)nblk,
                     1.0 / ((double)tokens * (double)cout), loss_acc);
  return check_launch(who);
}

extern "C" int moq_awq_err_gemm(const void* x, const void* w, const void* out_actual, const void* bias,
                                int64_t tokens, int64_t cout, int64_t cin, int dt, float* partial,
                                float* loss_acc, void* stream) {
  return err_gemm_common(x, w, out_actual, bias, tokens, cout, cin, dt, 1, 0, 0, partial, loss_acc, stream,
                         "moq_awq_err_gemm");
}

extern "C" int moq_awq_err_gemm_multi(const void* x, const void* w, const void* out_actual, const void* bias,
                                      int64_t tokens, int64_t cout, int64_t cin, int dt, int n_cand,
                                      int64_t x_stride, int64_t w_stride, float* partial, float* loss_acc,
                                      void* stream) {
  return err_gemm_common(x, w, out_actual, bias, tokens, cout, cin, dt, n_cand, x_stride, w_stride, partial,
                         loss_acc, stream, "moq_awq_err_gemm_multi");
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
  const int64_t nblk = dt == MOQ_F32 ? moq_f32_launch_store(x, w, bias, out, tokens, cout, cin, stream)
                                     : launch_gemm<1>(x, w, nullptr, bias, out, nullptr, tokens, cout, cin, dt, 1, 0, 0, stream);
  if (nblk < 0) return (int)nblk;
  return check_launch("moq_gemm_nt");
}

// out[i, j] = out[j, i] for i > j... the lower triangle of an fp32 [n, n] matrix becomes the mirror of the upper one
// (in the kernel's tile orientation `upper` = column-tile >= row-tile of the [t, n] = [row, col] view)
__global__ __launch_bounds__(256) void symmetrize_kernel(float* __restrict__ h, int64_t n) {
  __shared__ float tile[64][65];
  const int64_t bi = blockIdx.y, bj = blockIdx.x;
  if (bj < bi) return;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int64_t r0 = bi * 64, c0 = bj * 64;
  for (int i = ty; i < 64; i += 4) {
    const int64_t r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < n && c < n) ? h[r * n + c] : 0.0f;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int64_t r = c0 + i, c = r0 + tx;  // mirrored position of tile element [tx][i]
    if (r < n && c < n && r > c) h[r * n + c] = tile[tx][i];
  }
}

extern "C" int moq_symmetrize(float* h, int64_t n, void* stream) {
  if (h == nullptr || n < 0) {
    set_error("moq_symmetrize: bad arguments");
    return MOQ_ERR_INVALID;
  }
  if (n == 0) return MOQ_OK;
  const unsigned nb = (unsigned)((n + 63) / 64);
  if (nb > 65535) {
    set_error("moq_symmetrize: matrix too large");
    return MOQ_ERR_UNSUPPORTED;
  }
  hipLaunchKernelGGL(symmetrize_kernel, dim3(nb, nb), dim3(256), 0, S(stream), h, n);
  return check_launch("moq_symmetrize");
}

extern "C" int moq_hessian_accum(const void* xt, int64_t cin, int64_t tokens, int dt, float* hessian, float decay,
                                 float scale, int upper_only, void* stream) {
  int rc = gemm_check(xt, xt, cin, cin, tokens, dt, "moq_hessian_accum");
  if (rc != MOQ_OK) return rc;
  if (dt == MOQ_F32) {
    set_error("moq_hessian_accum: the Gram / Hessian accumulation takes bf16 / f16 activations");
    return MOQ_ERR_UNSUPPORTED;
  }
  if (hessian == nullptr || (reinterpret_cast<uintptr_t>(hessian) & 15u) != 0) {
    set_error("moq_hessian_accum: hessian must be a non-NULL 16-byte aligned pointer");
    return MOQ_ERR_INVALID;
  }
  const int64_t nblk = launch_gemm<2>(xt, xt, nullptr, nullptr, hessian, nullptr, cin, cin, tokens, dt, 1, 0, 0, stream,
                                      decay, scale, upper_only ? 1 : 0);
  if (nblk < 0) return (int)nblk;
  return check_launch("moq_hessian_accum");
}

extern "C" int moq_awq_quadform(const void* a, const void* b, const float* ref, int64_t rows, int64_t cols, int64_t k,
                                int dt, float* partial, float* loss_acc, double inv_count, void* stream) {
  int rc = gemm_check(a, b, rows, cols, k, dt, "moq_awq_quadform");
  if (rc != MOQ_OK) return rc;
  if (ref == nullptr || partial == nullptr || loss_acc == nullptr || (reinterpret_cast<uintptr_t>(ref) & 15u) != 0) {
    set_error("moq_awq_quadform: ref (16-byte aligned) / partial / loss_acc must not be NULL");
    return MOQ_ERR_INVALID;
  }
  if (rows == 0) return MOQ_OK;
  const int64_t nblk = dt == MOQ_F32 ? moq_f32_launch_dot(a, b, ref, partial, rows, cols, k, stream)
                                     : launch_gemm<3>(a, b, ref, nullptr, nullptr, partial, rows, cols, k, dt, 1, 0, 0, stream);
  if (nblk < 0) return (int)nblk;
  hipLaunchKernelGGL(err_finalize_kernel, dim3(1), dim3(256), 0, S(stream), partial, (int)nblk, inv_count, loss_acc);
  return check_launch("moq_awq_quadform");
}
