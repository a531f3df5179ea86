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

// Tile geometries.  Waves form a WN x WT grid; each wave owns NI x NJ MFMA tiles of 32 x 32.
//   GEO 0: 128 x 128, 4 waves, ONE 32 KiB LDS stage, ~4 workgroups per CU hide each other's DMA waits
//   GEO 1: 128 x 128, 4 waves, two stages; all fragments of a K-step are read before the next DMA is issued
//          (hipcc orders an LDS-DMA before every later ds_read with vmcnt(0), so reads must come first)
//   GEO 2: 256 x 256, 8 waves, two 64 KiB stages, ONE workgroup per CU; the DMA is issued from inline asm right
//          after the barrier (the compiler then neither sees a pending LDS write nor drains it early) and lands
//          under the K-step's 32 MFMAs per wave; half the L2->LDS bytes per flop of the 128 x 128 tile
template <int GEO> struct Geo;
template <> struct Geo<0> { static constexpr int TILE = 128, WAVES = 4, WN = 2, NI = 2, NJ = 2, STAGES = 1; };
template <> struct Geo<1> { static constexpr int TILE = 128, WAVES = 4, WN = 2, NI = 2, NJ = 2, STAGES = 2; };
template <> struct Geo<2> { static constexpr int TILE = 256, WAVES = 8, WN = 2, NI = 4, NJ = 2, STAGES = 2; };
//   GEO 3: 256 x 256, 8 waves, K-step 32 with FOUR 32 KiB stages: three tiles (96 KiB) stay in flight and the wait
//          before a step is a counted vmcnt(8) (never 0 in steady state) -- the GEO 2 loop keeps one 64 KiB tile in
//          flight and drains the queue every step, which parks its waves ~35 % of the time (SQ_WAIT_ANY)
template <> struct Geo<3> { static constexpr int TILE = 256, WAVES = 8, WN = 2, NI = 4, NJ = 2, STAGES = 4; };
//   GEO 4: GEO 2 with the K-loop rotated by one sub-step: the MFMAs of the LAST sub-step of tile k run after the barrier
//          that opens tile k + 1, underneath that tile's first fragment reads -- the matrix cores no longer idle
//          through "wait for the DMA, barrier, first ds_reads" at every K-step boundary
template <> struct Geo<4> { static constexpr int TILE = 256, WAVES = 8, WN = 2, NI = 4, NJ = 2, STAGES = 2; };
//   GEO 5 (experiment): the GEO 4 loop with FOUR waves of 128 x 128 (4 x 4 MFMA tiles, 256 accumulator registers, one
//          wave per SIMD): 8 fragment reads per 16 MFMAs instead of 6 per 8 -- a third less LDS read traffic
template <> struct Geo<5> { static constexpr int TILE = 256, WAVES = 4, WN = 2, NI = 4, NJ = 4, STAGES = 2; };
//   GEO 7 (experiment): GEO 4 with the next tile's eight LDS-DMA pieces issued ONE AT A TIME after every second MFMA of
//          the first two sub-steps (pinned with sched_barrier) instead of as one block of eight between two MFMA groups
template <> struct Geo<7> { static constexpr int TILE = 256, WAVES = 8, WN = 2, NI = 4, NJ = 2, STAGES = 2; };
//   GEO 10 (default): the GEO 4 loop as ONE pinned stream in which every MFMA is followed by one memory instruction --
//          a fragment read of the NEXT sub-step or an LDS-DMA piece of the next tile (32 MFMAs : 24 reads + 8 pieces per
//          wave and K-tile = 1 : 1, the recipe of the hand-scheduled kernels); the pieces go out in the first two
//          sub-steps so that they have two sub-steps of lead before the tile-boundary wait
template <> struct Geo<10> { static constexpr int TILE = 256, WAVES = 8, WN = 2, NI = 4, NJ = 2, STAGES = 2; };
//   GEO 8 / 9 (diagnostics, wrong results by construction -- timing only, tools/gemm_bench.py): the GEO 4 loop with ONLY
//          its LDS-DMA traffic (8: no fragment reads, no MFMAs) or ONLY its compute (9: no DMA after the prologue)
template <> struct Geo<8> { static constexpr int TILE = 256, WAVES = 8, WN = 2, NI = 4, NJ = 2, STAGES = 2; };
template <> struct Geo<9> { static constexpr int TILE = 256, WAVES = 8, WN = 2, NI = 4, NJ = 2, STAGES = 2; };
//   GEO 17 / 18 (round 3, correct results, 5-8 % SLOWER than GEO 10: profiles/r03_gemm_geo_regprefetch.md): the GEO 10 stream
//          with the operands going HBM/L2 -> VGPRs -> LDS (buffer_load_dwordx4 + ds_write_b128) instead of the LDS-DMA
template <> struct Geo<17> { static constexpr int TILE = 256, WAVES = 8, WN = 2, NI = 4, NJ = 2, STAGES = 2; };
template <> struct Geo<18> { static constexpr int TILE = 256, WAVES = 8, WN = 2, NI = 4, NJ = 2, STAGES = 2; };
//   GEO 20 (round 3): the GEO 10 stream with FOUR waves of 128 x 128 (4 x 4 MFMA tiles, 256 accumulator registers, one
//          wave per SIMD) -- the geometry of the library's MT256x256x64 kernel (WG32_8_1, MIWT8_8 in its name): 16 MFMAs per
//          sub-step against 8 fragment reads (GEO 10: 8 against 6), 16 LDS-DMA pieces per wave and K-tile in the first two
//          sub-steps, one after every second MFMA.  GEO 5 was this geometry with the block-issue loop and the branchy
//          pieces of round 1
template <> struct Geo<20> { static constexpr int TILE = 256, WAVES = 4, WN = 2, NI = 4, NJ = 4, STAGES = 2; };
//   GEO 21 (round 3): GEO 20 with ALL FOUR sub-steps' fragments of a K-tile resident in registers (128 VGPRs beside the 256
//          accumulators): a stage is read out during the first quarter of its tile's MFMAs and handed back to the LDS-DMA
//          right away, so the pieces of tile i + 2 are issued a quarter into tile i and waited for three quarters into tile
//          i + 1 -- 1.0-1.5 K-tiles of lead instead of 0.5-1.0 with the same two 64 KiB stages.  Why: PMC of the library
//          kernel on the same shape shows its waves parked (s_waitcnt / barrier) 6 % of the time, GEO 10's 35 %
//          (profiles/r03_gemm_geo_regprefetch.md)
template <> struct Geo<21> { static constexpr int TILE = 256, WAVES = 4, WN = 2, NI = 4, NJ = 4, STAGES = 2; };
//   GEO 22 (round 3): GEO 21 with the sixteen pieces of a tile spread EVENLY over a whole K-tile of MFMAs (one after every
//          fourth MFMA, from a quarter into tile i to a quarter into tile i + 1) instead of one after every second MFMA of
//          two sub-steps: a request stream without bursts
template <> struct Geo<22> { static constexpr int TILE = 256, WAVES = 4, WN = 2, NI = 4, NJ = 4, STAGES = 2; };
template <int GEO> constexpr int row_bytes() { return GEO == 3 ? 64 : kRowBytes; }
template <int GEO> constexpr int tile_bytes() { return Geo<GEO>::TILE * row_bytes<GEO>(); }
template <int GEO> constexpr int stage_bytes() { return 2 * tile_bytes<GEO>(); }
template <int GEO> constexpr int lds_bytes() { return Geo<GEO>::STAGES * stage_bytes<GEO>(); }

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

// one K-step (64 k) of a wave's NI x NJ tiles from the stage at `stage`.  Fragments are double-buffered in
// registers: the reads of sub-step ks+1 are in flight while the MFMAs of sub-step ks issue.  `after_first`
// runs once the first sub-step's reads have been issued (GEO 2 issues the next tile's DMA there, so its address
// arithmetic hides under the LDS latency instead of delaying the first MFMA after the barrier).
template <int DT, int GEO, class F>
__device__ __forceinline__ void k_step(const uint8_t* stage, int wn, int wt, int fr, int fh,
                                       f32x16_t (&acc)[Geo<GEO>::NI][Geo<GEO>::NJ], F&& after_first) {
  constexpr int NI = Geo<GEO>::NI, NJ = Geo<GEO>::NJ;
  const uint8_t* la = stage + (wn * NI * 32) * kRowBytes;
  const uint8_t* lb = stage + tile_bytes<GEO>() + (wt * NJ * 32) * kRowBytes;
  Pack16 a[2][NI], b[2][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i) a[0][i] = read_frag(la, i * 32 + fr, fh);
#pragma unroll
  for (int j = 0; j < NJ; ++j) b[0][j] = read_frag(lb, j * 32 + fr, fh);
  after_first();
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int cur = ks & 1, nxt = cur ^ 1;
    if (ks < 3) {
      const int c = (ks + 1) * 2 + fh;
#pragma unroll
      for (int i = 0; i < NI; ++i) a[nxt][i] = read_frag(la, i * 32 + fr, c);
#pragma unroll
      for (int j = 0; j < NJ; ++j) b[nxt][j] = read_frag(lb, j * 32 + fr, c);
    }
#ifdef MOQ_GEMM_SETPRIO
    __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = mfma32<DT>(a[cur][i], b[cur][j], acc[i][j]);
#ifdef MOQ_GEMM_SETPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
#ifndef MOQ_GEMM_NOPIN
    // pin the issue order hipcc would otherwise collapse: the next sub-step's reads go out first, then this
    // sub-step's MFMAs run while they are in flight (mask 0x100 = DS read, 0x008 = MFMA)
    if (ks < 3) __builtin_amdgcn_sched_group_barrier(0x100, NI + NJ, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, NI * NJ, 0);
#endif
  }
}

// ---- GEO 3: 64-byte LDS rows (32 k).  Chunk c (0..3) of row r sits at position c ^ ((r >> 2) & 3): the 16 rows of a
// ds_read_b128 service group then cover 16 distinct 16-byte slots of the 256-byte bank row (conflict-free).
// One wave-instruction of the DMA moves 16 rows x 64 B: lane l fills (row l >> 2, position l & 3).
__device__ __forceinline__ void stage_tile3(const TileDesc& desc, uint8_t* lds_tile, int64_t ld_bytes, int k0, int K,
                                            int wave, int lane) {
  constexpr int PER_WAVE = 256 / 16 / 8;  // 2
  const int r_local = lane >> 2, pos = lane & 3;
  const i32x4_t rs = desc.words;
#pragma unroll
  for (int j = 0; j < PER_WAVE; ++j) {
    const int rbase = (wave * PER_WAVE + j) * 16;
    const int r = rbase + r_local;
    const int c = pos ^ ((r >> 2) & 3);
    const int k = k0 + c * 8;
    const int voff = k < K ? (int)(r * ld_bytes + k * 2) : 0x7FFFFFF0;
    uint8_t* dst = lds_tile + rbase * 64;
    const uint32_t m0v = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_u8_t)dst);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                 :: "s"(m0v), "v"(voff), "s"(rs) : "memory");
  }
}
__device__ __forceinline__ Pack16 read_frag3(const uint8_t* lds_tile, int r, int c) {
  return *reinterpret_cast<const Pack16*>(lds_tile + r * 64 + ((c ^ ((r >> 2) & 3)) << 4));
}
// one K-step of 32 k (two MFMA sub-steps) from the stage at `stage`
template <int DT, class F>
__device__ __forceinline__ void k_step3(const uint8_t* stage, int wn, int wt, int fr, int fh,
                                        f32x16_t (&acc)[4][2], F&& after_first) {
  constexpr int NI = 4, NJ = 2;
  const uint8_t* la = stage + (wn * NI * 32) * 64;
  const uint8_t* lb = stage + tile_bytes<3>() + (wt * NJ * 32) * 64;
  Pack16 a[2][NI], b[2][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i) a[0][i] = read_frag3(la, i * 32 + fr, fh);
#pragma unroll
  for (int j = 0; j < NJ; ++j) b[0][j] = read_frag3(lb, j * 32 + fr, fh);
  after_first();
#pragma unroll
  for (int i = 0; i < NI; ++i) a[1][i] = read_frag3(la, i * 32 + fr, 2 + fh);
#pragma unroll
  for (int j = 0; j < NJ; ++j) b[1][j] = read_frag3(lb, j * 32 + fr, 2 + fh);
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = mfma32<DT>(a[ks][i], b[ks][j], acc[i][j]);
  }
}

// one LDS-DMA piece: rows rbase .. rbase + 7 of an operand tile for the K-step starting at k0 (stage_tile's loop body)
__device__ __forceinline__ void stage_piece(const TileDesc& desc, uint8_t* lds_tile, int64_t ld_bytes, int k0, int K,
                                            int rbase, int lane) {
  const int r = rbase + (lane >> 3), pos = lane & 7;
  const int c = pos ^ ((r >> 1) & 7);
  const int k = k0 + c * 8;
  const int voff = k < K ? (int)(r * ld_bytes + k * 2) : 0x7FFFFFF0;
  const i32x4_t rs = desc.words;
  uint8_t* dst = lds_tile + rbase * kRowBytes;
  const uint32_t m0v = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_u8_t)dst);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
               :: "s"(m0v), "v"(voff), "s"(rs) : "memory");
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
__global__ __launch_bounds__(Geo<GEO>::WAVES * 64, GEO == 0 ? 4 : (GEO == 5 || GEO == 20 || GEO == 21 || GEO == 22 ? 1 : 2))
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
  tile_of_block(blockIdx.x, tiles_t, tiles_n, MODE == 2 ? (upper_only >> 1) : (upper_only & 0xFFFF), tn, tt);  // group size travels in upper_only
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

  if constexpr (GEO == 21 || GEO == 22) {
    static_assert(NI == 4 && NJ == 4 && Geo<GEO>::WAVES == 4, "GEO 21 / 22: four waves of 4 x 4 MFMA tiles");
    const uint8_t* la0 = smem + (wn * NI * 32) * kRowBytes;
    const uint8_t* lb0 = smem + TB + (wt * NJ * 32) * kRowBytes;
    Pack16 a[4][NI], b[4][NJ];  // one register buffer per sub-step of a K-tile
    constexpr int PP = 8;
    const uint32_t lds_wave = __builtin_amdgcn_readfirstlane(
        (uint32_t)(uintptr_t)(lds_u8_t)smem + (uint32_t)(wave * PP * 8 * kRowBytes));
    const bool k_ragged = (K & (kBK - 1)) != 0;
    int voff[PP], voff_tail[PP];
#pragma unroll
    for (int j = 0; j < PP; ++j) {
      const int r = (wave * PP + j) * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((r >> 1) & 7);
      voff[j] = (int)(r * ld_bytes + c * 16);
      voff_tail[j] = (nk - 1) * kBK + c * 8 < K ? voff[j] : 0x7FFFFFF0;
    }
    const i32x4_t rsw = rs_w.words, rsx = rs_x.words;
    auto piece = [&](int sn, int k0, auto P, bool live = true) {
      constexpr int p = decltype(P)::value, j = p & (PP - 1);
      const uint32_t m0v = lds_wave + (uint32_t)(sn * SB + (p < PP ? 0 : TB) + j * 8 * kRowBytes);
      const int koff = k0 * 2;
      const int vfull = voff[j], vtail = voff_tail[j];
      const int vo = (k_ragged && k0 + kBK > K) ? vtail : vfull;
      i32x4_t rr = p < PP ? rsw : rsx;
      rr.z = live ? rr.z : 0;
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                   :: "s"(m0v), "v"(vo), "s"(rr), "s"(koff) : "memory");
    };
    auto pieces8 = [&](int sn, int k0, auto P0, bool live) {
      constexpr int p0 = decltype(P0)::value;
      piece(sn, k0, IC<p0 + 0>{}, live); piece(sn, k0, IC<p0 + 1>{}, live); piece(sn, k0, IC<p0 + 2>{}, live);
      piece(sn, k0, IC<p0 + 3>{}, live); piece(sn, k0, IC<p0 + 4>{}, live); piece(sn, k0, IC<p0 + 5>{}, live);
      piece(sn, k0, IC<p0 + 6>{}, live); piece(sn, k0, IC<p0 + 7>{}, live);
    };
    // one fragment read: sub-step KS of the tile in stage offset `st`, fragment f (0..3: a[ks][f], 4..7: b[ks][f - 4])
    auto frag = [&](auto KS, auto F, int st) {
      constexpr int ks = decltype(KS)::value, f = decltype(F)::value;
      const int c = ks * 2 + fh;
      if constexpr (f < 4) a[ks][f] = read_frag(la0 + st, f * 32 + fr, c);
      else b[ks][f - 4] = read_frag(lb0 + st, (f - 4) * 32 + fr, c);
    };
    // sixteen MFMAs of sub-step buffer BUF; after MFMA n the caller's `after(n)` issues that slot's memory instruction
    auto group = [&](auto BUF, auto after) {
      constexpr int buf = decltype(BUF)::value;
      auto one = [&](auto NC) {
        constexpr int n = decltype(NC)::value;
        acc[n >> 2][n & 3] = mfma32<DT>(a[buf][n >> 2], b[buf][n & 3], acc[n >> 2][n & 3]);
        __builtin_amdgcn_sched_barrier(0);
        after(NC);
        __builtin_amdgcn_sched_barrier(0);
      };
      one(IC<0>{}); one(IC<1>{}); one(IC<2>{}); one(IC<3>{}); one(IC<4>{}); one(IC<5>{}); one(IC<6>{}); one(IC<7>{});
      one(IC<8>{}); one(IC<9>{}); one(IC<10>{}); one(IC<11>{}); one(IC<12>{}); one(IC<13>{}); one(IC<14>{}); one(IC<15>{});
    };
    // prologue: tiles 0 and 1 on their way, tile 0 landed, its first two sub-steps in registers
    pieces8(0, 0, IC<0>{}, true);
    pieces8(0, 0, IC<8>{}, true);
    pieces8(1, kBK, IC<0>{}, nk > 1);
    pieces8(1, kBK, IC<8>{}, nk > 1);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int f = 0; f < 8; ++f) {
      const int c0 = 0 * 2 + fh, c1 = 1 * 2 + fh;
      if (f < 4) { a[0][f] = read_frag(la0, f * 32 + fr, c0); a[1][f] = read_frag(la0, f * 32 + fr, c1); }
      else { b[0][f - 4] = read_frag(lb0, (f - 4) * 32 + fr, c0); b[1][f - 4] = read_frag(lb0, (f - 4) * 32 + fr, c1); }
    }
    for (int kt = 0; kt < nk; ++kt) {
      const int so = (kt & 1) * SB, sno = ((kt + 1) & 1) * SB;
      const int k2 = (kt + 2) * kBK;
      const bool live2 = kt + 2 < nk;
      // first quarter: sub-step 0's MFMAs; the other two sub-steps of THIS tile come out of its stage
      // (GEO 22: the last four pieces of tile kt + 1 -- whose stage was handed over a K-tile ago -- ride along)
      group(IC<0>{}, [&](auto NC) {
        constexpr int n = decltype(NC)::value;
        if constexpr (n < 8) frag(IC<2>{}, IC<n>{}, so);
        else frag(IC<3>{}, IC<n - 8>{}, so);
        if constexpr (GEO == 22 && (n & 3) == 3) {
          if (kt > 0) piece((kt + 1) & 1, (kt + 1) * kBK, IC<12 + (n >> 2)>{}, kt + 1 < nk);
        }
      });
      // the stage of tile kt is read out (own reads done; the barrier makes it everyone's): it goes back to the DMA
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_barrier" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (GEO == 21) {
        // second and third quarter: the sixteen pieces of tile kt + 2 into that stage, one after every second MFMA
        group(IC<1>{}, [&](auto NC) {
          constexpr int n = decltype(NC)::value;
          if constexpr ((n & 1) != 0) piece(kt & 1, k2, IC<(n >> 1)>{}, live2);
        });
        group(IC<2>{}, [&](auto NC) {
          constexpr int n = decltype(NC)::value;
          if constexpr ((n & 1) != 0) piece(kt & 1, k2, IC<8 + (n >> 1)>{}, live2);
        });
        // tile kt + 1 (issued a K-tile ago) has landed once only the sixteen pieces above are still outstanding
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      } else {
        // one piece of tile kt + 2 after every fourth MFMA: four per quarter, the last four in the next tile's first quarter
        group(IC<1>{}, [&](auto NC) {
          constexpr int n = decltype(NC)::value;
          if constexpr ((n & 3) == 3) piece(kt & 1, k2, IC<(n >> 2)>{}, live2);
        });
        group(IC<2>{}, [&](auto NC) {
          constexpr int n = decltype(NC)::value;
          if constexpr ((n & 3) == 3) piece(kt & 1, k2, IC<4 + (n >> 2)>{}, live2);
        });
        // tile kt + 1: its last four pieces went out in this tile's first quarter; the eight of tile kt + 2 above are newer
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_barrier" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      // last quarter: sub-step 3's MFMAs; the first two sub-steps of the NEXT tile come out of the other stage
      group(IC<3>{}, [&](auto NC) {
        constexpr int n = decltype(NC)::value;
        if constexpr (n < 8) frag(IC<0>{}, IC<n>{}, sno);
        else frag(IC<1>{}, IC<n - 8>{}, sno);
        if constexpr (GEO == 22 && (n & 3) == 3) piece(kt & 1, k2, IC<8 + (n >> 2)>{}, live2);
      });
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // dead pieces / reads of the tail before the epilogue reuses the LDS
    __builtin_amdgcn_sched_barrier(0);
  } else
  if constexpr (GEO == 20) {
    static_assert(NI == 4 && NJ == 4 && Geo<GEO>::WAVES == 4, "GEO 20: four waves of 4 x 4 MFMA tiles");
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
    // LDS-DMA pieces as in GEO 10 (branch-free, three instructions), EIGHT per operand and wave: piece p (0..15) = 8 rows
    // of W (p < 8) or of x (p >= 8)
    constexpr int PP = 8;
    const uint32_t lds_wave = __builtin_amdgcn_readfirstlane(
        (uint32_t)(uintptr_t)(lds_u8_t)smem + (uint32_t)(wave * PP * 8 * kRowBytes));
    const bool k_ragged = (K & (kBK - 1)) != 0;
    int voff[PP], voff_tail[PP];
#pragma unroll
    for (int j = 0; j < PP; ++j) {
      const int r = (wave * PP + j) * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((r >> 1) & 7);
      voff[j] = (int)(r * ld_bytes + c * 16);
      voff_tail[j] = (nk - 1) * kBK + c * 8 < K ? voff[j] : 0x7FFFFFF0;
    }
    const i32x4_t rsw = rs_w.words, rsx = rs_x.words;
    auto piece = [&](int sn, int k0, auto P, bool live = true) {
      constexpr int p = decltype(P)::value, j = p & (PP - 1);
      const uint32_t m0v = lds_wave + (uint32_t)(sn * SB + (p < PP ? 0 : TB) + j * 8 * kRowBytes);
      const int koff = k0 * 2;
      const int vfull = voff[j], vtail = voff_tail[j];
      const int vo = (k_ragged && k0 + kBK > K) ? vtail : vfull;
      i32x4_t rr = p < PP ? rsw : rsx;
      rr.z = live ? rr.z : 0;
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                   :: "s"(m0v), "v"(vo), "s"(rr), "s"(koff) : "memory");
    };
    // sixteen MFMAs of register buffer BUF, each of the first eight followed by one fragment read of sub-step KS into the
    // other buffer and, when P0 >= 0, every second one by one LDS-DMA piece (pieces p0 .. p0 + 7)
    auto group = [&](auto BUF, auto KS, bool on, auto P0, int so, int sn, int k0) {
      constexpr int buf = decltype(BUF)::value, ks = decltype(KS)::value, nb = buf ^ 1, p0 = decltype(P0)::value;
      const int c = ks * 2 + fh;
      auto one = [&](auto NC) {
        constexpr int n = decltype(NC)::value;
        acc[n >> 2][n & 3] = mfma32<DT>(a[buf][n >> 2], b[buf][n & 3], acc[n >> 2][n & 3]);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (n < 4) a[nb][n] = read_frag(la0 + so, n * 32 + fr, c);
        else if constexpr (n < 8) b[nb][n - 4] = read_frag(lb0 + so, (n - 4) * 32 + fr, c);
        if constexpr ((n & 1) != 0 && p0 >= 0) piece(sn, k0, IC<(p0 >= 0 ? p0 : 0) + (n >> 1)>{}, on);
        __builtin_amdgcn_sched_barrier(0);
      };
      one(IC<0>{}); one(IC<1>{}); one(IC<2>{}); one(IC<3>{}); one(IC<4>{}); one(IC<5>{}); one(IC<6>{}); one(IC<7>{});
      one(IC<8>{}); one(IC<9>{}); one(IC<10>{}); one(IC<11>{}); one(IC<12>{}); one(IC<13>{}); one(IC<14>{}); one(IC<15>{});
    };
    auto pieces8 = [&](int sn, int k0, auto P0) {
      constexpr int p0 = decltype(P0)::value;
      piece(sn, k0, IC<p0 + 0>{}); piece(sn, k0, IC<p0 + 1>{}); piece(sn, k0, IC<p0 + 2>{}); piece(sn, k0, IC<p0 + 3>{});
      piece(sn, k0, IC<p0 + 4>{}); piece(sn, k0, IC<p0 + 5>{}); piece(sn, k0, IC<p0 + 6>{}); piece(sn, k0, IC<p0 + 7>{});
    };
    pieces8(0, 0, IC<0>{});
    pieces8(0, 0, IC<8>{});
    for (int kt = 0; kt < nk; ++kt) {
      const int so = (kt & 1) * SB, sn = (kt + 1) & 1, k0 = (kt + 1) * kBK;
      const bool more = kt + 1 < nk;
      // tile kt landed (own pieces; the barrier makes it everyone's) and every fragment read of tile kt - 1 has completed
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __syncthreads();
      if (kt > 0) {
        group(IC<1>{}, IC<0>{}, more, IC<0>{}, so, sn, k0);  // last sub-step of tile kt - 1 under the first reads of tile kt
      } else {
        read_sub(0, so, 0);
        if (more) pieces8(sn, k0, IC<0>{});
      }
      group(IC<0>{}, IC<1>{}, more, IC<8>{}, so, sn, k0);
      group(IC<1>{}, IC<2>{}, false, IC<-1>{}, so, sn, k0);
      group(IC<0>{}, IC<3>{}, false, IC<-1>{}, so, sn, k0);
    }
    mma_sub(1);  // last sub-step of the last tile
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // "dead" pieces of the last tile before the epilogue reuses the LDS
  } else
  if constexpr (GEO == 4 || GEO == 5 || GEO == 7 || GEO == 8 || GEO == 9 || GEO == 10 || GEO == 17 || GEO == 18) {
    const uint8_t* la0 = smem + (wn * NI * 32) * kRowBytes;
    const uint8_t* lb0 = smem + TB + (wt * NJ * 32) * kRowBytes;
    Pack16 a[2][NI], b[2][NJ];
    auto read_sub = [&](int buf, int stage_off, int ks) {
      if constexpr (GEO == 8) return;
      const int c = ks * 2 + fh;
#pragma unroll
      for (int i = 0; i < NI; ++i) a[buf][i] = read_frag(la0 + stage_off, i * 32 + fr, c);
#pragma unroll
      for (int j = 0; j < NJ; ++j) b[buf][j] = read_frag(lb0 + stage_off, j * 32 + fr, c);
    };
    auto mma_sub = [&](int buf) {
      if constexpr (GEO == 8) return;
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
    constexpr bool kFourPerOperand = Geo<GEO>::WAVES == 8;
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
    // (piece indices must be compile-time constants: they select the lane-offset register)
    auto mma_sub_staged = [&](int buf, int sn, int k0, auto P0, bool on) {
      constexpr int p0 = decltype(P0)::value;
      static_assert(NI == 4, "one piece after every second MFMA");
      acc[0][0] = mfma32<DT>(a[buf][0], b[buf][0], acc[0][0]);
      acc[0][1] = mfma32<DT>(a[buf][0], b[buf][1], acc[0][1]);
      __builtin_amdgcn_sched_barrier(0);
      piece(sn, k0, IC<p0 + 0>{}, on);
      __builtin_amdgcn_sched_barrier(0);
      acc[1][0] = mfma32<DT>(a[buf][1], b[buf][0], acc[1][0]);
      acc[1][1] = mfma32<DT>(a[buf][1], b[buf][1], acc[1][1]);
      __builtin_amdgcn_sched_barrier(0);
      piece(sn, k0, IC<p0 + 1>{}, on);
      __builtin_amdgcn_sched_barrier(0);
      acc[2][0] = mfma32<DT>(a[buf][2], b[buf][0], acc[2][0]);
      acc[2][1] = mfma32<DT>(a[buf][2], b[buf][1], acc[2][1]);
      __builtin_amdgcn_sched_barrier(0);
      piece(sn, k0, IC<p0 + 2>{}, on);
      __builtin_amdgcn_sched_barrier(0);
      acc[3][0] = mfma32<DT>(a[buf][3], b[buf][0], acc[3][0]);
      acc[3][1] = mfma32<DT>(a[buf][3], b[buf][1], acc[3][1]);
      __builtin_amdgcn_sched_barrier(0);
      piece(sn, k0, IC<p0 + 3>{}, on);
      __builtin_amdgcn_sched_barrier(0);
    };
    if constexpr (GEO == 17 || GEO == 18) {
      // ---- GEO 17 / 18 (round 3 experiment, correct results): the GEO 10 stream with the operands going HBM/L2 -> VGPRs -> LDS
      // (buffer_load_dwordx4 + ds_write_b128) instead of through the LDS-DMA.  Why: an LDS-DMA piece costs 60-185
      // cycles of issue among MFMAs (MI355X_MICROARCH.md, "LDS-DMA piece issue cost"), eight per wave and K-tile
      // against the tile's 1024 cycles of MFMA issue per wave; and its landing must be waited for with vmcnt(0) at
      // the tile boundary, right before the barrier.  Here the loads of tile kt + 2 are issued in the last sub-step of
      // tile kt into 32 registers and are waited for (by the compiler's own vmcnt bookkeeping) three sub-steps later,
      // where tile kt + 1's third sub-step stores them into the stage tile kt just left; the tile boundary waits for
      // LDS traffic only.  Same lane -> row / chunk mapping as the DMA pieces, so the fragment reads are unchanged.
      typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
      u32x4_t R[8];
      const __amdgpu_buffer_rsrc_t bw = rs_w.rsrc, bx = rs_x.rsrc;
      const int k_last = (nk - 1) * kBK;
      auto gload = [&](int k0, auto P) {
        constexpr int p = decltype(P)::value, j = p & 3;
        const int kk = k0 < k_last ? k0 : k_last;  // past the end: the last tile again (stored into an idle stage)
        const int vfull = voff[j], vtail = voff_tail[j];  // (locals: a select between array elements goes through scratch)
        const int vo = (k_ragged && kk + kBK > K) ? vtail : vfull;
        R[p] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(p < 4 ? bw : bx, vo, kk * 2, 0));
      };
      uint8_t* const lds_lane = smem + wave * 4 * 8 * kRowBytes + lane * 16;
      auto lstore = [&](int sn, auto P) {
        constexpr int p = decltype(P)::value, j = p & 3;
        *reinterpret_cast<u32x4_t*>(lds_lane + sn * SB + (p < 4 ? 0 : TB) + j * 8 * kRowBytes) = R[p];
      };
      // eight MFMAs of register buffer BUF, each followed by one fragment read of sub-step KS into the other buffer
      // (six) and, after every second one, by two stores (MODE_ 1) or two loads (MODE_ 2) of pieces 2q, 2q + 1
      auto group = [&](auto BUF, auto KS, auto MODE_, int so, int sn, int k0) {
        constexpr int buf = decltype(BUF)::value, ks = decltype(KS)::value, nb = buf ^ 1, md = decltype(MODE_)::value;
        const int c = ks * 2 + fh;
        auto one = [&](auto NC) {
          constexpr int n = decltype(NC)::value;
          acc[n >> 1][n & 1] = mfma32<DT>(a[buf][n >> 1], b[buf][n & 1], acc[n >> 1][n & 1]);
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (ks >= 0 && n < 4) a[nb][n] = read_frag(la0 + so, n * 32 + fr, c);
          else if constexpr (ks >= 0 && n < 6) b[nb][n - 4] = read_frag(lb0 + so, (n - 4) * 32 + fr, c);
          if constexpr ((n & 1) != 0 && md == 1) { lstore(sn, IC<n - 1>{}); lstore(sn, IC<n>{}); }
          if constexpr ((n & 1) != 0 && md == 2) { gload(k0, IC<n - 1>{}); gload(k0, IC<n>{}); }
          if constexpr ((n & 1) != 0 && md == 3) {  // GEO 18: a piece is stored and its registers reloaded right away
            lstore(sn, IC<n - 1>{}); lstore(sn, IC<n>{});
            gload(k0, IC<n - 1>{}); gload(k0, IC<n>{});
          }
          __builtin_amdgcn_sched_barrier(0);
        };
        one(IC<0>{}); one(IC<1>{}); one(IC<2>{}); one(IC<3>{}); one(IC<4>{}); one(IC<5>{}); one(IC<6>{}); one(IC<7>{});
      };
      // prologue: tile 0 through the registers into stage 0, tile 1 on its way
      gload(0, IC<0>{}); gload(0, IC<1>{}); gload(0, IC<2>{}); gload(0, IC<3>{});
      gload(0, IC<4>{}); gload(0, IC<5>{}); gload(0, IC<6>{}); gload(0, IC<7>{});
      lstore(0, IC<0>{}); lstore(0, IC<1>{}); lstore(0, IC<2>{}); lstore(0, IC<3>{});
      lstore(0, IC<4>{}); lstore(0, IC<5>{}); lstore(0, IC<6>{}); lstore(0, IC<7>{});
      gload(kBK, IC<0>{}); gload(kBK, IC<1>{}); gload(kBK, IC<2>{}); gload(kBK, IC<3>{});
      gload(kBK, IC<4>{}); gload(kBK, IC<5>{}); gload(kBK, IC<6>{}); gload(kBK, IC<7>{});
      for (int kt = 0; kt < nk; ++kt) {
        const int so = (kt & 1) * SB, sn = (kt + 1) & 1;
        // every LDS store of tile kt and every fragment read of tile kt - 1 of this wave is done; the barrier makes
        // that true for the workgroup.  Loads of tile kt + 1 stay in flight across it.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (kt > 0) group(IC<1>{}, IC<0>{}, IC<0>{}, so, sn, 0);  // last sub-step of tile kt - 1 under the first reads
        else read_sub(0, so, 0);
        group(IC<0>{}, IC<1>{}, IC<0>{}, so, sn, 0);
        if constexpr (GEO == 17) {
          group(IC<1>{}, IC<2>{}, IC<1>{}, so, sn, 0);                 // tile kt + 1: registers -> stage sn
          group(IC<0>{}, IC<3>{}, IC<2>{}, so, sn, (kt + 2) * kBK);    // tile kt + 2: HBM / L2 -> registers
        } else {  // GEO 18: both in the last sub-step -- the loads get a whole K-tile of lead
          group(IC<1>{}, IC<2>{}, IC<0>{}, so, sn, 0);
          group(IC<0>{}, IC<3>{}, IC<3>{}, so, sn, (kt + 2) * kBK);
        }
      }
      mma_sub(1);  // last sub-step of the last tile
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the epilogue reuses the LDS
      __builtin_amdgcn_sched_barrier(0);
    } else {
    stage_tile<GEO, true>(rs_w, smem, ld_bytes, 0, K, wave, lane);
    stage_tile<GEO, true>(rs_x, smem + TB, ld_bytes, 0, K, wave, lane);
    // DIAGNOSTIC (MOQ_TUNE_GEMM_STAT = 1 .. 4, GEO 10, timing only: the loss is replaced by the statistic): s_memtime stamps of
    // wave 0 around the tile-boundary wait -- 1: ticks per K-tile, 2: ticks parked at the boundary (s_waitcnt + barrier) per
    // K-tile, 3: ticks between the issue of a tile's first LDS-DMA piece and the boundary wait that needs the tile (the lead),
    // 4: ticks from kernel start to the end of the K loop, 5: the same in s_memrealtime ticks (constant 100 MHz)
    const int stat_mode = (GEO == 10 && MODE == 0) ? ((upper_only >> 26) & 7) : 0;
    unsigned long long st_prev = 0, st_issue = 0, st_sum = 0, st_begin = 0;
    if (stat_mode) st_begin = stat_mode == 5 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime();
    for (int kt = 0; kt < nk; ++kt) {
      const int so = (kt & 1) * SB;
      // tile kt landed (own part; the barrier makes it everyone's) and every fragment read of tile kt - 1 has
      // completed -- its last sub-step sits in a[1] / b[1], not yet multiplied
      unsigned long long st_t1 = 0;
      if (stat_mode) st_t1 = __builtin_amdgcn_s_memtime();
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __syncthreads();
      if (stat_mode) {
        const unsigned long long st_t2 = __builtin_amdgcn_s_memtime();
        if (kt > 0) {
          if (stat_mode == 1) st_sum += st_t2 - st_prev;
          if (stat_mode == 2) st_sum += st_t2 - st_t1;
          if (stat_mode == 3) st_sum += st_t1 - st_issue;
        }
        st_prev = st_t2;
        st_issue = st_t2;  // GEO 10 issues the next tile's first piece within a few MFMAs of the boundary
      }
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
      if constexpr (GEO == 7) {
        const int sn = (kt + 1) & 1;
        const bool more = kt + 1 < nk;
        const int k0 = (kt + 1) * kBK;
        if (kt > 0) {
          mma_sub_staged(1, sn, k0, std::integral_constant<int, 0>{}, more);
        } else if (more) {
          piece(sn, k0, IC<0>{}); piece(sn, k0, IC<1>{}); piece(sn, k0, IC<2>{}); piece(sn, k0, IC<3>{});
        }
        read_sub(1, so, 1);
        mma_sub_staged(0, sn, k0, std::integral_constant<int, 4>{}, more);
        read_sub(0, so, 2);
        __builtin_amdgcn_sched_group_barrier(0x100, NI + NJ, 0);
        mma_sub(1);
        __builtin_amdgcn_sched_group_barrier(0x008, NI * NJ, 0);
        read_sub(1, so, 3);
        __builtin_amdgcn_sched_group_barrier(0x100, NI + NJ, 0);
        mma_sub(0);
        __builtin_amdgcn_sched_group_barrier(0x008, NI * NJ, 0);
        continue;
      }
      if (kt > 0) {
        mma_sub(1);  // last sub-step of tile kt - 1, under the reads above
        __builtin_amdgcn_sched_group_barrier(0x008, NI * NJ, 0);
      }
      if (GEO != 9 && kt + 1 < nk) {  // next tile's DMA, issued in the gaps of the MFMAs above
        const int sn = (kt + 1) & 1;
        if constexpr (kFourPerOperand) {
          const int k0 = (kt + 1) * kBK;
          piece(sn, k0, IC<0>{}); piece(sn, k0, IC<1>{}); piece(sn, k0, IC<2>{}); piece(sn, k0, IC<3>{});
          piece(sn, k0, IC<4>{}); piece(sn, k0, IC<5>{}); piece(sn, k0, IC<6>{}); piece(sn, k0, IC<7>{});
        } else {
          uint8_t* nxt = smem + sn * SB;
          stage_tile<GEO, true>(rs_w, nxt, ld_bytes, (kt + 1) * kBK, K, wave, lane);
          stage_tile<GEO, true>(rs_x, nxt + TB, ld_bytes, (kt + 1) * kBK, K, wave, lane);
        }
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
    if (stat_mode) {
      if (stat_mode == 4) st_sum = __builtin_amdgcn_s_memtime() - st_begin;
      if (stat_mode == 5) st_sum = __builtin_amdgcn_s_memrealtime() - st_begin;  // constant 100 MHz
      if (threadIdx.x == 0)
        partial[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = stat_mode >= 4 ? (float)st_sum : (float)st_sum / (float)(nk > 1 ? nk - 1 : 1);
      return;
    }
    }
  } else if constexpr (GEO == 3) {
    constexpr int BK3 = 32;
    const int nk3 = (K + BK3 - 1) / BK3;
    // prologue: three tiles in flight
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      if (p < nk3) {
        stage_tile3(rs_w, smem + p * SB, ld_bytes, p * BK3, K, wave, lane);
        stage_tile3(rs_x, smem + p * SB + TB, ld_bytes, p * BK3, K, wave, lane);
      }
    }
    for (int kt = 0; kt < nk3; ++kt) {
      // tile kt has landed once at most the DMA pieces of the (up to two) younger tiles are outstanding:
      // 4 pieces per tile per wave
      if (kt + 2 < nk3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (kt + 1 < nk3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();  // everyone's part of tile kt landed; everyone finished reading stage (kt - 1) % 4
      k_step3<DT>(smem + (kt & 3) * SB, wn, wt, fr, fh, acc, [&]() {
        if (kt + 3 < nk3) {
          uint8_t* nxt = smem + ((kt + 3) & 3) * SB;
          stage_tile3(rs_w, nxt, ld_bytes, (kt + 3) * BK3, K, wave, lane);
          stage_tile3(rs_x, nxt + TB, ld_bytes, (kt + 3) * BK3, K, wave, lane);
        }
      });
    }
  } else if constexpr (GEO == 2) {
    stage_tile<GEO, true>(rs_w, smem, ld_bytes, 0, K, wave, lane);
    stage_tile<GEO, true>(rs_x, smem + TB, ld_bytes, 0, K, wave, lane);
    for (int kt = 0; kt < nk; ++kt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's part of tile kt has landed
      __syncthreads();  // ... everyone's has, and everyone is done reading the other stage (K-step kt-1)
      k_step<DT, GEO>(smem + (kt & 1) * SB, wn, wt, fr, fh, acc, [&]() {
        if (kt + 1 < nk) {
          uint8_t* nxt = smem + ((kt + 1) & 1) * SB;
          stage_tile<GEO, true>(rs_w, nxt, ld_bytes, (kt + 1) * kBK, K, wave, lane);
          stage_tile<GEO, true>(rs_x, nxt + TB, ld_bytes, (kt + 1) * kBK, K, wave, lane);
        }
      });
    }
  } else if constexpr (GEO == 1) {
    stage_tile<GEO, false>(rs_w, smem, ld_bytes, 0, K, wave, lane);
    stage_tile<GEO, false>(rs_x, smem + TB, ld_bytes, 0, K, wave, lane);
    for (int kt = 0; kt < nk; ++kt) {
      __builtin_amdgcn_s_waitcnt(0);
      __syncthreads();  // tile kt landed for everyone; everyone finished reading the other stage
      const uint8_t* cur = smem + (kt & 1) * SB;
      const uint8_t* la = cur + (wn * 64) * kRowBytes;
      const uint8_t* lb = cur + TB + (wt * 64) * kRowBytes;
      Pack16 a0[4], a1[4], b0[4], b1[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int c = ks * 2 + fh;
        a0[ks] = read_frag(la, fr, c); a1[ks] = read_frag(la, 32 + fr, c);
        b0[ks] = read_frag(lb, fr, c); b1[ks] = read_frag(lb, 32 + fr, c);
      }
      if (kt + 1 < nk) {
        uint8_t* nxt = smem + ((kt + 1) & 1) * SB;
        stage_tile<GEO, false>(rs_w, nxt, ld_bytes, (kt + 1) * kBK, K, wave, lane);
        stage_tile<GEO, false>(rs_x, nxt + TB, ld_bytes, (kt + 1) * kBK, K, wave, lane);
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
    for (int kt = 0; kt < nk; ++kt) {
      stage_tile<GEO, false>(rs_w, smem, ld_bytes, kt * kBK, K, wave, lane);
      stage_tile<GEO, false>(rs_x, smem + TB, ld_bytes, kt * kBK, K, wave, lane);
      __builtin_amdgcn_s_waitcnt(0);
      __syncthreads();
      k_step<DT, GEO>(smem, wn, wt, fr, fh, acc, []() {});
      __syncthreads();  // all fragment reads done before the next tile overwrites the stage
    }
  }

  if constexpr (MODE != 2) {
    // DIAGNOSTIC (timing only, wrong results): MOQ_TUNE_GEMM_NO_EPILOGUE=1 replaces the loss epilogue -- whose `ref` reads
    // are 8 bytes per lane from 32 different rows per wave instruction: 7 M partial-line L2 requests per 4096 x 14336 launch
    // beside the 29 M of the operand stream -- by a plain sum of the accumulators
    if (upper_only & (1 << 30)) {
      float sq = 0.0f;
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) sq += acc[i][j][e];
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) sq += __shfl_xor(sq, off, 64);
      if (lane == 0) atomicAdd(&partial[(int64_t)blockIdx.y * gridDim.x + blockIdx.x], sq);
      return;
    }
  }
  gemm_epilogue<DT, MODE, NI, NJ, Geo<GEO>::WAVES>(acc, ref, bias, out, partial, smem, T, N, n0, t0, tn, tt, wn, wt, fr, fh,
                                                   lane, wave, decay, scale, upper_only);
}

// ---- GEO 6: the 256 x 256 x 64 tile as a two-group ping-pong (cdna_hip_programming.md 5, "8-phase" idea restated on
// this kernel's 32x32x16 tiles and whole-tile double buffer).
//
// What limited GEO 4 (PMC, profiles/r01_gemm_table.md): all eight waves meet at ONE barrier per K-tile and then do the
// same thing at the same time -- wait for the DMA, read fragments, multiply -- so the matrix pipe of every SIMD idles
// through each "wait, barrier, first reads" stretch (MFMA busy 46 %, waves parked 36 %).  Here the two waves that share
// a SIMD (wave w and w + 4: one of each GROUP) never do the same thing: a K-tile is four phases of
//     R: fragment reads + two LDS-DMA pieces      | barrier |      M: one quadrant of the wave's 128 x 64 tile, 8 MFMAs
// and group 1 runs ONE barrier behind group 0, so in every interval between two barriers one wave of each SIMD
// multiplies while the other reads and stages.  The DMA never drains: the pieces of tile kt + 1 / kt + 2 are issued two
// per phase, and the single wait per K-tile (`vmcnt(2)` at the start of the tile's last phase) leaves the newest two in
// flight; a tile is first read two barriers after every wave waited for its own pieces of it.
//
// Per wave and K-tile: R1 reads a0 b0 b1 (16 x ds_read_b128), R2 reads a1 (8); all reads of tile kt are retired before
// the wave's third phase, so from there on tile kt's stage takes the pieces of tile kt + 2.  Piece order of a wave
// (8 per tile: A rows of its own group's half, B rows of its eighth of the token tile, alternating):
//     R3(kt): (kt+2)[0,1]   R4(kt): wait, (kt+2)[2,3]   R1(kt+1): (kt+2)[4,5]   R2(kt+1): (kt+2)[6,7]
// Hazards.  RAW: wave waits vmcnt at R4(kt) for ALL its pieces of tile kt + 1 (<= 2 newer ones outstanding); group 0
// reads that tile two barriers later, group 1 three.  WAR: a stage is re-filled only after every reader passed an
// `s_waitcnt lgkmcnt(0)` (end of each M segment) and a barrier.
#define MOQ_BAR()                                   \
  do {                                              \
    asm volatile("s_barrier" ::: "memory");         \
    __builtin_amdgcn_sched_barrier(0);              \
  } while (0)
#define MOQ_M_END()                                            \
  do {                                                         \
    __builtin_amdgcn_s_setprio(0);                             \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         \
    __builtin_amdgcn_sched_barrier(0);                         \
  } while (0)

template <int DT, int MODE>
__global__ __launch_bounds__(512, 2)
void err_gemm6_kernel(const void* __restrict__ x, const void* __restrict__ w, const void* __restrict__ ref,
                      const void* __restrict__ bias, void* __restrict__ out, float* __restrict__ partial, int T, int N,
                      int K, int tiles_t, int tiles_n, int64_t x_stride, int64_t w_stride, float decay, float scale,
                      int upper_only) {
  constexpr int TILE = 256, NI = 4, NJ = 2;
  constexpr int TB = TILE * kRowBytes, SB = 2 * TB;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  int tn, tt;
  tile_of_block(blockIdx.x, tiles_t, tiles_n, MODE == 2 ? (upper_only >> 1) : upper_only, tn, tt);  // group size travels in upper_only
  if constexpr (MODE == 2) {
    // tiles_t / tiles_n describe the folded triangle here: (n + 1) columns x ceil(n / 2) row pairs (gram_tile)
    if (!gram_tile(tn, tt, tiles_t - 1, tn, tt)) return;
  }
  const int n0 = tn * TILE, t0 = tt * TILE;
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

  const int g = wave >> 2, wq = wave & 3;  // group = the wave's n half (A rows g * 128 ..), wq = its 64 tokens
  const int fr = lane & 31, fh = lane >> 5;
  const int nk = (K + kBK - 1) / kBK;
  const uint8_t* la0 = smem + (g * NI * 32) * kRowBytes;
  const uint8_t* lb0 = smem + TB + (wq * NJ * 32) * kRowBytes;
  Pack16 a[NI][4], b[NJ][4];

  // piece p (0..7) of this wave for K-tile `kt`: even = 8 rows of the group's A half, odd = 8 rows of the B tile
  auto issue = [&](int kt, int p) {
    uint8_t* st = smem + (kt & 1) * SB;
    const int j = p >> 1;
    if (p & 1) stage_piece(rs_x, st + TB, ld_bytes, kt * kBK, K, (wave * 4 + j) * 8, lane);
    else stage_piece(rs_w, st, ld_bytes, kt * kBK, K, g * 128 + (wq * 4 + j) * 8, lane);
  };
  auto read_a = [&](int so, int i0) {  // row blocks i0, i0 + 1 of the wave's A half, all four k sub-steps
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) a[i0 + i][ks] = read_frag(la0 + so, (i0 + i) * 32 + fr, ks * 2 + fh);
  };
  auto read_b = [&](int so, int j) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) b[j][ks] = read_frag(lb0 + so, j * 32 + fr, ks * 2 + fh);
  };
  auto quadrant = [&](int i0, int j) {  // 8 MFMAs: two accumulators, four dependent k sub-steps each
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      acc[i0][j] = mfma32<DT>(a[i0][ks], b[j][ks], acc[i0][j]);
      acc[i0 + 1][j] = mfma32<DT>(a[i0 + 1][ks], b[j][ks], acc[i0 + 1][j]);
    }
  };

  // prologue: tile 0 whole, the first half of tile 1
#pragma unroll
  for (int p = 0; p < 8; ++p) issue(0, p);
  if (nk > 1) {
#pragma unroll
    for (int p = 0; p < 4; ++p) issue(1, p);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  MOQ_BAR();
  if (g == 1) MOQ_BAR();  // group 1 runs one barrier behind group 0 from here on

  for (int kt = 0; kt < nk; ++kt) {
    const int so = (kt & 1) * SB;
    const bool has1 = kt + 1 < nk, has2 = kt + 2 < nk;
    // ---- phase 1
    read_b(so, 0);
    read_b(so, 1);
    read_a(so, 0);
    if (has1) { issue(kt + 1, 4); issue(kt + 1, 5); }
    MOQ_BAR();
    quadrant(0, 0);
    MOQ_M_END();
    MOQ_BAR();
    // ---- phase 2
    read_a(so, 2);
    if (has1) { issue(kt + 1, 6); issue(kt + 1, 7); }
    MOQ_BAR();
    quadrant(0, 1);
    MOQ_M_END();
    MOQ_BAR();
    // ---- phase 3: every read of tile kt is retired (M_END above + barrier): its stage takes tile kt + 2
    if (has2) { issue(kt + 2, 0); issue(kt + 2, 1); }
    MOQ_BAR();
    quadrant(2, 1);
    MOQ_M_END();
    MOQ_BAR();
    // ---- phase 4: this wave's pieces of tile kt + 1 have landed (the two just issued may still fly)
    if (has1) {
      if (has2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (has2) { issue(kt + 2, 2); issue(kt + 2, 3); }
    MOQ_BAR();
    quadrant(2, 0);
    MOQ_M_END();
    MOQ_BAR();
  }
  if (g == 0) MOQ_BAR();  // same number of barriers for both groups
  gemm_epilogue<DT, MODE, NI, NJ, 8>(acc, ref, bias, out, partial, smem, T, N, n0, t0, tn, tt, g, wq, fr, fh, lane, wave,
                                     decay, scale, upper_only);
}
#undef MOQ_BAR
#undef MOQ_M_END

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

// ---- GEO 12: 256 x 256 x 64 with FOUR waves (one per SIMD), the token operand direct to registers.
//
// What bounds the 8-wave loops (profiles/r02_gemm_table.md): two 64 KiB LDS stages are all that fits, so a K-tile's
// operands are requested ONE tile ahead; every first touch of an operand slice by an XCD misses its L2 (~19 % of the
// requests), takes longer than a tile and parks the whole workgroup at the tile's barrier.  Here only the WEIGHT tile
// (256 rows, shared by the waves) goes through the LDS -- 32 KiB stages, FOUR of them -- and each wave owns a 64-token
// column slab of the tile whose x rows nobody else reads: they go from HBM/L2 straight to the wave's registers in MFMA
// fragment layout (`buffer_load_dwordx4`, 32 VGPRs per K-tile, three register buffers).  Both operands are then
// requested THREE tiles ahead (~2.5 us of matrix work); the wait before a tile is a counted `vmcnt(32)` that never
// drains the queue, and near the end of K "dead" loads against an empty descriptor keep that count uniform.
//
// Per wave and K-tile: 64 MFMAs (8 weight blocks x 2 token blocks x 4 sub-steps) : 32 ds_read_b128 + 8 LDS-DMA pieces
// + 8 register loads.  LDS traffic drops to 32 KiB written + 128 KiB read per tile (8-wave loops: 64 + 192).
// k order inside a tile: sub-step j multiplies, for the half-wave h = lane >> 5, k = 32 h + 8 j .. + 7 -- a lane's four
// register loads of a row are then 64 contiguous bytes, issued back to back so that the row's line is fetched once.
// The same permutation is applied to the weight fragments (chunk 4 h + j); a sum over k does not care.
//
// Hazards.  RAW: wave waits `vmcnt(32)` -- everything it requested for tile kt has landed -- then the barrier makes
// the weight stage everyone's.  WAR (LDS): stage (kt + 3) & 3 = (kt - 1) & 3 is refilled after the barrier that opens
// tile kt, which every wave passes only after its fragment reads of tile kt - 1 were waited for (lgkmcnt before the
// MFMAs that consume them).  WAR (registers): buffer kt % 3 is reloaded after the last MFMA that reads it was issued.
template <int DT>
__device__ __forceinline__ f32x16_t mfma32v(const i32x4_t& a, const i32x4_t& b, f32x16_t c) {
  if constexpr (DT == MOQ_BF16) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c,
                                                   0, 0, 0);
  } else {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0,
                                                  0, 0);
  }
}

// DIAG (timing only, wrong results): 1 = MFMAs alone, 2 = MFMAs + fragment reads, 3 = MFMAs + HBM requests,
// 4 = HBM requests alone
template <int DT, int MODE, int DIAG = 0>
__global__ __launch_bounds__(256, 1)
void err_gemm12_kernel(const void* __restrict__ x, const void* __restrict__ w, const void* __restrict__ ref,
                       const void* __restrict__ bias, void* __restrict__ out, float* __restrict__ partial, int T, int N,
                       int K, int tiles_t, int tiles_n, int64_t x_stride, int64_t w_stride, float decay, float scale,
                       int upper_only) {
  constexpr int TILE = 256, NI = 8, NJ = 2;
  constexpr int TB = TILE * kRowBytes;  // one weight stage: 32 KiB
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  int tn, tt;
  tile_of_block(blockIdx.x, tiles_t, tiles_n, MODE == 2 ? (upper_only >> 1) : upper_only, tn, tt);  // group size travels in upper_only
  if constexpr (MODE == 2) {
    // tiles_t / tiles_n describe the folded triangle here: (n + 1) columns x ceil(n / 2) row pairs (gram_tile)
    if (!gram_tile(tn, tt, tiles_t - 1, tn, tt)) return;
  }
  const int n0 = tn * TILE, t0 = tt * TILE;
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
  const i32x4_t rsw = rs_w.words, rsx = rs_x.words;
  const int fr = lane & 31, fh = lane >> 5;
  const int nk = (K + kBK - 1) / kBK;
  const bool k_ragged = (K & (kBK - 1)) != 0;

  f32x16_t acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  // weight pieces: wave w stages rows (8 w + j) * 8 .. + 7, j = 0 .. 7 (lane -> row / chunk as in stage_tile)
  const uint32_t lds_wave = __builtin_amdgcn_readfirstlane(
      (uint32_t)(uintptr_t)(lds_u8_t)smem + (uint32_t)(wave * 8 * 8 * kRowBytes));
  int voff_a[8], voff_a_tail[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int r = (wave * 8 + j) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    voff_a[j] = (int)(r * ld_bytes + c * 16);
    voff_a_tail[j] = (nk - 1) * kBK + c * 8 < K ? voff_a[j] : 0x7FFFFFF0;
  }
  auto piece = [&](int kt, auto J) {  // weight piece J of tile kt (dead past the last tile)
    constexpr int j = decltype(J)::value;
    if constexpr (DIAG == 1 || DIAG == 2) return;
    const bool live = kt < nk;
    const uint32_t m0v = lds_wave + (uint32_t)((kt & 3) * TB + j * 8 * kRowBytes);
    const int koff = kt * kBK * 2;
    const int vfull = voff_a[j], vtail = voff_a_tail[j];
    const int vo = (k_ragged && kt == nk - 1) ? vtail : vfull;
    i32x4_t rr = rsw;
    rr.z = live ? rr.z : 0;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(m0v), "v"(vo), "s"(rr), "s"(koff) : "memory");
  };
  // token rows: lane (fr, fh) of block jb holds row 64 w + 32 jb + fr, bytes [64 fh, 64 fh + 64) of the tile's 128
  int voff_b[NJ];
#pragma unroll
  for (int jb = 0; jb < NJ; ++jb) voff_b[jb] = (int)((wave * 64 + jb * 32 + fr) * ld_bytes + fh * 64);
  const int kh = fh * 32;
  i32x4_t bq[3][NJ][4];
  auto load_b = [&](int kt, auto BUF, auto JB, auto J) {
    constexpr int buf = decltype(BUF)::value, jb = decltype(JB)::value, j = decltype(J)::value;
    if constexpr (DIAG == 1 || DIAG == 2) {
      if (kt < 3) bq[buf][jb][j] = i32x4_t{lane, kt, j, jb};
      return;
    }
    const bool live = kt < nk;
    const int koff = kt * kBK * 2;
    const int vb = voff_b[jb];
    const int vo = kt * kBK + kh + j * 8 < K ? vb : 0x7FFFFFF0;  // k tail: chunks at or past K read as zero
    i32x4_t rr = rsx;
    rr.z = live ? rr.z : 0;
    i32x4_t v;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4"
                 : "=v"(v) : "v"(vo), "s"(rr), "s"(koff), "n"(j * 16) : "memory");
    bq[buf][jb][j] = v;
  };
  auto request = [&](int kt, auto BUF) {  // prologue form: everything of tile kt at once
    piece(kt, IC<0>{}); piece(kt, IC<1>{}); piece(kt, IC<2>{}); piece(kt, IC<3>{});
    piece(kt, IC<4>{}); piece(kt, IC<5>{}); piece(kt, IC<6>{}); piece(kt, IC<7>{});
    load_b(kt, BUF, IC<0>{}, IC<0>{}); load_b(kt, BUF, IC<0>{}, IC<1>{});
    load_b(kt, BUF, IC<0>{}, IC<2>{}); load_b(kt, BUF, IC<0>{}, IC<3>{});
    load_b(kt, BUF, IC<1>{}, IC<0>{}); load_b(kt, BUF, IC<1>{}, IC<1>{});
    load_b(kt, BUF, IC<1>{}, IC<2>{}); load_b(kt, BUF, IC<1>{}, IC<3>{});
  };

  // weight fragments: row 32 i + fr, chunk 4 fh + j at position chunk ^ ((fr >> 1) & 7)
  const int sw = (fr >> 1) & 7;
  i32x4_t a[2][NI];
  auto read_a = [&](const uint8_t* la, auto BUF, auto J, auto I) {
    constexpr int buf = decltype(BUF)::value, j = decltype(J)::value, i = decltype(I)::value;
    if constexpr (DIAG == 1 || DIAG == 3 || DIAG == 4) {
      a[buf][i] = i32x4_t{lane + i, j, buf, 1};
      return;
    }
    a[buf][i] = *reinterpret_cast<const i32x4_t*>(la + (i * 32 + fr) * kRowBytes + (((fh * 4 + j) ^ sw) << 4));
  };

  request(0, IC<0>{});
  request(1, IC<1>{});
  request(2, IC<2>{});

  auto tile = [&](int kt, auto BUF) {
    constexpr int buf = decltype(BUF)::value;
    if constexpr (DIAG == 0 || DIAG == 3)
      asm volatile("s_waitcnt vmcnt(32)" ::: "memory");  // all of tile kt landed; tiles kt + 1, kt + 2 stay in flight
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    const uint8_t* la = smem + (kt & 3) * TB;
    read_a(la, IC<0>{}, IC<0>{}, IC<0>{}); read_a(la, IC<0>{}, IC<0>{}, IC<1>{});
    read_a(la, IC<0>{}, IC<0>{}, IC<2>{}); read_a(la, IC<0>{}, IC<0>{}, IC<3>{});
    read_a(la, IC<0>{}, IC<0>{}, IC<4>{}); read_a(la, IC<0>{}, IC<0>{}, IC<5>{});
    read_a(la, IC<0>{}, IC<0>{}, IC<6>{}); read_a(la, IC<0>{}, IC<0>{}, IC<7>{});
    __builtin_amdgcn_sched_barrier(0);
    // sub-step J: 16 MFMAs; after every second one a fragment read of sub-step J + 1, after the 4th and 12th a weight
    // piece of tile kt + 3
    auto sub = [&](auto J) {
      constexpr int j = decltype(J)::value, cur = j & 1, nxt = cur ^ 1;
      auto pair = [&](auto I) {
        constexpr int i = decltype(I)::value;
        if constexpr (DIAG != 4) {
          acc[i][0] = mfma32v<DT>(a[cur][i], bq[buf][0][j], acc[i][0]);
          acc[i][1] = mfma32v<DT>(a[cur][i], bq[buf][1][j], acc[i][1]);
        } else if constexpr (i == 0 && j == 0) {  // keep the loaded registers alive: one cheap use per tile
          acc[0][0][0] += __builtin_bit_cast(float, bq[buf][0][0].x ^ bq[buf][1][3].w ^ bq[buf][0][1].y ^ bq[buf][1][2].z);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (j < 3) read_a(la, IC<nxt>{}, IC<(j + 1) & 3>{}, I);
        if constexpr (i == 1) piece(kt + 3, IC<2 * j>{});
        if constexpr (i == 5) piece(kt + 3, IC<2 * j + 1>{});
        __builtin_amdgcn_sched_barrier(0);
      };
      pair(IC<0>{}); pair(IC<1>{}); pair(IC<2>{}); pair(IC<3>{});
      pair(IC<4>{}); pair(IC<5>{}); pair(IC<6>{}); pair(IC<7>{});
    };
    sub(IC<0>{}); sub(IC<1>{}); sub(IC<2>{}); sub(IC<3>{});
    // the buffer is free: its reload for tile kt + 3, a row's four loads back to back
    load_b(kt + 3, BUF, IC<0>{}, IC<0>{}); load_b(kt + 3, BUF, IC<0>{}, IC<1>{});
    load_b(kt + 3, BUF, IC<0>{}, IC<2>{}); load_b(kt + 3, BUF, IC<0>{}, IC<3>{});
    load_b(kt + 3, BUF, IC<1>{}, IC<0>{}); load_b(kt + 3, BUF, IC<1>{}, IC<1>{});
    load_b(kt + 3, BUF, IC<1>{}, IC<2>{}); load_b(kt + 3, BUF, IC<1>{}, IC<3>{});
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int kt = 0; kt < nk; kt += 3) {
    tile(kt, IC<0>{});
    if (kt + 1 < nk) tile(kt + 1, IC<1>{});
    if (kt + 2 < nk) tile(kt + 2, IC<2>{});
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // dead loads: registers and LDS are reused by the epilogue

  gemm_epilogue<DT, MODE, NI, NJ, 4>(acc, ref, bias, out, partial, smem, T, N, n0, t0, tn, tt, 0, wave, fr, fh, lane,
                                     wave, decay, scale, upper_only);
}

template <int MODE, int DIAG = 0>
static void launch_geo12(const void* x, const void* w, const void* ref, const void* bias, void* out, float* partial,
                         int64_t tokens, int64_t cout, int64_t cin, int dt, int n_cand, int64_t x_stride,
                         int64_t w_stride, void* stream, float decay, float scale, int upper_only) {
  constexpr int TILE = 256, LDS = 4 * TILE * kRowBytes;  // four weight stages: 128 KiB
  int tiles_t = (int)((tokens + TILE - 1) / TILE), tiles_n = (int)((cout + TILE - 1) / TILE);
  if (MODE == 2) {  // the folded upper triangle (gram_tile): (n + 1) x ceil(n / 2) workgroups
    const int n = tiles_n;
    tiles_t = n + 1;
    tiles_n = (n + 1) / 2;
  }
  static std::atomic<uint64_t> attr_set{0};
  int device = 0;
  (void)hipGetDevice(&device);
  const uint64_t bit = 1ull << (device & 63);
  if (!(attr_set.load(std::memory_order_acquire) & bit)) {
    (void)hipFuncSetAttribute((const void*)err_gemm12_kernel<MOQ_BF16, MODE, DIAG>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    (void)hipFuncSetAttribute((const void*)err_gemm12_kernel<MOQ_F16, MODE, DIAG>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_set.fetch_or(bit, std::memory_order_release);
  }
  const dim3 grid((unsigned)(tiles_t * tiles_n), (unsigned)n_cand), block(256);
  if (dt == MOQ_BF16) {
    hipLaunchKernelGGL((err_gemm12_kernel<MOQ_BF16, MODE, DIAG>), grid, block, LDS, S(stream), x, w, ref, bias, out, partial,
                       (int)tokens, (int)cout, (int)cin, tiles_t, tiles_n, x_stride, w_stride, decay, scale, upper_only);
  } else {
    hipLaunchKernelGGL((err_gemm12_kernel<MOQ_F16, MODE, DIAG>), grid, block, LDS, S(stream), x, w, ref, bias, out, partial,
                       (int)tokens, (int)cout, (int)cin, tiles_t, tiles_n, x_stride, w_stride, decay, scale, upper_only);
  }
}

static int gemm_geo() {
  // MOQ_TUNE_GEMM_GEO selects the tile geometry / loop structure (A/B knob, read once).  Default: GEO 10 -- the 1 : 1
  // MFMA / memory stream; with the branch-free three-instruction LDS-DMA pieces it runs 3-5 % ahead of GEO 4
  // (profiles/r02_gemm_table.md)
  static const int geo = [] {
    const char* e = getenv("MOQ_TUNE_GEMM_GEO");
    const int g = e ? atoi(e) : 10;
    return g < 0 || g > 22 ? 10 : g;
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
    (void)hipFuncSetAttribute((const void*)err_gemm_kernel<MOQ_BF16, MODE, GEO>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes<GEO>());
    (void)hipFuncSetAttribute((const void*)err_gemm_kernel<MOQ_F16, MODE, GEO>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes<GEO>());
    attr_set.fetch_or(bit, std::memory_order_release);
  }
  const dim3 grid(nblk, (unsigned)n_cand), block(Geo<GEO>::WAVES * 64);
  if (dt == MOQ_BF16) {
    hipLaunchKernelGGL((err_gemm_kernel<MOQ_BF16, MODE, GEO>), grid, block, lds_bytes<GEO>(), S(stream), x, w, ref,
                       bias, out, partial, (int)tokens, (int)cout, (int)cin, tiles_t, tiles_n, x_stride, w_stride,
                       decay, scale, upper_only);
  } else {
    hipLaunchKernelGGL((err_gemm_kernel<MOQ_F16, MODE, GEO>), grid, block, lds_bytes<GEO>(), S(stream), x, w, ref,
                       bias, out, partial, (int)tokens, (int)cout, (int)cin, tiles_t, tiles_n, x_stride, w_stride,
                       decay, scale, upper_only);
  }
}

template <int MODE>
static void launch_geo6(const void* x, const void* w, const void* ref, const void* bias, void* out, float* partial,
                        int64_t tokens, int64_t cout, int64_t cin, int dt, int n_cand, int64_t x_stride,
                        int64_t w_stride, void* stream, float decay, float scale, int upper_only) {
  constexpr int TILE = 256, LDS = 2 * 2 * TILE * kRowBytes;  // two stages of an A and a B tile: 128 KiB
  int tiles_t = (int)((tokens + TILE - 1) / TILE), tiles_n = (int)((cout + TILE - 1) / TILE);
  if (MODE == 2) {  // the folded upper triangle (gram_tile): (n + 1) x ceil(n / 2) workgroups
    const int n = tiles_n;
    tiles_t = n + 1;
    tiles_n = (n + 1) / 2;
  }
  static std::atomic<uint64_t> attr_set{0};
  int device = 0;
  (void)hipGetDevice(&device);
  const uint64_t bit = 1ull << (device & 63);
  if (!(attr_set.load(std::memory_order_acquire) & bit)) {
    (void)hipFuncSetAttribute((const void*)err_gemm6_kernel<MOQ_BF16, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    (void)hipFuncSetAttribute((const void*)err_gemm6_kernel<MOQ_F16, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_set.fetch_or(bit, std::memory_order_release);
  }
  const dim3 grid((unsigned)(tiles_t * tiles_n), (unsigned)n_cand), block(512);
  if (dt == MOQ_BF16) {
    hipLaunchKernelGGL((err_gemm6_kernel<MOQ_BF16, MODE>), grid, block, LDS, S(stream), x, w, ref, bias, out, partial,
                       (int)tokens, (int)cout, (int)cin, tiles_t, tiles_n, x_stride, w_stride, decay, scale, upper_only);
  } else {
    hipLaunchKernelGGL((err_gemm6_kernel<MOQ_F16, MODE>), grid, block, LDS, S(stream), x, w, ref, bias, out, partial,
                       (int)tokens, (int)cout, (int)cin, tiles_t, tiles_n, x_stride, w_stride, decay, scale, upper_only);
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
    // workgroup -> tile order in it (tile_of_block).  MOQ_TUNE_GEMM_GROUP = 1 restores the plain row-major order.
    static const int group = [] {
      const char* e = getenv("MOQ_TUNE_GEMM_GROUP");
      const int g = e ? atoi(e) : kTileGroup;
      return g < 1 || g > 64 ? kTileGroup : g;
    }();
    upper_only = group | (moq_tune("MOQ_TUNE_GEMM_NO_EPILOGUE", 0) ? (1 << 30) : 0) | (((int)moq_tune("MOQ_TUNE_GEMM_STAT", 0) & 7) << 26);
  } else {
    // Gram mode: bit 0 stays the upper_only flag, the tile-group size rides above it
    static const int group2 = [] {
      const char* e = getenv("MOQ_TUNE_GEMM_GROUP");
      const int g = e ? atoi(e) : kTileGroup;
      return g < 1 || g > 64 ? kTileGroup : g;
    }();
    upper_only = (upper_only ? 1 : 0) | (group2 << 1);
  }
  const int tile = geo >= 2 ? 256 : 128;
  const int64_t nblk = n_tiles_for(tokens, cout, tile);
  if (nblk > 0x7FFFFFFF) {
    set_error("gemm: too many tiles");
    return MOQ_ERR_UNSUPPORTED;
  }
  switch (geo) {
    case 0: launch_geo<MODE, 0>(x, w, ref, bias, out, partial, tokens, cout, cin, dt, n_cand, x_stride, w_stride, stream, decay, scale, upper_only); break;
    case 1: launch_geo<MODE, 1>(x, w, ref, bias, out, partial, tokens, cout, cin, dt, n_cand, x_stride, w_stride, stream, decay, scale, upper_only); break;
    case 3: launch_geo<MODE, 3>(x, w, ref, bias, out, partial, tokens, cout, cin, dt, n_cand, x_stride, w_stride, stream, decay, scale, upper_only); break;
    case 5: launch_geo<MODE, 5>(x, w, ref, bias, out, partial, tokens, cout, cin, dt, n_cand, x_stride, w_stride, stream, decay, scale, upper_only); break;
    case 2: launch_geo<MODE, 2>(x, w, ref, bias, out, partial, tokens, cout, cin, dt, n_cand, x_stride, w_stride, stream, decay, scale, upper_only); break;
    case 10: launch_geo<MODE, 10>(x, w, ref, bias, out, partial, tokens, cout, cin, dt, n_cand, x_stride, w_stride, stream, decay, scale, upper_only); break;
    case 8: launch_geo<MODE, 8>(x, w, ref, bias, out, partial, tokens, cout, cin, dt, n_cand, x_stride, w_stride, stream, decay, scale, upper_only); break;
    case 9: launch_geo<MODE, 9>(x, w, ref, bias, out, partial, tokens, cout, cin, dt, n_cand, x_stride, w_stride, stream, decay, scale, upper_only); break;
    case 7: launch_geo<MODE, 7>(x, w, ref, bias, out, partial, tokens, cout, cin, dt, n_cand, x_stride, w_stride, stream, decay, scale, upper_only); break;
    case 12: launch_geo12<MODE>(x, w, ref, bias, out, partial, tokens, cout, cin, dt, n_cand, x_stride, w_stride, stream, decay, scale, upper_only); break;
    case 13: launch_geo12<MODE, 1>(x, w, ref, bias, out, partial, tokens, cout, cin, dt, n_cand, x_stride, w_stride, stream, decay, scale, upper_only); break;
    case 14: launch_geo12<MODE, 2>(x, w, ref, bias, out, partial, tokens, cout, cin, dt, n_cand, x_stride, w_stride, stream, decay, scale, upper_only); break;
    case 16: launch_geo12<MODE, 4>(x, w, ref, bias, out, partial, tokens, cout, cin, dt, n_cand, x_stride, w_stride, stream, decay, scale, upper_only); break;
    case 15: launch_geo12<MODE, 3>(x, w, ref, bias, out, partial, tokens, cout, cin, dt, n_cand, x_stride, w_stride, stream, decay, scale, upper_only); break;
    case 17: launch_geo<MODE, 17>(x, w, ref, bias, out, partial, tokens, cout, cin, dt, n_cand, x_stride, w_stride, stream, decay, scale, upper_only); break;
    case 18: launch_geo<MODE, 18>(x, w, ref, bias, out, partial, tokens, cout, cin, dt, n_cand, x_stride, w_stride, stream, decay, scale, upper_only); break;
    case 20: launch_geo<MODE, 20>(x, w, ref, bias, out, partial, tokens, cout, cin, dt, n_cand, x_stride, w_stride, stream, decay, scale, upper_only); break;
    case 21: launch_geo<MODE, 21>(x, w, ref, bias, out, partial, tokens, cout, cin, dt, n_cand, x_stride, w_stride, stream, decay, scale, upper_only); break;
    case 22: launch_geo<MODE, 22>(x, w, ref, bias, out, partial, tokens, cout, cin, dt, n_cand, x_stride, w_stride, stream, decay, scale, upper_only); break;
    case 6: launch_geo6<MODE>(x, w, ref, bias, out, partial, tokens, cout, cin, dt, n_cand, x_stride, w_stride, stream, decay, scale, upper_only); break;
    default: launch_geo<MODE, 4>(x, w, ref, bias, out, partial, tokens, cout, cin, dt, n_cand, x_stride, w_stride, stream, decay, scale, upper_only); break;
  }
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
  const int64_t nblk = launch_gemm<0>(x, w, out_actual, bias, nullptr, partial, tokens, cout, cin, dt, n_cand,
                                      x_stride, w_stride, stream);
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
  const int64_t nblk = launch_gemm<1>(x, w, nullptr, bias, out, nullptr, tokens, cout, cin, dt, 1, 0, 0, stream);
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
  const int64_t nblk = launch_gemm<3>(a, b, ref, nullptr, nullptr, partial, rows, cols, k, dt, 1, 0, 0, stream);
  if (nblk < 0) return (int)nblk;
  hipLaunchKernelGGL(err_finalize_kernel, dim3(1), dim3(256), 0, S(stream), partial, (int)nblk, inv_count, loss_acc);
  return check_launch("moq_awq_quadform");
}
