// moq_chunk.h -- the chunk skeleton shared by the HBM-streaming kernels (moq_stream.hip, moq_formats.hip).
//
// A *chunk* is MOQ_MT_CHUNK = 8192 consecutive elements; a 256-thread workgroup owns a chunk at a time and moves
// it with 16-byte lane accesses: packet u of thread t covers elements (u*256 + t)*kVec ..., so every wave
// instruction touches one contiguous KiB.  All loads of a chunk are issued before the first use; read-once /
// write-once streams carry the non-temporal hint; copy-shaped passes use a large strided grid (copy_grid).
#pragma once

#include <stdlib.h>

#include "moq_common.h"

namespace moq {

template <int DT>
struct Chunk {
  static constexpr int kVec = Elem<DT>::kVec;
  static constexpr int kPackets = MOQ_MT_CHUNK / (kBlock * kVec);  // 4 (16-bit) or 8 (f32)
  static_assert(kPackets * kBlock * kVec == MOQ_MT_CHUNK, "chunk must tile exactly");
};

// element offset (inside the chunk) of packet u of this thread
template <int DT>
__device__ __forceinline__ int packet_off(int u) {
  return (u * kBlock + (int)threadIdx.x) * Elem<DT>::kVec;
}

// Guarded / unaligned packet access.  FAST = chunk fully inside the tensor and base 16-byte aligned.
template <int DT, bool FAST, bool NT = true>
__device__ __forceinline__ Pack16 ld_packet(const void* base, int64_t e, int64_t n) {
  constexpr int V = Elem<DT>::kVec;
  constexpr int ES = 16 / V;
  if constexpr (FAST) {
    // streaming data is read exactly once: non-temporal hint (measured on MI355X, 14 GiB streams:
    // read-only 6.2 -> 7.0 TB/s, copy 5.9 -> 6.5 TB/s; tools/exp/stream_probe.hip).  NT = false keeps the
    // lines in L2 / Infinity Cache for a second pass over the same tensor (grouped calibrate -> QDQ).
    if constexpr (NT) return load16_nt(reinterpret_cast<const char*>(base) + e * ES);
    else return load16(reinterpret_cast<const char*>(base) + e * ES);
  } else {
    float f[V];
#pragma unroll
    for (int i = 0; i < V; ++i) f[i] = (e + i < n) ? load1<DT>(base, e + i) : 0.0f;
    if constexpr (DT == MOQ_F32) {
      return pack<DT>(f);
    } else {
      // keep the exact 16-bit patterns (no re-rounding): rebuild from raw storage
      Pack16 p;
      const uint16_t* b = reinterpret_cast<const uint16_t*>(base);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        uint32_t lo = (e + 2 * i < n) ? b[e + 2 * i] : 0u;
        uint32_t hi = (e + 2 * i + 1 < n) ? b[e + 2 * i + 1] : 0u;
        p.w[i] = lo | (hi << 16);
      }
      return p;
    }
  }
}
template <int DT, bool FAST>
__device__ __forceinline__ void st_packet(void* base, int64_t e, int64_t n, const Pack16& p) {
  constexpr int V = Elem<DT>::kVec;
  constexpr int ES = 16 / V;
  if constexpr (FAST) {
    store16_nt(reinterpret_cast<char*>(base) + e * ES, p);
  } else {
    if constexpr (DT == MOQ_F32) {
      float* b = reinterpret_cast<float*>(base);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (e + i < n) b[e + i] = __uint_as_float(p.w[i]);
    } else {
      uint16_t* b = reinterpret_cast<uint16_t*>(base);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (e + 2 * i < n) b[e + 2 * i] = (uint16_t)(p.w[i] & 0xFFFFu);
        if (e + 2 * i + 1 < n) b[e + 2 * i + 1] = (uint16_t)(p.w[i] >> 16);
      }
    }
  }
}

__device__ __forceinline__ bool aligned16(const void* p) {
  return (reinterpret_cast<uintptr_t>(p) & 15u) == 0;
}

// ------------------------------------------------------------------------------------------------
// multi-tensor kernels: a block owns a contiguous run of chunks of the segment table, so per-tensor
// reductions cost ~one atomic per (block, tensor) instead of one per chunk (same-address L2 atomics
// retire at only ~80 M/s on this chip -- MI355X_MICROARCH.md "fanin"/"dequeue" rows).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int find_segment(const int64_t* __restrict__ blk_start, int n_seg,
                                            int64_t chunk) {
  int lo = 0, hi = n_seg;  // invariant: blk_start[lo] <= chunk < blk_start[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (blk_start[mid] <= chunk) lo = mid; else hi = mid;
  }
  return lo;
}
struct ChunkRange {
  int64_t begin, end;
};
__device__ __forceinline__ ChunkRange block_range(int64_t n_chunks) {
  const int64_t per = (n_chunks + gridDim.x - 1) / gridDim.x;
  ChunkRange r;
  r.begin = (int64_t)blockIdx.x * per;
  r.end = r.begin + per < n_chunks ? r.begin + per : n_chunks;
  return r;
}
// Cursor over the segment table: the current tensor's fields stay in (scalar) registers and are reloaded
// only when a workgroup's chunk index crosses into another tensor, so the vector loads of a chunk never
// wait on a table lookup.
struct SegCursor {
  const moq_seg* segs;
  const int64_t* blk_start;
  int s;
  int64_t c_begin, c_end;  // chunk range of the current segment
  moq_seg sg;
  bool aligned;
  __device__ __forceinline__ void init(const moq_seg* sgs, const int64_t* bs, int n_seg, int64_t chunk) {
    segs = sgs;
    blk_start = bs;
    s = find_segment(bs, n_seg, chunk);
    load();
  }
  __device__ __forceinline__ void load() {
    c_begin = blk_start[s];
    c_end = blk_start[s + 1];
    sg = segs[s];
    aligned = aligned16(sg.x) && aligned16(sg.y);
  }
  // returns true when the segment changed (chunk indices only ever grow)
  __device__ __forceinline__ bool seek(int64_t chunk) {
    if (chunk < c_end) return false;
    do { ++s; } while (chunk >= blk_start[s + 1]);
    load();
    return true;
  }
};

// Index of the group (quantization block / scale entry) an element belongs to, for kernels that walk chunks:
// one 64-bit division per chunk (uniform), 32-bit shift / division per packet.
struct GroupIndex {
  int64_t q0;
  uint32_t r0, g;
  int shift;  // log2(g) when g is a power of two, else -1
  __device__ __forceinline__ void seek(int64_t e0) {
    q0 = e0 / (int64_t)g;
    r0 = (uint32_t)(e0 - q0 * (int64_t)g);
  }
  __device__ __forceinline__ int64_t at(uint32_t off) const {
    const uint32_t t = r0 + off;
    return q0 + (int64_t)(shift >= 0 ? (t >> shift) : (t / g));
  }
};
}  // namespace moq

static inline int log2_or_neg(int64_t g) {
  if (g <= 0 || (g & (g - 1)) != 0) return -1;
  int s = 0;
  while ((1LL << s) < g) ++s;
  return s;
}

// Copy-shaped passes (read + write): ONE chunk per workgroup, grid = n_chunks -- the machine's resident workgroups form a single
// dense window that moves through the tensors in dispatch order.  Rounds 1-3 ran ~8 chunks per workgroup (chunk = blockIdx +
// k * gridDim: eight windows 1.6 GB apart), which is faster on SOME placements of the tensors (0.80 of 8 TB/s) and slower on
// others (0.68), by box and by allocation; the single window runs at 0.76-0.77 wherever the driver put the data, on every box
// measured (profiles/r04_pool_placement.md: eight differently placed copies of the Llama-3-8B weights per process, five boxes;
// more windows are monotonically worse on the boxes of the slow class whatever their distance).  The kernels keep their
// grid-stride loops (a grid is capped at 2^31 - 1 workgroups); MOQ_TUNE_* exist in the experiment build only.
static inline int64_t copy_grid64(int64_t n_chunks) {
  const int64_t div = moq_tune("MOQ_TUNE_CHUNKS_PER_WG", 1);
  int64_t g = n_chunks / (div > 0 ? div : 1);
  const int64_t cap = moq_tune("MOQ_TUNE_COPY_GRID_CAP", 0x7FFFFFFF);
  if (g < 2048) g = 2048;
  if (g > cap) g = cap;
  if (g > n_chunks) g = n_chunks;
  return g < 1 ? 1 : g;
}
static inline int copy_grid(int64_t n_chunks) { return (int)copy_grid64(n_chunks); }
// read-only sweeps (whole-model abs-max): one chunk per workgroup as well -- 0.86 of 8 TB/s against 0.83 for the strided order
// of rounds 1-3 on the same allocations (profiles/r04_pool_placement.md)
static inline int read_grid(int64_t n_chunks) {
  const int64_t div = moq_tune("MOQ_TUNE_READ_CHUNKS_PER_WG", 1);
  int64_t g = n_chunks / (div > 0 ? div : 1);
  if (g < 2048) g = 2048;
  if (g > 0x7FFFFFFF) g = 0x7FFFFFFF;
  if (g > n_chunks) g = n_chunks;
  return (int)(g < 1 ? 1 : g);
}
// Dynamic LDS requested by the whole-model read + write launches although their kernels use none: 32 KiB per 256-thread
// workgroup caps a CU at 5 resident workgroups instead of 8 (a 20 MiB window over the chip instead of 32 MiB).  Measured per
// kernel on the same allocations (profiles/r04_pool_placement.md, "occupancy"): FP8 / INT-k map 0.758 -> 0.768, fused INT4 g128
// 0.736 -> 0.772, 2:4 mask 0.777 -> 0.787 at 32 KiB; the MX kernel (more registers per thread) is best at 24 KiB: 0.729 -> 0.739.
static inline size_t copy_lds(size_t dflt = 32 * 1024) { return (size_t)moq_tune("MOQ_TUNE_COPY_LDS", (long long)dflt); }
// The same cap for single-tensor read + write launches, where it measured above the run-to-run noise (tools/kbench.py with the
// experiment library at 0 / 24 / 32 KiB, profiles/r04t_kernel_ab.md): column scaling 0.68 -> 0.73, per-group INT QDQ and the
// AWQ scale + QDQ 0.71-0.75 -> 0.75-0.77, 2:4 mask 0.80 -> 0.83, INT8 row packer 0.79 -> 0.82 at 32 KiB; MX at 24 KiB
// 0.77 -> 0.79.  Launches that lose (INT4 / MXFP4 packers and unpackers: 0.67 -> 0.53 at 32 KiB; read-only per-group abs-max
// 0.82 -> 0.71) or do not move (FP8 QDQ / packers) pass the default 0.
static inline size_t copy_lds_1t(size_t dflt = 0) { return (size_t)moq_tune("MOQ_TUNE_COPY_LDS_1T", (long long)dflt); }
