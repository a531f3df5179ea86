// stream_probe2.hip -- follow-up to stream_probe.hip for the "slow box" question: on some nodes the read-only stream
// runs at full speed (6.7-7.0 TB/s) while the read+write stream drops from ~6.4 to ~5.5 TB/s.  Separates the load and
// the store hint, adds a write-only stream and a read/write mix with the two streams in different address regions.
// build : hipcc --offload-arch=gfx950 -O3 -o stream_probe2 stream_probe2.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <bool NT> __device__ __forceinline__ u32x4 ld(const u32x4* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st(u32x4* p, u32x4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// MODE 0: copy x -> y, 1: read only, 2: write only, 3: in place (y == x)
template <int P, bool NTL, bool NTS, int MODE>
__global__ __launch_bounds__(256) void probe(const u32x4* __restrict__ x, u32x4* __restrict__ y, int64_t n_chunks, uint32_t* sink) {
  uint32_t acc = 0;
  for (int64_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    const int64_t base = c * (256 * P) + threadIdx.x;
    u32x4 v[P];
    if (MODE != 2) {
#pragma unroll
      for (int u = 0; u < P; ++u) v[u] = ld<NTL>(x + base + u * 256);
    } else {
#pragma unroll
      for (int u = 0; u < P; ++u) v[u] = u32x4{(uint32_t)base, 1u, 2u, 3u};
    }
#pragma unroll
    for (int u = 0; u < P; ++u) {
      if (MODE == 1) acc |= v[u].x | v[u].y | v[u].z | v[u].w;
      else { u32x4 o = v[u]; o.x ^= 0x80008000u; st<NTS>((MODE == 3 ? const_cast<u32x4*>(x) : y) + base + u * 256, o); }
    }
  }
  if (MODE == 1 && acc == 0x12345678u) *sink = acc;
}

template <int P, bool NTL, bool NTS, int MODE>
static float run(const u32x4* x, u32x4* y, int64_t n_packets, int grid, uint32_t* sink, int reps) {
  int64_t n_chunks = n_packets / (256 * P);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  probe<P, NTL, NTS, MODE><<<grid, 256>>>(x, y, n_chunks, sink);
  hipEventRecord(a);
  for (int i = 0; i < reps; ++i) probe<P, NTL, NTS, MODE><<<grid, 256>>>(x, y, n_chunks, sink);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

int main() {
  const int64_t bytes = (int64_t)14 << 30;
  const int64_t n_packets = bytes / 16;
  u32x4 *x, *y; uint32_t* sink;
  CK(hipMalloc(&x, bytes)); CK(hipMalloc(&y, bytes)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(x, 0x3c, bytes)); CK(hipMemset(y, 0, bytes));
  const int reps = 5;
  static const char* names[] = {"copy", "read", "write", "inplace"};
  printf("%-8s %2s %3s %3s %7s %9s %9s\n", "kind", "P", "NTL", "NTS", "grid", "ms", "GB/s");
#define ROW(P, NTL, NTS, MODE, GRID) { float ms = run<P, NTL, NTS, MODE>(x, y, n_packets, GRID, sink, reps); \
    double gb = ((MODE == 0 || MODE == 3) ? 2.0 : 1.0) * bytes / 1e9; \
    printf("%-8s %2d %3d %3d %7d %9.3f %9.1f\n", names[MODE], P, (int)NTL, (int)NTS, GRID, ms, gb / (ms * 1e-3)); }
  for (int grid : {8192, 32768, 131072}) {
    ROW(4, true, true, 1, grid)
    ROW(4, true, true, 2, grid) ROW(4, true, false, 2, grid)
    ROW(4, true, true, 0, grid) ROW(4, true, false, 0, grid) ROW(4, false, true, 0, grid) ROW(4, false, false, 0, grid)
    ROW(4, true, true, 3, grid) ROW(4, true, false, 3, grid)
    ROW(8, true, true, 0, grid) ROW(2, true, true, 0, grid)
  }
  return 0;
}
