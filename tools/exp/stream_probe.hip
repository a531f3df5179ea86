// stream_probe.hip -- standalone HBM streaming probe for MI355X (not part of the product library).
// Measures what this box's HBM actually delivers for the access shapes our kernels use, so that kernel
// numbers can be read against a measured ceiling, and A/Bs launch-shape variants:
//   copy  : 16 B/lane read + write (the QDQ shape), read : 16 B/lane read-only (the amax shape)
// knobs : P packets in flight per lane, NT non-temporal hint, grid size, contiguous-range vs interleaved.
// build : hipcc --offload-arch=gfx950 -O3 -o stream_probe stream_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <bool NT> __device__ __forceinline__ u32x4 ld(const u32x4* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st(u32x4* p, u32x4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// chunk = 256 threads * P packets; CONTIG: block owns a contiguous run of chunks, else grid-stride
template <int P, bool NT, bool CONTIG, bool WRITE>
__global__ __launch_bounds__(256) void probe(const u32x4* __restrict__ x, u32x4* __restrict__ y, int64_t n_chunks, uint32_t* sink) {
  int64_t c0, c1, step;
  if (CONTIG) { int64_t per = (n_chunks + gridDim.x - 1) / gridDim.x; c0 = blockIdx.x * per; c1 = c0 + per < n_chunks ? c0 + per : n_chunks; step = 1; }
  else { c0 = blockIdx.x; c1 = n_chunks; step = gridDim.x; }
  uint32_t acc = 0;
  for (int64_t c = c0; c < c1; c += step) {
    const int64_t base = c * (256 * P) + threadIdx.x;
    u32x4 v[P];
#pragma unroll
    for (int u = 0; u < P; ++u) v[u] = ld<NT>(x + base + u * 256);
#pragma unroll
    for (int u = 0; u < P; ++u) {
      if (WRITE) { u32x4 o = v[u]; o.x ^= 0x80008000u; st<NT>(y + base + u * 256, o); }
      else acc |= v[u].x | v[u].y | v[u].z | v[u].w;
    }
  }
  if (!WRITE && acc == 0x12345678u) *sink = acc;
}

template <int P, bool NT, bool CONTIG, bool WRITE>
static float run(const u32x4* x, u32x4* y, int64_t n_packets, int grid, uint32_t* sink, int reps) {
  int64_t n_chunks = n_packets / (256 * P);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  probe<P, NT, CONTIG, WRITE><<<grid, 256>>>(x, y, n_chunks, sink);
  hipEventRecord(a);
  for (int i = 0; i < reps; ++i) probe<P, NT, CONTIG, WRITE><<<grid, 256>>>(x, y, n_chunks, sink);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

int main(int argc, char** argv) {
  const int64_t bytes = (int64_t)14 << 30;  // 14 GiB in, 14 GiB out: far past the 256 MiB Infinity Cache
  const int64_t n_packets = bytes / 16;
  u32x4 *x, *y; uint32_t* sink;
  CK(hipMalloc(&x, bytes)); CK(hipMalloc(&y, bytes)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(x, 0x3c, bytes)); CK(hipMemset(y, 0, bytes));
  const int reps = 5;
  printf("%-6s %2s %2s %-7s %5s %9s %9s\n", "kind", "P", "NT", "order", "grid", "ms", "GB/s");
#define ROW(P, NT, CONTIG, WRITE, GRID) { float ms = run<P, NT, CONTIG, WRITE>(x, y, n_packets, GRID, sink, reps); \
    double gb = (WRITE ? 2.0 : 1.0) * bytes / 1e9; printf("%-6s %2d %2d %-7s %5d %9.3f %9.1f\n", WRITE ? "copy" : "read", P, (int)NT, CONTIG ? "contig" : "stride", GRID, ms, gb / (ms * 1e-3)); }
  const int all4 = (int)(n_packets / (256 * 4)), all2 = (int)(n_packets / (256 * 2)), all1 = (int)(n_packets / 256);
  for (int grid : {4096, 8192, 16384, 32768, 65536, 131072, all4}) {
    ROW(4, true, false, true, grid) ROW(4, false, false, true, grid)
    ROW(4, true, false, false, grid)
  }
  for (int grid : {8192, 32768, all2}) { ROW(2, true, false, true, grid) }
  for (int grid : {8192, 32768, all1}) { ROW(1, true, false, true, grid) }
  for (int grid : {8192, 32768}) { ROW(8, true, false, true, grid) ROW(8, true, false, false, grid) }
  return 0;
}
