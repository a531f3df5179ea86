// LDS atomic throughput probe (gfx950): how many ds_add_u32 lane-operations per clock a CU retires, by address pattern.
// Decides the histogram kernel's design (moq_inputq.hip).  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/exp/lds_atomic_probe.hip -o /tmp/lds_probe && /tmp/lds_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <int MODE>
__global__ __launch_bounds__(1024) void probe(uint32_t* out, int iters, int slots_mask) {
  extern __shared__ uint32_t lds[];
  for (int i = threadIdx.x; i <= slots_mask; i += 1024) lds[i] = 0;
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63;
  uint32_t r = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  for (int it = 0; it < iters; ++it) {
    uint32_t a;
    if (MODE == 0) a = lane;                                  // 64 distinct consecutive dwords: conflict-free
    else if (MODE == 1) a = (r >> 8) & slots_mask;            // uniformly random over the table
    else if (MODE == 2) a = (r >> 8) & 3;                     // 4 hot addresses (same-address pile-up)
    else if (MODE == 3) a = ((r >> 8) & 3) * 8 + (lane & 7);  // 4 hot bins x 8 interleaved copies
    else if (MODE == 4) a = (lane * 32 + ((r >> 8) & 31)) & slots_mask;  // random bank-row per lane, distinct rows
    else a = (lane & 31) + 32 * ((r >> 8) & 63);              // bank = lane & 31 (conflict-free), random row
    if (MODE == 6) { lds[(r >> 8) & slots_mask] += 1; }       // non-atomic read-modify-write for comparison
    else atomicAdd(&lds[a], 1u);
    r = r * 1664525u + 1013904223u;
  }
  __syncthreads();
  uint32_t s = 0;
  for (int i = threadIdx.x; i <= slots_mask; i += 1024) s += lds[i];
  if (s == 0xFFFFFFFFu) out[0] = s;
}

template <int MODE>
static void run(const char* name, uint32_t* out) {
  const int iters = 4096, blocks = 512, mask = 16383;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  probe<MODE><<<blocks, 1024, (mask + 1) * 4>>>(out, 64, mask);
  hipDeviceSynchronize();
  hipEventRecord(a);
  probe<MODE><<<blocks, 1024, (mask + 1) * 4>>>(out, iters, mask);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double ops = (double)blocks * 1024 * iters;
  printf("%-58s %8.3f ms  %7.2f lane-ops/clk/CU (at 2.4 GHz, 256 CUs)\n", name, ms, ops / (ms * 1e-3) / 2.4e9 / 256);
}

int main() {
  uint32_t* out;
  hipMalloc(&out, 4);
  run<0>("ds_add_u32, lane-consecutive (conflict-free)", out);
  run<5>("ds_add_u32, bank = lane & 31, random row", out);
  run<1>("ds_add_u32, uniformly random over 16384 slots", out);
  run<4>("ds_add_u32, random within the lane's own 32-slot row", out);
  run<3>("ds_add_u32, 4 hot bins x 8 copies", out);
  run<2>("ds_add_u32, 4 hot addresses", out);
  run<6>("plain LDS read-modify-write (not atomic), random", out);
  return 0;
}
