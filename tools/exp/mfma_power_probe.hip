// Which bf16 MFMA shape delivers more under the power cap?  One wave per SIMD on every CU issues nothing but MFMAs on
// register operands (random bf16 bits or zeros); TFLOP/s from hipEvents, shader clock from s_memtime against the 100 MHz
// s_memrealtime.  Build: hipcc --offload-arch=gfx950 -O3 tools/exp/mfma_power_probe.hip -o tools/exp/bin/mfma_power_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>  // 0: 32x32x16 (16 accumulators of 16 regs), 1: 16x16x32 (32 accumulators of 4 regs)
__global__ __launch_bounds__(256, 1) void mfma_loop(const uint4* __restrict__ src, float* __restrict__ sink,
                                                    unsigned long long* __restrict__ ticks, int iters) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  bf16x8 a[8], b[8];
  for (int i = 0; i < 8; ++i) {
    uint4 u = src[(size_t)tid * 16 + i], v = src[(size_t)tid * 16 + 8 + i];
    a[i] = *reinterpret_cast<bf16x8*>(&u);
    b[i] = *reinterpret_cast<bf16x8*>(&v);
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float out = 0.0f;
  if constexpr (SHAPE == 0) {
    f32x16 acc[16];
    for (int i = 0; i < 16; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 7], b[(i + 3) & 7], acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) for (int e = 0; e < 16; ++e) out += acc[i][e];
  } else {
    // sixteen accumulators, two MFMAs per accumulator and iteration (independent chains of length 2 are far enough apart: the
    // second use of an accumulator comes 16 MFMAs = 256+ cycles after the first)
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) for (int e = 0; e < 4; ++e) acc[i][e] = 0.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 7], b[(i + 3) & 7], acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(i + 1) & 7], b[(i + 5) & 7], acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) for (int e = 0; e < 4; ++e) out += acc[i][e];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { ticks[blockIdx.x * 2] = t1 - t0; ticks[blockIdx.x * 2 + 1] = r1 - r0; }
  sink[tid] = out;
}

int main() {
  const int grid = 256, threads = 256, iters = 20000;
  const size_t n = (size_t)grid * threads * 16;
  std::vector<uint4> h(n);
  uint4* d; float* sink; unsigned long long* ticks;
  hipMalloc(&d, n * sizeof(uint4)); hipMalloc(&sink, grid * threads * 4); hipMalloc(&ticks, grid * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  printf("| operands | MFMA | ms | TFLOP/s | shader clock | cycles per MFMA per SIMD |\n|---|---|---|---|---|---|\n");
  for (int data = 0; data < 3; ++data) {
    srand(1234);
    for (size_t i = 0; i < n; ++i) {
      auto rb = [&]() -> unsigned {  // two bf16 values: random sign / mantissa, exponent near 1.0 (data 0) | small ints | zero
        if (data == 2) return 0u;
        if (data == 1) { unsigned lo = 0x3F80u + ((rand() & 1) << 15), hi = 0x4000u + ((rand() & 1) << 15); return lo | (hi << 16); }
        unsigned lo = ((rand() & 1) << 15) | ((120 + (rand() % 12)) << 7) | (rand() & 0x7F);
        unsigned hi = ((rand() & 1) << 15) | ((120 + (rand() % 12)) << 7) | (rand() & 0x7F);
        return lo | (hi << 16);
      };
      h[i] = make_uint4(rb(), rb(), rb(), rb());
    }
    hipMemcpy(d, h.data(), n * sizeof(uint4), hipMemcpyHostToDevice);
    for (int shape = 0; shape < 2; ++shape) {
      for (int rep = 0; rep < 3; ++rep) {  // warm the power state; the last repetition is reported
        hipEventRecord(e0);
        if (shape == 0) hipLaunchKernelGGL(mfma_loop<0>, dim3(grid), dim3(threads), 0, 0, d, sink, ticks, iters);
        else hipLaunchKernelGGL(mfma_loop<1>, dim3(grid), dim3(threads), 0, 0, d, sink, ticks, iters * 2 / 2);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long t[2]; hipMemcpy(t, ticks, 16, hipMemcpyDeviceToHost);
      const double per_wave = (shape == 0 ? 16.0 * 32 * 32 * 16 * 2 : 32.0 * 16 * 16 * 32 * 2) * iters;
      const double flop = per_wave * grid * 4;
      const double n_mfma = (shape == 0 ? 16.0 : 32.0) * iters;
      printf("| %s | %s | %.3f | %.0f | %.0f MHz | %.1f |\n", data == 0 ? "random bf16" : data == 1 ? "+-1 / +-2" : "zeros",
             shape == 0 ? "32x32x16" : "16x16x32", ms, flop / ms / 1e9, (double)t[0] / t[1] * 100.0, (double)t[0] / n_mfma);
    }
  }
  return 0;
}
