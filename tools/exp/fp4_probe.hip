// Probe (round 6): do gfx950's scaled FP4 converters reproduce the reference's E2M1 quantize-dequantize?
//   q  = v_cvt_scalef32_pk_fp4_f32(x0, x1, S)   (two f32 -> two E2M1 nibbles)
//   y  = v_cvt_scalef32_pk_f32_fp4(q, S)        (back to f32)
// against  sign(x) * (round_E2M1(|x| * 2^-k) * 2^k)  (moq_mx.h: mx_qdq, tensor_quant_mx.cu:36-55) for EVERY bf16 pattern as input
// and block exponents k in [-126, 126], S = 2^k.  Prints the number of mismatching (pattern, k) pairs by class.
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/exp/fp4_probe.hip -o /tmp/fp4_probe && /tmp/fp4_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <math.h>
#include <string.h>
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ float ref_round_abs(float a) {  // E2M1: m = 1, emin = 0, max 6, RNE (mx_round_abs with f = {1, 0, 6, 0, 0})
  const int shift = 22;
  uint32_t u = __float_as_uint(a);
  const uint32_t half = 1u << (shift - 1);
  u += half - 1u + ((u >> shift) & 1u);
  u &= ~((1u << shift) - 1u);
  const float qn = __uint_as_float(u);
  const float qs = __builtin_rintf(a * 2.0f) * 0.5f;
  float q = a >= 1.0f ? qn : qs;
  q = q > 6.0f ? 6.0f : q;
  return a != a ? 6.0f : q;
}
__device__ float ref_qdq(float x, float scale, float unscale) {
  const float sign = x < 0.0f ? -1.0f : (x > 0.0f ? 1.0f : 0.0f);
  return sign * (ref_round_abs(__builtin_fabsf(x) * scale) * unscale);
}
__global__ void probe(unsigned long long* counts, uint32_t* first) {
  // blockIdx.x = k + 126 ; threads sweep the 65536 patterns
  const int k = (int)blockIdx.x - 126;
  const float S = __builtin_ldexpf(1.0f, k), inv = __builtin_ldexpf(1.0f, -k);
  for (uint32_t p = threadIdx.x; p < 65536u; p += blockDim.x) {
    const float x = __uint_as_float(p << 16);
    const uint32_t q = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(0u, x, x, S, 0);
    const f2 y = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(q, S, 0);
    const float want = ref_qdq(x, inv, S);
    const uint32_t a = __float_as_uint(y.x), b = __float_as_uint(want);
    if (a == b) continue;
    int cls;
    if (x != x) cls = 0;                                   // NaN input
    else if ((p & 0x7FFFu) == 0) cls = 1;                  // +-0 input
    else if ((p & 0x7F80u) == 0x7F80u) cls = 2;            // inf
    else if ((a & 0x7FFFFFFFu) == (b & 0x7FFFFFFFu)) cls = 3;  // sign only
    else if (k < -100 || k > 100) cls = 4;                 // extreme block exponent
    else cls = 5;                                          // a real difference
    atomicAdd(&counts[cls], 1ull);
    if (cls == 5 && atomicAdd(&first[0], 1u) < 16u) {
      const uint32_t slot = atomicAdd(&first[1], 1u);
      if (slot < 16) { first[2 + 4 * slot] = p; first[3 + 4 * slot] = (uint32_t)(k + 126); first[4 + 4 * slot] = a; first[5 + 4 * slot] = b; }
    }
  }
}
int main() {
  unsigned long long* c; uint32_t* f;
  hipMalloc(&c, 8 * 8); hipMalloc(&f, 4 * 80);
  hipMemset(c, 0, 64); hipMemset(f, 0, 320);
  hipLaunchKernelGGL(probe, dim3(253), dim3(256), 0, 0, c, f);
  unsigned long long hc[8]; uint32_t hf[80];
  hipMemcpy(hc, c, 64, hipMemcpyDeviceToHost); hipMemcpy(hf, f, 320, hipMemcpyDeviceToHost);
  printf("pairs tested: %d x 65536\n", 253);
  const char* names[] = {"NaN input", "+-0 input", "inf input", "sign only (nonzero input)", "extreme block exponent |k| > 100", "REAL difference"};
  for (int i = 0; i < 6; ++i) printf("  mismatches, %-36s: %llu\n", names[i], hc[i]);
  for (uint32_t s = 0; s < hf[1] && s < 16; ++s) {
    float x, a, b; uint32_t xb = hf[2 + 4 * s] << 16;
    memcpy(&x, &xb, 4); memcpy(&a, &hf[4 + 4 * s], 4); memcpy(&b, &hf[5 + 4 * s], 4);
    printf("    x = %g (0x%04x) k = %d : hw %g  want %g\n", x, hf[2 + 4 * s], (int)hf[3 + 4 * s] - 126, a, b);
  }
  return 0;
}
