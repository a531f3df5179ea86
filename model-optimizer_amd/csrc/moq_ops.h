// moq_ops.h -- per-element quantize-dequantize operators shared by the streaming kernels (moq_stream.hip) and the fused
// input-quantizer pass (moq_inputq.hip).
#pragma once

#include "moq_common.h"

namespace moq {

// ------------------------------------------------------------------------------------------------
// Per-element operators applied by the chunk loop
// ------------------------------------------------------------------------------------------------
struct OpIntQdq {  // a6 with one scalar amax
  float scale;
  IntQ q;
  SharedDiv sd;
  __device__ __forceinline__ void set(float amax) {
    scale = int_scale(amax, q.hi);
    sd = make_shared_div(scale);
  }
  __device__ __forceinline__ void operator()(float* f, int n) const {
    if (scale != 0.0f && sd.fast) {
      // the ordinary case, tested ONCE per packet (the scale is the tensor's): same arithmetic as qdq_int_shared
      // without its two uniform branches per element
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (i < n) {
          const float p = f[i] * scale;
          float t = __builtin_rintf(p);
          t = t < q.lo ? q.lo : t;
          t = __builtin_fminf(t, q.hi);
          t = (p != p) ? p : t;
          f[i] = shared_div_in_window(t, sd);
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (i < n) f[i] = qdq_int_shared(f[i], scale, sd, q);
    }
  }
};
struct OpFp8Qdq {  // a7 with one scalar amax
  Fp8Scale sc;
  __device__ __forceinline__ void operator()(float* f, int n) const {
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
      if (i < n) {
        float a = f[i] * sc.s, b = f[i + 1] * sc.s;
        // clamp(+-448) then RNE cast; NaN survives fmin/fmax-free med3 style clamp by re-injection
        float ca = __builtin_fminf(__builtin_fmaxf(a, -448.0f), 448.0f);
        float cb = __builtin_fminf(__builtin_fmaxf(b, -448.0f), 448.0f);
        ca = (a != a) ? a : ca;
        cb = (b != b) ? b : cb;
        float ra, rb;
        e4m3_roundtrip2(ca, cb, ra, rb);
        f[i] = ra * sc.inv;
        f[i + 1] = rb * sc.inv;
      }
    }
  }
};
struct OpFp8Cast {  // a7 with amax=None: plain (non-saturating) e4m3fn cast, |x| > 464 -> NaN like torch
  __device__ __forceinline__ void operator()(float* f, int n) const {
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
      if (i < n) {
        float a = f[i], b = f[i + 1];
        float ra, rb;
        // 464 = midpoint between 448 and the (non-existent) next value 480: RNE ties-to-even keeps 448
        e4m3_roundtrip2(__builtin_fminf(__builtin_fmaxf(a, -448.0f), 448.0f),
                        __builtin_fminf(__builtin_fmaxf(b, -448.0f), 448.0f), ra, rb);
        const float nanv = __uint_as_float(0x7FC00000u);
        f[i] = (__builtin_fabsf(a) > 464.0f || a != a) ? nanv : ra;
        f[i + 1] = (__builtin_fabsf(b) > 464.0f || b != b) ? nanv : rb;
      }
    }
  }
};

}  // namespace moq
