// moq_hist.h -- the |x| binning rule of torch.histc shared by the histogram kernel (moq_formats.hip) and the fused
// input-quantizer pass (moq_inputq.hip).
#pragma once

#include "moq_common.h"

namespace moq {

// Branch-free binning: one LDS atomic per element, invalid elements (outside [0, max_edge], NaN, skipped zeros,
// past the end) go to a trash slot instead of around a branch -- the exec-mask juggling of a guarded atomic costs
// more issue slots than the atomic itself.  SHARED selects the shared-denominator division (bit-identical to `/`
// while |a * bins| <= 2^16, checked by the caller): five full-rate FMAs instead of the IEEE sequence per element.
// LDS layout: R = 2^rshift interleaved copies of the histogram (copy = lane & (R - 1), slot = bin * R + copy) plus R
// trash slots at bin index `bins`: activations pile up in a few low bins and same-address LDS atomics of one wave
// serialise -- R copies cut that R-fold and spread a hot bin over R banks.
template <bool SHARED>
__device__ __forceinline__ int hist_bin(float a, int bins, float max_edge, const SharedDiv& sd, int skip_zeros) {
  // torch.histc: pos = (int)((v - min) * bins / (max - min)) in fp32, v == max -> last bin, outside -> skip
  const float num = a * (float)bins;
  float qf;
  if constexpr (SHARED) {  // shared_div without its range check (the host selected this instantiation)
    const float q0 = num * sd.y;
    const float r0 = __builtin_fmaf(-sd.d, q0, num);
    const float q1 = __builtin_fmaf(r0, sd.y, q0);
    const float r1 = __builtin_fmaf(-sd.d, q1, num);
    qf = __builtin_fmaf(r1, sd.y, q1);
  } else {
    qf = num / max_edge;
  }
  int pos = (int)qf;
  pos = pos < bins - 1 ? pos : bins - 1;
  // bitwise, not short-circuit: no exec-mask branches.  a <= max_edge also drops NaN.
  const bool ok = (a <= max_edge) & !((skip_zeros != 0) & (a == 0.0f));
  return ok ? pos : bins;
}
}  // namespace moq
