/*
 * moquant.h -- C-ABI of libmoquant.so: the MI355X (gfx950) PTQ calibration / quantize-dequantize hot path.
 *
 * This is the drop-in boundary.  Every entry point names the reference interface
 * it replaces (paths relative to the reference checkout of NVIDIA/Model-Optimizer).  Conventions, shared
 * by all functions:
 *
 *   - plain pointers + sizes, no torch types.  Pointers named x / y / w / out / amax ... are DEVICE
 *     pointers (HBM) unless the comment says "host".  The caller owns and allocates every buffer
 *     (the reference's pybind layer allocates with empty_like -- tensor_quant.cpp:50,59; here the Python
 *     adapter does that and hands over data_ptr()).
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream).  Kernels are enqueued on it and the
 *     call returns without synchronising (reference: c10::cuda::getCurrentCUDAStream(),
 *     tensor_quant_gpu.cu:78,90,131).
 *   - return value: MOQ_OK (0) or a negative moq_status; moq_last_error() gives a thread-local message.
 *     Unsupported layouts return MOQ_ERR_UNSUPPORTED so that the adapter can raise ValueError, which the
 *     reference's callers treat as "fall back to eager" (tensor_quant.py:386-389).
 *   - dtype codes follow moq_dtype.  All arithmetic is fp32 as in the reference kernels; results are
 *     rounded to the storage dtype with round-to-nearest-even.
 */
#ifndef MOQUANT_H_
#define MOQUANT_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MOQ_ABI_VERSION 1

typedef enum moq_status {
  MOQ_OK = 0,
  MOQ_ERR_INVALID = -1,     /* bad argument (null pointer, negative size, unknown enum) */
  MOQ_ERR_UNSUPPORTED = -2, /* valid request that this build has no kernel for          */
  MOQ_ERR_LAUNCH = -3       /* HIP reported an error at launch                          */
} moq_status;

typedef enum moq_dtype { MOQ_F32 = 0, MOQ_F16 = 1, MOQ_BF16 = 2 } moq_dtype;

/* Element/scale formats of the MX path; numbering mirrors `enum class Types`
 * (modelopt/torch/kernels/quantization/gemm/tensor_quant_mx.h:39). */
typedef enum moq_mx_type {
  MOQ_E4M3 = 0, MOQ_E5M2 = 1, MOQ_INT8 = 2, MOQ_E0M3 = 3, MOQ_E1M2 = 4,
  MOQ_E3M0 = 5, MOQ_E2M1 = 6, MOQ_E3M2 = 7, MOQ_E2M3 = 8, MOQ_E8M0 = 9
} moq_mx_type;

/* How an amax array maps onto the elements of a contiguous tensor. */
typedef enum moq_amax_mode {
  MOQ_AMAX_SCALAR = 0, /* amax[0] for every element                                     */
  MOQ_AMAX_AXIS = 1    /* amax[(i / inner) % axis_size]; per-group: inner=g, axis_size=n/g */
} moq_amax_mode;

/* Rounding used by the real INT4 packer (the reference has two twins, see moq_int4_pack). */
typedef enum moq_round { MOQ_ROUND_HALF_EVEN = 0, MOQ_ROUND_HALF_AWAY = 1 } moq_round;

int moq_abi_version(void);
const char* moq_last_error(void);
/* Device facts used by the host for grid sizing / reporting: CU count of the current device. */
int moq_device_cu_count(void);

/* ------------------------------------------------------------------ amax (a1, a2, a3) */

/* Per-tensor abs-max: out[0] = max|x| as fp32 (NaN if any element is NaN -- torch.max propagates NaN,
 * quantization/utils/core_utils.py:172-174).  accumulate != 0 fuses MaxCalibrator's running max
 * (calib/max.py:79-83): out[0] = max(out[0], max|x|); out must then hold a non-negative float or NaN.
 * Replaces reduce_amax(x, axis=None) -- core_utils.py:146-183. */
int moq_amax(const void* x, int64_t n, int dt, float* out, int accumulate, void* stream);

/* Abs-max over every dimension except one.  x is viewed as contiguous [outer, axis_size, inner];
 * out[a] = max over (o, i) of |x[o, a, i]|.  Covers per-channel weights (outer=1, axis_size=Cout,
 * inner=Cin), per-channel activations axis=-1 (outer=tokens, axis_size=Cin, inner=1) and static
 * per-group amax of a (-1, g) view (outer=1, axis_size=n/g, inner=g).
 * Replaces reduce_amax(x, axis=reduce_axes) -- core_utils.py:175-182 -- and, with accumulate,
 * MaxCalibrator.collect (calib/max.py:52-86). */
int moq_amax_axis(const void* x, int64_t outer, int64_t axis_size, int64_t inner, int dt, float* out,
                  int accumulate, void* stream);

/* ------------------------------------------------------------------ INT-k fake quant (a6) */

/* y = clamp(rint(x * s), lo, hi) / s with s = bound / amax, bound = 2^(bits-1+unsigned) - 1,
 * hi = bound, lo = unsigned ? 0 : -(bound + !narrow); amax <= 2^-24 gives 0.  fp32 math, IEEE division.
 * Replaces cuda_ext.fake_tensor_quant / fake_tensor_quant_with_axis (tensor_quant.cpp:40-61,
 * tensor_quant_gpu.cu:43-140) and eager _tensor_quant (quantization/tensor_quant.py:607-645).
 * x == y (in place) is allowed (fake_tensor_quant_). */
int moq_fake_quant_int(const void* x, void* y, int64_t n, int dt, const float* amax, int amax_mode,
                       int64_t axis_size, int64_t inner, int num_bits, int is_unsigned,
                       int narrow_range, void* stream);

/* Fused dynamic per-group abs-max + INT-k QDQ over a contiguous (n_groups, g) view -- ONE read, ONE
 * write of the tensor.  amax_out[n_groups] receives the fp32 group amax (may be NULL).
 * This is what TensorQuantizer._get_amax + fake_tensor_quant do back to back for a static-block
 * quantizer without an _amax buffer (nn/modules/tensor_quantizer.py:736-751, :937), i.e. the inner
 * loop of the AWQ search, and max_calibrate + one forward for INT4 g=128 weights. */
int moq_amax_qdq_int_group(const void* x, void* y, float* amax_out, int64_t n_groups, int g, int dt,
                           int num_bits, int is_unsigned, int narrow_range, void* stream);

/* ------------------------------------------------------------------ FP8-E4M3 fake quant (a7) */

/* amax != NULL: s = 448 / (amax <= 2^-24 ? 1 : amax); y = e4m3fn_rne(clamp(x*s, +-448)) * (1/s).
 * amax == NULL: y = e4m3fn_rne(x) (torch's non-saturating cast: |x| > 464 -> NaN).
 * Replaces cuda_ext_fp8.fake_e4m3fy / fake_e4m3fy_with_axis (tensor_quant_gpu_fp8.cu:35-107) and
 * eager _fp8_eager (quantization/tensor_quant.py:46-59). */
int moq_fake_quant_e4m3(const void* x, void* y, int64_t n, int dt, const float* amax, int amax_mode,
                        int64_t axis_size, int64_t inner, void* stream);

/* ------------------------------------------------------------------ multi-tensor weight passes */

/* One segment = one weight tensor living anywhere in HBM.  A table of segments (device memory) lets a
 * single launch calibrate / QDQ a whole layer or model (launch gaps otherwise dominate 8-100 MB
 * tensors at HBM speed).  `blk_start` (device, n_seg+1 entries) is the exclusive prefix sum of
 * ceil(n / MOQ_MT_CHUNK) per segment, built by moq_mt_plan on the host. */
#define MOQ_MT_CHUNK 8192 /* elements per workgroup-chunk */
typedef struct moq_seg {
  const void* x; /* input tensor                                    */
  void* y;       /* QDQ output (may equal x)                        */
  float* amax;   /* per-tensor: [1]; per-group: [n/g]               */
  int64_t n;     /* elements                                        */
} moq_seg;

/* host helper: fills blk_start_host[n_seg+1]; returns total chunk count or negative status. */
int64_t moq_mt_plan(const int64_t* n_host, int n_seg, int64_t* blk_start_host);

/* Per-tensor abs-max of every segment: segs[s].amax[0] = max|x_s| (a1 over a tensor list;
 * model_calib.py:187-199 weight_only_quantize loop).  Zeroes the amax slots first. */
int moq_mt_amax(const moq_seg* segs, const int64_t* blk_start, int n_seg, int64_t n_chunks, int dt,
                void* stream);
/* The same result through a caller-provided scratch of n_chunks floats: one value per chunk, no atomics, memory swept
 * as one dense window (read-only stream at ~7 TB/s instead of ~6.3), then one fold per tensor. */
int moq_mt_amax_ws(const moq_seg* segs, const int64_t* blk_start, int n_seg, int64_t n_chunks, int dt,
                   float* chunk_scratch, void* stream);
/* RUNNING per-tensor abs-max of many tensors for many calibrators in one sweep -- the collect side of a max-calibration
 * forward loop (MaxCalibrator.collect, calib/max.py:52-86: `self._calib_amax = torch.max(self._calib_amax, local_amax)`,
 * called from TensorQuantizer.forward, nn/modules/tensor_quantizer.py:1186-1196, once per quantizer per batch).  The
 * reference launches its reductions per quantizer per call; here the activations one decoder layer hands its quantizers
 * (q / k / v read ONE tensor, gate / up another: a tensor appears once in `segs` whatever the number of quantizers that
 * read it) are swept in one dense window (stage 1 of moq_mt_amax_ws), and fold f then merges segment folds[f].seg's
 * maximum into the calibrator's running value *folds[f].dst (fp32; bit-pattern maximum: NaN sticks, as in moq_amax with
 * accumulate).  segs[s].y / .amax are not read. */
typedef struct moq_amax_fold {
  float* dst;  /* a calibrator's running abs-max (device, fp32 [1]) */
  int64_t seg; /* index into segs                                   */
} moq_amax_fold;
int moq_mt_amax_running(const moq_seg* segs, const int64_t* blk_start, int n_seg, int64_t n_chunks, int dt,
                        float* chunk_scratch, const moq_amax_fold* folds, int n_folds, void* stream);
/* cuda_ext_mx.convert_to_exmy (tensor_quant_mx.cu:398 -> convert_to_types, tensor_quant_mx.h:163-186): y[i] = the value
 * of element format `fmt` nearest to x[i] (format's own tie rule, saturating), no scaling.  fp32 in / out. */
int moq_mx_convert(const float* x, float* y, int64_t n, int fmt, void* stream);
/* FP8QTensor.quantize / dequantize with block_sizes on BOTH axes of a 2-D tensor (qtensor/fp8_tensor.py:60-112, :114-151;
 * the FP8 2-D blockwise weight-only export, export/quant_utils.py:874-877): scales is [rows/br, cols/bc] row-major.
 * pack: byte = e4m3fn(x / scale); scale_dt == dt: the quotient is rounded to dt first (same-dtype division);
 * scale_dt == MOQ_F32 with a 16-bit x: fp32 quotient (torch promotes a dimensioned fp32 operand).
 * unpack: out = dt(q) * scale, scales in dt. */
int moq_fp8_pack_tile(const void* x, const void* scales, int scale_dt, uint8_t* out, int64_t rows, int64_t cols,
                      int br, int bc, int dt, void* stream);
int moq_fp8_unpack_tile(const uint8_t* q, const void* scales, void* out, int64_t rows, int64_t cols, int br, int bc,
                        int dt, void* stream);
/* out[o, i] = max_b |x[o, b, i]| for x viewed as [outer, mid, inner] (one step of reduce_block_amax,
 * quantization/utils/core_utils.py:43-90, on a dim that is not the last one). */
int moq_amax_mid(const void* x, int64_t outer, int64_t mid, int64_t inner, int dt, float* out, void* stream);
/* a4 `calibrate_weights` (calib/histogram.py:346-433): counts[r, b] += number of |x[r, :]| in bin b of
 * np.histogram(|x[r]|, bins, range=(first[r], last[r])) -- numpy's float32 edges and closed last bin, bit for bit.
 * counts is int32 [rows, bins] and is OVERWRITTEN (it may be uninitialised memory: the call clears what it accumulates
 * into). */
int moq_row_hist_np(const void* x, int64_t rows, int64_t cols, int dt, int bins, const float* first,
                    const float* last, int* counts, void* stream);
/* Per-tensor FP8-E4M3 QDQ of every segment with its own amax (a7 over a tensor list). */
int moq_mt_fake_quant_e4m3(const moq_seg* segs, const int64_t* blk_start, int n_seg, int64_t n_chunks,
                           int dt, void* stream);
/* Per-tensor INT-k QDQ of every segment with its own amax (a6 over a tensor list). */
int moq_mt_fake_quant_int(const moq_seg* segs, const int64_t* blk_start, int n_seg, int64_t n_chunks,
                          int dt, int num_bits, int is_unsigned, int narrow_range, void* stream);
/* Fused per-group amax + INT-k QDQ of every segment (n % g == 0 per segment). */
int moq_mt_amax_qdq_int_group(const moq_seg* segs, const int64_t* blk_start, int n_seg,
                              int64_t n_chunks, int g, int dt, int num_bits, int is_unsigned,
                              int narrow_range, void* stream);

/* 2:4 magnitude mask (see moq_mask_2to4) of every segment in one launch: segs[s].y is the uint8 mask [n] of the
 * weight segs[s].x (groups of 4 run along the flattened tensor: every row length % 4 == 0). */
int moq_mt_mask_2to4(const moq_seg* segs, const int64_t* blk_start, int n_seg, int64_t n_chunks, int dt,
                     void* stream);
/* 2:4 magnitude mask of every segment AND its application in one pass: segs[s].y receives the uint8 mask like
 * moq_mt_mask_2to4, segs[s].x (the weight) is rewritten IN PLACE as dtype(w * mask) -- what mts.sparsify leaves behind
 * a SparseModule's weight access (sparsity/weight_sparsity/module.py:97-101, weight * mask: a pruned negative weight
 * becomes -0.0, inf / NaN times 0 is NaN, as the tensor product gives).  chunk_scratch != NULL (uint32 [n_chunks]):
 * segs[s].amax[0] additionally receives the abs-max of the MASKED weight (the per-tensor statistic a calibration that
 * follows would read the model again for).  5 bytes per 16-bit element instead of 3 + 6 (mask, then a mask-typed
 * multiply) + 2. */
int moq_mt_mask_2to4_apply(const moq_seg* segs, const int64_t* blk_start, int n_seg, int64_t n_chunks, int dt,
                           void* chunk_scratch, void* stream);
/* MX dynamic-block QDQ (see moq_mx_fused_amax_convert) of every segment in one launch; block in
 * {kVec, 2 kVec, 4 kVec, 8 kVec} elements (8..64 for 16-bit types), every segment 16-byte aligned with
 * n % block == 0; E8M0 scales.  segs[s].amax is not used. */
int moq_mt_mx_fused_amax_convert(const moq_seg* segs, const int64_t* blk_start, int n_seg, int64_t n_chunks,
                                 int block, int dt, int fmt, void* stream);
/* SmoothQuant's weight fold composed with the dynamic MX block QDQ, every segment in ONE launch, one read and one write
 * per element:  y_s = MXQDQ_{block, fmt}( dt( x_s * scale_s[col] ) )  -- _apply_weight_pre_quant_scale
 * (quantization/model_calib.py:1208-1216: `(weight * pre_quant_scale[None, :]).to(weight.dtype)`, fp32 product, one
 * rounding) followed by the weight quantizer's MX fake quantization (tensor_quant_mx.cu:239-294, E8M0 block scales) -- the
 * MX twin of moq_awq_scale_qdq, bit-equal to moq_scale_cols followed by moq_mt_mx_fused_amax_convert (BASELINE configs[4]:
 * Llama-3-70B MXFP4 g32 + SmoothQuant moved 8 B/element that way, 4 here).  side[s] rides beside segs[s]:
 * scale = the segment's fp32 column vector [cols] (NULL: no fold for this segment), cols = its row length
 * (cols % block == 0).  Same layout rules as moq_mt_mx_fused_amax_convert; x == y allowed. */
typedef struct moq_fold_seg {
  const float* scale; /* fp32 [cols] multiplied into every row before the block QDQ / pack; NULL = none */
  int64_t cols;       /* row length of the segment's tensor                                              */
  uint8_t* e8m0;      /* moq_mt_fold_mxfp4_pack: the segment's E8M0 block scales [n / block]; else unused */
} moq_fold_seg;
int moq_mt_fold_mx_fused(const moq_seg* segs, const int64_t* blk_start, const moq_fold_seg* side, int n_seg,
                         int64_t n_chunks, int block, int dt, int fmt, void* stream);
/* The PTQ-to-checkpoint form of the same pass: MXFP4QTensor.quantize (qtensor/mxfp4_tensor.py:37-81, as moq_mxfp4_pack)
 * of dt(x_s * scale_s[col]) for every segment in one launch -- segs[s].y = the packed E2M1 nibbles (uint8 [n / 2]),
 * side[s].e8m0 = the block exponents (uint8 [n / block]).  2 B read, 0.5 + 1/block B written per element. */
int moq_mt_fold_mxfp4_pack(const moq_seg* segs, const int64_t* blk_start, const moq_fold_seg* side, int n_seg,
                           int64_t n_chunks, int block, int dt, void* stream);

/* ------------------------------------------------------------------ MX dynamic block QDQ (a8) */

/* Per `block` consecutive elements of the last dim (cols virtually right-padded with zeros to a block
 * multiple): block amax -> E8M0 scale 2^ceil(log2(amax/fmt_max)) -> y = sign * round_fmt(|x|*2^-e) * 2^e.
 * Replaces cuda_ext_mx.fused_amax_convert (tensor_quant_mx.cu:239-387, tensor_quant_mx.h:39-245).
 * scale_fmt == MOQ_E8M0 (MX formats): no global_amax.  scale_fmt = an element format (MOQ_E4M3 for NVFP4-style
 * configs): compute_scale (tensor_quant_mx.cu:139-152) or, with global_amax (one fp32 value, the tensor-wide amax),
 * compute_scale_with_global (:154-183).  E8M0 with a global amax -> MOQ_ERR_UNSUPPORTED. */
int moq_mx_fused_amax_convert(const void* x, void* y, int64_t rows, int64_t cols, int block, int dt,
                              int fmt, int scale_fmt, const float* global_amax, void* stream);

/* ------------------------------------------------------------------ histogram (a4) */

/* counts[b] += #{ i : bin(|x_i|) == b }, bin = (int)(|x| * bins / max_edge) computed in fp32 with the
 * last bin closed (|x| == max_edge -> bins-1); |x| > max_edge and NaN are dropped; skip_zeros drops
 * exact zeros.  This is torch.histc(|x|.float(), bins, min=0, max=max_edge) as used by
 * HistogramCalibrator.collect (calib/histogram.py:77-130), with exact 64-bit integer counts. */
int moq_hist_abs(const void* x, int64_t n, int dt, unsigned long long* counts, int bins,
                 float max_edge, int skip_zeros, void* stream);

/* The threshold searches over a collected histogram, on the device.
 * Entropy (_compute_amax_entropy, calib/histogram.py:210-283): divergences[c] = KL divergence of candidate
 * i = start_bin + c * stride (clip after source bin i - 1; c = 0 .. (n_bins - start_bin) / stride) in fp64, with the
 * reference's bins[0] = bins[1] substitution; num_quant_bins = 1 << (num_bits - 1 + unsigned), a power of two <= 4096.
 * The caller takes the LAST minimum and returns calib_bin_edges[i]; candidates within 1e-9 of the minimum are to be
 * re-scored with the reference's own summation order for a bit-exact choice (the Python host does). */
int moq_hist_entropy(const int64_t* hist, int64_t n_bins, int num_quant_bins, int start_bin, int stride,
                     double* divergences, void* stream);
/* Percentile (_compute_amax_percentile, calib/histogram.py:326-343; per row for calibrate_weights, :400-412):
 * idx[r] = np.searchsorted(np.cumsum(hist[r] / hist[r].sum()), q), q = percentile / 100, the running sum formed
 * sequentially in fp64 like np.cumsum -- bit-exact indices.  hist is int32 (elem_bytes 4) or int64 (8) [rows, bins];
 * idx = bins when the sum never reaches q. */
int moq_hist_percentile(const void* hist, int elem_bytes, int64_t rows, int64_t bins, double q, int64_t* idx,
                        void* stream);

/* INT8 checkpoint weights: out[r, c] = (int8) clamp(rint(w[r, c] / scale[r]), -128, 127), fp32 quotient, round half to
 * even -- to_quantized_weight for W8A8_SQ_PER_CHANNEL / INT8 weight-only (export/quant_utils.py:868-869) with the fp32
 * per-output-channel scaling factor of get_weight_scaling_factor.  Needs cols % (16 / sizeof(elem)) == 0. */
int moq_int8_pack_rows(const void* w, const float* scale, int8_t* out, int64_t rows, int64_t cols, int dt,
                       void* stream);

/* ------------------------------------------------------------------ fused input-quantizer pass (8f-3) */

/* TensorQuantizer.forward for a per-tensor input quantizer in ONE read of the activation x[rows, cols]
 * (nn/modules/tensor_quantizer.py:1119-1221); every stage is optional:
 *   pre_quant_scale != NULL : v = dtype(x * pre_quant_scale[col])  (:1143-1144; fp32 [cols] holding model-dtype values)
 *   amax_running   != NULL  : amax_running[0] = max(amax_running[0], max |v|)   (collect -> MaxCalibrator, calib/max.py:63-85;
 *                             NaN propagates like torch.max)
 *   hist_counts    != NULL  : hist_counts[bin(|v|)] += 1 with moq_hist_abs' binning over [0, hist_max_edge]
 *                             (HistogramCalibrator.collect with a known range, calib/histogram.py:95-130); 1 <= bins < 16384
 *   fmt 1 / 2               : y = INT-k / FP8-E4M3 quantize-dequantize of v with the per-tensor qdq_amax[0]
 *                             (moq_fake_quant_int / moq_fake_quant_e4m3 arithmetic, tensor_quant.py:607-645, :46-59)
 *   fmt 0                   : y = v when y != NULL and a pre_quant_scale is given (the calibration pass hands the scaled
 *                             activation on to the linear), otherwise nothing is written.
 * y may alias x.  pre_quant_scale needs cols % (16 / sizeof(elem)) == 0. */
int moq_input_quant(const void* x, const float* pre_quant_scale, void* y, int64_t rows, int64_t cols, int dt,
                    float* amax_running, const float* qdq_amax, int fmt, int num_bits, int is_unsigned,
                    int narrow_range, unsigned long long* hist_counts, int hist_bins, float hist_max_edge,
                    int hist_skip_zeros, void* stream);

/* ------------------------------------------------------------------ MSE amax sweep (a5) */

/* loss[k, a] (+)= sum over the elements governed by amax entry a of (x - QDQ(x, cand_amax[k, a]))^2 for all
 * n_cand (<= 64) candidates in ONE read of x.  x is viewed as contiguous [outer, axis_size, inner] exactly like
 * moq_amax_axis (per-tensor: axis_size = 1; per-channel: inner = Cin; static blocks: outer = 1, inner = g).
 * Arithmetic as MseCalibrator.collect with the quantizer's fake quant as quant_func (calib/mse.py:83-121,
 * model_calib.py:639-662): x upcast to fp32, QDQ in fp32 (fp8 != 0: E4M3, else INT-k with num_bits /
 * is_unsigned / narrow_range), squared error summed in fp32.  cand_amax, loss: fp32 [n_cand, axis_size].
 * `partial`: workspace of moq_mse_sweep_workspace() floats (may be NULL for the static-block layout). */
int64_t moq_mse_sweep_workspace(int64_t outer, int64_t axis_size, int64_t inner, int n_cand);
int moq_mse_sweep(const void* x, int64_t outer, int64_t axis_size, int64_t inner, int dt,
                  const float* cand_amax, int n_cand, float* loss, float* partial, int accumulate, int fp8,
                  int num_bits, int is_unsigned, int narrow_range, void* stream);

/* ------------------------------------------------------------------ 2:4 mask (a14) */

/* mask[r, c] in {0,1}: for every 4 consecutive elements of a row keep the 2-of-4 pattern with the
 * largest |w| sum, first-max tie break in the reference's pattern order.  cols % 4 == 0.
 * Replaces create_asp_mask / mn_1d_best (sparsity/weight_sparsity/magnitude.py:68-128). */
int moq_mask_2to4(const void* w, int64_t rows, int64_t cols, int dt, uint8_t* mask, void* stream);

/* ------------------------------------------------------------------ real INT4 (a15) + export pack */

/* q = round(x * scale[i/g]) clamped to [-8,7], byte = ((q_even+8) << 4) | (q_odd+8); arithmetic in the
 * storage dtype like the reference (scales have dtype dt).  rounding = MOQ_ROUND_HALF_EVEN restates the
 * eager twin (qtensor/int4_tensor.py:69-84), MOQ_ROUND_HALF_AWAY the CUDA kernel
 * (tensor_quant_gpu.cu:310-340: clamp first, then roundf(v + 8)).  n % g == 0, g even. */
int moq_int4_pack(const void* x, const void* scales, uint8_t* out, int64_t n, int g, int dt,
                  int rounding, void* stream);
/* out[2b] = ((q[b] >> 4) - 8) / scale, out[2b+1] = ((q[b] & 15) - 8) / scale, in dtype dt.
 * Replaces cuda_ext.INT4_dequantize (tensor_quant_gpu.cu:261-308). */
int moq_int4_unpack(const uint8_t* q, const void* scales, void* out, int64_t n_bytes, int g, int dt,
                    void* stream);
/* Checkpoint packer: q = clamp(rint(w[r,c] / wsf[r, c/g]), -8, 7) (fp32 division),
 * out[r/2, c] = (q[2*(r/2), c] & 15) | (q[2*(r/2)+1, c] << 4).  wsf is fp32 [rows, cols/g].
 * Replaces export/quant_utils.py:792-833 pack_int4_in_uint8 (2-D case). */
int moq_int4_pack_export(const void* w, const float* wsf, uint8_t* out, int64_t rows, int64_t cols,
                         int g, int dt, void* stream);

/* ------------------------------------------------------------------ AWQ / SmoothQuant helpers (a11, a12) */

/* y[r,c] = dtype(w[r,c] * s[c]) with an fp32 multiply (s fp32) -- _apply_weight_pre_quant_scale
 * (quantization/model_calib.py:1208-1216).  x == y allowed. */
int moq_scale_cols(const void* w, const float* s, void* y, int64_t rows, int64_t cols, int dt,
                   void* stream);
/* y[r,c] = dtype((w[r,c] * mul[c]) / div[c]) in fp32 (mul, div fp32 [cols]) -- _update_pre_quant_scale of the
 * checkpoint export's resmooth step: W * old_pre_quant_scale / new_pre_quant_scale
 * (export/quant_utils.py:1285-1296).  x == y allowed. */
int moq_rescale_cols(const void* w, const float* mul, const float* div, void* y, int64_t rows, int64_t cols,
                     int dt, void* stream);
/* y[a, r, c] = dtype(w[r,c] * s[a, c]) for a < n_scales (s fp32 [n_scales, cols]): one read, n_scales writes
 * -- the pre-scaled inputs x * (1/s_alpha) of all AWQ candidates (input_quantizer.pre_quant_scale,
 * nn/modules/tensor_quantizer.py:1143-1144 applied by model_calib.py:1552).  cols % (16 / elem size) == 0. */
int moq_scale_cols_multi(const void* w, const float* s, void* y, int64_t rows, int64_t cols, int n_scales,
                         int dt, void* stream);
/* Fused AWQ search inner op: t = dtype(w[r,c] * s[c]) (s already in dtype dt, product rounded to dt as
 * TensorQuantizer.forward does, tensor_quantizer.py:1143-1144), then per-group (g along cols) dynamic
 * amax + INT-k QDQ of t.  cols % g == 0.  Replaces model_calib.py:1552-1554 weight side. */
int moq_awq_scale_qdq(const void* w, const void* s, void* y, int64_t rows, int64_t cols, int g, int dt,
                      int num_bits, void* stream);
/* AWQ weight scale: out[c] = dtype(mean_r dtype(|w[r,c]| / dtype(gamax[r, c/g] + tiny_dtype))) as fp32,
 * gamax = abs-max of the g-column group (get_weight_scale, quantization/model_calib.py:1453-1469).
 * `partial`: caller-provided fp32 workspace of moq_col_stats_workspace(rows, cols) floats. */
int moq_awq_weight_scale(const void* w, int64_t rows, int64_t cols, int g, int dt, float* out,
                         float* partial, void* stream);
/* Column abs statistics of an activation batch x[tokens, cols]: sum_out[c] (+)= sum_t |x[t,c]| (fp32,
 * deterministic two-stage; `partial` is caller-provided fp32 workspace of moq_col_stats_workspace()
 * floats) and, if amax_out != NULL, amax_out[c] = max(amax_out[c], max_t |x[t,c]|).
 * Feeds get_act_scale (model_calib.py:1471-1472) and per-channel input amax (smoothquant :1309). */
int64_t moq_col_stats_workspace(int64_t tokens, int64_t cols);
int moq_col_abs_stats(const void* x, int64_t tokens, int64_t cols, int dt, float* sum_out,
                      float* amax_out, float* partial, int accumulate, void* stream);
/* acc[c] += (float) dtype( mean_t |x[t,c]| ): one step of awq_lite's running activation scale --
 * get_act_scale (model_calib.py:1471-1472: x.abs().mean(0) in the activation dtype, then .to(float32)) added to the
 * sum over calibration batches (:1527).  fp32 accumulation, IEEE division by `tokens`, one rounding to `dt`.
 * `partial` as for moq_col_abs_stats.  Needs a 16-byte aligned batch with cols % (16 / sizeof(elem)) == 0. */
int moq_col_abs_mean_accum(const void* x, int64_t tokens, int64_t cols, int dt, float* acc, float* partial,
                           void* stream);

/* ------------------------------------------------------------------ AWQ-lite search error GEMM (a12) */

/* One alpha step of the AWQ-lite search for one linear and one calibration batch, contraction and loss fused
 * on the matrix cores (MFMA, fp32 accumulate):
 *   out[t, n]  = dtype( sum_k x[t, k] * w[n, k]  (+ bias[n]) )            (F.linear of the patched forward)
 *   loss_acc[0] += (float) mean_{t,n} ( float(dtype(out[t, n] - out_actual[t, n])) ^ 2 )
 * which is update_loss() of the reference (quantization/model_calib.py:1489-1495) applied to the output of
 * the patched forward (:1552-1556); `out` itself is never written.  x = input * (1/awq_scale) [tokens, cin]
 * and w = QDQ(weight * awq_scale) [cout, cin] are row-major and 16-byte aligned; dt is MOQ_BF16 / MOQ_F16 (16-bit
 * matrix cores) or MOQ_F32 (fp32 models: v_mfma_f32_32x32x2_f32, every step in fp32 -- dtype() above is the identity;
 * out_actual, bias and the operands are fp32).  cin % 8 == 0, cout % 4 == 0.
 * `partial`: fp32 workspace of moq_awq_err_gemm_workspace(tokens, cout) floats (per-tile sums, reduced in a
 * fixed order: the result is deterministic).  Not bit-identical to the reference's BLAS accumulation order;
 * tolerance stated in tests/test_gpu_parity.py. */
int64_t moq_awq_err_gemm_workspace(int64_t tokens, int64_t cout);
int moq_awq_err_gemm(const void* x, const void* w, const void* out_actual, const void* bias,
                     int64_t tokens, int64_t cout, int64_t cin, int dt, float* partial, float* loss_acc,
                     void* stream);
/* All candidates of one linear in ONE grid (blockIdx.y = candidate): x[a] = x + a * x_stride elements,
 * w[a] = w + a * w_stride elements (strides multiples of 8; 0 shares the operand), loss_acc[a] += loss_a.
 * `partial`: n_cand * moq_awq_err_gemm_workspace(tokens, cout) floats.  Replaces the alpha loop of the
 * patched forward (quantization/model_calib.py:1535-1556) -- 11 launches and 11 host round trips become one. */
int moq_awq_err_gemm_multi(const void* x, const void* w, const void* out_actual, const void* bias,
                           int64_t tokens, int64_t cout, int64_t cin, int dt, int n_cand, int64_t x_stride,
                           int64_t w_stride, float* partial, float* loss_acc, void* stream);
/* The same MFMA main loops with a store epilogue: out[t, n] = dtype(sum_k x[t,k] * w[n,k] (+ bias[n])) --
 * torch.nn.functional.linear for the model dtype, bf16 / f16 / f32 (used to check the contractions on their own). */
int moq_gemm_nt(const void* x, const void* w, const void* bias, void* out, int64_t tokens, int64_t cout,
                int64_t cin, int dt, void* stream);

/* ------------------------------------------------------------------ AWQ-clip block search (a13) */

/* Per-(output channel, block) loss of every clip ratio, one pass over W (MFMA block dots):
 *   org[t]    = dt( sum_j x[t, b*g+j] * w[r, b*g+j] )
 *   cur_k[t]  = dt( sum_j x[t, b*g+j] * QDQ_int(w[r, b*g+j]; amax_k) ),  amax_k = amax_dt(amax[r, b] * shrinks[k])
 *   loss[k, b, r] += mean_t float( dt(cur_k[t] - org[t]) )^2
 * which is _clip_search's block branch of awq_clip (quantization/model_calib.py:1817-1868): the reference
 * loops over output-channel batches and shrinks, materialising [co, tokens, n_block, g] products each time.
 * x: [n_tok, cin] rows x_row_stride elements apart (the reference's token sub-sampling inputs[0::step],
 * :1820, is a stride here); w: [cout, cin]; amax: fp32 [cout, nblk] holding values of dtype amax_dt (the
 * dtype of the reference's w_amax: the weight dtype, or MOQ_F32 once the quantizer keeps fp32 amax);
 * nblk = ceil(cin / g), a ragged last block is zero padded like F.pad (:1825-1828); shrinks: device fp32
 * [n_shrink <= 32]; loss: fp32 [n_shrink, nblk, cout] (note: block-major, cout contiguous).
 * signed INT-num_bits, narrow_range False.  cin % (16 / elem size) == 0; g / (32 / elem size) in {2,4,8,16}. */
int moq_awq_clip_loss(const void* x, int64_t n_tok, int64_t x_row_stride, const void* w, int64_t cout,
                      int64_t cin, int g, int dt, const float* amax, int amax_dt, const float* shrinks,
                      int n_shrink, int num_bits, float* loss, void* stream);

/* ------------------------------------------------------------------ real FP8 / MXFP4 (a15) */

/* out[i] = e4m3fn( dt( x[i] / scale ) ): the quotient is rounded to the storage dtype first (both operands of the
 * reference's division have the model dtype), then cast with torch's NON-saturating RNE cast (|v| > 464 or NaN ->
 * 0x7F | sign).  scales: dtype scale_dt = dt (FP8QTensor: amax / 448 in the model dtype) or MOQ_F32 (checkpoint export:
 * fp32 weights_scaling_factor; the quotient is still rounded to dt); amax_mode MOQ_AMAX_SCALAR: scales[0]; MOQ_AMAX_AXIS: scales[(i / inner) %
 * axis_size] (per-channel rows: inner = Cin; 1-D blocks: inner = block, axis_size = n / block).
 * Replaces FP8QTensor.quantize's cast (quantization/qtensor/fp8_tensor.py:103-107) and the FP8 branch of
 * to_quantized_weight (export/quant_utils.py).  n % (16 / elem size) == 0, inner likewise. */
int moq_fp8_pack(const void* x, const void* scales, int scale_dt, uint8_t* out, int64_t n, int dt, int amax_mode,
                 int64_t axis_size, int64_t inner, void* stream);
/* out[i] = dt( dt(float(q[i])) * scale ) -- FP8QTensor.dequantize (fp8_tensor.py:151). */
int moq_fp8_unpack(const uint8_t* q, const void* scales, void* out, int64_t n, int dt, int amax_mode,
                   int64_t axis_size, int64_t inner, void* stream);
/* MXFP4QTensor.quantize (quantization/qtensor/mxfp4_tensor.py:37-81) over n_blocks consecutive blocks of `block`
 * elements: e = ceil(max(log2(amax / 6), -127)); e8m0[b] = e + 127; v = x / 2^e; nibble = (sign_bit << 3) +
 * #{bound < |v|} over {0.25, 0.75, 1.25, 1.75, 2.5, 3.5, 5} (strict: ties round down; an exact zero gets sign_bit
 * 1); packed[i / 2] = (nibble_odd << 4) + nibble_even.  block even. */
int moq_mxfp4_pack(const void* x, uint8_t* packed, uint8_t* e8m0, int64_t n_blocks, int block, int dt,
                   void* stream);
/* out = dt( sign * {0, .5, 1, 1.5, 2, 3, 4, 6}[nibble & 7] * 2^(e8m0 - 127) ) -- MXFP4QTensor.dequantize
 * (mxfp4_tensor.py:83-144). */
int moq_mxfp4_unpack(const uint8_t* packed, const uint8_t* e8m0, void* out, int64_t n_blocks, int block, int dt,
                     void* stream);

/* ------------------------------------------------------------------ SparseGPT */

/* y[c, r] = x[r, c] for 2-byte elements (bf16 / f16): turns an activation batch [tokens, cin] into the
 * K-contiguous operand [cin, tokens] of moq_hessian_accum. */
int moq_transpose16(const void* x, void* y, int64_t rows, int64_t cols, void* stream);
/* The same with a leading dimension for y (y[c * y_ld + r] = x[r, c], y_ld >= rows): transposes a calibration batch
 * straight into a column block of a wider [cols, y_ld] staging buffer (several batches per moq_hessian_accum launch). */
int moq_transpose16_ld(const void* x, void* y, int64_t rows, int64_t cols, int64_t y_ld, void* stream);
/* hessian[i, j] = hessian[i, j] * decay + scale * sum_t xt[i, t] * xt[j, t]   (fp32 [cin, cin], updated in place)
 * on the matrix cores: bf16 / f16 products are exact in fp32, accumulation is fp32 -- the arithmetic of the
 * reference's fp32 `inp.matmul(inp.t())` up to summation order (only the tiles on or above the diagonal are
 * contracted, the mirror images are written from the transposed accumulators).  One calibration batch of
 * SparseGPTSearcher._hook_compute_hessian (sparsity/weight_sparsity/sparsegpt.py:238-276): decay =
 * samples / (samples + new), scale = 2 / (samples + new).  xt: [cin, tokens] (dt = MOQ_BF16 | MOQ_F16, 16-byte
 * aligned, tokens % 8 == 0, cin % 4 == 0).
 * upper_only != 0: only the tiles on / above the diagonal are updated (half the contraction, no mirrored writes);
 * call moq_symmetrize once after the last batch.  upper_only == 0: the matrix is complete after every call. */
int moq_hessian_accum(const void* xt, int64_t cin, int64_t tokens, int dt, float* hessian, float decay,
                      float scale, int upper_only, void* stream);
/* h[r, c] = h[c, r] for r > c: completes an fp32 [n, n] matrix accumulated with upper_only. */
int moq_symmetrize(float* h, int64_t n, void* stream);
/* Column sweep of create_sgpt_mask over one column block [i1, i1 + bs) (sparsegpt.py:96-127), all rows at once:
 * w: fp32 [rows, ld] working weights (block overwritten with the pruned weights q), hinv: fp32 [ld, ld] upper
 * Cholesky factor of the damped inverse Hessian, delta: fp32 [rows, bs] receives err_j = (w_j - q_j) / hinv_jj.
 * Every prune_m consecutive columns lose their prune_n smallest w^2 / (hinv_kk^2 + 1e-9).  bs <= 128,
 * prune_m in {2, 4, 8}, bs % prune_m == 0.  The trailing update w[:, i2:] -= delta @ hinv[i1:i2, i2:] is
 * moq_sgpt_trailing_update. */
int moq_sgpt_block_sweep(float* w, int64_t rows, int64_t ld, int64_t i1, int bs, const float* hinv, float* delta,
                         int prune_n, int prune_m, void* stream);
/* Column sweep of GPTQ's blockwise weight update over one column block [i1, i1 + bs)
 * (quantization/utils/calib_utils.py:241-276, gptq_blockwise_update), all rows at once, for quantizers whose amax is
 * calibrated (static), i.e. elementwise: w: fp32 [rows, ld] working weights (block overwritten with the
 * quantized-dequantized columns q), hinv: fp32 [ld, ld] upper Cholesky factor of the damped inverse Hessian
 * (compute_hessian_inverse, calib_utils.py:80-113), delta: fp32 [rows, bs] receives err_j = (w_j - q_j) / hinv_jj.
 * Column j: q_j = QDQ(w_j) -- fmt 1: INT-num_bits (tensor_quant.py:607-645), fmt 2: FP8-E4M3 (:46-59) -- with the fp32 amax
 * entry amax[r * amax_row_stride + c / g] of element (r, c) (per tensor: stride 0 and g >= ld; per output channel:
 * stride 1 and g >= ld; static blocks of g columns: stride ld / g); then every column k >= j of the block gets
 * w_k -= fl(err_j * hinv[j, k]).  bs <= 128.
 * fmt 3: MX dynamic blocks (fused_amax_convert, tensor_quant_mx.cu:239-294): num_bits is the element format
 * (moq_mx_type), g the block size (a power of two <= 64 dividing i1 and bs); the E8M0 scale of the pivot's block comes from
 * the block's CURRENT abs-max at every column (what the reference's full-matrix call computes), amax may be NULL.
 * fmt 4: the same with block scales in an element format relative to a calibrated tensor-wide amax (NVFP4-style two-level
 * scales, compute_scale_with_global: tensor_quant_mx.cu:139-183): num_bits = element format, is_unsigned = SCALE format
 * (moq_mx_type both), amax[0] = the tensor-wide amax.
 * The update of the columns right of the block,
 * weight[:, i2:] -= errs @ hinv[i1:i2, i2:] (calib_utils.py:276), is moq_sgpt_trailing_update. */
int moq_gptq_block_sweep(float* w, int64_t rows, int64_t ld, int64_t i1, int bs, const float* hinv, float* delta,
                         const float* amax, int64_t amax_row_stride, int64_t g, int fmt, int num_bits, int is_unsigned,
                         int narrow, void* stream);
/* The update between two column blocks of create_sgpt_mask, w_rows[:, i2:] -= delta_blk.matmul(hessian_inv[i1:i2, i2:])
 * (sparsegpt.py:124; i2 = i1 + bs), with a DEFINED summation order instead of the BLAS library's:
 *     w[r, c] -= chain_{k = 0 .. bs-1, ascending} fmaf(delta[r, k], hinv[i1 + k, c], .)   started at +0,   c >= i2
 * on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, bit-for-bit an fmaf chain).  w: fp32 [rows, ld] working weights,
 * delta: fp32 [rows, bs] from moq_sgpt_block_sweep, hinv: fp32 [ld, ld].  bs <= 128. */
int moq_sgpt_trailing_update(float* w, int64_t rows, int64_t ld, int64_t i1, int bs, const float* delta,
                             const float* hinv, void* stream);

/* ------------------------------------------------------------------ AWQ-lite Gram-matrix search (a12) */

/* loss_acc[0] += inv_count * sum_{r,c} ( sum_k a[r, k] * b[c, k] ) * ref[r, c]   -- the dot product <a b^T, ref>
 * with the contraction on the matrix cores and the product with `ref` (fp32 [rows, cols]) fused into the epilogue.
 * Used for trace(E G E^T) = <E G, E> of the AWQ Gram search: a = [E_hi | E_hi | E_lo] (bf16 [Cout, 3 Cin]),
 * b = [G_hi | G_lo | G_hi] (bf16 [Cin, 3 Cin]; G symmetric), ref = E -- the K-concatenation sums the three
 * split-precision products in the fp32 accumulators.  a, b: dt = MOQ_BF16 | MOQ_F16, 16-byte aligned, k % 8 == 0,
 * cols % 4 == 0; or dt = MOQ_F32 (fp32 models): a = E fp32 [Cout, Cin], b = G fp32 [Cin, Cin], k = Cin, on the fp32
 * matrix cores.  partial: moq_awq_err_gemm_workspace(rows, cols) floats. */
/* One candidate of the Gram search from one read of W [rows, cols]:  e_out = dt(QDQ_g(dt(w * s[c]))) * r[c] - w  (fp32)
 * and a_out = the `a` operand of moq_awq_quadform, bf16 [rows, planes * cols] with hi = bf16(e), lo = bf16(e - hi):
 * planes 3: [hi | hi | lo] (against b = [G_hi | G_lo | G_hi]), 2: [hi | lo] (b = [G_hi | G_hi]), 1: [hi] (b = [G_hi]).
 * s: dtype dt [cols] (awq_scale.to(dtype), model_calib.py:1552), r: fp32 [cols] (float of (1/awq_scale).to(dtype),
 * :1551); dynamic per-group amax, signed INT-num_bits, narrow_range False.  cols % g == 0. */
int moq_awq_err_weight(const void* w, const void* s, const float* r, float* e_out, void* a_out, int64_t rows,
                       int64_t cols, int g, int dt, int num_bits, int planes, void* stream);
int moq_awq_quadform(const void* a, const void* b, const float* ref, int64_t rows, int64_t cols, int64_t k, int dt,
                     float* partial, float* loss_acc, double inv_count, void* stream);

/* ------------------------------------------------------------------ 2-D block amax / QDQ (a2) */

/* br x bc tiles of a contiguous [rows, cols] tensor; amax: fp32 [rows/br, cols/bc].
 *   mode 0: amax[i, j] = max |tile| (accumulate != 0: running max with the stored value)
 *   mode 1: y = QDQ(x) with the given tile amax      mode 2: amax (stored if amax != NULL) + QDQ, one read of x
 * fp8 != 0: FP8-E4M3 QDQ (as moq_fake_quant_e4m3), else INT-num_bits (as moq_fake_quant_int).  x == y allowed.
 * Replaces reduce_block_amax (quantization/utils/core_utils.py:43-90) and the eager fake-quant TensorQuantizer falls
 * back to for a two-axis block amax (nn/modules/tensor_quantizer.py:1018-1043, tensor_quant.py:80) -- the FP8 2-D
 * blockwise weight preset.  rows % br == 0, cols % bc == 0 (pad on the host).  Modes 1 and 2: br * bc <= 32768 elements
 * (16384 for fp32), bc % (16 / elem size) == 0, 16-byte aligned tensors.  Mode 0 takes every tile shape and alignment (tile
 * rows that are not whole 16-byte packets go through an element-wise tile kernel). */
int moq_block2d(const void* x, void* y, float* amax, int64_t rows, int64_t cols, int br, int bc, int dt, int mode,
                int accumulate, int fp8, int num_bits, int is_unsigned, int narrow_range, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MOQUANT_H_ */
