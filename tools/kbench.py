"""Per-kernel throughput table on one MI355X: every C-ABI kernel on Llama-3-70B-sized tensors.
Algorithmic bytes (SURVEY.md 8d / DESIGN.md) divided by the HIP-event time of the launch.
Usage (GPU box): python tools/kbench.py [> profiles/rNN_kernel_table.md]"""

import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _moa_import  # noqa: E402

moa = _moa_import.load()
ops = moa.ops
DEV = "cuda:0"
PEAK = 8000.0
# vector issue peak: 256 CUs x 4 SIMDs x 16 lanes per clock x 2.4 GHz lane-instructions per second (an fp32 FMA in every slot
# is the 78.6 TFLOP/s vector figure, packed FMAs the 157 TFLOP/s one)
VALU_PEAK = 256 * 4 * 16 * 2.4e9
# hot loop of moq_mse_sweep per (element, candidate), from the ISA (tools/isa_blocks.py, round 5): INT 415 / 64 (64 rndne +
# 64 med3 + 287 packed), FP8 351 / 64; round 4's loop: 13.6 / 11.4
MSE_SLOTS_INT, MSE_SLOTS_FP8 = 415 / 64, 351 / 64


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    torch.manual_seed(1234)
    rows, cols = 28672, 8192  # Llama-3-70B gate/up projection
    w = (torch.randn(rows, cols, device=DEV) * 0.02).to(torch.bfloat16)
    n = w.numel()
    x = torch.randn(8 * 512, 8192, device=DEV).to(torch.bfloat16)  # activation batch [B*S, H]
    nx = x.numel()
    amax1 = ops.reduce_amax(w).float().reshape(1)
    am_g = ops.reduce_amax(w.view(-1, 128), axis=(1,)).float()
    am_c = ops.reduce_amax(w, axis=(1,)).float()
    s_col = torch.rand(cols, device=DEV) + 0.5
    scales = (7 / am_g).to(torch.bfloat16).reshape(-1)
    wsf = (am_g / 7).reshape(rows, cols // 128)
    q4 = ops.int4_quantize(w.view(-1), scales, 128)
    counts = torch.zeros(2048, dtype=torch.int64, device=DEV)
    xmax = float(x.float().abs().max())
    s448 = (amax1 / 448.0).to(torch.bfloat16)
    s_tile = (ops.reduce_block_amax(w, {-1: 128, -2: 128}).float() / 448.0).to(torch.bfloat16)
    q8t = ops.fp8_quantize_tile(w, s_tile, 128, 128).view(torch.uint8)
    # activations with per-channel spread and a few massive channels: the abs-max is set by outliers, the
    # bulk of |x| sits in the lowest histogram bins
    chan = torch.exp(torch.randn(8192, device=DEV))
    chan[:4] *= 50.0
    xo = (torch.randn(8 * 512, 8192, device=DEV) * chan).to(torch.bfloat16)
    xo_max = float(xo.float().abs().max())
    xbig = (torch.randn(32 * 4096, 8192, device=DEV) * chan).to(torch.bfloat16)  # 2.1 GB: past the Infinity Cache
    xbig_max = float(xbig.float().abs().max())
    counts_big = torch.zeros(2048, dtype=torch.int64, device=DEV)
    run_amax = torch.zeros(1, dtype=torch.float32, device=DEV)
    pqs = (torch.rand(8192, device=DEV) + 0.5).to(torch.bfloat16)
    cand39 = torch.linspace(0.1, 4.0, 39, device=DEV).reshape(39, 1)
    q8 = ops.fp8_quantize(w, s448).view(torch.uint8)
    mxq = ops.mxfp4_quantize(w, 32)
    wsf_row = (am_c / 127.0).float()
    hsym = torch.randn(8192, 8192, device=DEV)
    s5 = torch.rand(5, 8192, device=DEV) + 0.5
    amax_x = torch.tensor(xo_max * 0.5, device=DEV)
    ybig = torch.empty_like(xbig)
    cases = [
        ("moq_hist_abs 2048 bins, activations with outlier channels (67 MB)", lambda: ops.hist_abs(xo, 2048, xo_max, False, counts), 2 * nx),
        ("moq_hist_abs 2048 bins, activations with outlier channels (2.1 GB)", lambda: ops.hist_abs(xbig, 2048, xbig_max, False, counts_big), 2 * xbig.numel()),
        ("moq_input_quant: amax + histogram, one read (2.1 GB)", lambda: ops.input_quant(xbig, amax_running=run_amax, hist_counts=counts_big, hist_max_edge=xbig_max), 2 * xbig.numel()),
        ("moq_input_quant: x * pqs -> amax -> FP8 QDQ (2.1 GB)", lambda: ops.input_quant(xbig, pqs, amax_running=run_amax, qdq_amax=amax_x, num_bits=(4, 3), out=ybig), 4 * xbig.numel()),
        ("moq_input_quant: x * pqs -> amax -> INT8 QDQ (2.1 GB)", lambda: ops.input_quant(xbig, pqs, amax_running=run_amax, qdq_amax=amax_x, num_bits=8, out=ybig), 4 * xbig.numel()),
        ("unfused: scale_cols + amax + FP8 QDQ (2.1 GB, 3 kernels)", lambda: (lambda v: (ops.reduce_amax(v), ops.scaled_e4m3(v, amax_x)))(ops.scale_cols(xbig, pqs)), 4 * xbig.numel()),
        ("moq_col_abs_mean_accum (awq act scale, 67 MB)", lambda: ops.col_abs_mean_accum(x, torch.zeros(8192, device=DEV)), 2 * nx),
        ("moq_amax (per-tensor)", lambda: ops.reduce_amax(w), 2 * n),
        ("moq_amax_axis rows (per-channel, axis 0)", lambda: ops.reduce_amax(w, axis=(1,)), 2 * n),
        ("moq_amax_axis groups g=128 (static per-group)", lambda: ops.reduce_amax(w.view(-1, 128), axis=(1,)), 2 * n + 4 * n / 128),
        ("moq_amax_axis columns (activation per-channel)", lambda: ops.reduce_amax(x, axis=(0,)), 2 * nx),
        ("moq_col_abs_stats (act sum + amax)", lambda: ops.col_abs_stats(x), 2 * nx),
        ("moq_fake_quant_int INT8 per-tensor", lambda: ops.fake_tensor_quant(w, amax1, 8, False, True), 4 * n),
        ("moq_fake_quant_int INT8 per-channel", lambda: ops.fake_tensor_quant(w, am_c.view(-1, 1), 8, False, True), 4 * n),
        ("moq_fake_quant_int INT4 static g=128", lambda: ops.fake_tensor_quant(w.view(-1, 128), am_g.view(-1, 1), 4, False, False), 4 * n + 4 * n / 128),
        ("moq_amax_qdq_int_group INT4 g=128 (fused)", lambda: ops.amax_qdq_int_group(w, 128, 4), 4 * n + 4 * n / 128),
        ("moq_fake_quant_e4m3 per-tensor", lambda: ops.scaled_e4m3(w, amax1), 4 * n),
        ("moq_fake_quant_e4m3 per-channel", lambda: ops.scaled_e4m3(w, am_c.view(-1, 1)), 4 * n),
        ("moq_mx_fused_amax_convert MXFP4 g=32", lambda: ops.fused_amax_convert(w, 32, "E2M1"), 4 * n),
        ("moq_mx_fused_amax_convert MXFP8 e4m3 g=32", lambda: ops.fused_amax_convert(w, 32, "E4M3"), 4 * n),
        ("moq_awq_scale_qdq INT4 g=128", lambda: ops.awq_scale_qdq(w, s_col, 128, 4), 4 * n),
        ("moq_scale_cols", lambda: ops.scale_cols(w, s_col), 4 * n),
        ("moq_mask_2to4", lambda: ops.mask_2to4(w), 3 * n),
        ("moq_int4_pack (qtensor)", lambda: ops.int4_quantize(w.view(-1), scales, 128), 2.5 * n + 2 * n / 128),
        ("moq_int4_unpack (qtensor)", lambda: ops.int4_dequantize(q4, scales, 128), 2.5 * n + 2 * n / 128),
        ("moq_int4_pack_export", lambda: ops.pack_int4_in_uint8(w, wsf), 2.5 * n + 4 * n / 128),
        ("moq_hist_abs 2048 bins", lambda: ops.hist_abs(x, 2048, xmax, False, counts), 2 * nx),
        ("moq_rescale_cols", lambda: ops.rescale_cols(w, s_col, s_col), 4 * n),
        ("moq_block2d amax 128x128 tiles (reduce_block_amax)", lambda: ops.reduce_block_amax(w, {-1: 128, -2: 128}), 2 * n),
        ("moq_block2d FP8 amax + QDQ 128x128 tiles", lambda: ops.block2d(w.view(rows // 128, 128, cols // 128, 128), 2), 4 * n),
        ("moq_fp8_pack per-tensor (qtensor / export)", lambda: ops.fp8_quantize(w, s448), 3 * n),
        ("moq_fp8_pack_tile 128x128 (fp8_pb_wo export)", lambda: ops.fp8_quantize_tile(w, s_tile, 128, 128), 3 * n),
        ("moq_fp8_unpack_tile 128x128", lambda: ops.fp8_dequantize_tile(q8t, s_tile, torch.bfloat16, 128, 128), 3 * n),
        ("moq_mxfp4_pack g=32", lambda: ops.mxfp4_quantize(w, 32), 2.5 * n + n / 32),
        ("moq_row_hist_np 2048 bins per channel (calibrate_weights)", lambda: ops.row_hist_np(w, 2048), 4 * n),
        ("moq_amax_mid (block amax over a middle dim)", lambda: ops.reduce_block_amax(w.view(rows // 64, 64, cols), {1: 64}), 2 * n),
        ("moq_mx_fused_amax_convert E2M1 / E4M3 scales g=16 (two-level)", lambda: ops.fused_amax_convert(w, 16, "E2M1", "E4M3", amax1), 4 * n),
        # entries that had no per-kernel timing before round 3
        ("moq_fp8_unpack per-tensor (qtensor dequantize)", lambda: ops.fp8_dequantize(q8, s448, torch.bfloat16), 3 * n),
        ("moq_mxfp4_unpack g=32", lambda: ops.mxfp4_dequantize(mxq[0], mxq[1], torch.bfloat16, 32), n // 2 + n // 32 + 2 * n),
        ("moq_int8_pack_rows (INT8 SmoothQuant export)", lambda: ops.int8_pack_rows(w, wsf_row), 3 * n),
        ("moq_awq_weight_scale g=128 (awq_lite get_weight_scale)", lambda: ops.awq_weight_scale(w, 128), 2 * n),
        ("moq_scale_cols_multi, 5 candidates (x / s_alpha copies of an AWQ search step)", lambda: ops.scale_cols_multi(x, s5), 2 * nx * 6),
        ("moq_transpose16 (activation transpose feeding the Gram accumulation)", lambda: ops.transpose16(x), 4 * nx),
        ("moq_symmetrize 8192 x 8192 fp32 (mirror of an upper-triangle Gram / Hessian)", lambda: ops.symmetrize(hsym), 8 * 8192 * 8192 // 2 * 1),
        # MseCalibrator.collect: 39 candidate amax values in ONE read (VALU-bound by design: the "GB/s" is the one read; the
        # reference makes ~5 passes per candidate = 195 x these bytes)
        # fourth entry: VALU instruction slots per (element, candidate) read off the kernel's hot loop (tools/isa_blocks.py;
        # a packed fp32 instruction is ONE slot for two elements) -> the kernel's own roofline, the vector issue rate
        ("moq_mse_sweep INT8 per-tensor, 39 candidates", lambda: ops.mse_sweep(w, cand39 * amax1, None, 8), 2 * n, 39 * MSE_SLOTS_INT),
        ("moq_mse_sweep INT8 per-channel, 39 candidates", lambda: ops.mse_sweep(w, cand39 * am_c.reshape(1, -1), (1,), 8), 2 * n, 39 * MSE_SLOTS_INT),
        ("moq_mse_sweep FP8 per-tensor, 39 candidates", lambda: ops.mse_sweep(w, cand39 * amax1, None, (4, 3)), 2 * n, 39 * MSE_SLOTS_FP8),
        ("moq_mse_sweep INT4 static g=128, 39 candidates", lambda: ops.mse_sweep(w.view(-1, 128), cand39 * am_g.reshape(1, -1), (1,), 4), 2 * n, 39 * MSE_SLOTS_INT),
    ]
    print(f"| kernel (bf16, {rows}x{cols} weight = {2 * n / 1e6:.0f} MB; activations {tuple(x.shape)}) | ms | algorithmic GB/s | frac of 8 TB/s |")
    print("|---|---|---|---|")
    # ops under ~40 us are timed from Python here = the host's launch rate (2-3 launches + a small allocation per call), not the
    # kernel: their kernel-only durations (rocprofv3) are in profiles/r05g_small_kernels_kernel_only.md
    only = sys.argv[1] if len(sys.argv) > 1 else None
    for name, fn, nbytes, *valu in cases:
        if only and not any(o in name for o in only.split(",")):  # comma-separated substrings
            continue
        ms = timed(fn)
        gbs = nbytes / ms / 1e6
        note = ""
        if valu:  # VALU-bound kernel: elements x slots per element against the chip's vector issue rate
            rate = n * valu[0] / (ms * 1e-3)
            note = f" -- VALU-bound: {valu[0]:.0f} instruction slots per element = {rate / 1e12:.1f} T lane-slots/s = {rate / VALU_PEAK:.2f} of the vector issue peak ({VALU_PEAK / 1e12:.1f} T/s)"
        if ms < 0.040:
            note += " -- HOST-launch-bound at this size; kernel-only: profiles/r05g_small_kernels_kernel_only.md"
        print(f"| {name}{note} | {ms:.3f} | {gbs:.0f} | {gbs / PEAK:.3f} |")


if __name__ == "__main__":
    main()
