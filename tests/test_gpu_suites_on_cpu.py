"""The golden-fixture tests of the `-m gpu` suites, run a second time WITHOUT a GPU: the same test functions, with
their module-level device switched to "cpu" and the C-ABI served by the oracle through tests/hostmem_backend.py.

What this checks: everything on the Python side of the boundary (wrappers, layouts, host algorithms, exports) against
the reference-run fixtures, in the CPU tier.  What it cannot check: the HIP kernels (the real `-m gpu` run does that);
comparisons of the library against the oracle become trivial here and are left out.  A test that needs an entry with
no oracle twin (the MFMA kernels) is skipped."""

import functools
import importlib

import pytest

import _moa_import
import hostmem_backend

moa = _moa_import.load()

# (module, test function names): fixture-based tests that fit the CPU tier's time budget
SELECTED = {
    "test_gpu_qtensor": ["test_fp8_and_mxfp4_kernels_match_reference_run", "test_int4_qtensor_round_trip_matches_kernels",
                         "test_reference_literal_vectors_for_real_quantization"],
    "test_gpu_host": ["test_quantize_max_calibration_matches_reference", "test_quantize_smoothquant_matches_reference",
                      "test_quantize_awq_lite_matches_reference",
                      "test_tensor_quantizer_block_matches_reference", "test_max_calibrator_matches_reference",
                      "test_histogram_calibrator_matches_reference", "test_awq_weight_scale_vs_oracle_and_reference",
                      "test_sequential_quantizer_w4a8_matches_reference", "test_tensor_quantizer_2d_blocks_match_reference",
                      "test_two_level_block_format_flow_and_dynamic_type", "test_histogram_mse_threshold_is_the_candidate_loop",
                      "test_local_hessian_calibrate_on_the_gpu_equals_the_reference_run",
                      "test_affine_offset_of_the_kv_cache_presets_matches_the_reference_run",
                      "test_tensor_quantizer_tiles_on_the_last_two_axes_of_any_rank"],
    "test_gpu_mse": ["test_mse_calibrator_matches_reference", "test_quantize_mse_flow_matches_reference"],
    "test_gpu_export": ["test_export_from_reference_state_is_byte_identical", "test_fp8_export_from_reference_state_is_byte_identical",
                        "test_mxfp4_export_is_byte_identical", "test_int8_smoothquant_export_from_reference_state_is_byte_identical",
                        "test_quantize_and_export_end_to_end",
                        "test_quantize_and_export_with_replayed_inputs_is_byte_identical",
    "test_w4a8_awq_with_replayed_inputs_is_byte_identical",
                        "test_smoothquant_mxfp4_composition_on_gpu"],
    "test_gpu_input_quant": ["test_tensor_quantizer_takes_the_fused_pass",
                             "test_histogram_calibrator_later_batches_are_one_pass"],
    "test_gpu_calibrate_weights": ["test_row_hist_equals_numpy_on_reference_weights", "test_calibrate_weights_matches_reference_run",
                                   "test_calibrate_weights_mse_threshold", "test_calibrate_weights_mse_matches_reference_run"],
    "test_gpu_fp8_2d": ["test_fp8_qtensor_2d_blocks_match_reference_run", "test_fp8_2d_blockwise_export_is_byte_identical",
                        "test_reduce_block_amax_and_padding"],
    "test_gpu_mxfp8": ["test_mxfp8_qtensor_matches_reference_run", "test_mxfp8_rejects_wrong_scale_dtype_and_block",
                       "test_mxfp8_preset_exports_e4m3_weights_with_e8m0_scales"],
    "test_gpu_reference_style": None,  # None: every test of the module
    "test_gpu_clip": ["test_clip_loss_matches_reference_run", "test_quantize_awq_clip_matches_reference"],
    "test_gpu_sparsegpt": ["test_create_sgpt_mask_matches_reference", "test_hessian_matches_reference_hook",
                           "test_sparsify_sparsegpt_flow", "test_sparsegpt_hessian_shared_between_linears_with_the_same_input",
                           "test_sgpt_mask_disagreements_are_ties_at_llama_width",
                           "test_create_sgpt_mask_is_bit_exact_against_the_oracle_given_the_inverse_factor"],
    # (the Gram-sharing test counts staged launches; the staging buffer is budgeted from free GPU memory and is off here)
    "test_gpu_awq_search": ["test_gram_search_equals_gemm_search", "test_unexercised_and_nan_linears_fall_back_to_max_calibration",
                            "test_awq_lite_ragged_input_width_equals_the_reference_run",
                            "test_self_checking_margin_on_adversarial_distributions"],
    "test_gpu_layerwise": None,
    "test_gpu_fold_weight": ["test_fold_weight_keep_attrs"],
    "test_gpu_kv_cache": ["test_fp8_kv_cache_calibration_and_export_match_reference"],
    "test_gpu_moe": ["test_mixtral_fp8_calibration_and_export_match_reference"],
    "test_gpu_gptq": ["test_blockwise_update_equals_the_reference_run"],
    # (host-side layout logic: the permuted copy in front of the per-row reduction, checked against torch's amax)
    "test_gpu_parity": ["test_amax_over_dims_that_leave_the_kept_dims_apart", "test_golden_reduce_amax"],
}


@pytest.fixture(autouse=True)
def _host_memory_backend(monkeypatch):
    hostmem_backend.install(monkeypatch, moa)
    for name in SELECTED:
        monkeypatch.setattr(importlib.import_module(name), "DEV", "cpu", raising=False)


def _on_cpu(fn):
    @functools.wraps(fn)
    def run(*args, **kwargs):
        try:
            return fn(*args, **kwargs)
        except NotImplementedError as e:
            if "no oracle twin" in str(e):
                pytest.skip(str(e))
            raise
    return run


for _mod_name, _names in SELECTED.items():
    _mod = importlib.import_module(_mod_name)
    for _n in (_names if _names is not None else [n for n in dir(_mod) if n.startswith(("test_", "Test"))]):
        _obj = getattr(_mod, _n, None)
        if _obj is None:
            raise AttributeError(f"{_mod_name}.{_n} does not exist (renamed?)")
        if isinstance(_obj, type):  # test classes: wrap their methods
            _cls = type(f"{_n}OnCpu", (_obj,), {k: _on_cpu(v) for k, v in vars(_obj).items() if k.startswith("test_") and callable(v)})
            globals()[f"{_n}OnCpu"] = _cls
        else:
            globals()[f"{_n}__{_mod_name[len('test_gpu_'):]}"] = _on_cpu(_obj)
