"""pytest plugin for the subprocess that runs the REFERENCE's own GPU test files on top of this library
(tests/test_gpu_reference_live.py, section C; `-p ref_seams_plugin`).  With MOQ_INSTALL_SEAMS=1 it calls
modelopt_plugin.install() in pytest_configure -- before collection, because the reference's test modules call
get_cuda_ext*() at import time (tests/gpu/torch/quantization/test_quantize_mxformats_cuda.py:26) -- and prints the seam
counters at the end.  Test infrastructure only."""

import os
import sys


def pytest_configure(config):
    root = os.environ.get("MOQ_REPO_ROOT")
    if root and root not in sys.path:
        sys.path.insert(0, root)
    if os.environ.get("MOQ_S7_HOSTMEM") in ("1", "all"):
        # CPU tier: ONLY the algorithm seam, the C-ABI served by the host-memory stand-in (tests/hostmem_backend.py) and the
        # seams' "is this a GPU tensor" gate opened -- the reference's own UNIT tests then calibrate through this package's
        # flows on CPU tensors (tests/test_algorithm_seam_cpu.py::test_the_references_own_unit_tests_...)
        import _moa_import

        moa = _moa_import.load()
        import hostmem_backend
        from model_optimizer_amd import modelopt_plugin

        class _Patch:
            @staticmethod
            def setattr(obj, name, value):
                setattr(obj, name, value)

        hostmem_backend.install(_Patch, moa)
        modelopt_plugin._takes = lambda t: t.device.type == "cpu"  # (stands in for `is_cuda`: meta / offloaded tensors stay out)
        every = os.environ["MOQ_S7_HOSTMEM"] == "all"  # (also S6 reduce_amax and the S5 mask seams, on host memory)
        config._moq_seams = modelopt_plugin.install(extensions=False, backend=False, utilities=every, sparsity_seam=every,
                                                    algorithms=True)
        return
    if os.environ.get("MOQ_INSTALL_SEAMS") != "1":
        return
    import _moa_import

    _moa_import.load()
    from model_optimizer_amd import modelopt_plugin

    # MOQ_INSTALL_ALGORITHMS=1: the algorithm seam (S7) on top -- the reference's tests then calibrate through this package's
    # fused flows wherever a model is adoptable, and through their own code (counted fallbacks) wherever it is not
    config._moq_seams = modelopt_plugin.install(algorithms=os.environ.get("MOQ_INSTALL_ALGORITHMS") == "1")


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    if not getattr(config, "_moq_seams", None):
        terminalreporter.write_line("[seams] not installed (the reference's own eager / extension-less path)")
        return
    from model_optimizer_amd import modelopt_plugin

    terminalreporter.write_line(f"[seams] installed: {config._moq_seams}")
    for k, v in sorted(modelopt_plugin.STATS.items()):
        terminalreporter.write_line(f"[seams] {k} = {v}")
