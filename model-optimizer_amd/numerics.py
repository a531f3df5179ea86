"""Where the SMALL-VECTOR scale math of the flows runs: on the host (default) or on the statistics' own device.

The calibration algorithms end in a few [Cin]- or [Cout]-sized fp32 formulas the reference writes as plain torch expressions --
AWQ's `x_max.pow(alpha) / (w_max.pow(1 - alpha) + tiny)` and its normalisation (model_calib.py:1474-1487), `act_scale /
num_cache_steps` (:1601), `1 / awq_scale` (:1551), the exporter's `amax / maxbound` (export/quant_utils.py:225-243, :1038-1040),
the MSE calibrator's `initial_amax * multiplier` (calib/mse.py:83-121).  torch evaluates these differently on a CPU and on a
GPU in the last bit: the GPU kernel turns `tensor / python_scalar` into a multiplication by the host-computed reciprocal, a
0-dim operand is cast to a 16-bit tensor's dtype before the product instead of after, and the device math library's `pow` is
not the host's.  So the REFERENCE ITSELF writes a different checkpoint from a GPU run than from a CPU run of the same model
(measured on the MI355X, tests/test_gpu_reference_live.py: every weight scale of a tiny Llama within 1 ulp, 28 of 39 INT4-AWQ
tensors not byte-identical between the reference's two runs).  There are therefore two things "identical to the reference"
can mean, and this switch selects one:

  "host"   (default) -- the formulas run in IEEE fp32 on the host whatever device the statistics live on.  Results do not
           depend on the device; they equal the reference's CPU run, which is what the committed fixtures
           (tests/golden/, generated in a GPU-less container) pin byte for byte.  Cost: one device -> host -> device round
           trip per flow stage (`scales` 0.13-0.33 s of the 8B AWQ flow).
  "device" -- the reference's own expressions as torch evaluates them on the tensors' device.  On a GPU the results equal
           the reference's run ON THAT GPU bit for bit (amax, logits and every checkpoint byte: section B of
           tests/test_gpu_reference_live.py); nothing depends on the host's vector ISA and no round trip is paid.

`MOQ_SCALE_MATH=device` selects the second form for a process; `scale_math("device")` for a block."""

from __future__ import annotations

import contextlib
import os

import torch

_MODES = ("host", "device")
_mode = os.environ.get("MOQ_SCALE_MATH", "host")
if _mode not in _MODES:
    raise ValueError(f"MOQ_SCALE_MATH must be one of {_MODES}, got {_mode!r}")


def mode() -> str:
    return _mode


def on_host() -> bool:
    return _mode == "host"


@contextlib.contextmanager
def scale_math(which: str):
    global _mode
    if which not in _MODES:
        raise ValueError(f"scale_math: one of {_MODES}, got {which!r}")
    before, _mode = _mode, which
    try:
        yield
    finally:
        _mode = before


def vec(t: torch.Tensor) -> torch.Tensor:
    """A statistics vector where the scale math runs: detached fp32, on the host ("host") or where it lives ("device")."""
    t = t.detach().float()
    return t.cpu() if _mode == "host" else t


def div_scalar(t: torch.Tensor, divisor: float) -> torch.Tensor:
    """`t.float() / python_scalar` as the selected mode evaluates it, on t's device (host: IEEE division; device: torch's
    own kernel -- a multiplication by the reciprocal on a GPU, like the reference's GPU run)."""
    return (vec(t) / divisor).to(t.device)
