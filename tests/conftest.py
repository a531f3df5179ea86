"""pytest configuration: the `gpu` marker, repo-root imports and golden-fixture helpers."""

import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

DT = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# One-line facts a test wants in the suite's tail (printed after the pass/fail summary, so they show up in the committed
# `pytest ... | tail` of a GPU run): host-pow pins of the byte-identical replay, SparseGPT tie-audit counts, ...
SUITE_NOTES: list = []


def note(line: str):
    SUITE_NOTES.append(line)


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    for line in SUITE_NOTES:
        terminalreporter.write_line(f"[note] {line}")


def from_bits(a: np.ndarray, dtype: torch.dtype) -> torch.Tensor:
    """Inverse of gen_golden.bits(): uint16 patterns -> bf16/f16 tensors."""
    if dtype in (torch.bfloat16, torch.float16):
        return torch.from_numpy(a.view(np.int16).copy()).view(dtype)
    return torch.from_numpy(a.copy())


class Golden:
    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN, f"{name}.npz"))
        self.cases = json.loads(str(self.z["cases"]))

    def t(self, key, dtype=torch.float32):
        return from_bits(self.z[key], dtype)

    def raw(self, key):
        return self.z[key]


@pytest.fixture(autouse=True)
def _fixed_seed():
    """Every test starts from the same generator state, whatever ran before it in the worker: a test that draws without a
    seed of its own (the reference-style tolerance checks do) gets the same numbers in every run and under any xdist
    schedule -- an unlucky draw cannot fail one run in fifty."""
    torch.manual_seed(1234)
    np.random.seed(1234)
    yield


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]

    return get


_INT_VIEW = {torch.float32: torch.int32, torch.float16: torch.int16, torch.bfloat16: torch.int16, torch.float64: torch.int64}


def _same_bits(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """Elementwise: identical bit patterns (so -0.0 != +0.0), except that any NaN equals any NaN."""
    if a.dtype in _INT_VIEW:
        ai, bi = a.contiguous().view(_INT_VIEW[a.dtype]), b.contiguous().view(_INT_VIEW[b.dtype])
        return (ai == bi) | (torch.isnan(a) & torch.isnan(b))
    return a == b


def bits_equal(a: torch.Tensor, b: torch.Tensor) -> bool:
    """Bit-exact equality (the sign of zero counts); NaN == NaN."""
    a, b = a.detach().cpu(), b.detach().cpu()
    if a.shape != b.shape or a.dtype != b.dtype:
        return False
    return bool(_same_bits(a, b).all())


def assert_bits_equal(a, b, what=""):
    a, b = a.detach().cpu(), b.detach().cpu()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    assert a.dtype == b.dtype, f"{what}: dtype {a.dtype} vs {b.dtype}"
    bad = ~_same_bits(a, b)
    if bad.any():
        i = bad.reshape(-1).nonzero()[0].item()
        raise AssertionError(
            f"{what}: {int(bad.sum())} of {a.numel()} elements differ; first at flat index {i}: "
            f"{a.reshape(-1)[i].item()!r} vs {b.reshape(-1)[i].item()!r}")
