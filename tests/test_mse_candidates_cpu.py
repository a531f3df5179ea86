"""MseCalibrator's candidate amax table (round 3: one broadcast host product, built once per calibrator) equals the
reference's per-candidate `amax * 0-dim multiplier` products -- including their dtype promotion (a dimensioned 16-bit amax
keeps its dtype, a 0-dim one promotes to fp32) -- bit for bit."""

import torch

import _moa_import

moa = _moa_import.load()
from model_optimizer_amd.calib import MseCalibrator  # noqa: E402


def test_candidate_table_equals_per_candidate_host_products():
    torch.manual_seed(0)
    for dt in (torch.bfloat16, torch.float16, torch.float32):
        for shape in ((), (1,), (7, 1), (4097,), (3, 1, 5, 1)):
            a = (torch.rand(shape) * 3 + 1e-3).to(dt)
            cal = MseCalibrator(a, fused_format=(8, False, False))
            cands = cal._generate_candidates("cpu")
            loop = torch.stack([cal._compute_candidate_amax(c).float().reshape(-1) for c in cands])
            tab = cal._candidate_amax_table(torch.device("cpu"))
            assert tab.dtype == torch.float32 and tab.shape == loop.shape
            assert torch.equal(loop, tab), (dt, shape, int((loop != tab).sum()))
            assert cal._candidate_amax_table(torch.device("cpu")) is tab  # cached
            cal.reset()
            assert cal._cand_table is None
