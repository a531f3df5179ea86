"""A flow_fuzz case whose MSE-calibrated amax differs between this package and the reference's eager run on the SAME device:
which channels, which candidates, and how far apart the REFERENCE's own fp32 losses of the two picks are (a near-tie decided
by the summation order of a 72-element row, or a defect?).  Test infrastructure (needs the staged reference)."""
import copy
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import flow_fuzz  # noqa: E402
import ref_shim  # noqa: E402

CASE = {"preset": "INT8_DEFAULT_CFG", "dims": [256, 72, 384], "bias": True, "dtype": "bfloat16", "batches": 1, "tokens": 8,
        "seed": 330213563, "algorithm": "mse", "outliers": True}


def main():
    moa = flow_fuzz.load_package()
    ref_shim.install()
    import modelopt.torch.quantization as mtq
    from modelopt.torch.quantization.calib import mse as ref_mse

    tables = {}
    orig = ref_mse.MseCalibrator.compute_amax

    def keep(self, verbose=False):  # the reference's loss table of every weight it calibrates
        out = orig(self, verbose)
        if self._losses_sum is not None:
            tables[len(tables)] = (torch.stack([l.detach().clone() for l in self._losses_sum]), self._candidates.clone(),
                                   self._initial_amax.detach().clone())
        return out

    ref_mse.MseCalibrator.compute_amax = keep
    want, _ = flow_fuzz.run(mtq.quantize, mtq, "TensorQuantizer", CASE)
    ref_mse.MseCalibrator.compute_amax = orig
    with moa.numerics.scale_math("device"):
        got, _ = flow_fuzz.run(moa.quantize, moa.model_quant, "TensorQuantizer", CASE)
    for k in want:
        if not k.endswith("weight_quantizer._amax"):
            continue
        a, b = got[k].float().reshape(-1), want[k].float().reshape(-1)
        diff = (a != b).nonzero().reshape(-1).tolist()
        print(f"{k}: {len(diff)} of {a.numel()} channels differ")
        if not diff:
            continue
        # the table of this weight: the one whose initial amax has this many channels and reproduces the reference's pick
        for losses, cand, a0 in tables.values():
            if a0.numel() != a.numel():
                continue
            a0 = a0.float().reshape(-1).cpu()
            losses, cand = losses.float().reshape(len(cand), -1).cpu(), cand.float().cpu()
            pick_ref = losses.argmin(0)
            if not torch.equal((a0 * cand[pick_ref]).to(want[k].dtype).float(), b):
                continue
            for ch in diff[:12]:
                ours = (a[ch] / a0[ch])
                i_ours = int((cand - ours).abs().argmin())
                i_ref = int(pick_ref[ch])
                l_ref, l_ours = float(losses[i_ref, ch]), float(losses[i_ours, ch])
                print("   ", json.dumps({"channel": ch, "reference_candidate": round(float(cand[i_ref]), 3),
                                         "our_candidate": round(float(cand[i_ours]), 3), "reference_loss_at_its_pick": l_ref,
                                         "reference_loss_at_our_pick": l_ours,
                                         "relative_gap": (l_ours - l_ref) / max(l_ref, 1e-30)}))
            break


if __name__ == "__main__":
    main()
