"""A flow_fuzz awq_clip case whose clipped amax differs between this package and the reference's eager run on the SAME device:
how many blocks, and how far apart the REFERENCE's own losses of the two picks are.  awq_clip's block dots are bf16 products
summed per block and ROUNDED TO bf16 before the difference is taken (model_calib.py:1839-1858): the summation order of the
device decides roundings, and with 8 tokens many clip ratios tie or nearly tie.  Test infrastructure (staged reference)."""
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

CASES = [{"preset": "INT4_AWQ_CFG", "dims": [256, 384, 256, 256], "bias": True, "dtype": "bfloat16", "batches": 1, "tokens": 8,
          "seed": 632734596, "algorithm": {"method": "awq_clip", "debug": True}, "outliers": False},
         {"preset": "INT4_AWQ_CFG", "dims": [256, 384, 256, 256], "bias": False, "dtype": "bfloat16", "batches": 1, "tokens": 8,
          "seed": 750530098, "algorithm": {"method": "awq_clip", "debug": True}, "outliers": False}]


def main():
    moa = flow_fuzz.load_package()
    ref_shim.install()
    import modelopt.torch.quantization as mtq

    for case in CASES:
        tables = {}
        orig_run = flow_fuzz.run

        model, batches, probe = flow_fuzz.build(case)
        import copy
        cfg = copy.deepcopy(mtq.INT4_AWQ_CFG)
        cfg["algorithm"] = copy.deepcopy(case["algorithm"])
        with torch.no_grad():
            q = mtq.quantize(model, cfg, lambda m: [m(b) for b in batches])
        want = {}
        for n, mod in q.named_modules():
            if hasattr(mod, "awq_clip"):
                h = mod.awq_clip
                ratios = sorted(h.loss)
                tables[n] = (ratios, torch.stack([h.loss[r].float().reshape(-1) for r in ratios]).cpu(), h.w_amax.float().reshape(-1).cpu())
                want[n] = mod.weight_quantizer._amax.float().reshape(-1).cpu()
        plain = dict(case, algorithm={"method": "awq_clip"})
        with moa.numerics.scale_math("device"):
            got, _ = flow_fuzz.run(moa.quantize, moa.model_quant, "TensorQuantizer", plain)
        for n, (ratios, losses, w_amax) in tables.items():
            a, b = got[f"{n}.weight_quantizer._amax"].float().reshape(-1), want[n]
            diff = (a != b).nonzero().reshape(-1)
            r_ours = (a / w_amax)
            gaps, exact_ties = [], 0
            rt = torch.tensor(ratios)
            for blk in diff.tolist():
                i_ref = int(losses[:, blk].argmin())
                i_ours = int((rt - r_ours[blk]).abs().argmin())
                l_ref, l_ours = float(losses[i_ref, blk]), float(losses[i_ours, blk])
                exact_ties += l_ours == l_ref
                gaps.append((l_ours - l_ref) / max(l_ref, 1e-30))
            gaps.sort()
            print(json.dumps({"seed": case["seed"], "linear": n, "blocks": int(a.numel()), "differ": int(diff.numel()),
                              "exact_ties_in_the_references_table": int(exact_ties),
                              "median_relative_gap": gaps[len(gaps) // 2] if gaps else 0.0,
                              "p90_relative_gap": gaps[int(len(gaps) * 0.9)] if gaps else 0.0,
                              "max_relative_gap": gaps[-1] if gaps else 0.0,
                              "ratio_steps_apart_max": float(((a - b).abs() / w_amax / 0.05).max())}))


if __name__ == "__main__":
    main()
