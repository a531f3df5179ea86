"""Seeded random calibration runs on the device, this package's calibrators against the reference's own (same tensors):
HistogramCalibrator -- counts and edges after 1-3 batches whose range grows or shrinks, then percentile / entropy / mse amax --
and quantize(..., algorithm "mse" / "max") of a two-layer MLP.  Test infrastructure (needs the staged reference).

    python tools/calib_fuzz.py [cases] [seed]

Mind the clock: the REFERENCE's entropy / mse threshold searches are host loops over ~1 900 candidates (seconds per case), so
the default is 20 cases; 150 cases ran into a 700 s timeout on the GPU box without finishing (round 5) -- not part of the suite."""
import copy
import json
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import _moa_import  # noqa: E402
import ref_shim  # noqa: E402

DEV = os.environ.get("MOQ_FUZZ_DEVICE", "cuda")  # "cpu": the host logic through tests/hostmem_backend.py (oracle-served C-ABI)


def load_package():
    """The package; on MOQ_FUZZ_DEVICE=cpu with its C-ABI served by the oracle (the CPU tier's stand-in), so that the same
    random cases run in the build container against the reference's CPU path."""
    moa = _moa_import.load()
    if DEV == "cpu":
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import hostmem_backend

        class _Setter:
            @staticmethod
            def setattr(obj, name, value):
                setattr(obj, name, value)

        hostmem_backend.install(_Setter, moa)
    return moa
DT = {"bfloat16": torch.bfloat16, "float16": torch.float16, "float32": torch.float32}


METHODS = ["percentile", "percentile", "percentile", "entropy", "mse"]


def hist_case(rng):
    case = _hist_case(rng)
    # a later batch 800 x the first grows the histogram to ~8e5 bins: the REFERENCE's threshold searches are loops over every
    # bin (minutes on a GPU, hours on a host), and round 6's first run of such a case (seed 6, case 27) took a GPU box down
    # through this package's then-unbounded one-pass form of the same search -- the growth is capped at 10 x here; the bounded
    # search has its own test (tests/test_host_round6_cpu.py)
    first = case["batches"][0]["scale"]
    for b in case["batches"][1:]:
        b["scale"] = min(b["scale"], first * 10.0)
    return case


def _hist_case(rng):
    return {"dtype": rng.choice(list(DT)), "bins": rng.choice([256, 1024, 2048]), "skip_zeros": rng.random() < 0.3,
            "unsigned": rng.random() < 0.2, "num_bits": rng.choice([8, 4]),
            "batches": [{"shape": [rng.randint(1, 64), rng.choice([64, 256, 1000, 4096])], "scale": rng.choice([0.05, 1.0, 3.0, 40.0]),
                         "zeros": rng.random() < 0.3, "seed": rng.randint(0, 1 << 30)} for _ in range(rng.randint(1, 3))],
            "method": rng.choice(METHODS), "percentile": rng.choice([99.0, 99.9, 99.99, 100.0])}


def hist_run(Cal, case):
    cal = Cal(num_bits=case["num_bits"], axis=None, unsigned=case["unsigned"], num_bins=case["bins"], skip_zeros=case["skip_zeros"])
    for b in case["batches"]:
        g = torch.Generator().manual_seed(b["seed"])
        x = torch.randn(*b["shape"], generator=g) * b["scale"]
        if b["zeros"]:
            x[torch.rand(*b["shape"], generator=g) < 0.3] = 0.0
        if case["unsigned"]:
            x = x.abs()
        cal.collect(x.to(DT[case["dtype"]]).to(DEV))
    kw = {"percentile": case["percentile"]} if case["method"] == "percentile" else {}
    amax = cal.compute_amax(case["method"], **kw)
    hist, edges = cal._calib_hist, cal._calib_bin_edges
    to_np = lambda t: t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
    return to_np(hist).astype(np.int64), to_np(edges).astype(np.float64), None if amax is None else amax.detach().float().cpu()


def main(n=20, seed=2025, verbose=True):
    moa = load_package()
    ref_shim.install()
    from modelopt.torch.quantization.calib import HistogramCalibrator as RefHist

    rng = random.Random(seed)
    st = {"cases": 0, "equal": 0, "both_refused": 0, "reference_refused": {}, "ours_refused": [], "different": []}
    for _ in range(n):
        case = hist_case(rng)
        st["cases"] += 1
        try:
            want = hist_run(RefHist, case)
        except Exception as e:
            want = e
        try:
            with moa.numerics.scale_math("device"):  # device vs device: the reference grows its edges on this GPU too
                got = hist_run(moa.calib.HistogramCalibrator, case)
        except Exception as e:
            got = e
        if isinstance(want, Exception):
            if isinstance(got, Exception):
                st["both_refused"] += 1
            else:
                why = f"{type(want).__name__}: {str(want)[:80]}"
                st["reference_refused"][why] = st["reference_refused"].get(why, 0) + 1
            continue
        if isinstance(got, Exception):
            st["ours_refused"].append({"case": case, "error": f"{type(got).__name__}: {got}"[:200]})
            continue
        same_hist = got[0].shape == want[0].shape and np.array_equal(got[0], want[0])
        same_edges = got[1].shape == want[1].shape and np.array_equal(got[1].astype(np.float32), want[1].astype(np.float32))
        same_amax = (got[2] is None and want[2] is None) or (got[2] is not None and want[2] is not None and torch.equal(got[2].reshape(-1), want[2].reshape(-1)))
        if verbose:
            print(f"  case {st['cases']}: {case['method']} bins {case['bins']} batches {len(case['batches'])} -> "
                  f"hist {bool(same_hist)} edges {bool(same_edges)} amax {bool(same_amax)}", flush=True)
        if same_hist and same_edges and same_amax:
            st["equal"] += 1
        else:
            st["different"].append({"case": {k: v for k, v in case.items() if k != "batches"}, "n_batches": len(case["batches"]),
                                    "hist": bool(same_hist), "edges": bool(same_edges), "amax": bool(same_amax),
                                    "amax_pair": [None if t is None else t.reshape(-1)[:1].tolist() for t in (got[2], want[2])]})
    if verbose:
        print("histogram", json.dumps({k: (v if not isinstance(v, list) else len(v)) for k, v in st.items()})[:400])
        for d in st["different"][:8] + st["ours_refused"][:8]:
            print("   ", json.dumps(d)[:500])
    return {"histogram": st}


if __name__ == "__main__":
    if len(sys.argv) > 3:
        METHODS[:] = sys.argv[3].split(",")
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 20, int(sys.argv[2]) if len(sys.argv) > 2 else 2025)
