import os, sys, time, torch, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import _moa_import
moa = _moa_import.load()
from model_optimizer_amd import calib
orig_pick = calib._pick_entropy_candidate
stats = []
def pick(div, hist_fn, *a):
    t0 = time.perf_counter()
    lo = np.nanmin(div)
    close = int((div <= lo + abs(lo) * calib.ENTROPY_TIE_RTOL + 1e-300).sum())
    r = orig_pick(div, hist_fn, *a)
    stats.append((len(div), close, time.perf_counter() - t0))
    return r
calib._pick_entropy_candidate = pick
ob, of = calib.HistogramCalibrator.begin_amax, calib.HistogramCalibrator.finish_amax
tb = [0.0, 0.0]
def b(self, *a, **k):
    t0 = time.perf_counter(); r = ob(self, *a, **k); tb[0] += time.perf_counter() - t0; return r
def f(self, t):
    t0 = time.perf_counter(); r = of(self, t); tb[1] += time.perf_counter() - t0; return r
calib.HistogramCalibrator.begin_amax, calib.HistogramCalibrator.finish_amax = b, f
import hf_flow_check
r = hf_flow_check.run(hf_flow_check.parse_args(["--layers", "4", "--batches", "16", "--qformat", "int8_entropy"]), moa)
print(r)
print("begin total %.3f s, finish total %.3f s" % tuple(tb))
print("n_cand / n_close / pick seconds:", [(n, c, round(t, 4)) for n, c, t in stats][:60])
