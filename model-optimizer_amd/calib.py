"""Calibrators -- same API as modelopt.torch.quantization.calib (collect / reset / compute_amax), with
the statistics kept ON THE DEVICE and every pass over the data done by one HIP kernel.

Differences from the reference that do not change results:
  * MaxCalibrator keeps one fp32 running-max buffer that the kernel updates in place (fused
    `torch.max(prev, new)`), and the three per-collect asserts of calib/max.py:69-77 (NaN / negative / inf,
    each a device->host sync) are evaluated ONCE in compute_amax from the accumulated amax itself:
    abs-max is NaN iff some element was NaN, inf iff some element was inf, never negative.
  * HistogramCalibrator counts in exact 64-bit integers (torch.histc counts in fp32).
"""

from __future__ import annotations

import math

import numpy as np
import torch

from . import _lib, numerics, ops


class _Calibrator:
    """calib/calibrator.py:25-69."""

    def __init__(self, num_bits=8, axis=None, unsigned=False):
        self._num_bits = num_bits
        self._axis = axis
        self._unsigned = unsigned

    def collect(self, x):
        raise NotImplementedError

    def reset(self):
        raise NotImplementedError

    def compute_amax(self, *args, **kwargs):
        raise NotImplementedError

    def __repr__(self):
        return f"num_bits={self._num_bits} axis={self._axis} unsigned={self._unsigned}"


def convert_quantization_axis_to_reduce_axis(input, axis):
    """core_utils.py:128-143."""
    if axis is None:
        return None
    axis = axis if isinstance(axis, (list, tuple)) else [axis]
    return [i for i in range(input.dim()) if i not in axis and (i - input.dim()) not in axis]


class DeferredAmax:
    """The per-tensor running abs-max requests of ONE decoder layer answered by ONE sweep (moq_mt_amax_running).

    A max-calibration forward loop asks for nine small reductions per decoder layer and batch (the inputs of q / k / v,
    o_proj, gate / up, down_proj, the key / value states: 33-117 MB each at Llama-3-8B shapes) -- 7-20 us kernels whose
    fixed ramp / tail is a third of their time, three of them reading a tensor another one has just read.  While an
    instance is `current`, MaxCalibrator.collect_per_tensor_fast only NOTES (tensor, running-max buffer); `flush()` -- at
    the end of every decoder layer (a forward hook set by model_calib.max_calibrate), when `limit_bytes` of tensors are
    held, and before anybody reads a calibrator -- sweeps the distinct tensors in one dense window and folds each maximum
    into every buffer that asked for it.  Statistics collection leaves the tensors unchanged (quantization is off while a
    calibrator collects), so the only thing deferral needs is that nobody writes to a noted tensor before the flush:
    the version counter of every tensor is checked at flush time and a write is an ERROR, not a wrong statistic.
    The tables of a flush (segment rows, chunk prefix sums, folds) are cached by their content: the allocator hands a
    decoder stack the same addresses batch after batch."""

    current = None  # the instance statistics collection runs under, or None (every request is its own launch)

    def __init__(self, device, limit_bytes: int = 1 << 30, probation: bool = False, flush_points: int | None = None):
        self.device = torch.device(device)
        self.limit_bytes = int(limit_bytes)
        self.entries = {}      # id(tensor) -> [tensor, version, [running-max buffers]]
        self.bytes = 0
        self.dt = None
        self.tables = {}       # content key -> (device int64 table, n_seg, n_chunks, n_folds, scratch)
        self.stats = {"requests": 0, "flushes": 0, "tensors": 0, "table_builds": 0, "bytes": 0}
        # probation (the AUTOMATIC choice of max_calibrate, defer_stats=None): nobody promised that the model leaves a noted
        # tensor alone until the end of its decoder layer, so the FIRST pass through every flush point answers each request with
        # its own launch (add() returns False) and only WATCHES the tensors' version counters; a write seen there switches
        # deferral off for the rest of the calibration -- no statistic is lost, nothing raises -- and a clean first pass
        # (whether a model writes its activations in place is a property of its code, not of the batch) switches it on
        self.probation = bool(probation)
        self.flush_points = flush_points  # how many distinct flush points one pass goes through (the decoder layers), if known
        self.watched = []      # probation: (tensor, version when its quantizer saw it)
        self.flushed_keys = set()
        self.pass_watched = 0  # requests watched in the current pass
        self.disabled = False

    def add(self, x, dt_code, buf) -> bool:
        if self.disabled or x.device != self.device or (self.dt is not None and dt_code != self.dt):
            return False
        if x.is_inference():  # (no version counter to watch: `forward_loop` under torch.inference_mode())
            return False
        if self.probation:
            self.watched.append((x, x._version))
            return False
        ent = self.entries.get(id(x))
        if ent is None or ent[0] is not x:
            self.entries[id(x)] = ent = [x, x._version, []]
            self.bytes += x.numel() * x.element_size()
            self.dt = dt_code
        ent[2].append(buf)
        self.stats["requests"] += 1
        if self.bytes >= self.limit_bytes:
            self.flush()
        return True

    def _tables(self, ents):
        key = tuple((e[0].data_ptr(), e[0].numel(), tuple(b.data_ptr() for b in e[2])) for e in ents)
        hit = self.tables.get(key)
        if hit is not None:
            return hit
        import ctypes

        n_seg = len(ents)
        sizes = [e[0].numel() for e in ents]
        blk = (ctypes.c_int64 * (n_seg + 1))()
        total = _lib.lib().moq_mt_plan((ctypes.c_int64 * n_seg)(*sizes), n_seg, blk)
        if total < 0:
            _lib.check(int(total))
        rows = []
        for e in ents:  # struct moq_seg: x, y, amax (unused by this entry), n
            rows += [e[0].data_ptr(), 0, 0, e[0].numel()]
        rows += list(blk)
        n_folds = 0
        for i, e in enumerate(ents):  # struct moq_amax_fold: dst, seg
            for b in e[2]:
                rows += [b.data_ptr(), i]
                n_folds += 1
        table = torch.tensor(rows, dtype=torch.int64).to(self.device)  # ONE upload: [segs | blk_start | folds]
        scratch = torch.empty(max(int(total), 1), dtype=torch.float32, device=self.device)
        if len(self.tables) >= 64:
            self.tables.clear()
        self.tables[key] = hit = (table, n_seg, int(total), n_folds, scratch)
        self.stats["table_builds"] += 1
        return hit

    def _run(self, ents):
        self._launch(*self._tables(ents))

    def _launch(self, table, n_seg, n_chunks, n_folds, scratch):
        base = table.data_ptr()
        rc = _lib.lib().moq_mt_amax_running(base, base + 32 * n_seg, n_seg, n_chunks, self.dt, scratch.data_ptr(),
                                            base + 32 * n_seg + 8 * (n_seg + 1), n_folds,
                                            torch._C._cuda_getCurrentRawStream(self.device.index))
        if rc:
            _lib.check(rc)

    def flush(self, key=None):
        """key: which flush point this is (a decoder layer); the second visit of a point ends the probation."""
        if self.probation:
            if any(x._version != v for x, v in self.watched):
                self.disabled, self.probation = True, False
                self.stats["disabled_by_inplace_write"] = True
            seen, self.watched = len(self.watched), []
            if key is None or self.disabled:
                self.pass_watched += seen
                return
            # A pass through every flush point counts only if requests were WATCHED in it (a calibrator's first collect takes
            # the general path and never asks: the first batch of a calibration shows nothing) -- then deferral is on from
            # here; else the next pass is watched.
            if key in self.flushed_keys:  # this point again: the previous pass ended before this flush
                if self.pass_watched > 0:
                    self.probation = False
                self.flushed_keys, self.pass_watched = {key}, seen
                return
            self.flushed_keys.add(key)
            self.pass_watched += seen
            if self.flush_points is not None and len(self.flushed_keys) >= self.flush_points:  # the pass ends with this flush
                if self.pass_watched > 0:
                    self.probation = False
                self.flushed_keys, self.pass_watched = set(), 0
            return
        if not self.entries:
            return
        ents, self.entries, self.bytes = list(self.entries.values()), {}, 0
        for x, version, _ in ents:
            if x._version != version:
                raise RuntimeError(
                    "deferred statistics: a tensor handed to a quantizer was written in place before the end of its decoder "
                    "layer, so its abs-max can no longer be taken; calibrate with max_calibrate(..., defer_stats=False)")
        self._run(ents)
        self.stats["flushes"] += 1
        self.stats["tensors"] += len(ents)
        self.stats["bytes"] += sum(e[0].numel() * e[0].element_size() for e in ents)
        self.dt = None


class MaxCalibrator(_Calibrator):
    """Running abs-max, per tensor or per kept axis -- calib/max.py:26-110."""

    def __init__(self, num_bits=8, axis=None, unsigned=False, track_amax=False):
        super().__init__(num_bits, axis, unsigned)
        self._track_amax = track_amax
        self._amaxs = [] if track_amax else None
        self._buf = None       # fp32 running max on the device (flat)
        self._shape = None     # shape the reference's amax would have (keepdims=True)
        self._dtype = None
        self._fast = None      # (dtype, dtype code, device index, pointer of _buf): collect_per_tensor_fast

    @property
    def amaxs(self):
        return self._amaxs

    def collect_per_tensor_fast(self, x) -> bool:
        """The hot call of a max-calibration forward loop -- running per-tensor abs-max of one more activation -- with
        nothing between Python and the C-ABI: no axis bookkeeping, no context managers, the library entry point called
        with the raw pointers on torch's current stream.  Returns False when this call is not of that shape (first
        collect, another dtype / device, non-contiguous input), and the general `collect` takes it.
        Why: the kernels of a Llama-3-8B FP8 calibration loop are 1.4 % of its GPU time, yet the loop ran 6.5 % over the
        plain forward -- all of it the host getting from `TensorQuantizer.forward` to the launch, 18 432 times
        (profiles/r03_flows_rocprof.md)."""
        fast = self._fast
        if fast is None or x.dtype is not fast[0] or not x.is_cuda:
            return False
        if not x.is_contiguous() and not ops._is_dense(x):
            # (dense = some permutation of the dims is contiguous, e.g. HF key / value states `.view(B, S, H, D)
            # .transpose(1, 2)`: numel elements from data_ptr on, which an abs-max may walk in any order)
            return False
        dev = x.device.index
        if dev != fast[2] or torch.cuda.current_device() != dev:
            return False
        buf = self._buf
        if buf is None or buf.data_ptr() != fast[3]:  # (a deep copy of the calibrator has its own buffer)
            self._fast = None
            return False
        batch = DeferredAmax.current
        if batch is not None and batch.add(x, fast[1], buf):
            return True  # answered by the layer's ONE launch (DeferredAmax.flush)
        rc = _lib._lib.moq_amax(x.data_ptr(), x.numel(), fast[1], fast[3], 1, torch._C._cuda_getCurrentRawStream(dev))
        if rc:
            _lib.check(rc)
        return True

    @torch.no_grad()
    def collect(self, x):
        reduce_axis = convert_quantization_axis_to_reduce_axis(x, self._axis)
        nd = x.dim()
        if reduce_axis is None or len(reduce_axis) == nd:
            shape = () if reduce_axis is None else ()  # reduce_amax squeezes scalars
            n = 1
        elif nd == 4 and sorted(a % nd for a in reduce_axis) == [1, 3]:
            # 2-D blocks: the (R/br, br, C/bc, bc) view reduced over the block dims (ops.block2d)
            shape = (x.shape[0], 1, x.shape[2], 1)
            n = x.shape[0] * x.shape[2]
            if n == 1:
                shape = ()  # a single tile: reduce_amax squeezes scalars (core_utils.py:181-182)
        else:
            # (kept axes that lie apart -- axis=(0, 2) of a rank-3 tensor -- are served by ops.reduce_amax through one
            # permuted copy; the buffer is the kept axes' product either way)
            red = {a % nd for a in reduce_axis}
            keep = [d for d in range(nd) if d not in red]
            shape = tuple(x.shape[d] if d in keep else 1 for d in range(nd))
            n = 1
            for d in keep:
                n *= x.shape[d]
            if n == 1:
                shape = ()
        if self._buf is None:
            self._buf = torch.zeros(n, dtype=torch.float32, device=x.device)
            self._shape, self._dtype = shape, x.dtype
        elif shape != self._shape:
            raise RuntimeError("amax shape changed!")  # calib/max.py:81-82
        ops.reduce_amax(x, axis=reduce_axis, out=self._buf, accumulate=True)
        if self._track_amax:
            self._amaxs.append(ops.reduce_amax(x, axis=reduce_axis).float().cpu().numpy())
        elif n == 1 and shape == () and x.is_cuda and x.dtype in ops._DT and hasattr(torch._C, "_cuda_getCurrentRawStream"):
            # later per-tensor collects of this dtype on this device may take collect_per_tensor_fast
            self._fast = (x.dtype, ops._DT[x.dtype], x.device.index, self._buf.data_ptr())

    def reset(self):
        self._buf = None
        self._shape = None
        self._fast = None

    def compute_amax(self, verified: bool = False):
        """verified: the caller has already checked this calibrator's buffer for NaN / inf (`verify_finite` over all
        calibrators of a model: ONE device -> host read instead of one per quantizer)."""
        if self._buf is None:
            return None
        amax = self._buf.to(self._dtype).reshape(self._shape)
        if not verified:
            # deferred form of the asserts in calib/max.py:69-77 (one sync per quantizer per calibration)
            bad = torch.stack([torch.isnan(self._buf).any(), torch.isinf(self._buf).any()]).tolist()
            assert not bad[0], "detected nan values in amax"
            assert not bad[1], "detected inf values in amax"
        return amax

    @staticmethod
    def verify_finite(calibrators) -> bool:
        """Are the running maxima of ALL these calibrators free of NaN / inf?  One concatenation, one host read.  False
        sends the caller back to the per-calibrator asserts, which name the offender (calib/max.py:69-77)."""
        by_dev = {}
        for c in calibrators:
            if c._buf is not None:
                by_dev.setdefault(c._buf.device, []).append(c._buf.reshape(-1))
        for dev, bufs in by_dev.items():
            flat = torch.cat(bufs)
            # the library's abs-max compares |x| bit patterns: NaN > inf > every finite value, so one reduction answers
            # "any NaN or inf?" (and it is a kernel the process has loaded already -- the first torch.isfinite of a
            # process costs 80 ms of code-object loading on ROCm, measured inside this very call)
            top = ops.reduce_amax(flat) if dev.type == "cuda" else flat.abs().max() if not torch.isnan(flat).any() else flat.new_tensor(float("nan"))
            if not math.isfinite(float(top)):
                return False
        return True

    def __str__(self):
        return f"MaxCalibrator(track_amax={self._track_amax})"


class BiasCalibrator(_Calibrator):
    """The offset of affine quantization (calib/bias.py:102-178): the quantizer subtracts it before the QDQ and adds it
    back afterwards (KV caches whose keys sit off-centre: FP8_AFFINE_KV_CFG).  `axis` lists the dims that are REDUCED
    (bias.py:41-50 -- the key / value states [batch, heads, tokens, head_dim] with axis (-2, -4) keep one offset per
    head and channel).  "mean": the mean of every collected tensor, averaged over the calls (fp32 running average,
    stored in the tensor's dtype); "max_min": the midpoint of the running extremes.

    A host mirror: torch reductions in the reference's order (the offset is a [1, heads, 1, head_dim]-sized statistic
    of a tensor the abs-max kernel reads anyway; no kernel of its own)."""

    def __init__(self, method: str = "mean", axis=None):
        super().__init__(axis=axis)
        self._method = method
        self.reset()

    def reset(self):
        self._calib_bias = self._calib_max = self._calib_min = None
        self._cnt = 0

    def _reduced_dims(self, x):
        return tuple(i for i in range(x.dim()) if i in self._axis or (i - x.dim()) in self._axis)

    def _extremes(self, x):
        if self._axis is None:
            return torch.max(x), torch.min(x)
        dims = self._reduced_dims(x)
        return torch.amax(x, dim=dims, keepdim=True), torch.amin(x, dim=dims, keepdim=True)

    def _of(self, x, method):
        if method != "mean":
            hi, lo = self._extremes(x)
            return (hi + lo) / 2
        return torch.mean(x) if self._axis is None else torch.mean(x, dim=self._reduced_dims(x), keepdim=True)

    def collect(self, x):
        if self._method == "mean":
            now = self._of(x, "mean")
            if self._calib_bias is None:
                self._calib_bias = now
            else:
                self._calib_bias = ((self._calib_bias.float() * self._cnt + now.float()) / (self._cnt + 1)).to(now.dtype)
            self._cnt += 1
        elif self._method == "max_min":
            hi, lo = self._extremes(x)
            self._calib_max = hi if self._calib_max is None else torch.max(self._calib_max, hi)
            self._calib_min = lo if self._calib_min is None else torch.min(self._calib_min, lo)
            self._calib_bias = (self._calib_max + self._calib_min) / 2
        else:
            raise ValueError(f"Unsupported method: {self._method}")

    def compute_bias(self):
        return self._calib_bias

    def compute_dynamic_bias(self, inputs):
        if self._method not in ("mean", "max_min"):
            raise ValueError(f"Unknown bias method: {self._method}")
        return self._of(inputs, self._method)

    def compute_amax(self, *args, **kwargs):  # not an amax calibrator
        raise NotImplementedError


class HistogramCalibrator(_Calibrator):
    """|x| histogram with the reference's growth rule -- calib/histogram.py:36-205."""

    def __init__(self, num_bits=8, axis=None, unsigned=False, num_bins=2048, grow_method=None,
                 skip_zeros=False, torch_hist=True):
        super().__init__(num_bits, axis, unsigned)
        if axis is not None:
            raise NotImplementedError("Calibrator histogram collection only supports per tensor scaling")
        self._num_bins = num_bins
        self._skip_zeros = skip_zeros
        self._calib_bin_edges = None  # torch fp32, like the reference's torch_hist=True branch
        self._calib_hist = None       # int64 counts on the device
        self._share_range = False     # data parallel: bin with rank 0's first-batch range (distributed.py)
        self._grown_to = None         # the abs-max the range was last extended for

    def share_range_across_ranks(self, on: bool = True):
        """Data-parallel calibration: the first collect broadcasts rank 0's first-batch abs-max, every rank bins with
        that width (growing its range by the reference's rule when a batch exceeds it), and the int64 counts of all
        ranks add up exactly (distributed.sync_calibrators_bucketed) to the histogram a single rank would have built
        from all batches starting with rank 0's first one."""
        self._share_range = bool(on)

    @torch.no_grad()
    def collect(self, x):
        x = x.detach()
        if x.dtype not in (torch.float32, torch.float16, torch.bfloat16):
            x = x.float()  # the reference histograms x.abs().float() whatever the dtype (calib/histogram.py:95-96)
        lo, hi = ops.INPUT_QUANT_HIST_BINS
        if (self._calib_hist is not None and x.is_contiguous() and lo <= self._num_bins < hi
                and x.dim() >= 1 and x.numel() > 0):
            # later batches: abs-max AND the histogram over the current range from ONE read of the batch
            # (ops.input_quant); only if the batch exceeds the range -- the growth step of calib/histogram.py:121-127 --
            # are the optimistic counts dropped and the batch binned again over the extended range
            amax = torch.zeros(1, dtype=torch.float32, device=x.device)
            counts = torch.zeros(self._num_bins, dtype=torch.int64, device=x.device)
            ops.input_quant(x.reshape(1, -1), amax_running=amax, hist_counts=counts,
                            hist_max_edge=float(self._calib_bin_edges[-1]), hist_skip_zeros=self._skip_zeros)
            x_max = amax.reshape(()).cpu()
            if not x_max > self._calib_bin_edges[-1]:
                self._calib_hist += counts
                return
        else:
            # x_max of what is histogrammed (|x| in fp32; zeros never raise the max)
            x_max = ops.reduce_amax(x).float().cpu()
        if self._calib_bin_edges is None and self._calib_hist is None:
            first_max = x_max
            if self._share_range:
                from . import distributed as mdist

                first_max = mdist.agree_histogram_range(x_max.clone().to(x.device)).cpu()
            self._calib_bin_edges = torch.linspace(0, first_max, self._num_bins + 1)
            self._calib_hist = torch.zeros(self._num_bins, dtype=torch.int64, device=x.device)
        if x_max > self._calib_bin_edges[-1]:
            self._grow_to(x_max)
        ops.hist_abs(x, self._num_bins, float(self._calib_bin_edges[-1]), self._skip_zeros, counts=self._calib_hist)

    def _grow_to(self, x_max: torch.Tensor):
        """Keep the bin width, extend the range to hold `x_max` (a 0-dim fp32 host tensor) -- calib/histogram.py:121-127."""
        width = self._calib_bin_edges[1] - self._calib_bin_edges[0]
        self._num_bins = int((x_max / width).ceil().item())
        dev = self._calib_hist.device
        if numerics.on_host() or dev.type != "cuda":
            self._calib_bin_edges = torch.arange(0, x_max + width, width)
        else:
            # numerics "device": the reference grows its edges with `torch.arange(..., device=x.device)`, and torch's GPU arange
            # accumulates in fp32 where the CPU one uses double -- edges past the first range differ in the last bit between
            # the reference's two runs (found by tools/calib_fuzz.py); this is its run on the device
            self._calib_bin_edges = torch.arange(0, x_max + width, width, device=dev).cpu()
        grown = torch.zeros(self._num_bins, dtype=torch.int64, device=self._calib_hist.device)
        grown[: self._calib_hist.numel()] = self._calib_hist
        self._calib_hist = grown
        self._grown_to = x_max.clone()

    def reset(self):
        self._calib_bin_edges = None
        self._calib_hist = None
        self._grown_to = None

    def merge(self, other_hist: torch.Tensor):
        """Add counts collected elsewhere (used by the cross-rank SUM all-reduce, distributed.py)."""
        self._calib_hist += other_hist.to(self._calib_hist.device)

    def compute_amax(self, method: str, *, stride: int = 1, start_bin: int = 128, percentile: float = 99.99):
        return self.finish_amax(self.begin_amax(method, stride=stride, start_bin=start_bin, percentile=percentile))

    def begin_amax(self, method: str, *, stride: int = 1, start_bin: int = 128, percentile: float = 99.99):
        """First half of compute_amax: queue the device work of the threshold search and return a ticket for
        `finish_amax`.  A caller with many calibrators (finish_stats_collection) begins them all before it finishes the
        first, so the searches of a model overlap and the host waits once instead of once per quantizer.

        Histograms on the GPU are searched there: percentile = one kernel, the index never leaves the device (the amax is a
        gather from the device edges); entropy = one kernel for the 1 921 divergences, of which the host only looks at
        the minimum (and re-scores exact ties with the reference's own arithmetic, `_pick_entropy_candidate`).  Histograms
        on the host (the CPU test tier) take the numpy restatement of the reference's loops."""
        if method not in ("entropy", "mse", "mse_qdq", "percentile"):
            raise TypeError(f"Unknown calibration method {method}")
        if self._calib_hist is None:
            return ("none", None)
        if method == "percentile" and (percentile < 0 or percentile > 100):
            raise ValueError("Invalid percentile. Must be in range 0 <= percentile <= 100.")
        hist, edges = self._calib_hist, self._calib_bin_edges
        if method in ("mse", "mse_qdq"):
            search = _compute_amax_mse if method == "mse" else _compute_amax_mse_qdq
            return ("value", search(hist, edges, self._num_bits, self._unsigned, stride, start_bin))
        nq_bits = self._num_bits - 1 + int(self._unsigned) if isinstance(self._num_bits, int) else None
        # (the counts live on the device, the bin edges on the host -- a linspace, as in the reference)
        if method == "percentile":
            if hist.is_cuda:
                return ("percentile", (ops.hist_percentile_index(hist.reshape(1, -1), percentile / 100), edges))
            return ("value", _compute_amax_percentile(hist.cpu().numpy(), edges.cpu().numpy(), percentile))
        if hist.is_cuda and nq_bits is not None and 0 <= nq_bits <= 12 and hist.numel() >= start_bin:
            div = ops.hist_entropy_divergences(hist, 1 << nq_bits, start_bin, stride)
            return ("entropy", (div, hist, edges, stride, start_bin))
        return ("value", _compute_amax_entropy(hist.cpu().numpy(), edges.cpu().numpy(), self._num_bits, self._unsigned,
                                               stride, start_bin))

    def finish_amax(self, ticket):
        """Second half of compute_amax: the one host read of the search (8 bytes for percentile, 15 KB for entropy) and the
        amax as a 0-dim fp32 host tensor, like the reference's `torch.tensor(calib_bin_edges[idx].item())`."""
        kind, payload = ticket
        if kind in ("none", "value"):
            return payload
        if kind == "percentile":
            idx, edges = payload
            return torch.tensor(edges[int(idx.cpu()[0])].item())
        div, hist, edges, stride, start_bin = payload
        pick = _pick_entropy_candidate(div.cpu().numpy(), lambda: hist.cpu().numpy(), self._num_bits, self._unsigned, stride,
                                       start_bin)
        return torch.tensor(edges[pick * stride + start_bin].item())


def _compute_amax_percentile(calib_hist, calib_bin_edges, percentile):
    """calib/histogram.py:326-343."""
    if percentile < 0 or percentile > 100:
        raise ValueError("Invalid percentile. Must be in range 0 <= percentile <= 100.")
    total = calib_hist.sum()
    cdf = np.cumsum(calib_hist / total)
    idx = np.searchsorted(cdf, percentile / 100)
    return torch.tensor(calib_bin_edges[idx].item())


ENTROPY_TIE_RTOL = 1e-9  # device divergences within this of the minimum are re-scored on the host (they agree to ~1e-13)


def _pick_entropy_candidate(div, hist_fn, num_bits, unsigned, stride, start_bin) -> int:
    """Index (into the candidate list) of the reference's choice -- the LAST minimum of the divergences (histogram.py:
    277-279) -- from the device's fp64 divergences `div`.  The device sums in its own order, so a value can differ from
    numpy's in the last bits; that only matters between candidates that tie: every candidate within ENTROPY_TIE_RTOL of
    the minimum is re-scored with the reference's own arithmetic (`_compute_amax_entropy(..., only=...)`) and the last
    minimum of THOSE values decides.  NaN divergences behave like np.argmin (a NaN is the minimum)."""
    if np.isnan(div).any():
        close = np.flatnonzero(np.isnan(div))
    else:
        lo = div.min()
        close = np.flatnonzero(div <= lo + abs(lo) * ENTROPY_TIE_RTOL + 1e-300) if np.isfinite(lo) else np.flatnonzero(div == lo)
    if close.size == 1:
        return int(close[0])
    exact = []
    _compute_amax_entropy(hist_fn(), None, num_bits, unsigned, stride, start_bin, divergences_out=exact,
                          only=[int(c) for c in close])
    exact = np.array(exact)
    return int(close[len(exact) - 1 - np.argmin(exact[::-1])])


def _kl_divergence(pk, qk, pk_total=None, qk_total=None):
    """scipy.stats.entropy(pk, qk) without its argument-policy wrapper (0.3 ms per call, most of the search): the
    same three statements on the same arrays -- normalise both (x / np.sum(x); scipy's leading `1.0 *` is the
    identity on fp64), special.rel_entr, sum.  The totals may be handed in when the caller already holds
    np.sum of the very same array."""
    from scipy.special import rel_entr
    pk_total = np.add.reduce(pk) if pk_total is None else pk_total
    qk_total = np.add.reduce(qk) if qk_total is None else qk_total
    with np.errstate(invalid="ignore"):
        pk = pk / pk_total
    qk = qk / qk_total
    return np.add.reduce(rel_entr(pk, qk))


def _compute_amax_entropy(calib_hist, calib_bin_edges, num_bits, unsigned, stride=1, start_bin=128,
                          divergences_out=None, only=None):
    """KL-divergence threshold search -- calib/histogram.py:210-283.

    Same arithmetic as the reference's loop, candidate by candidate, with its slow steps replaced by exact
    equivalents (about 5x faster on the host, identical divergences bit for bit -- tests/test_host_cpu.py):
      * `np.digitize(range(i), np.linspace(0, i, nbins + 1)) - 1`: nbins is a power of two, so every edge k*i/nbins
        is exact in fp64 and the bucket of source bin j is floor(j * nbins / i);
      * `np.add.at` of the counts: integer-valued fp64 sums below 2^53 are exact in any order (np.bincount);
      * `Counter(digitized.tolist())`: the number of non-empty source bins per bucket (np.bincount);
      * the two totals of the count check are the very sums scipy.stats.entropy normalises by (np.sum of the same
        array, same pairwise order), so each is formed once.
    The floating-point part (division, the two totals, the entropy formula) is the reference's, on identical arrays."""
    bins = calib_hist.astype(np.int64).copy()
    bins[0] = bins[1]
    total_data = np.sum(bins)
    nbins = 1 << (num_bits - 1 + int(unsigned))
    nonzero = bins != 0
    bins_f = bins.astype(np.float64)
    tail = np.concatenate([np.cumsum(bins[::-1])[::-1], [0]])  # tail[i] = sum(bins[i:]), exact
    j_scaled = np.arange(len(bins), dtype=np.int64) * nbins
    all_nonzero = bool(nonzero.all())
    divergences = []
    candidates = range(start_bin, len(bins) + 1, stride)
    if only is not None:  # (the exact tie-break of the device search: these candidate indices only)
        candidates = [candidates[c] for c in only]
    for i in candidates:
        valid = nonzero[:i]
        bucket = j_scaled[:i] // i
        weights = bins_f[:i]
        if not all_nonzero:
            bucket, weights = bucket[valid], weights[valid]
        sums = np.bincount(bucket, weights=weights, minlength=nbins)
        members = np.bincount(bucket, minlength=nbins)
        new_density_counts = np.zeros(nbins, dtype=np.float64)
        np.divide(sums, members, out=new_density_counts, where=members > 0)
        if all_nonzero:
            new_density = new_density_counts[bucket]
        else:
            new_density = np.zeros(i, dtype=np.float64)
            new_density[valid] = new_density_counts[bucket]
        new_sum = np.add.reduce(new_density)
        reference_density = bins_f[:i].copy()
        reference_density[-1] += tail[i]
        old_sum = np.add.reduce(reference_density)
        total_counts_new, total_counts_old = new_sum + tail[i], old_sum
        if round(total_counts_new) != total_data or round(total_counts_old) != total_data:
            raise RuntimeError(f"Count mismatch! total_counts_new={total_counts_new}, "
                               f"total_counts_old={total_counts_old}, total_data={total_data}")
        # NB: the reference's _normalize_distr rebinds a local and normalises nothing (histogram.py:217-220);
        # scipy.stats.entropy normalises both arguments itself, so the result is the same either way.
        divergences.append(_kl_divergence(reference_density, new_density, old_sum, new_sum))
    divergences = np.array(divergences)
    if divergences_out is not None:
        divergences_out.extend(divergences.tolist())
    if only is not None:
        return None
    last_argmin = len(divergences) - 1 - np.argmin(divergences[::-1])
    return torch.tensor(calib_bin_edges[last_argmin * stride + start_bin].item())


_MSE_SEARCH_CHUNK = 1 << 24    # elements of one [candidates, bins] pass of the histogram MSE search (64 MB fp32 per temporary)
_MSE_SEARCH_BUDGET = 1 << 30   # candidate x bin products evaluated exhaustively (~10 s of host time)


def _compute_amax_mse(counts, edges, num_bits, unsigned, stride=1, start_bin=128):
    """calib/histogram.py:286-323 AS IT COMPUTES, for integer formats: the amax the reference returns for
    `compute_amax("mse")`, bit for bit (pinned by the reference-run `hist` fixture and the live differential test).

    The reference calls `fake_tensor_quant(centers, amax, num_bits, unsigned)`, whose positional slots are
    (inputs, amax, bias, num_bits, ...) (`tensor_quant.py:349-360`): the bit width lands in `bias` and the signedness in
    `num_bits`.  What it evaluates per candidate is therefore a quantizer of int(unsigned) bits around a bias:
      signed   (0 bits): bound = 2^-1 - 1 = -0.5, clamp(., +0.5, -0.5) = -0.5 for every centre, so the "quantized centres"
                         are the constant -0.5 / (-0.5 / amax) + num_bits and the count-weighted error is smallest where
                         amax + num_bits meets the count-weighted mean of the centres;
      unsigned (1 bit):  bound = 0, scale = 0, 0 / 0 = NaN for every centre: all errors are NaN and numpy's argmin takes the
                         first candidate, centres[start_bin].
    Neither is a mean-squared-error search (the MSE search of this path is MseCalibrator, calib/mse.py); a user who
    switches from the reference gets the amax the reference gave.  The documented intent -- QDQ of the centres at each
    candidate amax -- is `_compute_amax_mse_qdq` (method "mse_qdq"), which also serves (4, 3): the reference's FP8 branch
    calls scaled_e4m3 one argument short and raises TypeError, so there is no value to match.

    The arithmetic below follows `_tensor_quant` (`tensor_quant.py:607-645`) step by step in fp32 on the host (2 048 bins:
    microseconds), as the reference's CPU path does; one [candidates, bins] pass instead of its loop."""
    if not isinstance(num_bits, int):
        if tuple(num_bits) != (4, 3):
            raise TypeError("Invalid num_bits. num_bits must be a positive integer or tuple (4,3).")
        return _compute_amax_mse_qdq(counts, edges, num_bits, unsigned, stride, start_bin)
    if num_bits < 0:
        raise TypeError("Invalid num_bits. num_bits must be a positive integer or tuple (4,3).")
    dev = counts.device if isinstance(counts, torch.Tensor) else torch.device("cpu")
    c = torch.as_tensor(counts).detach().to("cpu").float()
    e = torch.as_tensor(edges).detach().to("cpu").float()
    centers = (e[1:] + e[:-1]) / 2
    n_bins = centers.numel()
    idx = torch.arange(start_bin, n_bins, stride)
    if idx.numel() == 0:
        raise ValueError(f"mse threshold search: start_bin={start_bin} leaves no candidate among {n_bins} bins")
    amax = centers[idx]                                      # [candidates]
    if bool((amax < 0).any()):  # (an all-zero channel under np.histogram's widened (-0.5, 0.5) range)
        raise ValueError("Negative values in amax")
    slot_bits = int(bool(unsigned))                          # what arrives in `num_bits`
    bound = torch.tensor((2.0 ** (slot_bits - 1)) - 1.0)     # max_bound; `unsigned` itself keeps its default False
    scale = bound / amax
    tiny = amax <= 1.0 / (1 << 24)
    mul = torch.where(tiny, torch.zeros_like(scale), scale).reshape(-1, 1)
    div = torch.where(tiny, torch.ones_like(scale), scale).reshape(-1, 1)
    cen, cnt = centers.reshape(1, -1), c.reshape(1, -1)

    def exact(lo, hi):  # the reference's arithmetic for candidates [lo, hi): one [hi - lo, bins] pass
        x = cen - num_bits                                   # `inputs - bias`
        y = torch.clamp((x * mul[lo:hi]).round_(), -bound, bound)
        q = y / div[lo:hi] + num_bits
        return ((q - cen) ** 2 * cnt).mean(dim=1)

    # Memory and time are bounded however far the histogram grew (a calibrator whose range grew 800-fold holds 8e5 bins: the
    # one-pass form asked the host for candidates x bins = 2.7 TB and took the machine down -- found by tools/calib_fuzz.py,
    # seed 6 case 27).  Up to _MSE_SEARCH_BUDGET candidate x bin products: every candidate in chunks, exactly.
    n_cand = idx.numel()
    rows = max(1, _MSE_SEARCH_CHUNK // n_bins)
    if n_cand * n_bins <= _MSE_SEARCH_BUDGET:
        mse = torch.cat([exact(lo, min(lo + rows, n_cand)) for lo in range(0, n_cand, rows)])
        pick = int(np.argmin(mse.numpy()))                   # first minimum; the first NaN when there is one
        return centers[idx[pick]].clone().to(dev)
    # Beyond the budget (the reference's own loop needs minutes on a GPU and hours on a host there): the clamp bounds are
    # inverted or zero (above), so a candidate's "quantized centres" are ONE value q_c whatever the centre, and its error is
    # the parabola (A q^2 - 2 B q + C) / bins in q_c with A, B, C the count-weighted moments of the centres.  The parabola
    # (float64) screens; the candidates within 1e-4 of its minimum -- far beyond the fp32 noise of the sums -- are evaluated
    # with the reference's arithmetic, and the first minimum of those is taken.  NaN rules as np.argmin's: the first NaN wins.
    q_c = (torch.clamp(((cen[:, :1] - num_bits) * mul).round_(), -bound, bound) / div + num_bits).reshape(-1).double()
    has_empty_bin = bool((c == 0).any())
    nan_like = torch.isnan(q_c) | (torch.isinf(q_c) if has_empty_bin else torch.zeros_like(q_c, dtype=torch.bool))
    if bool(nan_like.any()):
        return centers[idx[int(nan_like.to(torch.int8).argmax())]].clone().to(dev)
    c64, cen64 = c.double(), centers.double()
    a_, b_, c_ = c64.sum(), (c64 * cen64).sum(), (c64 * cen64 * cen64).sum()
    par = (a_ * q_c * q_c - 2.0 * b_ * q_c + c_) / n_bins
    par = torch.where(torch.isfinite(par), par, torch.full_like(par, float("inf")))
    best = float(par.min())
    near = (par <= best + abs(best) * 1e-4 + 1e-300).nonzero().reshape(-1)
    lo, hi = int(near.min()), int(near.max()) + 1
    cap = max(1, _MSE_SEARCH_BUDGET // n_bins)
    if hi - lo > cap:  # (a flat vertex wider than the budget: centre the window on the parabola's minimum)
        mid = int(par.argmin())
        lo = max(0, min(mid - cap // 2, n_cand - cap))
        hi = min(n_cand, lo + cap)
    mse = torch.cat([exact(a, min(a + rows, hi)) for a in range(lo, hi, rows)])
    pick = lo + int(np.argmin(mse.numpy()))
    return centers[idx[pick]].clone().to(dev)


def _compute_amax_mse_qdq(counts, edges, num_bits, unsigned, stride=1, start_bin=128):
    """The MSE threshold search calib/histogram.py:286-323 documents: QDQ of the bin centres at each candidate amax,
    count-weighted squared error, first minimum (`compute_amax("mse_qdq")`; also what "mse" runs for (4, 3), where the
    reference raises).  NOT what the reference computes for integer formats -- see _compute_amax_mse.

    All candidates of a chunk are one [candidates, bins] per-row QDQ launch (amax [candidates, 1]) instead of the
    reference-shaped loop of one QDQ + reduction + device->host read per candidate (1 920 of them for 2 048 bins);
    nothing is read back: the result is a device tensor."""
    if not isinstance(num_bits, int) and tuple(num_bits) != (4, 3):
        raise TypeError("Invalid num_bits. num_bits must be a positive integer or tuple (4,3).")
    dev = counts.device
    c = counts.float()
    e = edges.float().to(dev)
    centers = ((e[1:] + e[:-1]) / 2).contiguous()
    n_bins = centers.numel()
    idx = torch.arange(start_bin, n_bins, stride, device=dev)
    if idx.numel() == 0:
        raise ValueError(f"mse threshold search: start_bin={start_bin} leaves no candidate among {n_bins} bins")
    chunk = max(1, (1 << 25) // n_bins)  # <= 32 M elements (128 MB fp32) per launch, however far the histogram grew
    mse = []
    for lo in range(0, idx.numel(), chunk):
        amax = centers[idx[lo:lo + chunk]].reshape(-1, 1)
        x = centers.reshape(1, -1).expand(amax.shape[0], n_bins).contiguous()
        if isinstance(num_bits, int):
            q = ops.fake_tensor_quant(x, amax, num_bits, unsigned)
        else:
            q = ops.scaled_e4m3(x, amax)
        mse.append((((q - x) ** 2) * c).mean(dim=1))
    mse = torch.cat(mse)
    mse = torch.where(torch.isnan(mse), torch.full_like(mse, float("inf")), mse)  # a NaN never wins (`mse < best`)
    first_min = (mse == mse.min()).to(torch.int32).argmax()  # first minimum, as the strict `<` of the loop form
    return centers[idx[first_min]].clone()


@torch.no_grad()
def calibrate_weights(model, method="percentile", perchannel=True, percentile=99.99, num_bins=2048):
    """calib/histogram.py:346-433: set the amax of every `weight_quantizer` from a histogram of its weight -- one
    histogram per output channel (axis 0; transposed convolutions are outside this path) or one per tensor.

    The reference moves every channel to the host and calls np.histogram on it (Cout numpy calls per weight); here
    all channel histograms of a weight come from ONE kernel with numpy's float32 edges (ops.row_hist_np) and the
    `percentile` reduction -- cumsum / searchsorted in the reference's float64 arithmetic, row by row -- from a second
    one (ops.hist_percentile_index): nothing but the amax leaves the device."""
    for _, module in model.named_modules():
        if not (hasattr(module, "weight") and hasattr(module, "weight_quantizer")):
            continue
        wq = module.weight_quantizer
        w = module.weight.detach()
        axis = 0 if perchannel else None
        if method == "max":
            reduce_axis = convert_quantization_axis_to_reduce_axis(w, axis)
            amax = ops.reduce_amax(w, axis=reduce_axis)
        elif method in ("percentile", "mse"):
            if method == "percentile" and (percentile < 0 or percentile > 100):
                raise ValueError("Invalid percentile. Must be in range 0 <= percentile <= 100.")
            counts, edges = ops.row_hist_np(w if perchannel else w.reshape(1, -1), num_bins)
            if method == "percentile":
                # cumsum / searchsorted of every row in the reference's float64 arithmetic, on the device: the [Cout, bins]
                # counts (235 MB for a 28672-row weight) never travel to the host
                idx = ops.hist_percentile_index(counts, percentile / 100)
                vals = edges.gather(1, idx.reshape(-1, 1)).reshape(-1)
            else:
                if isinstance(wq._num_bits, int):  # the search is host arithmetic: ONE device -> host copy per weight
                    counts, edges = counts.cpu(), edges.cpu()
                vals = torch.stack([_compute_amax_mse(counts[r].to(torch.int64), edges[r], wq._num_bits, wq._unsigned)
                                    for r in range(counts.shape[0])]).cpu()
            if perchannel:
                amax = vals.reshape([w.shape[0]] + [1] * (w.dim() - 1))
            else:
                amax = vals.reshape(())
        else:
            raise TypeError(f"Unsupported calibration method {method}")
        if amax.numel() == 1:
            amax = amax.reshape(())
        if hasattr(wq, "_amax"):
            wq.reset_amax()
        wq.amax = amax.to(w.device)


class MseCalibrator(_Calibrator):
    """amax multiplier sweep minimising the QDQ error -- calib/mse.py:31-172.  quant_func(x, amax) is
    supplied by the quantizer (model_calib.py:639-662) and runs our QDQ kernels; the candidate loop and the
    argmin stay as in the reference."""

    def __init__(self, amax, axis=None, step_size=0.1, start_multiplier=0.25, stop_multiplier=4.0,
                 quant_func=None, error_func=None, fused_format=None):
        """fused_format = (num_bits, unsigned, narrow_range) of the quantizer behind `quant_func` (INT-k, or
        num_bits == (4, 3) for FP8): with the default squared error all candidates are then evaluated by ONE
        kernel (ops.mse_sweep) instead of the reference's per-candidate QDQ / error / reduce passes."""
        super().__init__(num_bits=None, axis=axis, unsigned=None)
        self._fused_format = fused_format
        self._initial_amax = amax
        self._initial_amax_host = None
        self._num_steps = math.ceil((stop_multiplier - start_multiplier) / step_size) + 1
        self._start_multiplier, self._stop_multiplier = start_multiplier, stop_multiplier
        self._quant_func, self._error_func = quant_func, error_func
        self._losses_sum = None
        self._candidates = None
        self._amax = None
        self._cand_table = None  # fp32 [K, n] on the device: every candidate amax, built once per calibrator
        self._loss_acc = None    # fp32 [K, n]: the fused path's running losses (rows are the entries of _losses_sum)

    def _candidate_amax_table(self, device):
        """All K candidate amax values as ONE fp32 [K, n] device tensor.  Values: exactly what _compute_candidate_amax
        gives per candidate (0-dim fp32 multiplier x amax tensor on the HOST: float product, one rounding to the amax
        dtype), formed as one broadcast fp32 product on the tensor's device and rounded to that dtype once -- instead of K host products,
        K device->host reads of the multiplier and K uploads per collect (a static-block weight has ~460 K amax entries:
        65 ms per quantizer; an input quantizer paid the 39 synchronisations on every batch)."""
        from . import numerics

        if self._cand_table is None or self._cand_table.device != device:
            a = self._initial_amax.detach().to(device)
            if not numerics.on_host():
                # numerics "device": the reference's run on THIS device -- its own expression, candidate by candidate (a GPU
                # casts the 0-dim multiplier to a 16-bit amax's dtype BEFORE the product, and its linspace is the device's);
                # K small launches, once per calibrator
                mult = self._generate_candidates(device)
                self._cand_table = torch.stack([(a * m).float().reshape(-1) for m in mult.unbind(0)])
                return self._cand_table
            # the multipliers come from the HOST linspace (torch's CPU and GPU linspace differ in the last ulp); the product
            # itself is one IEEE fp32 multiply and one round-to-nearest-even conversion per entry -- the same on any device
            mult = torch.linspace(self._start_multiplier, self._stop_multiplier, steps=self._num_steps)  # as _generate_candidates
            # dtype of `amax * 0-dim fp32 multiplier`: the amax dtype for a dimensioned amax, fp32 for a 0-dim one
            out_dt = torch.result_type(a, mult[0])
            self._cand_table = (a.float().reshape(1, -1) * mult.to(device).reshape(-1, 1)).to(out_dt).float()
        return self._cand_table

    def _generate_candidates(self, device):
        # calib/mse.py:69-73.  The multipliers are generated on the host and copied: torch's CPU and GPU linspace
        # differ in the last ulp for some steps, which moves bf16 / f16 candidate amax values by one ulp; the host
        # values are the ones the (CPU-run) reference fixtures are pinned to, and they are device independent.
        # (numerics "device": the reference's own call, on the device -- its run there)
        from . import numerics

        if not numerics.on_host():
            return torch.linspace(self._start_multiplier, self._stop_multiplier, steps=self._num_steps, device=device)
        return torch.linspace(self._start_multiplier, self._stop_multiplier, steps=self._num_steps).to(device)

    def _compute_candidate_amax(self, candidates):
        if candidates.ndim != 0:  # final compute_amax: dimensioned x dimensioned promotes to fp32, unambiguous
            candidates = candidates.view_as(self._initial_amax)
            return self._initial_amax * candidates
        # 0-dim multiplier x amax tensor keeps the amax dtype (bf16 / f16 for 16-bit weights), and torch's CPU and
        # GPU kernels round that product differently (the GPU casts the 0-dim operand to the 16-bit dtype first).
        # One ulp of amax moves a clipping-dominated loss by >10 %, so the product is formed with the host
        # arithmetic the CPU-run reference fixtures are pinned to -- identical on every device.
        # (numerics "device": torch's own product on the amax's device -- the reference's run on that device)
        from . import numerics

        if not numerics.on_host():
            return self._initial_amax * candidates.to(self._initial_amax.device)
        if self._initial_amax_host is None:
            self._initial_amax_host = self._initial_amax.detach().cpu()
        return (self._initial_amax_host * candidates.detach().cpu()).to(self._initial_amax.device)

    @torch.no_grad()
    def collect(self, x):
        if self._quant_func is None and self._fused_format is None:
            raise RuntimeError("Quantization function not set.")
        candidates = self._generate_candidates(x.device)
        if self._candidates is None:
            self._candidates = candidates
            self._losses_sum = [None] * len(candidates)
        reduce_axis = convert_quantization_axis_to_reduce_axis(x, self._axis)
        if self._fused_format is not None and self._error_func is None and x.is_cuda and len(candidates) <= 64:
            # candidate amax values with the reference's dtype promotion (0-dim multiplier x amax tensor), then
            # one pass over x for all of them; the fp32 upcast of x (mse.py:92) happens in registers
            cand = self._candidate_amax_table(x.device)
            nb, uns, narrow = self._fused_format
            # the running sums of all K candidates live in ONE [K, n] tensor (mse_sweep accumulates into it); the list the
            # reference keeps (`_losses_sum`, read by compute_amax) holds views of its rows
            first = self._loss_acc is None
            self._loss_acc = ops.mse_sweep(x.detach(), cand, reduce_axis, nb, uns, narrow, loss=self._loss_acc)
            if first:
                rows = self._loss_acc.unbind(0)
                self._losses_sum = [r.reshape(()) for r in rows] if reduce_axis is None else list(rows)
            return
        x = x.detach().to(dtype=torch.float32)
        for step, candidate in enumerate(candidates):
            xq = self._quant_func(x, self._compute_candidate_amax(candidate))
            error = self._error_func(x, xq) if self._error_func is not None else (x - xq) ** 2
            loss = error.sum() if reduce_axis is None else error.sum(dim=reduce_axis)
            self._losses_sum[step] = loss.clone() if self._losses_sum[step] is None else self._losses_sum[step] + loss

    def reset(self):
        self._losses_sum = None
        self._candidates = None
        self._amax = None
        self._cand_table = None
        self._loss_acc = None

    @torch.no_grad()
    def compute_amax(self, verbose=False):
        if self._losses_sum is None or not any(v is not None for v in self._losses_sum):
            return None
        losses = torch.stack([v for v in self._losses_sum])
        best = torch.argmin(losses, dim=0)
        self._amax = self._compute_candidate_amax(self._candidates[best])
        return self._amax
