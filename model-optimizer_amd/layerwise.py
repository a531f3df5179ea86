"""Layer-by-layer calibration with checkpoint / resume -- the data path of model_calib.layerwise_calibrate
(quantization/model_calib.py:2050-2190, utils/layerwise_calib.py): calibrate decoder layer N on its own cached
inputs, hand its outputs to layer N + 1, persist per-layer quantizer state so that an interrupted run resumes at the
first unfinished layer.  Only one layer's activations are alive at a time, which is what lets models beyond one
GPU's HBM (or long calibration sets) through the same kernels.

Two ways of producing layer N + 1's inputs:

* ``capture="parent"`` (default; the reference's skip / run / capture strategy, utils/layerwise_calib.py:107-466): the
  PARENT model's forward is run again for every layer.  A `DecoderWalk` gives each decoder layer a role for that run --
  finished layers return shape-only placeholders on the meta device (no compute, no memory), the layer just before the
  target replays the inputs recorded for it (whatever the parent passes in is ignored), the target records its call
  and stops the forward.  Whatever the parent computes BETWEEN the blocks -- per-layer attention masks (sliding /
  full), per-layer position tables, per-layer keyword arguments -- therefore reaches every layer as in a whole-model
  pass.  A parent that runs real-device arithmetic on the hidden state between two blocks fails on the placeholders;
  the error says so and names the other mode.
* ``capture="handover"``: the captured positional / keyword arguments of the FIRST layer call are replayed for every
  layer with the hidden states (first positional argument) replaced by the previous layer's output -- no parent
  re-run; exact for stacks that call every block with the same arguments (the Llama family of BASELINE.json).
"""

from __future__ import annotations

import json
import os
import warnings
from collections import deque

import torch
from torch import nn

from .tensor_quantizer import TensorQuantizer


class _EarlyStop(Exception):
    pass


def get_decoder_layers(model: nn.Module):
    """The first nn.ModuleList of >= 2 same-typed children (LayerActivationCollector.get_decoder_layers)."""
    for m in model.modules():
        if isinstance(m, nn.ModuleList) and len(m) >= 2 and len({type(c) for c in m}) == 1:
            return m
    return None


def _capture_inputs(model, layer, forward_loop):
    """(args, kwargs) of every call of `layer` during forward_loop; the model forward stops at that layer."""
    calls = []

    def hook(mod, args, kwargs):
        calls.append((tuple(a.detach() if isinstance(a, torch.Tensor) else a for a in args), dict(kwargs)))
        raise _EarlyStop

    handle = layer.register_forward_pre_hook(hook, with_kwargs=True)
    orig_call = model.__class__.__call__

    def guarded(m):  # forward_loop may call the model many times: swallow the early stop of each call
        class _Guard:
            def __call__(self_inner, *a, **k):
                try:
                    return orig_call(m, *a, **k)
                except _EarlyStop:
                    return None

            def __getattr__(self_inner, name):
                return getattr(m, name)

        return _Guard()

    try:
        forward_loop(guarded(model))
    finally:
        handle.remove()
    return calls


def without_cache(kwargs: dict) -> dict:
    """Keyword arguments of a decoder-layer call without its KV-cache objects: a layer replayed outside its model must
    not append to (or attend over) a cache another call filled -- the second replay would see the keys / values of the
    first, computed from weights a weight-mutating calibration has since changed."""
    kwargs = dict(kwargs)
    for k in ("past_key_values", "past_key_value"):
        if kwargs.get(k) is not None:
            kwargs[k] = None
    if kwargs.get("use_cache"):
        kwargs["use_cache"] = False
    return kwargs


def _first_tensor(out):
    return out[0] if isinstance(out, (tuple, list)) else out


# ------------------------------------------------------------------------------------------- the parent-forward walk
def _describe(out):
    """Shape-only description of a layer output (tensors -> (shape, dtype); tuples / lists keep their structure; other
    values ride along as they are): what a finished layer's placeholder is rebuilt from, and what a checkpoint stores
    for it (utils/layerwise_calib.py:176-216)."""
    if isinstance(out, torch.Tensor):
        return ("tensor", tuple(out.shape), out.dtype)
    if isinstance(out, tuple):
        return ("tuple", tuple(_describe(o) for o in out))
    if isinstance(out, list):
        return ("list", [_describe(o) for o in out])
    return ("other", out)


def _placeholder(desc):
    kind = desc[0]
    if kind == "tensor":
        return torch.zeros(desc[1], dtype=desc[2], device="meta")
    if kind == "tuple":
        return tuple(_placeholder(d) for d in desc[1])
    if kind == "list":
        return [_placeholder(d) for d in desc[1]]
    return desc[1]


class DecoderWalk:
    """Per-layer roles for runs of the PARENT model's forward (context manager: the layers' and the model's `forward`
    are shadowed on the instances while it is entered and put back on exit).

    PASS    the layer's own forward (its output shape is noted);
    SKIP    a meta-device placeholder of the noted output shape: a finished layer costs nothing;
    REPLAY  the next recorded (args, kwargs) of this layer go through its own forward, the parent's arguments are
            dropped -- upstream placeholders are never consumed;
    RECORD  the call's arguments are stored (KV-cache objects removed) and the model's forward ends there.
    """

    PASS, SKIP, REPLAY, RECORD = "pass", "skip", "replay", "record"

    def __init__(self, model: nn.Module, layers):
        self.model = model
        self.layers = list(layers)
        n = len(self.layers)
        self.role = [self.PASS] * n
        self.inputs: list = [None] * n  # the recorded calls of a layer while they are still needed
        self.out_desc: list = [None] * n
        self._queue: list = [deque() for _ in range(n)]
        self._recorded: list = [[] for _ in range(n)]
        self._shadowed: list = []
        self._entered = False

    # -- install / remove
    def _shadow(self, module, make):
        had = "forward" in module.__dict__
        self._shadowed.append((module, had, module.__dict__.get("forward")))
        module.forward = make(module.forward)

    def __enter__(self):
        try:
            for i, layer in enumerate(self.layers):
                self._shadow(layer, lambda own, i=i: self._layer_forward(i, own))
            self._shadow(self.model, self._model_forward)
        except Exception:
            self.__exit__(None, None, None)
            raise
        self._entered = True
        return self

    def __exit__(self, *exc):
        for module, had, previous in reversed(self._shadowed):
            if had:
                module.forward = previous
            elif "forward" in module.__dict__:
                del module.forward
        self._shadowed = []
        self._entered = False
        return False

    def _layer_forward(self, i, own):
        def forward(*args, **kwargs):
            role = self.role[i]
            if role == self.SKIP:
                if self.out_desc[i] is None:
                    raise RuntimeError(f"decoder layer {i} is skipped but its output shape was never noted")
                return _placeholder(self.out_desc[i])
            if role == self.RECORD:
                self._recorded[i].append((tuple(a.detach() if isinstance(a, torch.Tensor) else a for a in args),
                                          without_cache(kwargs)))
                raise _EarlyStop
            if role == self.REPLAY:
                if not self._queue[i]:
                    raise RuntimeError(f"forward_loop reached decoder layer {i} more often than when its inputs were "
                                       "recorded: layer-by-layer calibration needs a forward_loop that makes the same "
                                       "model calls every time it is run")
                args, kwargs = self._queue[i].popleft()
            out = own(*args, **kwargs)
            self.out_desc[i] = _describe(out)
            return out

        return forward

    def _model_forward(self, own):
        def forward(*args, **kwargs):
            try:
                return own(*args, **kwargs)
            except _EarlyStop:
                return None
            except (RuntimeError, NotImplementedError) as e:
                if "meta" not in str(e).lower() or not any(r == self.SKIP for r in self.role):
                    raise
                raise RuntimeError("layer-by-layer calibration stands in for finished decoder layers with meta-device "
                                   "placeholders; this model computes on the hidden state BETWEEN its decoder blocks, "
                                   "which a placeholder cannot serve.  Use capture='handover' (hidden states passed "
                                   "from layer to layer) or a whole-model calibration") from e

        return forward

    # -- one run of the parent
    def capture(self, target: int, forward_loop) -> list:
        """Inputs of layer `target`: layers before target - 1 are skipped, layer target - 1 replays its own recorded
        inputs (so its CURRENT weights and quantizer state shape what the target sees), the target records."""
        assert self._entered, "DecoderWalk.capture outside its context"
        n = len(self.layers)
        for i in range(n):
            self.role[i] = self.SKIP if i < target - 1 else self.PASS
        if target > 0:
            feed = self.inputs[target - 1]
            if not feed:
                raise RuntimeError(f"decoder layer {target - 1} has no recorded inputs to replay: layers are walked in order")
            self.role[target - 1] = self.REPLAY
            self._queue[target - 1] = deque(feed)
            if target > 1:
                self.inputs[target - 2] = None
        self.role[target] = self.RECORD
        self._recorded[target] = []
        try:
            forward_loop(self.model)
        finally:
            got, self._recorded[target] = self._recorded[target], []
            left = len(self._queue[target - 1]) if target > 0 else 0
            if target > 0:
                self._queue[target - 1] = deque()
            for i in range(n):
                self.role[i] = self.SKIP if i < target - 1 else self.PASS
        if not got:
            raise RuntimeError(f"forward_loop never reached decoder layer {target}")
        if left:
            raise RuntimeError(f"forward_loop reached decoder layer {target - 1} {left} time(s) less often than when its "
                               "inputs were recorded: layer-by-layer calibration needs a forward_loop that makes the "
                               "same model calls every time it is run")
        self.inputs[target] = got
        return got

    def seed(self, target: int, inputs: list, out_descs: list):
        """Resume: the walk continues at `target` with inputs from a checkpoint; earlier layers are finished."""
        for i in range(target):
            self.out_desc[i] = out_descs[i]
            self.role[i] = self.SKIP
            self.inputs[i] = None
        self.inputs[target] = inputs


# ------------------------------------------------------------------------------------------------ checkpoint files
def _quantizer_state(layer):
    return {n: {k: v.detach().cpu() for k, v in q.state_dict().items()}
            for n, q in layer.named_modules() if isinstance(q, TensorQuantizer)}


def _load_quantizer_state(layer, state):
    for n, q in layer.named_modules():
        if isinstance(q, TensorQuantizer) and n in state:
            for k, v in state[n].items():
                name = k.lstrip("_")
                dev = next(layer.parameters()).device
                if name == "amax":
                    q.replace_amax(v.to(dev))
                elif name == "pre_quant_scale":
                    q.pre_quant_scale = v.to(dev)


def _atomic_save(obj, path: str):
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        torch.save(obj, f)
        f.flush()
        os.fsync(f.fileno())
    os.replace(tmp, path)


def _to_device(obj, device):
    """Tensors inside tuples / lists / dicts moved to `device`; everything else as it is."""
    if isinstance(obj, torch.Tensor):
        return obj.to(device)
    if isinstance(obj, dict):
        return {k: _to_device(v, device) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_device(v, device) for v in obj)
    return obj


def _read_manifest(checkpoint_dir, n_layers, save_every, calib_mutates_weights):
    """(completed layers, manifest) of a checkpoint directory.  A directory written under other settings is refused
    (utils/layerwise_calib.py:617-629): per-layer files of another model depth, another window or another restore mode
    do not compose with this run's."""
    path = os.path.join(checkpoint_dir, "manifest.json")
    if not os.path.exists(path):
        return 0, None
    try:
        with open(path) as f:
            man = json.load(f)
    except (json.JSONDecodeError, OSError):
        return 0, None
    for key, now in (("num_layers", n_layers), ("save_every", save_every), ("calib_mutates_weights", calib_mutates_weights)):
        was = man.get(key)
        if was is not None and was != now:
            raise ValueError(f"Checkpoint {key} mismatch: manifest has {was!r} but new run uses {now!r}. "
                             "Use a fresh checkpoint directory.")
    return int(man.get("completed", 0)), man


@torch.no_grad()
def _replay_to_layer(model, layers, start: int, forward_loop, qdq_from_prev_layer: bool = False):
    """Inputs of layer `start` when the saved ones cannot be trusted: the forward loop is run up to that layer through
    the restored (calibrated) layers 0 .. start-1 -- with their quantizers bypassed, as the saved inputs were produced,
    or ACTIVE when the run feeds every layer the quantized output of its predecessor (`qdq_from_prev_layer`).

    The replay is exact for calibrations that leave the weights alone (max / mse / histogram).  After a
    weight-mutating calibration (AWQ fold, SparseGPT, GPTQ-style updates) the restored layers carry their FINAL weights,
    whereas the lost inputs came from layer N-1 before / while it was calibrated: the replayed activations then differ
    from an uninterrupted run's by that layer's update (the caller warns)."""
    qs = [q for lyr in list(layers)[:start] for q in lyr.modules() if isinstance(q, TensorQuantizer)]
    saved = [(q._disabled, q._if_calib) for q in qs]
    for q in qs:
        if not qdq_from_prev_layer:
            q._disabled = True
        q._if_calib = False
    try:
        inputs = _capture_inputs(model, layers[start], forward_loop)
    finally:
        for q, (d, c) in zip(qs, saved):
            q._disabled, q._if_calib = d, c
    if not inputs:
        raise RuntimeError(f"forward_loop never reached decoder layer {start}")
    return inputs


class _QuantizersOff:
    """The layer's quantizers bypassed and not collecting (set_quantizer_by_cfg_context(layer, "*": enable False),
    model_calib.py:2152-2155): the activations handed on are those of a whole-model pass."""

    def __init__(self, layer):
        self.qs = [q for q in layer.modules() if isinstance(q, TensorQuantizer)]

    def __enter__(self):
        self.saved = [(q._disabled, q._if_calib) for q in self.qs]
        for q in self.qs:
            q._disabled, q._if_calib = True, False

    def __exit__(self, *exc):
        for q, (d, c) in zip(self.qs, self.saved):
            q._disabled, q._if_calib = d, c
        return False


@torch.no_grad()
def layerwise_calibrate(model: nn.Module, forward_loop, calib_func, layers=None, checkpoint_dir: str | None = None,
                        get_qdq_activations_from_prev_layer: bool = False, calib_mutates_weights: bool = True,
                        save_every: int = 1, capture: str = "parent", **calib_kwargs):
    """calib_func(layer, layer_forward_loop, **calib_kwargs) is run once per decoder layer, e.g.
    model_calib.max_calibrate / mse_calibrate / awq.  Returns the number of layers calibrated in THIS call
    (smaller than len(layers) after a resume).

    get_qdq_activations_from_prev_layer: layer N + 1's inputs are taken AFTER layer N was calibrated, quantizers
    active (GPTQ's default: quantization error and weight updates propagate); otherwise BEFORE, quantizers bypassed
    (the activations of a whole-model pass).  save_every: the per-layer file is written after every layer, the next
    layer's inputs and the manifest only every `save_every` layers (and after the last): a crash inside a window
    resumes at the window's start (utils/layerwise_calib.py:742-798)."""
    if forward_loop is None:
        raise ValueError("forward_loop must not be None for layerwise calibration.")
    if capture not in ("parent", "handover"):
        raise ValueError(f"capture must be 'parent' or 'handover', got {capture!r}")
    if not isinstance(save_every, int) or save_every < 1:
        raise ValueError(f"save_every must be an integer >= 1, got {save_every!r}")
    layers = layers if layers is not None else get_decoder_layers(model)
    if layers is None or len(layers) == 0:
        raise ValueError("Could not find transformer layers in model.")
    n_layers = len(layers)
    start, inputs, out_descs = 0, None, [None] * n_layers
    manifest_path = os.path.join(checkpoint_dir, "manifest.json") if checkpoint_dir else None
    if checkpoint_dir:
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            # (utils/layerwise_calib.py:574-579) every rank would write the same files; a resumed rank would restore
            # state its peers do not have
            raise RuntimeError("Layerwise calibration checkpointing is not supported in multi-process distributed jobs. "
                               "Use single-process calibration or disable checkpointing.")
        os.makedirs(checkpoint_dir, exist_ok=True)
        start, _ = _read_manifest(checkpoint_dir, n_layers, save_every, bool(calib_mutates_weights))
        for i in range(start):  # restore finished layers
            blob = torch.load(os.path.join(checkpoint_dir, f"layer_{i:04d}.pt"), weights_only=False)
            _load_quantizer_state(layers[i], blob["quantizers"])
            if blob.get("weights") is not None:
                layers[i].load_state_dict(blob["weights"], strict=False)
            out_descs[i] = blob.get("output")
        if 0 < start < n_layers:
            dev = next(layers[start].parameters()).device
            # the checkpoint directory is the user's own (pickled kwargs of the layer calls: weights_only
            # cannot apply); what is checked is that the inputs belong to the layer the manifest resumes at
            nxt = torch.load(os.path.join(checkpoint_dir, "next_inputs.pt"), weights_only=False)
            if not isinstance(nxt, dict) or nxt.get("for_layer") != start:
                found = nxt.get("for_layer") if isinstance(nxt, dict) else "an older format"
                warnings.warn(f"layerwise_calibrate: {checkpoint_dir}/next_inputs.pt holds the inputs of layer "
                              f"{found}, the manifest resumes at layer {start} (interrupted checkpoint write); "
                              "re-capturing the inputs by replaying the finished layers"
                              + ("; the calibration mutates weights, so the replayed activations are those of "
                                 "the FINAL weights of the finished layers -- approximate, not those of an "
                                 "uninterrupted run" if calib_mutates_weights else ""))
                inputs = _replay_to_layer(model, layers, start, forward_loop, get_qdq_activations_from_prev_layer)
            else:
                inputs = _to_device(nxt["inputs"], dev)
    if start >= n_layers:
        return 0

    def first_tensor_desc(args):  # handover mode / old checkpoints: a finished layer returned its hidden states' shape
        return _describe(args[0]) if args and isinstance(args[0], torch.Tensor) else None

    walk = DecoderWalk(model, layers) if capture == "parent" else None
    done = 0
    try:
        if walk is not None:
            walk.__enter__()
        if inputs is None:
            inputs = walk.capture(0, forward_loop) if walk is not None else _capture_inputs(model, layers[0], forward_loop)
            if not inputs:
                raise RuntimeError("forward_loop never reached the first decoder layer")
        inputs = [(args, without_cache(kwargs)) for args, kwargs in inputs]
        if walk is not None and start > 0:
            for i in range(start):
                if out_descs[i] is None:
                    out_descs[i] = first_tensor_desc(inputs[0][0])
            walk.seed(start, inputs, out_descs)
        elif walk is not None:
            walk.inputs[0] = inputs
        for idx in range(start, n_layers):
            layer = layers[idx]

            def layer_loop(m, _inputs=inputs):
                for args, kwargs in _inputs:
                    m(*args, **kwargs)

            def next_layer_inputs(_inputs=inputs, _idx=idx, _layer=layer):
                if walk is not None:
                    return walk.capture(_idx + 1, forward_loop)
                outs = []
                for args, kwargs in _inputs:
                    h = _first_tensor(_layer(*args, **kwargs)).detach()
                    outs.append(((h, *args[1:]), kwargs))
                return outs

            is_last = idx + 1 >= n_layers
            next_inputs = None
            if not is_last and not get_qdq_activations_from_prev_layer:
                # inputs of layer N+1 from the un-quantized layer N: same activations as a whole-model calibration pass
                with _QuantizersOff(layer):
                    next_inputs = next_layer_inputs()
            calib_func(layer, layer_loop, **calib_kwargs)
            if not is_last and get_qdq_activations_from_prev_layer:
                next_inputs = next_layer_inputs()  # with quantizers active and any weight updates (GPTQ-style)
            if checkpoint_dir:
                # Crash-safe: every file is written under a temporary name and moved into place (os.replace is atomic),
                # the manifest LAST; the saved inputs carry the index of the layer they feed, so a resume that finds
                # inputs of another layer than the manifest names (a crash between the two moves) refuses them instead
                # of calibrating a layer on its successor's activations.
                desc = walk.out_desc[idx] if walk is not None else None
                blob = {"quantizers": _quantizer_state(layer),
                        "weights": {k: v.detach().cpu() for k, v in layer.state_dict().items()
                                    if "quantizer" not in k} if calib_mutates_weights else None,
                        "output": desc if desc is not None else first_tensor_desc(inputs[0][0])}
                _atomic_save(blob, os.path.join(checkpoint_dir, f"layer_{idx:04d}.pt"))
                if is_last or (idx + 1) % save_every == 0:
                    if next_inputs is not None:
                        _atomic_save({"for_layer": idx + 1, "inputs": _to_device(next_inputs, "cpu")},
                                     os.path.join(checkpoint_dir, "next_inputs.pt"))
                    tmp = manifest_path + ".tmp"
                    with open(tmp, "w") as f:
                        json.dump({"num_layers": n_layers, "completed": idx + 1, "save_every": save_every,
                                   "calib_mutates_weights": bool(calib_mutates_weights)}, f)
                        f.flush()
                        os.fsync(f.fileno())
                    os.replace(tmp, manifest_path)
            inputs = next_inputs
            done += 1
    finally:
        if walk is not None:
            walk.__exit__(None, None, None)
    return done
