"""Layer-by-layer calibration with checkpoint / resume -- the data path of model_calib.layerwise_calibrate
(quantization/model_calib.py:2050-2190, utils/layerwise_calib.py): calibrate decoder layer N on its own cached
inputs, hand its outputs to layer N + 1, persist per-layer quantizer state so that an interrupted run resumes at the
first unfinished layer.  Only one layer's activations are alive at a time, which is what lets models beyond one
GPU's HBM (or long calibration sets) through the same kernels.

Scope: the reference re-runs the parent model's forward with skip / run / capture dummies so that inter-layer glue
executes naturally; here the captured positional / keyword arguments of the first layer call are replayed for every
layer with the hidden states (first positional argument) replaced by the previous layer's output -- the decoder-stack
contract of the Llama-family models of BASELINE.json.
"""

from __future__ import annotations

import json
import warnings
import os

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
                    if hasattr(q, "_amax"):
                        delattr(q, "_amax")
                    q.amax = v.to(dev)
                elif name == "pre_quant_scale":
                    q.pre_quant_scale = v.to(dev)


def _atomic_save(obj, path: str):
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        torch.save(obj, f)
        f.flush()
        os.fsync(f.fileno())
    os.replace(tmp, path)


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


@torch.no_grad()
def layerwise_calibrate(model: nn.Module, forward_loop, calib_func, layers=None, checkpoint_dir: str | None = None,
                        get_qdq_activations_from_prev_layer: bool = False, calib_mutates_weights: bool = True,
                        **calib_kwargs):
    """calib_func(layer, layer_forward_loop, **calib_kwargs) is run once per decoder layer, e.g.
    model_calib.max_calibrate / mse_calibrate / awq.  Returns the number of layers calibrated in THIS call
    (smaller than len(layers) after a resume)."""
    if forward_loop is None:
        raise ValueError("forward_loop must not be None for layerwise calibration.")
    layers = layers if layers is not None else get_decoder_layers(model)
    if layers is None or len(layers) == 0:
        raise ValueError("Could not find transformer layers in model.")
    n_layers = len(layers)
    start, inputs = 0, None
    manifest_path = os.path.join(checkpoint_dir, "manifest.json") if checkpoint_dir else None
    if checkpoint_dir:
        os.makedirs(checkpoint_dir, exist_ok=True)
        if os.path.exists(manifest_path):
            with open(manifest_path) as f:
                man = json.load(f)
            if man.get("num_layers") == n_layers:
                start = int(man["completed"])
                for i in range(start):  # restore finished layers
                    blob = torch.load(os.path.join(checkpoint_dir, f"layer_{i:04d}.pt"), weights_only=False)
                    _load_quantizer_state(layers[i], blob["quantizers"])
                    if blob.get("weights") is not None:
                        layers[i].load_state_dict(blob["weights"], strict=False)
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
                        inputs = _replay_to_layer(model, layers, start, forward_loop,
                                                  get_qdq_activations_from_prev_layer)
                    else:
                        inputs = [(tuple(a.to(dev) if isinstance(a, torch.Tensor) else a for a in args), kwargs)
                                  for args, kwargs in nxt["inputs"]]
    if start >= n_layers:
        return 0
    if inputs is None:
        inputs = _capture_inputs(model, layers[start], forward_loop)
        if not inputs:
            raise RuntimeError("forward_loop never reached the first decoder layer")
    inputs = [(args, without_cache(kwargs)) for args, kwargs in inputs]
    done = 0
    for idx in range(start, n_layers):
        layer = layers[idx]

        def layer_loop(m, _inputs=inputs):
            for args, kwargs in _inputs:
                m(*args, **kwargs)

        def outputs_of(lyr, _inputs=inputs):
            outs = []
            for args, kwargs in _inputs:
                h = _first_tensor(lyr(*args, **kwargs)).detach()
                outs.append(((h, *args[1:]), kwargs))
            return outs

        is_last = idx + 1 >= n_layers
        next_inputs = None
        if not is_last and not get_qdq_activations_from_prev_layer:
            # inputs of layer N+1 from the un-quantized layer N: same activations as a whole-model calibration pass
            qs = [q for q in layer.modules() if isinstance(q, TensorQuantizer)]
            saved = [(q._disabled, q._if_calib) for q in qs]
            for q in qs:
                q._disabled, q._if_calib = True, False
            try:
                next_inputs = outputs_of(layer)
            finally:
                for q, (d, c) in zip(qs, saved):
                    q._disabled, q._if_calib = d, c
        calib_func(layer, layer_loop, **calib_kwargs)
        if not is_last and get_qdq_activations_from_prev_layer:
            next_inputs = outputs_of(layer)  # with quantizers active and any weight updates (GPTQ-style)
        if checkpoint_dir:
            # Crash-safe: every file is written under a temporary name and moved into place (os.replace is atomic), the
            # manifest LAST; the saved inputs carry the index of the layer they feed, so a resume that finds inputs of
            # another layer than the manifest names (a crash between the two moves) refuses them instead of
            # calibrating a layer on its successor's activations.
            blob = {"quantizers": _quantizer_state(layer),
                    "weights": {k: v.detach().cpu() for k, v in layer.state_dict().items()
                                if "quantizer" not in k} if calib_mutates_weights else None}
            _atomic_save(blob, os.path.join(checkpoint_dir, f"layer_{idx:04d}.pt"))
            if next_inputs is not None:
                _atomic_save({"for_layer": idx + 1,
                              "inputs": [(tuple(a.cpu() if isinstance(a, torch.Tensor) else a for a in args), kwargs)
                                         for args, kwargs in next_inputs]},
                             os.path.join(checkpoint_dir, "next_inputs.pt"))
            tmp = manifest_path + ".tmp"
            with open(tmp, "w") as f:
                json.dump({"num_layers": n_layers, "completed": idx + 1}, f)
                f.flush()
                os.fsync(f.fileno())
            os.replace(tmp, manifest_path)
        inputs = next_inputs
        done += 1
    return done
