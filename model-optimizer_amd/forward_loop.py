"""The calibration forward loop of the path's callers (modelopt/torch/utils/dataset_utils.py:963-1255):
`create_forward_loop(dataloader=...)` returns the `forward_loop(model)` every algorithm of model_calib takes -- batches
of keyword tensors through the model under no_grad with KV caching switched off, a batch that runs out of memory halved
and retried, and the largest size that worked remembered so that later batches are cut before they fail.

What is not here: the dataset side of the reference's helper (`dataset_name` -> a Hugging Face hub download + tokenizer,
dataset_utils.py:257-940).  There is no network on this system; `synthetic_token_batches` is the stand-in the bench and the
tools use (random token ids of BASELINE.json's calibration shape), and a caller with real data passes its own DataLoader.
"""

from __future__ import annotations

import contextlib
import warnings

import torch

_NESTED_CONFIGS = ("text_config",)  # configs that keep the language model's settings one level down (:943-960)
_GENERATE_FAMILIES = ("t5", "bart", "whisper", "diffusiongemma")  # calibrated through .generate (:1258-1271)


def model_type_is_enc_dec(model) -> bool:
    name = model.__class__.__name__.lower()
    return any(family in name for family in _GENERATE_FAMILIES)


@contextlib.contextmanager
def use_cache_disabled(model):
    """config.use_cache (and text_config.use_cache) False inside the block, put back -- or removed again, if the config
    never had the attribute -- afterwards (:963-993).  A calibration pass has no use for a KV cache: it costs memory, and
    a cache object that survives into a replayed layer call carries keys of weights a calibration may since have changed."""
    touched, seen = [], set()
    top = getattr(model, "config", None)
    for cfg in ([top] + [getattr(top, a, None) for a in _NESTED_CONFIGS]) if top is not None else []:
        if cfg is None or id(cfg) in seen:
            continue
        seen.add(id(cfg))
        had = hasattr(cfg, "use_cache")
        touched.append((cfg, had, cfg.use_cache if had else None))
        cfg.use_cache = False
    try:
        yield
    finally:
        for cfg, had, before in reversed(touched):
            if had:
                cfg.use_cache = before
            else:
                with contextlib.suppress(AttributeError):
                    delattr(cfg, "use_cache")


def _rows(batch: dict, lo: int, hi: int) -> dict:
    return {k: (None if v is None else v[lo:hi, ...]) for k, v in batch.items()}


def run_batch(batch: dict, infer, largest_ok: int | None = None, allowed_non_tensor_keys=None) -> int:
    """One batch (a dict of tensors with the samples on dim 0) through `infer(**batch)`; returns the largest batch size
    known to work afterwards (dataset_utils.py:1083-1154, call for call).

    * larger than `largest_ok`: cut into pieces of that size first;
    * out of memory: halve (first half the larger one) and run the halves, the second with what the first learned;
      a single sample that does not fit is an error.
    The reference steps through a pre-cut batch with the size it knew at the start while it slices with the size it knows
    NOW; when a piece runs out of memory after all, the samples between the two are not run.  Kept as it is (the same
    samples reach the model as under the reference); on 288 GB the branch is hard to reach."""
    allowed = allowed_non_tensor_keys or set()
    assert all(torch.is_tensor(v) or v is None or k in allowed for k, v in batch.items()), \
        f"batch_data values must be tensors or None, except for keys: {allowed}."
    n = batch[next(iter(batch))].shape[0]
    if largest_ok is not None and n > largest_ok:
        for lo in range(0, n, largest_ok):
            largest_ok = run_batch(_rows(batch, lo, min(lo + largest_ok, n)), infer, largest_ok, allowed_non_tensor_keys)
        return largest_ok
    try:
        infer(**batch)
        return n if largest_ok is None else max(n, largest_ok)
    except torch.cuda.OutOfMemoryError:
        assert n > 1, ("CUDA out of memory error occurred while processing a single sample. This indicates the model is too "
                       "large for the available GPU memory. Consider reducing the model size, using a smaller "
                       "max_sample_length, or using a GPU with more memory.")
    mid = (n + 1) // 2
    warnings.warn(f"CUDA out of memory with batch size {n}, trying with batch size {mid}")
    # (None entries stay None in the halves; the reference's halving indexes them and fails)
    learned = run_batch(_rows(batch, 0, mid), infer, allowed_non_tensor_keys=allowed_non_tensor_keys)
    return run_batch(_rows(batch, mid, n), infer, learned, allowed_non_tensor_keys)


def _forward_loop(model, dataloader, allowed_non_tensor_keys=None):
    with use_cache_disabled(model), torch.no_grad():
        # the module is CALLED (hooks run), not its .forward; generate() families calibrate through their decoding loop
        infer = model.generate if model_type_is_enc_dec(model) else model
        largest_ok = None
        for batch in dataloader:
            largest_ok = run_batch(batch, infer, largest_ok, allowed_non_tensor_keys)


def create_forward_loop(model=None, dataset_name: str | None = None, tokenizer=None, batch_size: int = 1,
                        num_samples: int = 512, max_sample_length: int = 512, device=None, include_labels: bool = False,
                        dataloader=None, allowed_non_tensor_keys=None):
    """dataset_utils.create_forward_loop (:1183-1255): `forward_loop(model)` over `dataloader` (any iterable of dicts of
    tensors).  Without a dataloader the reference builds one from a hub dataset; that needs the network and is refused
    here with the way out."""
    if dataloader is None:
        raise ValueError("create_forward_loop: pass `dataloader=` (an iterable of {'input_ids': ..., ...} batches). "
                         f"Building one from dataset_name={dataset_name!r} downloads from the Hugging Face hub, which this "
                         "system cannot reach; forward_loop.synthetic_token_batches(...) gives random token ids of the "
                         "same shape")
    return lambda m: _forward_loop(m, dataloader, allowed_non_tensor_keys)


def synthetic_token_batches(vocab_size: int, num_samples: int = 512, max_sample_length: int = 512, batch_size: int = 8,
                            device=None, seed: int = 1234) -> list:
    """Random token ids in the shape of BASELINE.json's calibration set (512 samples x 512 tokens): a list of
    {"input_ids": int64 [batch, length]} batches, the last one ragged.  Seeded like the reference's example
    (hf_ptq.py:98)."""
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(0, vocab_size, (num_samples, max_sample_length), generator=g)
    if device is not None:
        ids = ids.to(device)
    return [{"input_ids": ids[i:i + batch_size]} for i in range(0, num_samples, batch_size)]


def get_max_batch_size(model, max_sample_length: int = 512, sample_memory_usage_ratio: float = 1.0,
                       sample_input_single_batch=None, enable_grad: bool = False) -> int:
    """The calibration batch size the device's free memory allows (:996-1080): one sample is run, the memory it took
    (free-memory drop or allocator peak, whichever is larger) divides the free memory, the candidate is tried (halved
    while it runs out of memory) and rounded down to 1, 2 or a multiple of 4, at most 512."""
    def free_and_peak():
        least, peak = torch.cuda.get_device_properties(0).total_memory, 0
        for d in range(torch.cuda.device_count()):
            free = torch.cuda.mem_get_info(d)[0]
            if free < least:
                least, peak = free, torch.cuda.max_memory_allocated(d)
        return least, peak

    torch.cuda.empty_cache()
    free0, peak0 = free_and_peak()
    infer = model.generate if model_type_is_enc_dec(model) else model
    one = sample_input_single_batch
    if one is None:
        one = torch.ones([1, max_sample_length], dtype=torch.int32, device=model.device) * 100
    with use_cache_disabled(model):
        with torch.set_grad_enabled(enable_grad):
            infer(one)
        free1, peak1 = free_and_peak()
        per_sample = max(free0 - free1, peak1 - peak0) * sample_memory_usage_ratio
        target = max(int(free0 / per_sample), 1) if per_sample > 0 else 1
        while target > 1:
            with torch.set_grad_enabled(enable_grad):
                try:
                    infer(one.expand([target, *one.shape[1:]]))
                    break
                except torch.cuda.OutOfMemoryError:
                    target //= 2
                    torch.cuda.empty_cache()
    if target < 2:
        return 1
    if target < 4:
        return 2
    return target // 4 * 4 if target < 512 else 512
