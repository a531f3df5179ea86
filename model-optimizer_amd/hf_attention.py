"""KV-cache quantizers on Hugging Face attention modules -- the mirror of `_QuantAttention` and
`register_hf_attentions_on_the_fly` (quantization/plugins/huggingface.py:78-360, :416-470) for this path.

An attention module that goes through transformers' attention interface (`ALL_ATTENTION_FUNCTIONS` /
`eager_attention_forward`, transformers >= 4.48) gets `q_bmm_quantizer`, `k_bmm_quantizer`, `v_bmm_quantizer`
(`TensorQuantizer`s, disabled until a config entry such as `*[kv]_bmm_quantizer` enables them).  During the module's
forward the interface function is replaced by one that passes the query / key / value states through those quantizers
first, and restored afterwards -- also when the forward raises.  The statistics and the QDQ are the same kernels as
for every other activation (per-tensor abs-max `moq_amax`, FP8 QDQ `moq_fake_quant_e4m3`); key / value states are
permuted dense views (`[B, heads, S, D]` of a `[B, S, heads, D]` buffer), which the per-tensor entries walk in place
(ops._flat_alias) instead of copying.

Attention classes that multiply their states themselves (no interface: GPT-J, CodeGen, MPT) get the quantizer calls
written into their own source, the reference's rule for which product takes which quantizer (legacy_quant_attention_class).

Not mirrored: the softmax (`p_bmm_quantizer`) path, which the reference runs on its own Triton / "kitchen" flash
attention kernels (huggingface.py:92-197) -- enabling it raises; T5's matmul patching (:315-352).
"""

from __future__ import annotations

import inspect
from functools import partial

from torch import nn

from .tensor_quantizer import QuantizerAttributeConfig, TensorQuantizer

_BMM_QUANTIZERS = ("q_bmm_quantizer", "k_bmm_quantizer", "v_bmm_quantizer", "p_bmm_quantizer")
_quant_classes: dict[type, type] = {}


def _attention_module_of(cls: type):
    return inspect.getmodule(cls)


class _QuantAttentionMixin:
    """forward = original forward with the attention interface wrapped by `_quantized_attention`."""

    def _setup_bmm_quantizers(self):
        for name in _BMM_QUANTIZERS:
            if not hasattr(self, name):
                setattr(self, name, TensorQuantizer(QuantizerAttributeConfig(enable=False)))

    @staticmethod
    def _quantized_attention(original_attention_interface, self, query_states, key_states, value_states, *args,
                             **kwargs):
        # huggingface.py:222-236
        q, k, v = self.q_bmm_quantizer, self.k_bmm_quantizer, self.v_bmm_quantizer
        plain = TensorQuantizer  # (a chain or a subclass takes the call)
        if not (type(q) is plain and q.hands_back()):  # the KV-cache presets leave the query quantizer disabled
            query_states = q(query_states)
        if not (type(k) is plain and k.hands_back()):
            key_states = k(key_states)
        if not (type(v) is plain and v.hands_back()):
            value_states = v(value_states)
        if self.p_bmm_quantizer.is_enabled:
            raise NotImplementedError("p_bmm_quantizer (softmax QDQ inside flash attention) is outside this path")
        return original_attention_interface(self, query_states, key_states, value_states, *args, **kwargs)

    def forward(self, *args, **kwargs):
        # huggingface.py:283-334: pick the function this forward is going to call, patch, run, restore
        config = getattr(self, "config", None)
        if config is None:
            config = next((getattr(m, "config", None) for m in self.children() if hasattr(m, "config")), None)
        impl = getattr(config, "_attn_implementation", None) if config is not None else None
        eager = impl is None or impl == "eager" or (impl == "sdpa" and kwargs.get("output_attentions", False))
        module = _attention_module_of(self._moq_original_cls)
        if eager:
            if not hasattr(module, "eager_attention_forward"):
                raise AssertionError(f"Module {module} does not have `eager_attention_forward` to enable KV Cache "
                                     "quantization. Please use a different attention implementation such as `sdpa`.")
            original = module.eager_attention_forward
            module.eager_attention_forward = partial(self._quantized_attention, original)
        else:
            original = module.ALL_ATTENTION_FUNCTIONS[impl]
            module.ALL_ATTENTION_FUNCTIONS[impl] = partial(self._quantized_attention, original)
        try:
            return super().forward(*args, **kwargs)
        finally:
            if eager:
                module.eager_attention_forward = original
            else:
                module.ALL_ATTENTION_FUNCTIONS[impl] = original


def is_compatible_attention(cls: type) -> bool:
    """huggingface.py:336-343: the class's module uses the attention interface."""
    return getattr(_attention_module_of(cls), "ALL_ATTENTION_FUNCTIONS", None) is not None


def _wraps_nested_attention(module: nn.Module) -> bool:
    """huggingface.py:355-368: a wrapper (e.g. ViTAttention around ViTSelfAttention) is not patched itself."""
    return any(child is not module and type(child).__name__.endswith("Attention") for _, child in module.named_modules())


def is_quantized_attention(m) -> bool:
    return isinstance(m, (_QuantAttentionMixin, _LegacyQuantAttentionMixin))


def convert_attention(module: nn.Module) -> nn.Module:
    """In-place class swap to a `Quant<cls>` subclass (one per attention class) + the bmm quantizers."""
    cls = type(module)
    if issubclass(cls, _QuantAttentionMixin):
        return module
    qcls = _quant_classes.get(cls)
    if qcls is None:
        qcls = type(f"Quant{cls.__name__}", (_QuantAttentionMixin, cls), {"_moq_original_cls": cls})
        _quant_classes[cls] = qcls
    module.__class__ = qcls
    module._setup_bmm_quantizers()
    return module


# ---------------------------------------------------------------------------------------------- attention without the interface
_BMM_CALLS = ("matmul", "bmm", "baddbmm")
_legacy_classes: dict[type, type | None] = {}


class _LegacyQuantAttentionMixin:
    """An attention class that multiplies its states itself (GPT-J, CodeGen, MPT, ...): the quantizer calls are written INTO
    the methods that hold the two products (legacy_quant_attention_class); nothing to patch at call time."""

    def _setup_bmm_quantizers(self):
        for name in _BMM_QUANTIZERS[:3]:
            if not hasattr(self, name):
                setattr(self, name, TensorQuantizer(QuantizerAttributeConfig(enable=False)))


def legacy_quant_attention_class(cls: type):
    """plugins/attention.py:45-210 (register_attention_for_kv_quant): for attention classes that do not go through the attention
    interface, the class's SOURCE is rewritten -- walked breadth-first like `ast.walk`, it must hold exactly two
    matmul / bmm / baddbmm calls, or exactly two `@` products, or one scaled_dot_product_attention call.  Of two calls the FIRST
    found gets `self.v_bmm_quantizer(.)` around its second operand and the second gets the query / key quantizers around its
    two operands, each between a transpose(-1, -2) pair (per-token layout for the key states); of two `@` products the second
    found takes q / k and the first v; the fused call takes all three.  Returns the subclass carrying the rewritten methods, or
    None when the source is not available or holds none of the three shapes."""
    if cls in _legacy_classes:
        return _legacy_classes[cls]
    import ast
    import textwrap
    import types

    _legacy_classes[cls] = None
    try:
        tree = ast.parse(textwrap.dedent(inspect.getsource(cls)))
    except (OSError, TypeError, SyntaxError):
        return None
    walk = list(ast.walk(tree))
    calls = [n for n in walk if isinstance(n, ast.Call) and isinstance(n.func, ast.Attribute)]
    bmm = [n for n in calls if n.func.attr in _BMM_CALLS]
    sdpa = [n for n in calls if n.func.attr == "scaled_dot_product_attention"]
    products = [n for n in walk if isinstance(n, ast.BinOp) and isinstance(n.op, ast.MatMult)]
    if len(bmm) != 2 and len(sdpa) != 1 and len(products) != 2:
        return None

    def through(name, operand, transposed):
        def swapped(x):
            return ast.Call(func=ast.Attribute(value=x, attr="transpose", ctx=ast.Load()),
                            args=[ast.Constant(value=-1), ast.Constant(value=-2)], keywords=[])

        q = ast.Attribute(value=ast.Name(id="self", ctx=ast.Load()), attr=name, ctx=ast.Load())
        inner = swapped(operand) if transposed else operand
        out = ast.Call(func=q, args=[inner], keywords=[])
        return swapped(out) if transposed else out

    touched = []
    if len(bmm) == 2:
        if min(len(bmm[0].args), len(bmm[1].args)) < 2:
            return None  # operands passed by keyword (Bloom's baddbmm): the reference's rewrite stops with an IndexError here
        bmm[0].args[1] = through("v_bmm_quantizer", bmm[0].args[1], False)
        bmm[1].args[0] = through("q_bmm_quantizer", bmm[1].args[0], True)
        bmm[1].args[1] = through("k_bmm_quantizer", bmm[1].args[1], True)
        touched += bmm
    if len(products) == 2:
        products[1].left = through("q_bmm_quantizer", products[1].left, False)
        products[1].right = through("k_bmm_quantizer", products[1].right, True)
        products[0].right = through("v_bmm_quantizer", products[0].right, False)
        touched += products
    if len(sdpa) == 1:
        if len(sdpa[0].args) < 3:
            return None
        for i, name in enumerate(_BMM_QUANTIZERS[:3]):
            sdpa[0].args[i] = through(name, sdpa[0].args[i], False)
        touched += sdpa
    classdef = next(n for n in tree.body if isinstance(n, ast.ClassDef))
    rewritten = [f.name for f in classdef.body if isinstance(f, (ast.FunctionDef, ast.AsyncFunctionDef))
                 and any(t is n for n in ast.walk(f) for t in touched)]
    classdef.name = "_MoqRewritten" + cls.__name__
    scope = vars(_attention_module_of(cls))  # the methods keep the defining module's globals
    try:
        exec(compile(ast.fix_missing_locations(tree), f"<KV-cache quantizers for {cls.__qualname__}>", "exec"), scope)  # noqa: S102
        body = vars(scope.pop(classdef.name))
    except Exception:  # noqa: BLE001 -- a class body this rewrite cannot re-evaluate: the caller warns
        scope.pop(classdef.name, None)
        return None
    methods = {n: body[n] for n in rewritten if isinstance(body.get(n), types.FunctionType)}
    if len(methods) != len(rewritten):
        return None
    qcls = type(f"Quant{cls.__name__}", (_LegacyQuantAttentionMixin, cls), {"_moq_original_cls": cls, **methods})
    _legacy_classes[cls] = qcls
    return qcls


def _is_supported_hf_model(model) -> bool:
    """Is `model` a transformers.PreTrainedModel?  Only a process that has ALREADY imported transformers can hold one, so
    the package is looked up in sys.modules and never imported from here: on a fresh box the cold import of transformers
    pages in for 17-30 s (measured inside quantize()'s convert stage: 30.1 s cold vs 1.2 s warm, profiles/
    r04_awq_unstaged.md) -- which a model made of plain nn.Linear modules paid for nothing."""
    import sys

    defining = sys.modules.get("transformers.modeling_utils")  # (the lazy top-level module would import it on access)
    base = getattr(defining, "PreTrainedModel", None) if defining is not None else None
    return base is not None and isinstance(model, base)


def register_hf_attentions_on_the_fly(model: nn.Module) -> int:
    """Convert every `*Attention` module of a Hugging Face model that calls the attention interface
    (huggingface.py:371-415).  Returns how many modules were converted."""
    if not _is_supported_hf_model(model):
        return 0
    n = 0
    legacy = []
    for _, m in list(model.named_modules()):
        cls = type(m)
        if is_quantized_attention(m) or not cls.__name__.endswith("Attention"):
            continue
        if _wraps_nested_attention(m):
            continue
        if not is_compatible_attention(cls):
            legacy.append(m)
            continue
        convert_attention(m)
        n += 1
    if n or not legacy:
        return n
    # no attention of this model goes through the interface (huggingface.py:452-470): rewrite the classes' own products
    failed = set()
    for m in legacy:
        qcls = legacy_quant_attention_class(type(m))
        if qcls is None:
            failed.add(type(m).__name__)
            continue
        m.__class__ = qcls
        m._setup_bmm_quantizers()
        n += 1
    if not n:
        import warnings

        warnings.warn(f"Could not create a quantized attention class for {sorted(failed)} from this model. To enable KV Cache "
                      "quantization, write a quantized attention class for it (the attention interface, or two matmul / bmm "
                      "calls, two `@` products or one scaled_dot_product_attention call in the class's own source)")
    return n
