"""Shared by the CPU (host-memory stand-in) and GPU tiers: the end-to-end INT4-AWQ checkpoint of the tiny Llama with the
reference run's per-linear inputs REPLAYED (tests/golden/export_llama_replay.npz, gen_golden.gen_export_replay).

The model's own GEMMs (attention / MLP glue: library kernels whose summation order differs between machines) are taken
out of the comparison: the calibration loop hands every quantized linear exactly the tensor the reference's linear
received in that batch.  Everything the path computes from there -- activation / weight statistics, the alpha search,
the fold, the per-group amax, the resmooth + norm fusion + nibble packing of the export -- must then reproduce the
reference's model.safetensors byte for byte (export/unified_export_hf.py:1491, export/quant_utils.py:792-833)."""

import copy

import torch

from conftest import assert_bits_equal, from_bits


def replay_loop(replay, device):
    """forward_loop(model): feeds every quantized linear the reference's input of each batch.  Linears that read one
    tensor in the model (q / k / v, gate / up) get the SAME tensor object, as in a real forward."""
    rc = replay.cases
    cache = {}

    def x_of(name, b):
        owner = rc["alias"].get(name, name)
        if (owner, b) not in cache:
            shape = [int(v) for v in replay.raw(f"in_shape/{owner}/{b}")]
            cache[(owner, b)] = from_bits(replay.raw(f"in/{owner}/{b}"), torch.bfloat16).reshape(shape).to(device)
        return cache[(owner, b)]

    def loop(model):
        for b in range(rc["n_batches"]):
            for name in rc["linears"]:
                model.get_submodule(name)(x_of(name, b))

    return loop


def stage_report(q, g, replay):
    """Per linear: which stage first differs from the reference run (None = all equal).  Diagnostics for a failing
    byte comparison; the stages are in the order the flow computes them."""
    out = {}
    for name in replay.cases["linears"]:
        lin = q.get_submodule(name)
        h = lin.awq_lite
        stages = [
            ("act_scale", h.act_scale.float().cpu(), from_bits(replay.raw(f"ref/{name}.act_scale"), torch.float32)),
            ("weight_scale", h.weight_scale.float().cpu(), from_bits(replay.raw(f"ref/{name}.weight_scale"), torch.float32)),
            ("best_alpha", torch.tensor(float(h.best_alpha)), torch.tensor(float(g.raw(f"pre/{name}.best_alpha")))),
            ("best_scale", h.best_scale.float().cpu(), from_bits(replay.raw(f"ref/{name}.best_scale"), torch.float32)),
            ("pre_quant_scale", lin.input_quantizer._pre_quant_scale.cpu(),
             from_bits(g.raw(f"pre/{name}.pre_quant_scale"), torch.bfloat16)),
            ("weight", lin.weight.detach().cpu(), from_bits(g.raw(f"pre/{name}.weight"), torch.bfloat16)),
            ("amax", lin.weight_quantizer._amax.float().cpu().reshape(-1),
             from_bits(g.raw(f"pre/{name}.amax"), torch.float32).reshape(-1)),
        ]
        first = None
        for what, got, want in stages:
            got, want = got.reshape(-1), want.reshape(-1)
            if got.shape != want.shape or not torch.equal(got, want):
                n = int((got != want).sum()) if got.shape == want.shape else -1
                first = f"{what} ({n} of {want.numel()} values)"
                break
        out[name] = first
    return out


def awq_ragged_check(moa, golden, device):
    """tests/golden/awq_ragged.npz (gen_golden.gen_awq_ragged): INT4-AWQ of one bf16 linear with Cin = 192 (block 128)."""
    g = golden("awq_ragged")

    class One(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.fc = torch.nn.Linear(192, 96, bias=False)

        def forward(self, x):
            return self.fc(x)

    model = One().to(torch.bfloat16)
    with torch.no_grad():
        model.fc.weight.copy_(from_bits(g.raw("w"), torch.bfloat16).reshape(96, 192))
    model = model.to(device)
    batches = [from_bits(g.raw(f"x{i}"), torch.bfloat16).reshape(24, 192).to(device) for i in range(g.cases["n_batches"])]
    for search in ("auto", "gemm"):
        m = copy.deepcopy(model)
        cfg = copy.deepcopy(moa.model_quant.INT4_AWQ_CFG)
        cfg["algorithm"] = {"method": "awq_lite", "search": search}
        with torch.no_grad():
            q = moa.quantize(m, cfg, lambda mm: [mm(b) for b in batches])
            y0 = q(batches[0])
        h = q.fc.awq_lite
        assert not h.use_gram, "a ragged input width stays on the error-GEMM engine"
        assert float(h.best_alpha) == float(g.raw("best_alpha"))
        assert_bits_equal(h.weight_scale.cpu(), from_bits(g.raw("weight_scale"), torch.float32), "weight_scale")
        assert_bits_equal(h.act_scale.cpu(), from_bits(g.raw("act_scale"), torch.float32), "act_scale")
        ref_loss = torch.from_numpy(g.raw("loss")).float()
        assert ((h.loss_buf.cpu() - ref_loss).abs() <= 1e-2 * ref_loss).all(), "per-alpha losses"
        assert_bits_equal(q.fc.input_quantizer._pre_quant_scale.cpu().reshape(-1),
                          from_bits(g.raw("pre_quant_scale"), torch.bfloat16).reshape(-1), "pre_quant_scale")
        assert_bits_equal(q.fc.weight.detach().cpu().reshape(-1), from_bits(g.raw("folded"), torch.bfloat16).reshape(-1), "folded weight")
        assert list(q.fc.weight_quantizer._amax.shape) == g.cases["amax_shape"]
        assert_bits_equal(q.fc.weight_quantizer._amax.float().cpu().reshape(-1), from_bits(g.raw("amax"), torch.float32).reshape(-1), "block amax")
        assert_bits_equal(y0.cpu().reshape(-1), from_bits(g.raw("y0"), torch.bfloat16).reshape(-1), "fake-quantized output")


def check_w4a8_state(q, g):
    """Quantizer state of a W4A8-AWQ run right before export against gen_golden.gen_export_w4a8's record."""
    for name in g.cases["linears"]:
        lin = q.get_submodule(name)
        iq, stages = lin.input_quantizer, list(lin.weight_quantizer)
        assert iq.is_enabled and iq.axis is None, name
        for what, got, key in [("per-channel input amax", iq._amax_for_smoothing, "in_amax_channels"),
                               ("input amax", iq._amax, "in_amax"), ("FP8 stage amax", stages[1]._amax, "amax2")]:
            want = from_bits(g.raw(f"pre/{name}.{key}"), torch.float32).reshape(-1)
            assert torch.equal(got.detach().float().cpu().reshape(-1), want), f"{name}: {what} differs"


def pin_host_pow(monkeypatch, replay):
    """get_scale's pow / divide run in torch's HOST math library (model_calib.get_scale), whose last bit depends on the
    CPU's vector ISA (Sleef AVX2 vs AVX-512 vs the scalar tail): third-party arithmetic outside the path, like the
    model's GEMMs.  The fixture was generated on the build container's CPU; on another host a scale vector may come out
    one fp32 ulp away.  For the byte comparison the replay pins it: a freshly computed scale vector that agrees with one
    of the reference run's best_scale vectors to within FOUR ulp everywhere (pow of the entry, pow of the max and of the
    min that normalise it, the square root) is replaced by that vector; anything further off is left alone and fails
    the stage report.  Returns the list of (linear, differing entries) it pinned; `check_pins` bounds it."""
    from model_optimizer_amd import model_calib

    refs = {name: from_bits(replay.raw(f"ref/{name}.best_scale"), torch.float32).reshape(-1)
            for name in replay.cases["linears"]}
    pinned = []
    orig = model_calib.get_scale

    def get_scale(x_max, w_max, alpha):
        out = orig(x_max, w_max, alpha)
        o = out.detach().float().cpu()
        for name, ref in refs.items():
            if ref.numel() != o.numel():
                continue
            ulp = torch.maximum(ref.abs(), o.abs()) * 2.0 ** -23
            if bool(((ref - o).abs() <= MAX_PIN_ULP * ulp).all()):
                n = int((ref != o).sum())
                if n:
                    worst = float(((ref - o).abs() / ulp).max())
                    pinned.append((name, n, int(ref.numel()), round(worst, 2)))
                    return ref.to(out.device)
                break
        return out

    monkeypatch.setattr(model_calib, "get_scale", get_scale)
    pinned_total = sum(int(r.numel()) for r in refs.values())
    pinned.append(("__total__", 0, pinned_total, 0.0))
    return pinned


# What a pin may be: a last-bits difference of torch's HOST pow / sqrt (a third-party library whose result depends on the
# CPU's vector ISA).  The bound is on the SIZE of the difference and on HOW MUCH of the model it touches -- not on which or
# how many vectors happen to differ on the box at hand (round 3 bounded the vector count by what had been observed: 2):
MAX_PIN_ULP = 4             # per entry, in fp32 ulps (pow of the entry, of the normalising max and min, the square root)
MAX_PINNED_FRACTION = 0.25  # of all scale-vector entries of the fixture model (observed on the GPU boxes: 309 of 4096 = 7.5 %, build container: 0)


def check_pins(pinned, what: str):
    """"Byte-identical" must not quietly rest on pins: every pinned entry lies within MAX_PIN_ULP ulps of the reference's
    (by construction of the pin), and the pinned entries are at most MAX_PINNED_FRACTION of the model's scale entries;
    entries / total, vectors and the largest difference go to the suite's tail (conftest.note)."""
    import conftest

    total = next((t for n, _, t, _ in pinned if n == "__total__"), 0)
    real = sorted({p for p in pinned if p[0] != "__total__"})
    entries = sum(n for _, n, _, _ in real)
    worst = max((u for _, _, _, u in real), default=0.0)
    conftest.note(f"{what}: host-pow pins = {entries} of {total} scale entries in {len({n for n, _, _, _ in real})} vector(s), "
                  f"largest difference {worst} ulp {[(n, k) for n, k, _, _ in real]}")
    assert worst <= MAX_PIN_ULP, f"{what}: a pinned entry is {worst} ulp from the reference's"
    assert total and entries <= MAX_PINNED_FRACTION * total, \
        f"{what}: {entries} of {total} scale entries needed pinning to the reference's host pow: {real}"
