"""Stream contract of the boundary ("kernels enqueue on the current torch stream ... no host sync inside").

Every entry is launched on a NON-default stream behind a long spin kernel and a late copy that produces its input:
  * ordering   -- the result equals the result on the final input (a launch on any other stream would read the
                  zero-filled buffer);
  * no sync    -- when the Python call returns, the stream still has the spin kernel pending;
  * capture    -- the same calls record into a HIP graph and replay on new data (no allocation / sync / stream
                  switch inside the C-ABI).
"""

import pytest
import torch

import _moa_import
from conftest import assert_bits_equal

pytestmark = pytest.mark.gpu
moa = _moa_import.load()
from model_optimizer_amd import ops  # noqa: E402

DEV = "cuda:0"
SPIN = int(4e8)  # ~0.2 s of device time at 2 GHz


def _weight(seed=0, shape=(1024, 4096), dtype=torch.bfloat16):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * 0.02).to(dtype).to(DEV)


def _table(ws, outs):
    return moa.multi_tensor.SegmentTable(ws, outs)


# name -> (aux tensors built BEFORE the stream section: a host->device copy of a Python scalar is torch's own sync,
#          entry under test)
CASES = {
    "reduce_amax": (lambda x: None, lambda x, a: ops.reduce_amax(x)),
    "amax_axis": (lambda x: None, lambda x, a: ops.reduce_amax(x, axis=[1])),
    "fp8_qdq": (lambda x: torch.tensor([0.09], device=DEV), lambda x, a: ops.scaled_e4m3(x, a)),
    "int8_qdq": (lambda x: torch.tensor([0.09], device=DEV), lambda x, a: ops.fake_tensor_quant(x, a, 8, False, True)),
    "int4_group": (lambda x: None, lambda x, a: ops.amax_qdq_int_group(x, 128, num_bits=4, narrow_range=False)[0]),
    "mxfp4": (lambda x: None, lambda x, a: ops.fused_amax_convert(x, 32, "E2M1")),
    "mask24": (lambda x: None, lambda x, a: ops.mask_2to4(x)),
    "hist": (lambda x: None, lambda x, a: ops.hist_abs(x, 2048, 0.2)),
    "col_stats": (lambda x: None, lambda x, a: torch.cat(ops.col_abs_stats(x))),
    "int4_pack": (lambda x: torch.full((x.numel() // 128, 1), 0.01, device=DEV, dtype=x.dtype),
                  lambda x, a: ops.int4_quantize(x.view(-1), a, 128)),
    "fp8_pack": (lambda x: torch.tensor([2e-4], device=DEV, dtype=x.dtype),
                 lambda x, a: ops.fp8_quantize(x, a).view(torch.uint8)),
    "mxfp4_pack": (lambda x: None, lambda x, a: torch.cat([t.reshape(-1) for t in ops.mxfp4_quantize(x, 32)])),
    "scale_cols": (lambda x: torch.linspace(0.5, 2.0, x.shape[1], device=DEV), lambda x, a: ops.scale_cols(x, a)),
    "gemm_nt": (lambda x: None, lambda x, a: ops.gemm_nt(x, x[:256])),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_entry_runs_on_the_current_stream_without_host_sync(name):
    make_aux, fn = CASES[name]
    src = _weight(3)
    aux = make_aux(src)
    want = fn(src, aux)
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=DEV)
    x = torch.zeros_like(src)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        torch.cuda._sleep(SPIN)
        x.copy_(src, non_blocking=True)   # the input only becomes valid on `side`, after the spin
        got = fn(x, aux)
        pending = not side.query()         # the call returned while the stream was still busy
    side.synchronize()
    assert pending, f"{name}: the call blocked the host until the stream drained"
    if got.dtype.is_floating_point:
        assert_bits_equal(got, want, name)
    else:
        assert torch.equal(got, want), name


def test_multi_tensor_table_on_side_stream():
    srcs = [_weight(i, s) for i, s in enumerate([(256, 1024), (3, 8192), (1024, 1024)])]
    ref = _table(srcs, None)
    ref.calibrate_amax()
    want = [t.clone() for t in ref.fake_quant_e4m3()]
    want_amax = ref.amax_flat.clone()
    torch.cuda.synchronize()
    xs = [torch.zeros_like(s) for s in srcs]
    tab = _table(xs, None)
    tab.calibrate_amax()  # allocates the scratch outside the timed section
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=DEV)
    with torch.cuda.stream(side):
        torch.cuda._sleep(SPIN)
        for x, s in zip(xs, srcs):
            x.copy_(s, non_blocking=True)
        tab.calibrate_amax()
        outs = tab.fake_quant_e4m3()
        pending = not side.query()
    side.synchronize()
    assert pending
    assert torch.equal(tab.amax_flat, want_amax)
    for o, w in zip(outs, want):
        assert_bits_equal(o, w, "mt fp8 on a side stream")


def test_calibrate_and_qdq_replay_from_a_hip_graph():
    """The multi-tensor calibrate + QDQ and the fused group kernel are graph-capturable; a replay on new weights
    gives the new weights' result."""
    shapes = [(512, 1024), (64, 8192), (1024, 1024)]
    xs = [torch.zeros(s, dtype=torch.bfloat16, device=DEV) for s in shapes]
    tab = _table(xs, None)
    tabg = moa.multi_tensor.SegmentTable(xs, None, group_size=128)
    tab.calibrate_amax(); tab.fake_quant_e4m3(); tabg.amax_qdq_int_group(4, False, False)
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=DEV)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            tab.calibrate_amax()
            tab.fake_quant_e4m3()
            tabg.amax_qdq_int_group(4, False, False)
    for seed in (11, 12):
        srcs = [_weight(seed + i, s) for i, s in enumerate(shapes)]
        for x, s in zip(xs, srcs):
            x.copy_(s)
        graph.replay()
        torch.cuda.synchronize()
        for i, s in enumerate(srcs):
            a = ops.reduce_amax(s).float().reshape(1)
            assert torch.equal(tab.amax[i], a)
            assert_bits_equal(tab.outputs[i], ops.scaled_e4m3(s, a), f"graph fp8 {i}")
            yg, ag = ops.amax_qdq_int_group(s, 128, num_bits=4, narrow_range=False)
            assert_bits_equal(tabg.outputs[i], yg, f"graph int4 {i}")
            assert torch.equal(tabg.amax[i].reshape(-1), ag.reshape(-1))
