"""world_size-2 gloo tests (CPU) of the multi-GPU path: bucketed MAX/SUM all-reduce of calibration state,
round-robin sharding of per-layer tensors.  The code under test is device-agnostic host logic; on the GPU
node the same calls run over RCCL (backend 'nccl')."""

import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import _moa_import


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, fn_name, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        moa = _moa_import.load()
        globals()[fn_name](rank, world, moa)
        ret[rank] = "ok"
    except Exception as e:  # noqa: BLE001
        ret[rank] = f"{type(e).__name__}: {e}"
    finally:
        dist.destroy_process_group()


def _spawn(fn_name, world=2):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), fn_name, ret), nprocs=world, join=True)
    assert dict(ret) == {r: "ok" for r in range(world)}, dict(ret)


class _FakeQuantizer:
    def __init__(self, amax):
        self._amax = amax


def _job_amax(rank, world, moa):
    qs = [_FakeQuantizer(torch.tensor([1.0 + rank, 5.0 - rank])), _FakeQuantizer(torch.tensor(3.0 * (rank + 1))),
          _FakeQuantizer(None), _FakeQuantizer(torch.tensor([[0.5], [float("nan") if rank == 1 else 2.0]]).bfloat16())]
    moa.distributed.sync_amax_bucketed(qs)
    assert torch.equal(qs[0]._amax, torch.tensor([2.0, 5.0]))
    assert qs[1]._amax.item() == 6.0
    assert qs[3]._amax.dtype == torch.bfloat16 and torch.isnan(qs[3]._amax).all()  # NaN flags the whole tensor
    # every rank ends with identical values (tests/unit/torch/quantization/test_dist.py:27-47 property)
    for q in qs:
        if q._amax is not None:
            ref = q._amax.float().clone()
            dist.all_reduce(ref, op=dist.ReduceOp.MAX)
            assert torch.equal(torch.nan_to_num(ref, nan=-1), torch.nan_to_num(q._amax.float(), nan=-1))


def _job_calibrators(rank, world, moa):
    from model_optimizer_amd.calib import HistogramCalibrator, MaxCalibrator

    m = MaxCalibrator(8, None, False)
    m._buf, m._shape, m._dtype = torch.tensor([1.0 + rank]), (), torch.float32
    h = HistogramCalibrator(8, None, False, num_bins=4)
    width = 0.5
    n = 4 + 2 * rank  # rank 1 has grown its histogram
    h._calib_hist = torch.arange(n, dtype=torch.int64) + rank
    h._num_bins = n
    h._calib_bin_edges = torch.arange(0, n + 1, dtype=torch.float32) * width
    h._grown_to = torch.tensor(3.0) if rank == 1 else None  # the abs-max rank 1 extended its range for
    moa.distributed.sync_calibrators_bucketed([m, h])
    assert m._buf.item() == 2.0
    want = torch.zeros(6, dtype=torch.int64)
    want[:4] += torch.arange(4)
    want += torch.arange(6) + 1
    assert torch.equal(h._calib_hist, want), h._calib_hist
    assert h._calib_bin_edges.numel() == 7 and h._calib_bin_edges[-1].item() == 3.0
    r = moa.distributed.agree_histogram_range(torch.tensor(1.0 + rank))
    assert r.item() == 1.0  # rank 0's first-batch range: the width a single rank starting with that batch would use


def _job_bucket_and_shard(rank, world, moa):
    a, b = torch.full((3,), float(rank + 1)), torch.full((2, 2), 10.0 * (rank + 1))
    moa.distributed.all_reduce_bucket([a, b], dist.ReduceOp.SUM, average=True)
    assert torch.equal(a, torch.full((3,), 1.5)) and torch.equal(b, torch.full((2, 2), 15.0))
    items = list(range(7))
    mine = moa.distributed.shard_list(items)
    assert mine == [i for i in items if i % world == rank]
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    assert sorted(sum(gathered, [])) == items  # a partition: nothing lost, nothing duplicated


def _job_missing_amax(rank, world, moa):
    """A quantizer that saw no data on one rank (routed expert without tokens from that rank's shard): the bucket
    membership is agreed first, the rank without the amax adopts the group's value; strict mode raises the
    reference's error on every rank instead of dead-locking."""
    def make():
        return [_FakeQuantizer(torch.tensor([1.0 + rank])),
                _FakeQuantizer(torch.tensor([[2.0], [7.0]]).bfloat16() if rank == 0 else None),   # missing on rank 1
                _FakeQuantizer(None),                                                               # missing everywhere
                _FakeQuantizer(torch.tensor(4.0) if rank == 1 else None)]                           # missing on rank 0
    qs = make()
    moa.distributed.sync_amax_bucketed(qs)
    assert qs[0]._amax.item() == 2.0
    assert qs[1]._amax.dtype == torch.bfloat16 and torch.equal(qs[1]._amax.float(), torch.tensor([[2.0], [7.0]]))
    assert qs[2]._amax is None
    assert qs[3]._amax.shape == () and qs[3]._amax.item() == 4.0
    with pytest.raises(RuntimeError, match="MoE calibration incomplete"):
        moa.distributed.sync_amax_bucketed(make(), on_missing="raise")
    # a real TensorQuantizer gets a registered buffer through its amax setter
    tq = moa.TensorQuantizer(moa.QuantizerAttributeConfig(num_bits=8, axis=None))
    if rank == 0:
        tq.amax = torch.tensor(3.0)
    moa.distributed.sync_amax_bucketed([tq])
    assert tq.amax.item() == 3.0 and "_amax" in dict(tq.named_buffers())


def _job_awq_scales(rank, world, moa):
    cin = [4, 3, 2, 5]
    act = [torch.full((4,), 1.0 + rank),                       # both ranks: average
           torch.full((3,), 5.0) if rank == 0 else None,       # rank 1 saw no tokens: rank 0's value
           None,                                               # nobody: disabled
           torch.tensor([1.0, float("nan") if rank == 1 else 2.0, 3.0, 4.0, 5.0])]   # NaN on one rank: disabled
    wsc = [torch.ones(c) for c in cin]
    synced, enabled = moa.distributed.sync_awq_act_scales(act, wsc, cin, torch.device("cpu"))
    assert enabled == [True, True, False, False]
    assert torch.equal(synced[0], torch.full((4,), 1.5)) and torch.equal(synced[1], torch.full((3,), 5.0))
    assert synced[2] is None
    # NaN in the weight scale votes too
    wsc[0][1] = float("nan") if rank == 0 else 1.0
    _, enabled = moa.distributed.sync_awq_act_scales(act, wsc, cin, torch.device("cpu"))
    assert enabled == [False, True, False, False]


def test_sync_amax_with_missing_members_gloo():
    _spawn("_job_missing_amax")


def test_sync_awq_act_scales_gloo():
    _spawn("_job_awq_scales")


def test_sync_amax_bucketed_gloo():
    _spawn("_job_amax")


def test_sync_calibrators_bucketed_gloo():
    _spawn("_job_calibrators")


def test_bucket_average_and_sharding_gloo():
    _spawn("_job_bucket_and_shard")


def test_single_process_is_a_noop():
    moa = _moa_import.load()
    q = _FakeQuantizer(torch.tensor([1.0]))
    moa.distributed.sync_amax_bucketed([q])
    assert q._amax.item() == 1.0
    assert moa.distributed.shard_list([1, 2, 3]) == [1, 2, 3]
