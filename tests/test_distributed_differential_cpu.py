"""world_size-2 gloo DIFFERENTIAL test against the reference (build container only): every rank holds a replica of the same
tiny Hugging Face model and calibrates on ITS share of the batches, once under the reference's `mtq.quantize` (whose
`max_calibrate(distributed_sync=True)` all-reduces every quantizer's amax with MAX over the data-parallel group -- the
world when nothing else is declared, utils/distributed.py:441-458, model_calib.py:378-407) and once under this package's
`quantize` (one bucketed MAX all-reduce, distributed.sync_amax_bucketed).  Per rank: every amax, the fake-quantized logits
and (on rank 0) every byte of the exported checkpoint must be the reference's.

What it cannot cover: the AWQ flows -- the reference averages the ranks' activation scales with ReduceOp.AVG
(model_calib.py:1588-1593), which gloo does not implement, so the reference itself cannot run them on this tier (RCCL does;
the property tests of tests/test_distributed_flows_cpu.py cover this package's side) -- and histogram calibrators, which the
reference leaves unsynchronised (calib/histogram.py:158-163; DESIGN.md section 8 states the difference)."""

import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import ref_shim  # noqa: E402

pytestmark = pytest.mark.skipif(not ref_shim.reference_available(), reason="reference checkout not present (GPU box)")

CASES = [("FP8_DEFAULT_CFG", "bfloat16", True, "llama"), ("INT8_SMOOTHQUANT_CFG", "float32", False, "llama"),
         ("INT8_DEFAULT_CFG", "float32", False, "mixtral"), ("FP8_DEFAULT_CFG", "float16", "affine", "qwen2"),
         ("W4A8_MXFP4_FP8_CFG", "bfloat16", "cast", "llama")]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ret):
    try:
        sys.path.insert(0, HERE)
        torch.set_num_threads(max(1, (os.cpu_count() or 2) // world))
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        import hostmem_backend
        import test_differential_cpu as diff

        hostmem_backend.install(pytest.MonkeyPatch(), diff.moa)
        every = [torch.randint(0, diff.CFG["vocab_size"], (3, 24), generator=torch.Generator().manual_seed(70 + i)) for i in range(4)]
        every[3][0, :] = 5  # rank 1's share holds a repeated token: its activations differ in range from rank 0's
        single = {}
        for case in CASES[:2]:  # before the group exists: what ONE process over all batches gives
            diff._batches = lambda: every
            single[case] = diff._our_run(case[0], getattr(torch, case[1]), case[2], case[3], None)[0]
        dist.init_process_group("gloo", rank=rank, world_size=world)
        diff._batches = lambda: every[rank::world]
        notes = []
        for preset, dtype, with_kv, arch in CASES:
            dt = getattr(torch, dtype)
            # the checkpoint on rank 0 only: under an initialised process group transformers' save_pretrained leaves config.json
            # to the main process, and the reference's exporter then fails on the others (unified_export_hf.py:1477-1479)
            ref_amax, ref_state = diff._reference_run(preset, dt, with_kv, arch, None, export=rank == 0)
            our_amax, our_state = diff._our_run(preset, dt, with_kv, arch, None, export=rank == 0)
            what = f"rank {rank} {preset} {arch}"
            assert sorted(ref_amax) == sorted(our_amax), f"{what}: calibrated quantizers {set(ref_amax) ^ set(our_amax)}"
            for n, a in ref_amax.items():
                assert torch.equal(our_amax[n].reshape(-1), a.reshape(-1)), f"{what}: amax of {n} differs"
            # the synchronisation did something: the input amax equals the single-process run over all batches, and every
            # rank holds the same
            if (preset, dtype, with_kv, arch) in single:
                for n, a in single[(preset, dtype, with_kv, arch)].items():
                    assert torch.equal(our_amax[n].reshape(-1), a.reshape(-1)), f"{what}: amax of {n} is not the all-batches amax"
            for n, a in our_amax.items():
                other = a.clone()
                dist.all_reduce(other, op=dist.ReduceOp.MAX)
                assert torch.equal(other, a), f"{what}: amax of {n} differs between the ranks"
            ref_json, our_json = ref_state.pop("__quant_json__", None), our_state.pop("__quant_json__", None)
            ref_logits, our_logits = ref_state.pop("__logits__"), our_state.pop("__logits__")
            if ref_logits is not None:
                assert torch.equal(our_logits, ref_logits), f"{what}: logits differ"
            assert sorted(our_state) == sorted(ref_state), f"{what}: checkpoint keys {set(our_state) ^ set(ref_state)}"
            if ref_json is not None and ref_json[0] is not None:
                diff._assert_same_quant_json(our_json, ref_json, what)
            for k, want in ref_state.items():
                got = our_state[k].detach().cpu()
                assert got.dtype == want.dtype and tuple(got.shape) == tuple(want.shape), f"{what}: {k} {got.dtype} {tuple(got.shape)}"
                assert torch.equal(got.contiguous().reshape(-1).view(torch.uint8), want.contiguous().reshape(-1).view(torch.uint8)), f"{what}: {k} differs"
            notes.append(f"{preset}/{arch}: {len(ref_amax)} amax, {len(ref_state)} tensors")
        ret[rank] = "ok " + "; ".join(notes)
    except Exception as e:  # noqa: BLE001
        import traceback

        ret[rank] = f"{type(e).__name__}: {e}\n{traceback.format_exc()}"
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_data_parallel_max_calibration_equals_the_references_data_parallel_run_gloo(world):
    """world 3: the four batches split 2 / 1 / 1 -- ranks with different amounts of data still end on the all-batches amax."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert all(str(ret.get(r, "")).startswith("ok") for r in range(world)), "\n".join(f"rank {r}: {v}" for r, v in dict(ret).items())
