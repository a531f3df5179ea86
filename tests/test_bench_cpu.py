"""bench.py's host logic without a GPU: how the pool of per-layer weight tensors is partitioned over the ranks in the
two scaling modes, the model shape tables behind the metric's byte counts, and the committed-PMC lookup."""

import importlib.util
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("moq_bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def test_model_tables_give_the_survey_sizes():
    # Llama-3-8B 6.98 G linear parameters in 224 tensors, Llama-3-70B 68.45 G in 560
    for model, n_tensors, elems in [("llama3-8b", 224, 6_979_321_856), ("llama3-70b", 560, 68_451_041_280)]:
        shapes = [s for _ in range(bench.MODELS[model][2]) for s in bench.layer_shapes(model)]
        assert len(shapes) == n_tensors
        assert sum(r * c for r, c in shapes) == elems
    assert len(bench.layer_shapes("mixtral-8x7b")) == 4 + 3 * 8


@pytest.mark.parametrize("world", [1, 2, 4, 8])
@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_pool_partition(world, scaling, monkeypatch):
    """Every pool tensor lives on exactly one rank; weak: every rank holds one model's worth (same shapes in the same
    order, different values), strong: one model's list dealt largest-first round-robin; a tensor's values depend on its
    pool index only."""
    monkeypatch.setitem(bench.MODELS, "tiny", (16, 24, 3, 8))
    per_rank = [bench.make_weights("tiny", 3, "cpu", rank=r, world=world, scaling=scaling) for r in range(world)]
    n_model = 3 * 7
    pool = n_model * (world if scaling == "weak" else 1)
    assert all(n == pool for _, _, n in per_rank)
    owned = sorted(i for _, idx, _ in per_rank for i in idx)
    assert owned == list(range(pool))
    shapes = [s for _ in range(3) for s in bench.layer_shapes("tiny")]
    for r, (ws, idx, _) in enumerate(per_rank):
        assert [tuple(w.shape) for w in ws] == [shapes[i % n_model] for i in idx]
        assert all(w.dtype == torch.bfloat16 for w in ws)
        if scaling == "weak":
            assert idx == list(range(r * n_model, (r + 1) * n_model))
        else:
            assert idx == bench.deal(shapes, r, world) == sorted(idx)
    if world > 1:
        # the same pool index gives the same tensor whoever generates it; different indices differ
        again, idx, _ = bench.make_weights("tiny", 3, "cpu", rank=1, world=world, scaling=scaling)
        assert all(torch.equal(a, b) for a, b in zip(again, per_rank[1][0]))
        if scaling == "weak":
            assert not torch.equal(per_rank[0][0][0], per_rank[1][0][0])
    if scaling == "strong" or world == 1:
        single, _, _ = bench.make_weights("tiny", 3, "cpu")
        for ws, idx, _ in per_rank:
            assert all(torch.equal(w, single[i]) for w, i in zip(ws, idx))


@pytest.mark.parametrize("model", ["llama3-8b", "llama3-70b", "mixtral-8x7b"])
@pytest.mark.parametrize("world", [2, 4, 8])
def test_strong_deal_is_balanced(model, world):
    """The strong-scaling deal gives every rank the same number of tensors of every shape -- the ranks' bytes are equal,
    so max-over-ranks time is not an imbalance artefact."""
    shapes = [s for _ in range(bench.MODELS[model][2]) for s in bench.layer_shapes(model)]
    loads = [sum(shapes[i][0] * shapes[i][1] for i in bench.deal(shapes, r, world)) for r in range(world)]
    assert len(set(loads)) == 1
    assert sorted(i for r in range(world) for i in bench.deal(shapes, r, world)) == list(range(len(shapes)))


def test_default_model_is_the_baseline_config_of_the_format():
    """N = 1: configs[1]'s Llama-3-8B; N > 1: the multi-GPU configuration BASELINE.json names for the format."""
    for wl in bench.SCALE_MODEL:
        assert bench.default_model(wl, 1) == "llama3-8b"
    for n in (2, 4, 8):
        assert bench.default_model("fp8", n) == bench.default_model("mask24", n) == "mixtral-8x7b"
        assert bench.default_model("int4g128", n) == bench.default_model("mxfp4-sq", n) == "llama3-70b"
    args = bench.build_parser().parse_args([])
    assert (args.gpus, args.scaling, args.model, args.workload) == (1, "strong", None, "fp8")


def test_bare_contract_command_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher in the environment re-executes itself under torch.distributed.run:
    here (no GPU) both ranks then stop with the no-GPU message -- which only ranks started by the launcher print twice."""
    import subprocess
    import sys

    cmd = bench.launch_command(2, ["--gpus", "2"], port=29999)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=2" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-3:] == [os.path.join(ROOT, "bench.py"), "--gpus", "2"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HIP_VISIBLE_DEVICES"] = ""  # (a GPU box: the ranks must stop at the same place)
    env["CUDA_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-extra", "--no-cpu-baseline"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0
    # both ranks stop with the no-GPU message; on a loaded host the launcher may SIGTERM the slower rank (once the first one has
    # failed) before it gets to print its own -- then the launcher's failure report names the second rank instead
    said = r.stderr.count("bench.py needs a GPU")
    assert said == 2 or (said == 1 and "local_rank: 1" in r.stderr and "local_rank: 0" in r.stderr), r.stderr[-2000:]


def test_cpu_baseline_only_mode_prints_one_object(monkeypatch):
    """`--cpu-baseline-only` (the process the main run starts LAST for the CPU baseline) touches no GPU and prints the
    cpu_baseline object: kind "reference" (the reference's own eager path timed on this host, the C port beside it) wherever
    the reference is present -- /root/reference here, the staged archive on the GPU box -- kind "port" otherwise."""
    import json
    import subprocess
    import sys

    code = ("import importlib.util, sys, json; spec = importlib.util.spec_from_file_location('b', sys.argv[1]); "
            "b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b); "
            "b.cpu_baseline.__defaults__ = (0.2,); sys.argv = ['bench.py', '--cpu-baseline-only', '--workload', 'int4g128']; "
            "b.main()")
    r = subprocess.run([sys.executable, "-c", code, os.path.join(ROOT, "bench.py")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    obj = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert obj["unit"] == "GB/s" and obj["value"] > 0 and obj["cores"] >= 1
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import ref_shim

    if ref_shim.reference_available():
        assert obj["kind"] == "reference" and obj["reference_source"] in ("checkout", "staged")
        assert "max_calibrate(TensorQuantizer)" in obj["sample"] and obj["port"]["value"] > 0
    else:
        assert obj["kind"] == "port"


def test_committed_pmc_traffic_matches_the_algorithmic_bytes():
    """roofline.traffic of the default line comes from the committed PMC profile of the same launch: present, and within
    0.1 % of the 4 B/element the FP8 QDQ launch must move (anything above would be wasted re-reads)."""
    traffic, src = bench.pmc_traffic("fp8", "llama3-8b", 32)
    assert traffic is not None and os.path.exists(os.path.join(ROOT, src.split(",")[0].split(" ")[0]))
    alg = 6_979_321_856 * 4
    assert abs(traffic - alg) / alg < 1e-3
    assert bench.pmc_traffic("fp8", "llama3-8b", 4)[0] is None  # another workload size: no committed counter run


def test_scale_n_builds_the_drivers_command_lines_and_refuses_lines_that_are_not_multi_gpu_measurements(capsys):
    """tools/scale_n.py: N = 1 is the bare contract command, N > 1 the driver's torch.distributed.run form; a line counts for a
    SCALE record only with every contract field and, at N > 1, a `collective` object that says RCCL joined N ranks on N
    distinct devices (the one-GPU gloo debug mode -- the only way N > 1 has ever run here -- must NOT pass)."""
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import scale_n

    assert scale_n.main(["--dry-run", "--gpus", "1,2,8", "--steps", "7", "--warmup", "3"]) == 0
    lines = capsys.readouterr().out.strip().splitlines()
    assert lines[0].split()[1:] == [os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "7", "--warmup", "3"]
    two = lines[1].split()
    assert two[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=2" in two
    assert two[two.index("--master-addr") + 1] == "127.0.0.1" and two[-6:] == ["--gpus", "2", "--steps", "7", "--warmup", "3"]
    assert "--nproc-per-node=8" in lines[2]
    line = {k: 0 for k in scale_n.REQUIRED}
    line.update(n_gpus=2, collective={"backend": "nccl", "rccl_ranks_seen": 2, "distinct_devices": 2, "multi_gpu_valid": True})
    assert scale_n.check_line(line, 2) == []
    debug = dict(line, collective={"backend": "gloo", "rccl_ranks_seen": 2, "distinct_devices": 1, "multi_gpu_valid": False})
    assert any("not a multi-GPU measurement" in p for p in scale_n.check_line(debug, 2))
    assert any("no `collective`" in p for p in scale_n.check_line(dict(line, collective=None), 2))
    assert any("n_gpus" in p for p in scale_n.check_line(dict(line, n_gpus=1), 2))
    assert any("missing field roofline" in p for p in scale_n.check_line({k: v for k, v in line.items() if k != "roofline"}, 2))
