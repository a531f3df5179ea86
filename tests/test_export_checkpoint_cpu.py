"""export_hf_checkpoint end to end on CPU (host-memory stand-in of the C-ABI): what lands next to the tensors --
hf_quant_config.json, config.json with its embedded quantization_config, the file list -- against what the reference's
export_hf_checkpoint wrote for the same twelve presets on the same tiny Llama (tests/golden/export_configs.npz)."""

import copy
import json
import os

import pytest
import torch

import _moa_import
import hostmem_backend
from conftest import from_bits

moa = _moa_import.load()
mq = moa.model_quant


@pytest.fixture
def hostmem(monkeypatch):
    return hostmem_backend.install(monkeypatch, moa)


def _mixed():
    cfg = copy.deepcopy(mq.FP8_DEFAULT_CFG)
    cfg["quant_cfg"]["*mlp*weight_quantizer"] = {"num_bits": 8, "axis": 0}
    cfg["quant_cfg"]["*mlp*input_quantizer"] = {"enable": False}
    return cfg


RUNS = {"int4_awq": (lambda: mq.INT4_AWQ_CFG, True), "fp8": (lambda: mq.FP8_DEFAULT_CFG, True),
        "fp8_kv": (lambda: mq.update_quant_cfg_with_kv_cache_quant(mq.FP8_DEFAULT_CFG, mq.FP8_KV_CFG["quant_cfg"]), True),
        "int8_sq": (lambda: mq.INT8_SMOOTHQUANT_CFG, True), "int8_wo": (lambda: mq.INT8_WEIGHT_ONLY_CFG, True),
        "w4a8_awq": (lambda: mq.W4A8_AWQ_BETA_CFG, True), "fp8_pc_pt": (lambda: mq.FP8_PER_CHANNEL_PER_TOKEN_CFG, True),
        "mxfp4": (lambda: mq.MXFP4_DEFAULT_CFG, False), "w4a8_mxfp4_fp8": (lambda: mq.W4A8_MXFP4_FP8_CFG, False),
        "mxfp4_mlp": (lambda: mq.MXFP4_MLP_WEIGHT_ONLY_CFG, False), "fp8_2d": (lambda: mq.FP8_2D_BLOCKWISE_WEIGHT_ONLY_CFG, False),
        "mixed_fp8_int8wo": (_mixed, True)}


@pytest.mark.parametrize("name", sorted(RUNS))
def test_export_hf_checkpoint_writes_what_the_reference_writes(golden, hostmem, tmp_path, name):
    from safetensors import safe_open
    from transformers import LlamaConfig, LlamaForCausalLM

    g = golden("export_configs")
    want = g.cases["runs"][name]
    cfg = LlamaConfig(architectures=["LlamaForCausalLM"], **g.cases["config"])
    model = LlamaForCausalLM(cfg).to(torch.bfloat16).eval()
    sd = {k[len("orig/"):]: from_bits(g.raw(k), torch.bfloat16) for k in g.z.files if k.startswith("orig/")}
    assert not model.load_state_dict(sd, strict=False).unexpected_keys
    batches = [torch.from_numpy(g.raw(f"tokens{i}")) for i in range(g.cases["n_batches"])]
    make, calib = RUNS[name]
    with torch.no_grad():
        moa.quantize(model, copy.deepcopy(make()), (lambda m: [m(b) for b in batches]) if calib else None)
        quant = moa.export.export_hf_checkpoint(model, export_dir=str(tmp_path))
    assert sorted(os.listdir(tmp_path)) == want["files"]
    assert quant["quantization"] == want["hf_quant_config"]["quantization"]
    on_disk = json.load(open(tmp_path / "hf_quant_config.json"))
    assert on_disk["quantization"] == want["hf_quant_config"]["quantization"]
    cj = json.load(open(tmp_path / "config.json"))
    assert sorted(cj) == want["config_keys"]
    got_qc, want_qc = dict(cj["quantization_config"]), dict(want["quantization_config"])
    got_qc.pop("producer"), want_qc.pop("producer")  # (the producer's name and version are each library's own)
    assert got_qc == want_qc
    # the tensor file opens and holds the packed linears under checkpoint names
    with safe_open(str(tmp_path / "model.safetensors"), "pt") as f:
        keys = set(f.keys())
    assert "model.layers.0.mlp.down_proj.weight" in keys and "lm_head.weight" in keys


def test_convert_hf_quant_config_format_groups_mixed_precision_layers():
    cfg = {"producer": {"name": "x", "version": "1"},
           "quantization": {"quant_algo": "MIXED_PRECISION", "kv_cache_quant_algo": "FP8",
                            "quantized_layers": {"a.q": {"quant_algo": "FP8"}, "a.k": {"quant_algo": "FP8"},
                                                 "a.up": {"quant_algo": "W4A16_AWQ", "group_size": 64, "has_zero_point": False,
                                                          "pre_quant_scale": True}}}}
    out = moa.export.convert_hf_quant_config_format(cfg)
    assert out["quant_method"] == "modelopt" and out["ignore"] == [] and out["quant_algo"] == "MIXED_PRECISION"
    assert out["kv_cache_scheme"] == {"dynamic": False, "num_bits": 8, "type": "float"}
    groups = out["config_groups"]
    assert groups["group_0"]["targets"] == ["a.k", "a.q"] and groups["group_0"]["weights"]["type"] == "float"
    assert groups["group_1"]["targets"] == ["a.up"] and groups["group_1"]["weights"] == {"dynamic": False, "num_bits": 4, "type": "int", "group_size": 64}
    assert "input_activations" not in groups["group_1"]
