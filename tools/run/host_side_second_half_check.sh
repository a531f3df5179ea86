#!/bin/bash
# (gpurun call for the START of the next round) the host-side work of round 4's second half was written after the round's
# GPU minutes were spent: its GPU tests ran only in the CPU tier (host-memory stand-in).  This runs them on the device and
# times the two flows whose default changed: layer-by-layer calibration through the parent's forward (layerwise.DecoderWalk)
# against the hand-over mode on a Hugging Face Llama-3-8B-shaped stack of 4 layers, and quantize()'s `validate` stage.
set -u
mkdir -p gpurun_out
python3 -m pytest tests/test_gpu_layerwise.py tests/test_gpu_host.py tests/test_gpu_kv_cache.py tests/test_gpu_moe.py -q -m gpu \
  -k "layerwise or affine or last_two_axes or 2d_blocks or kv_cache or mixtral" 2>&1 | tail -5 | tee gpurun_out/host_side_second_half_tests.txt
python3 - <<'P' 2>&1 | tee gpurun_out/host_side_second_half_timing.txt
import copy, os, sys, time, torch
sys.path.insert(0, os.getcwd())
import _moa_import
moa = _moa_import.load()
import transformers as tf
dev = torch.device("cuda", 0)
cfg = tf.LlamaConfig(hidden_size=4096, intermediate_size=14336, num_hidden_layers=4, num_attention_heads=32, num_key_value_heads=8,
                     vocab_size=32000, max_position_embeddings=4096, architectures=["LlamaForCausalLM"])
torch.manual_seed(0)
model = tf.LlamaForCausalLM(cfg).to(torch.bfloat16).to(dev).eval()
batches = moa.forward_loop.synthetic_token_batches(32000, num_samples=64, max_sample_length=512, batch_size=8, device=dev)
loop = moa.forward_loop.create_forward_loop(dataloader=batches)
def sync():
    torch.cuda.synchronize(); return time.perf_counter()
for label, algo in (("whole model", "max"), ("layerwise, parent walk", {"method": "max", "layerwise": {"enable": True}}),
                    ("layerwise, hand-over", {"method": "max", "layerwise": {"enable": True, "capture": "handover"}})):
    m = copy.deepcopy(model)
    for rep in range(2):
        mm = copy.deepcopy(m)
        t0 = sync()
        moa.quantize(mm, {**moa.model_quant.FP8_DEFAULT_CFG, "algorithm": algo}, loop)
        t1 = sync()
    print(f"FP8 max calibration, 4 Llama-3-8B layers, 64 x 512 tokens, {label}: {t1 - t0:.3f} s; stages {moa.model_quant.QUANTIZE_STATS.get('stages_s')}", flush=True)
    del mm, m
P
