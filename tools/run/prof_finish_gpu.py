import sys, cProfile, pstats, copy, torch, pytest
import os; sys.path.insert(0, os.getcwd())
import _moa_import
moa = _moa_import.load()


from transformers import LlamaConfig, LlamaForCausalLM
cfg = LlamaConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=32, num_attention_heads=4, num_key_value_heads=2, vocab_size=100)
m = LlamaForCausalLM(cfg).to(torch.bfloat16).eval().cuda()
mq = moa.model_quant
qcfg = mq.update_quant_cfg_with_kv_cache_quant(mq.FP8_DEFAULT_CFG, mq.FP8_KV_CFG["quant_cfg"])
toks = [torch.randint(0, 100, (2, 16), device='cuda') for _ in range(2)]
from model_optimizer_amd import model_calib
moa.nn.replace_quant_module(m); mq.set_quantizer_by_cfg(m, qcfg["quant_cfg"])
model_calib.enable_stats_collection(m)
model_calib.weight_only_quantize(m)
with torch.no_grad():
    for t in toks: m(t)
torch.cuda.synchronize(); pr = cProfile.Profile(); pr.enable()
model_calib.finish_stats_collection(m)
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(25)
