import os, sys, torch
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests'))
import _moa_import
moa=_moa_import.load(); ops=moa.ops
from conftest import Golden, from_bits
import test_gpu_export as T
g=Golden('export_llama'); cases=g.cases
model=T._build_llama(g,cases)
mq=moa.model_quant
moa.nn.replace_quant_module(model); mq.set_quantizer_by_cfg(model, mq.INT4_AWQ_CFG["quant_cfg"])
for name in cases["linears"]:
    lin=model.get_submodule(name)
    with torch.no_grad(): lin.weight.copy_(from_bits(g.raw(f"pre/{name}.weight"), torch.bfloat16).cuda())
    wq=lin.weight_quantizer; wq(lin.weight)
    wq.amax=from_bits(g.raw(f"pre/{name}.amax"), torch.float32).cuda(); wq.promote_static_block()
    lin.input_quantizer._enable_pre_quant_scale=True
    lin.input_quantizer.pre_quant_scale=from_bits(g.raw(f"pre/{name}.pre_quant_scale"), torch.bfloat16).cuda()
for key in g.z.files:
    if key.startswith("pre/") and key.endswith("norm.weight"):
        mod=model.get_submodule(key[len("pre/"):-len(".weight")])
        with torch.no_grad(): mod.weight.copy_(from_bits(g.raw(key), torch.bfloat16).cuda())
fused=moa.export.requantize_resmooth_fused_llm_layers(model, T._dummy_forward(model))
print('fused groups', fused)
state=moa.export.export_state_dict(model, torch.bfloat16, None)
for key,dts in cases["dtypes"].items():
    want=T._tensor(g, f"exp/{key}", T._TD[dts]); got=state[key].detach().cpu()
    if got.shape!=want.shape or got.dtype!=want.dtype: print(key,'SHAPE/DTYPE',got.dtype,got.shape,want.dtype,want.shape); continue
    f=(got.contiguous().view(torch.uint8).reshape(-1)==want.contiguous().view(torch.uint8).reshape(-1)).float().mean().item()
    if f<1: print(key, f)
