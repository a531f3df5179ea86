import os, sys, json, torch, numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests'))
import _moa_import
moa=_moa_import.load(); ops=moa.ops
from conftest import Golden, DT
g=Golden('mse'); name='int8_chan_bf16'; c=g.cases[name]
w=g.t(f"{name}_w", DT[c['dtype']]).cuda()
tq=moa.tensor_quantizer
q=tq.TensorQuantizer(tq.QuantizerAttributeConfig(narrow_range=False, num_bits=8, axis=0))
lin=torch.nn.Linear(w.shape[1], w.shape[0], bias=False)
moa.nn.QuantLinear.convert(lin); lin.weight_quantizer=q; lin.input_quantizer.disable()
lin=lin.cuda().to(w.dtype)
with torch.no_grad(): lin.weight.copy_(w)
orig=ops.mse_sweep
def spy(x, cand, reduce_axis, nb, uns, narrow, loss=None):
    print('mse_sweep args: x', x.dtype, tuple(x.shape), 'cand', cand.dtype, tuple(cand.shape), 'reduce', reduce_axis, nb, uns, narrow)
    init=g.t(f"{name}_init_amax").to(torch.bfloat16).reshape(c['init_shape'])
    mult=torch.linspace(0.25,4.0,steps=39)
    cref=torch.stack([(init*m).float().reshape(-1) for m in mult])
    d=(cand.cpu()-cref).abs(); print('cand max diff', d.max().item(), 'at', divmod(d.argmax().item(),40))
    out=orig(x,cand,reduce_axis,nb,uns,narrow,loss)
    want=g.t(f"{name}_losses"); r=((out.cpu()-want).abs()/want.abs().clamp_min(1e-20)); i=r.argmax()
    print('rel', r.max().item(), 'at', divmod(i.item(),40), out.cpu().reshape(-1)[i].item(), want.reshape(-1)[i].item())
    return out
ops.mse_sweep=spy
print('weight eq', torch.equal(lin.weight.data, w), lin.weight.dtype)
moa.model_calib.mse_calibrate(lin, None, distributed_sync=False)
print('amax', q._amax.dtype, q._amax.shape)
