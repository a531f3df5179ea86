import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import _moa_import
moa = _moa_import.load()
from model_optimizer_amd import ops
from oracle import oracle
DEV="cuda:0"
SHRINKS = [round(float(k), 2) for k in torch.arange(0.5, 1.0, 0.05)] + [1.0]
def block_amax(w,g):
    cout,cin=w.shape; pad=(-cin)%g
    return torch.nn.functional.pad(w.float(),(0,pad)).view(cout,-1,g).abs().amax(-1)
for dn,dt in [("f16",torch.float16),("bf16",torch.bfloat16)]:
  for cout,cin,g,ntok,step in [(128, 512, 128, 64, 2),(96, 200, 64, 100, 1),(64,256,128,64,1)]:
    gen = torch.Generator().manual_seed(cout * 7 + cin + ntok)
    w = (torch.randn(cout, cin, generator=gen) * 0.02).to(dt)
    x = (torch.randn(ntok, cin, generator=gen) * torch.exp(torch.randn(cin, generator=gen) * 0.5)).to(dt)
    amax=block_amax(w,g).to(dt); xs=x[0::step].contiguous()
    want=oracle.awq_clip_loss(xs,w,amax,SHRINKS,g,4)
    nblk=-(-cin//g)
    loss=torch.zeros(len(SHRINKS),nblk,cout,device=DEV)
    ops.awq_clip_loss(x.to(DEV),w.to(DEV),amax.to(DEV),torch.tensor(SHRINKS,device=DEV),g,4,loss,token_step=step)
    got=loss.transpose(1,2).cpu()
    rel=((got-want).abs()/want.abs().clamp_min(1e-30))
    i=rel.argmax().item(); k=i//(cout*nblk); r=(i//nblk)%cout; b=i%nblk
    print(dn,(cout,cin,g,ntok,step),'max rel',rel.max().item(),'at k,r,b',k,r,b,'median',rel.median().item())
    print('   got ',[f"{v:.5g}" for v in got[:,r,b].tolist()])
    print('   want',[f"{v:.5g}" for v in want[:,r,b].tolist()])
    print('   rel>0.05 count per shrink', (rel>0.05).sum((1,2)).tolist(), 'per block', (rel>0.05).sum((0,1)).tolist(), 'rows', (rel>0.05).sum((0,2)).nonzero().reshape(-1).tolist()[:20])
