import sys, os, torch, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests/golden')
import golden_inputs as gi
from spair_pytorch_amd import config as cfg
from spair_pytorch_amd.models import SPAIR
from spair_pytorch_amd.optim import FusedAdam
from spair_pytorch_amd.data import scattered_digits
def run(dtype, I, B, steps, gs0):
    cfg.set_grid(I, (2,2,2,1,1,1))
    torch.manual_seed(3)
    m = SPAIR([1,I,I], None, torch.device("cuda"), compute_dtype=dtype).to("cuda")
    opt = FusedAdam(m, lr=1e-4)
    x = torch.from_numpy(scattered_digits(5, B, I, 3 if I==48 else 11)[0]).cuda()
    torch.manual_seed(11)
    out=[]
    for s in range(steps):
        opt.zero_grad()
        loss = m(x, gs0+s)[0]
        loss.backward()
        opt.step()
        out.append(loss.item())
    return np.array(out)
for (I,B,steps,gs0) in ((48,16,200,1000),(128,256,60,2000)):
    a=run("f32",I,B,steps,gs0); b=run("bf16",I,B,steps,gs0)
    rel=np.abs(a-b)/np.abs(a)
    print(I,B,"f32 first/last",a[0],a[-1],"bf16",b[0],b[-1])
    print(" rel diff at steps", {k: float(rel[k]) for k in (0,1,5,10,20,50,steps-1) if k<steps}, "max", float(rel.max()))
