import sys, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo/tests/golden')
import test_engine_gpu as T
from spair_pytorch_amd.data import scattered_digits
x = torch.from_numpy(scattered_digits(1234, 256, 128, 11)[0]).cuda()
out={}
for dtype in ("f32","bf16"):
    m=T._bench_model(dtype,128); torch.manual_seed(7); m.zero_grad()
    loss=m(x,2000)[0]; loss.backward()
    out[dtype]=(m.flat_gradients().double().cpu().numpy().copy(), dict(m._slices), loss.item())
    del m
a,b=out["f32"],out["bf16"]
print("loss", a[2], b[2], abs(a[2]-b[2])/abs(a[2]))
for k,(off,cnt,shp) in a[1].items():
    if k.startswith("attn."): continue
    ga,gb=a[0][off:off+cnt],b[0][off:off+cnt]
    print("%-42s cos %.5f norm %.4f" % (k, np.dot(ga,gb)/(np.linalg.norm(ga)*np.linalg.norm(gb)+1e-30), np.linalg.norm(gb)/np.linalg.norm(ga)))
