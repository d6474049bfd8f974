"""Are two builds' fused steps bit-identical?  usage: python tools/exp/chain_biteq.py libA.so libB.so  (runs each in a child process)"""
import hashlib, os, subprocess, sys
CHILD = r'''
import sys, hashlib, torch
sys.path.insert(0, ".")
from spair_pytorch_amd import config as cfg, models
from spair_pytorch_amd.data import scattered_digits
cfg.set_grid(128, (2, 2, 2, 1, 1, 1))
torch.manual_seed(3)
m = models.SPAIR([1, 128, 128], None, torch.device("cuda"), compute_dtype="bf16").to("cuda")
x = torch.from_numpy(scattered_digits(1, 64, 128, 11)[0]).cuda()
torch.manual_seed(5)
m.zero_grad(); loss, recon, zw, zp = m(x, 2000); loss.backward()
h = hashlib.sha256()
for t in (m.loss_terms(), recon, zw, zp, m.flat_gradients()): h.update(t.detach().cpu().numpy().tobytes())
print("HASH", h.hexdigest(), float(loss))
'''
out = []
for lib in sys.argv[1:]:
    env = dict(os.environ)
    if lib != "default": env["SPAIR_HIP_LIB"] = os.path.abspath(lib)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("HASH")]
    print(lib, line[0] if line else r.stderr[-500:])
    out.append(line[0] if line else None)
print("bit-identical" if len(set(out)) == 1 and out[0] else "DIFFERENT")
