import sys, os, ctypes, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo/tests/golden')
import golden_inputs as gi
from helpers import load_case
from spair_pytorch_amd import config as cfg, models, _lib as L
for name in ("ref_default_b2_step1001", "c2_b2_step1001"):
    z, case = load_case(name)
    cfg.set_grid(case["I"], case["strides"])
    models.STEP_FLAGS = 2
    m = models.SPAIR([1, case["I"], case["I"]], None, torch.device("cuda"), compute_dtype="bf16").to("cuda")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in gi.make_weights(case["wseed"], case["wscale"]).items()})
    x = torch.from_numpy(z["x"]).cuda()
    noise = {k: torch.from_numpy(z[k]).cuda() for k in ("eps_box", "eps_attr", "eps_depth", "u_pres")}
    loss = m(x, int(z["global_step"]), noise=noise)[0]
    torch.cuda.synchronize()
    e = m._last["engine"]
    out = torch.zeros(4096, dtype=torch.int64, device="cuda")
    L.check(L.lib().spair_chain_stamps(ctypes.byref(e["dims"]), L.ptr(e["workspace"]), L.ptr(out), 4096, L.stream()), "stamps")
    print(name, "mismatches:", int(out[4000].item()), "loss", loss.item())
