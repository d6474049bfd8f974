import sys, os, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo/tests/golden')
import test_chain_gpu as T
from helpers import load_case
for name in ("c2_b2_step1001", "ref_default_b2_step1001", "c4_b1_step1001"):
    z, case = load_case(name)
    a = T.run(case, z, flags=0); b = T.run(case, z, flags=1)
    print(name, {k: float((a[k]-b[k]).abs().max()) for k in ("z_where","z_pres","z_attr","z_depth","recon")}, float(abs(a["terms"][0]-b["terms"][0])/abs(b["terms"][0])))
