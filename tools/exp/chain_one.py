import sys, os, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo/tests/golden')
import test_chain_gpu as T
from helpers import load_case
z, case = load_case("ref_default_b2_step1001")
a = T.run(case, z, flags=0)
print(float(a["terms"][0]))
