"""The fused decoder data-gradient kernel (dec_fused_bwd.hip) alone at configs[1]'s N = 65,536 rows through its C-ABI entry point (GPU box only;
developer tool).  DB_NO=<n_out> shortens the first layer's K: with 64 / 512 / 1568 logits the launch takes 0.040 / 0.054 / 0.101 ms, i.e. ~20 us per
workgroup round of fixed cost (epilogues: gate loads, dH2 / dH1 / d z_attr stores, layers 1 and 0) + 1.27 us per 64-deep stage."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spair_pytorch_amd import _lib as L
lib = L.lib(); bf = torch.bfloat16
N, A, LDR, NO = 65536, 50, 56, int(os.environ.get("DB_NO", "1568"))
dL = (torch.randn(N, NO, device="cuda") * 0.05).to(bf)
W2t = (torch.randn(256, NO, device="cuda") * 0.08).to(bf); W1t = (torch.randn(128, 256, device="cuda") * 0.1).to(bf); W0t = (torch.randn(A, 128, device="cuda") * 0.2).to(bf)
H2 = torch.relu(torch.randn(N, 256, device="cuda")).to(bf); H1 = torch.relu(torch.randn(N, 128, device="cuda")).to(bf)
dH2 = torch.empty(N, 256, device="cuda", dtype=bf); dH1 = torch.empty(N, 128, device="cuda", dtype=bf); dza = torch.empty(N, LDR, device="cuda")
def run():
    L.check(lib.spair_decoder_bwd16(L.ptr(dL), NO, L.ptr(W2t), NO, L.ptr(W1t), L.ptr(W0t), L.ptr(H2), L.ptr(H1), L.ptr(dH2), L.ptr(dH1), L.ptr(dza), LDR,
                                    ctypes.c_longlong(N), A, NO, L.stream()), "bwd")
run(); run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 20
fl = 2.0 * N * (NO * 256 + 256 * 128 + 128 * A)
print("fused decoder data-gradient chain, N = %d, n_out = %d: %.3f ms (%.0f TFLOP/s)" % (N, NO, t, fl / t / 1e9))
