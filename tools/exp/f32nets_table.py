"""Which net's bf16 operands cost the gradient direction?  Per-wavefront path (flags bit 0) with ONE net's forward layers on fp32 operands
(experiment build libspair_f32nets.so: flags 128 box, 256 encoder, 512 z, 1024 obj)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bf16_parity_table as T
from spair_pytorch_amd import models
import io, contextlib
for name in sys.argv[1:] or ["c1_b8_step7001", "ref_default_b2_step1001", "c2_b2_step1001", "c1_b16_step1", "c1_b8_step1001", "c4_b1_step1001"]:
    for fl, lab in ((1, "bf16 per-wavefront"), (1 | 128, "+ box fp32"), (1 | 256, "+ enc fp32"), (1 | 128 | 256, "+ box, enc fp32")):
        models.STEP_FLAGS = fl
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            T.run(name)
        out = buf.getvalue().splitlines()
        print("%-24s %-22s %s | %s" % (name, lab, out[0].split(":")[1][:60], out[-1].strip()))
models.STEP_FLAGS = 0
