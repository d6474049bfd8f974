#!/bin/bash
# usage (GPU box): tools/exp/ab_step.sh lib1 lib2 ...  -> step time + forward-side kernel events per library (`base` = the in-tree library)
cd "$GRAFT_REPO_ROOT"
for lib in "$@"; do
  if [ "$lib" == "base" ]; then unset SPAIR_HIP_LIB; else export SPAIR_HIP_LIB=$PWD/build/libspair_$lib.so; fi
  python bench.py --no-cpu-baseline --no-sweep --no-config3 --steps 40 --warmup 10 --repeat 3 > gpurun_out/abs_$lib.log 2>&1
  python - "$lib" <<'PY'
import json,sys
lib=sys.argv[1]
try:
    d=json.loads(open("gpurun_out/abs_%s.log"%lib).read().strip().splitlines()[-1])
    k=d["kernels"]; sb=d.get("step_breakdown_ms",{})
    print("%-8s step %.3f ms (min %.3f)  dec_fwd %.4f render_fwd %.4f  breakdown: %s" % (lib, d["ms_per_step"], d["ms_per_step_min"], k["decoder_fwd"]["avg_ms"], k["render_fwd"]["avg_ms"], {a:round(b,3) for a,b in sb.items() if a in ("decoder_fwd","render_fwd","count_kl","loss","cells_fwd")}))
except Exception as e:
    print(lib, "failed:", e); print(open("gpurun_out/abs_%s.log"%lib).read()[-600:])
PY
done
