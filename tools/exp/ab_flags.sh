#!/bin/bash
# usage (GPU box): tools/exp/ab_flags.sh flags1 flags2 ...  -> step time + step breakdown per SPAIR_STEP_FLAGS value (in-tree library), each twice
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for fl in "$@"; do
  SPAIR_STEP_FLAGS=$fl python bench.py --no-cpu-baseline --no-sweep --no-config3 --steps 40 --warmup 10 --repeat 3 > gpurun_out/abf_$fl.log 2>&1
  python - "$fl" <<'PY'
import json,sys
fl=sys.argv[1]
try:
    d=json.loads(open("gpurun_out/abf_%s.log"%fl).read().strip().splitlines()[-1])
    sb=d.get("step_breakdown_ms",{})
    print("flags %-4s step %.3f ms (min %.3f)  breakdown: %s" % (fl, d["ms_per_step"], d["ms_per_step_min"], {a:round(b,3) for a,b in sb.items()}))
except Exception as e:
    print(fl, "failed:", e); print(open("gpurun_out/abf_%s.log"%fl).read()[-600:])
PY
done
done
