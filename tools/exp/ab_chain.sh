#!/bin/bash
# usage (GPU box): tools/exp/ab_chain.sh lib1 lib2 ...  -> chain kernel times and step time per library (timing only: variants may be numerically wrong)
cd "$GRAFT_REPO_ROOT"
for lib in "$@"; do
  if [ "$lib" == "base" ]; then unset SPAIR_HIP_LIB; else export SPAIR_HIP_LIB=$PWD/build/libspair_$lib.so; fi
  python bench.py --no-cpu-baseline --no-sweep --no-config3 --steps 40 --warmup 10 --repeat 2 > gpurun_out/ab_$lib.log 2>&1
  python - "$lib" <<'PY'
import json,sys
lib=sys.argv[1]
try:
    d=json.loads(open("gpurun_out/ab_%s.log"%lib).read().strip().splitlines()[-1])
    k=d["kernels"]
    print("%-8s step %.3f ms (min %.3f)  chain fwd %.4f bwd %.4f" % (lib, d["ms_per_step"], d["ms_per_step_min"], k["chain_fwd"]["avg_ms"], k["chain_bwd"]["avg_ms"]))
except Exception as e:
    print(lib, "failed:", e); print(open("gpurun_out/ab_%s.log"%lib).read()[-800:])
PY
done
