#!/bin/bash
# usage (GPU box): tools/exp/ab_tail.sh lib...  -> step time and backbone-backward breakdown per library (timing only)
cd "$GRAFT_REPO_ROOT"
for lib in "$@"; do
  if [ "$lib" == "base" ]; then unset SPAIR_HIP_LIB; else export SPAIR_HIP_LIB=$PWD/build/libspair_$lib.so; fi
  python bench.py --no-cpu-baseline --no-sweep --no-config3 --steps 40 --warmup 10 --repeat 3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); sb=d.get('step_breakdown_ms',{})
print('$lib', 'step %.3f (min %.3f)' % (d['ms_per_step'], d['ms_per_step_min']), {k:round(v,3) for k,v in sb.items() if 'bwd' in k or 'wgrad' in k})"
done
