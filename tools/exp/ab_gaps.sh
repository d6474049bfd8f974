#!/bin/bash
# usage (GPU box): tools/exp/ab_gaps.sh lib1 lib2 ...  -> step time (bench) and the idle gaps of a traced step per library (`base` = in-tree)
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for lib in "$@"; do
  if [ "$lib" == "base" ]; then unset SPAIR_HIP_LIB; else export SPAIR_HIP_LIB=$PWD/build/libspair_$lib.so; fi
  python bench.py --no-cpu-baseline --no-sweep --no-config3 --steps 40 --warmup 10 --repeat 3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', 'step %.3f (min %.3f)' % (d['ms_per_step'], d['ms_per_step_min']))"
  rm -rf gpurun_out/ps_g_$lib
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ps_g_$lib -- python3 bench.py --no-cpu-baseline --no-sweep --no-config3 --steps 20 --warmup 5 --repeat 1 > gpurun_out/ps_g_$lib.log 2>&1
  python3 tools/step_gaps.py gpurun_out/ps_g_$lib | tail -8
done
