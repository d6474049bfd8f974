#!/bin/bash
# usage (GPU box): tools/exp/ab_c3.sh lib1 lib2 ...  -> configs[3] step and chain kernel times per library
cd "$GRAFT_REPO_ROOT"
for lib in "$@"; do
  if [ "$lib" == "base" ]; then unset SPAIR_HIP_LIB; else export SPAIR_HIP_LIB=$PWD/build/libspair_$lib.so; fi
  python bench.py --no-cpu-baseline --no-sweep --no-config3 --image 256 --batch 64 --steps 20 --warmup 5 --repeat 3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('$lib', 'c3 step %.3f ms  chain fwd %.4f bwd %.4f  stn stage %.4f' % (d['ms_per_step'], k['chain_fwd']['avg_ms'], k['chain_bwd']['avg_ms'], k.get('stn_fwd',{}).get('stage_ms',0)))"
done
