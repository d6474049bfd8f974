#!/bin/bash
# usage (GPU box): tools/exp/ab_dense.sh lib1 lib2 ...  -> chain kernel times on DENSE images (bench.py --dense-input) per library
cd "$GRAFT_REPO_ROOT"
for lib in "$@"; do
  if [ "$lib" == "base" ]; then unset SPAIR_HIP_LIB; else export SPAIR_HIP_LIB=$PWD/build/libspair_$lib.so; fi
  python bench.py --dense-input --no-cpu-baseline --no-sweep --no-config3 --steps 40 --warmup 10 --repeat 2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('$lib dense: step %.3f ms  chain fwd %.4f bwd %.4f' % (d['ms_per_step'], k['chain_fwd']['avg_ms'], k['chain_bwd']['avg_ms']))"
done
