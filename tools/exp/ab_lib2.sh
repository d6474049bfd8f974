#!/bin/bash
# usage (GPU box): tools/exp/ab_lib2.sh <libA|default> <libB> [bench args]: alternating bench runs of two builds on one box
a=$1; b=$2; shift 2
for i in 1 2 3; do
  for l in $a $b; do
    if [ "$l" == "default" ]; then unset SPAIR_HIP_LIB; else export SPAIR_HIP_LIB=$l; fi
    python bench.py --no-cpu-baseline --no-sweep --no-config3 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$l', round(d['ms_per_step'],4), [round(v,4) for v in d['ms_per_step_repeats']])"
  done
done
