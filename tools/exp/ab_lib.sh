# usage: tools/exp/ab_lib.sh <variant> : same-box A/B of build/libspair_<variant>.so against the product library (bench main loop, 2 rounds)
for i in 1 2; do for v in "" $1; do
  if [ -z "$v" ]; then lib=""; else lib="build/libspair_$v.so"; fi
  SPAIR_HIP_LIB=$lib python bench.py --repeat 1 --no-sweep --no-config3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
b=d['step_breakdown_ms']
print('${v:-product}', round(d['ms_per_step'],4), {k:round(b[k],3) for k in ('decoder_bwd','cells_bwd','cells_wgrad','backbone_bwd','dec_out_wgrad')})
"; done; done
