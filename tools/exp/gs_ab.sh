for gs in 2000 4000 2000 4000; do python bench.py --repeat 1 --no-sweep --no-config3 --no-cpu-baseline --global-step $gs 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
b=d['step_breakdown_ms']
print('$gs', round(d['ms_per_step'],4), {k:round(v,3) for k,v in b.items()})
"; done
