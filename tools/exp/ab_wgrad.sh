for i in 1 2; do for v in wgold wgnew; do SPAIR_HIP_LIB=build/libspair_$v.so python bench.py --repeat 1 --no-sweep --no-config3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
b=d['step_breakdown_ms']
print('$v', round(d['ms_per_step'],4), 'cells_wgrad', round(b['cells_wgrad'],4), 'backbone_bwd', round(b['backbone_bwd'],4), 'dec_out_wgrad', round(b['dec_out_wgrad'],4))
"; done; done
