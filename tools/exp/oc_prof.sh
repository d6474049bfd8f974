#!/bin/bash
# kernel stats of tools/exp/objconv_time.py (GPU box)
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/oc_prof; rm -rf $o; mkdir -p $o
rocprofv3 --kernel-trace --stats --output-format csv -d $o -- python3 tools/exp/objconv_time.py ${1:-256} > $o/run.log 2>&1
tail -2 $o/run.log
python3 - <<PY
import csv,glob
f=glob.glob("$o/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:16]: print("%-100s calls %6s total %9.2f ms avg %8.1f us" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3))
PY
