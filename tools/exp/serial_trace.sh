#!/bin/bash
# usage (GPU box): tools/exp/serial_trace.sh  -> gpurun_out/serial_timeline.txt: the step's kernels with the helper stream off
# (SPAIR_STEP_FLAGS=4: every kernel alone on the caller's stream), i.e. each kernel's isolated duration inside the real step
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/ps_serial; rm -rf $o; mkdir -p $o
export SPAIR_STEP_FLAGS=4
rocprofv3 --kernel-trace --stats --output-format csv -d $o/trace -- python3 bench.py --no-cpu-baseline --no-sweep --no-config3 --steps 20 --warmup 5 --repeat 1 > $o/trace.log 2>&1
(python3 tools/step_trace.py $o/trace 15; python3 tools/step_gaps.py $o/trace) > gpurun_out/serial_timeline.txt
tail -4 gpurun_out/serial_timeline.txt
