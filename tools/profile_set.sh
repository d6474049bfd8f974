#!/bin/bash
# usage (GPU box): tools/profile_set.sh <tag> [bench args...]  -> gpurun_out/<tag>_{kernel_stats.csv,step_timeline.txt,pmc_traffic.txt/.json,sq_counters.txt,l2_requests.txt}
# every pass runs bench.py's configs[1] timed loop only (--no-sweep --no-config3: the default line's extra records would mix other
# shapes' launches into the per-kernel averages); kernel trace and every counter pass are separate rocprofv3 runs (no pass combines --pmc with a trace domain)
set -e
tag=$1; shift
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/ps_$tag
rm -rf $o; mkdir -p $o
rocprofv3 --kernel-trace --stats --output-format csv -d $o/trace -- python3 bench.py --no-cpu-baseline --no-sweep --no-config3 --steps 20 --warmup 5 --repeat 1 "$@" > $o/trace.log 2>&1
cp $(ls $o/trace/*/*kernel_stats.csv | head -1) gpurun_out/${tag}_kernel_stats.csv
(python3 tools/step_trace.py $o/trace 15; python3 tools/step_gaps.py $o/trace) > gpurun_out/${tag}_step_timeline.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $o/fetch -- python3 bench.py --no-cpu-baseline --no-sweep --no-config3 --steps 3 --warmup 1 --repeat 1 "$@" > $o/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $o/write -- python3 bench.py --no-cpu-baseline --no-sweep --no-config3 --steps 3 --warmup 1 --repeat 1 "$@" > $o/write.log 2>&1
python3 tools/pmc_summary.py $(ls $o/fetch/*/*counter_collection.csv | head -1) $(ls $o/write/*/*counter_collection.csv | head -1) gpurun_out/${tag}_pmc_traffic.json > gpurun_out/${tag}_pmc_traffic.txt
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $o/sq1 -- python3 bench.py --no-cpu-baseline --no-sweep --no-config3 --steps 3 --warmup 1 --repeat 1 "$@" > $o/sq1.log 2>&1
rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $o/sq2 -- python3 bench.py --no-cpu-baseline --no-sweep --no-config3 --steps 3 --warmup 1 --repeat 1 "$@" > $o/sq2.log 2>&1
python3 tools/sq_summary.py $o/sq1 $o/sq2 > gpurun_out/${tag}_sq_counters.txt
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $o/l2 -- python3 bench.py --no-cpu-baseline --no-sweep --no-config3 --steps 3 --warmup 1 --repeat 1 "$@" > $o/l2.log 2>&1
python3 tools/l2_summary.py $(ls $o/l2/*/*counter_collection.csv | head -1) > gpurun_out/${tag}_l2_requests.txt
tail -3 gpurun_out/${tag}_step_timeline.txt; head -12 gpurun_out/${tag}_pmc_traffic.txt
