#!/bin/bash
# usage: tools/pmc_run.sh <outdir under gpurun_out> <filter> -- program args...   (GPU box; two SQ counter passes)
set -e
out=gpurun_out/$1; flt=$2; shift 3
cd /tmp 2>/dev/null || true
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf "$out"; mkdir -p "$out"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d "$out/p1" -- "$@" > "$out/p1.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS --output-format csv -d "$out/p2" -- "$@" > "$out/p2.log" 2>&1
python3 tools/pmc_all.py --filter "$flt" "$out/p1" "$out/p2" | tee "$out/summary.txt"
