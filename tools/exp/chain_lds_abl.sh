#!/bin/bash
# usage (GPU box): per ablation build of the chain kernels, the LDS counters of k_chain_fwd / k_chain_bwd (which stage owns the bank conflicts?)
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for v in base NOGL NOSTORE NOW; do
  o=gpurun_out/clds_$v; rm -rf $o; mkdir -p $o
  SPAIR_HIP_LIB=build/libspair_abl_$v.so BWD=1 rocprofv3 --pmc SQ_WAVES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $o/p -- python3 tools/exp/chain_ablate.py > $o/log 2>&1
  python3 tools/pmc_all.py --filter k_chain $o/p | grep -v "^ *$" | sed "s/^/$v: /"
done
