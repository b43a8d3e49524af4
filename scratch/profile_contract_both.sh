#!/bin/bash
# the SQ / GRBM counters of the degree-3 contraction in its two forms (MIMI_HIP_P3_CONTRACT = cxx | asm), one box
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05c
mkdir -p $O
B="python bench.py --workload cfg3 --no-cpu-baseline --no-other-configs --steps 3 --warmup 1"
for V in cxx asm; do
  export MIMI_HIP_P3_CONTRACT=$V
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $O/${V}_pipe -- $B > $O/${V}_pipe.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/${V}_mix -- $B > $O/${V}_mix.log 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${V}_stats -- $B > $O/${V}_stats.log 2>&1
  echo "== $V" >> $O/both.txt
  grep -h "tp3_contract" $O/${V}_stats/*/*kernel_stats.csv >> $O/both.txt
  python scratch/pmc_summary.py $O/${V}_pipe $O/${V}_mix 2>/dev/null | grep "tp3_contract" >> $O/both.txt
done
cat $O/both.txt
