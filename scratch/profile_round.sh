#!/bin/bash
# The round's profile set, on the GPU box (under gpurun): kernel-trace stats + the three PMC passes for the north-star and cfg3
# workloads (residual+Jacobian step, and the residual-only assembly), then profiles/r05_traffic.json.  Outputs under
# gpurun_out/r05p/ (copy what is to be kept into profiles/).  usage: bash scratch/profile_round.sh [ROUND_TAG]
set -e
R=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${R}p
mkdir -p $O
for W in northstar cfg3; do
  for M in grad residual; do
    X=""; [ $M = residual ] && X="--residual-only"
    B="python bench.py --workload $W --no-cpu-baseline --no-other-configs --steps 3 --warmup 1 $X"
    T=${W}_${M}
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_stats -- $B > $O/${T}_stats.log 2>&1
    timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $O/${T}_pipe -- $B > $O/${T}_pipe.log 2>&1
    timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${T}_fetch -- $B > $O/${T}_fetch.log 2>&1
    timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${T}_write -- $B > $O/${T}_write.log 2>&1
    cp $O/${T}_stats/*/*kernel_stats.csv $O/${R}_${T}_kernel_stats.csv
    python scratch/pmc_summary.py $O/${T}_pipe $O/${T}_fetch $O/${T}_write > $O/${R}_${T}_pmc.txt
    echo "$T done"
  done
done
python scratch/make_traffic_json.py $O/${R}_traffic.json \
  northstar/neohookean/grad/n1:$O/northstar_grad_pipe:$O/northstar_grad_fetch:$O/northstar_grad_write:262144:59012 \
  cfg3/j2/grad/n1:$O/cfg3_grad_pipe:$O/cfg3_grad_fetch:$O/cfg3_grad_write:262144:319240 \
  northstar/neohookean/residual/n1:$O/northstar_residual_pipe:$O/northstar_residual_fetch:$O/northstar_residual_write:262144:6524 \
  cfg3/j2/residual/n1:$O/cfg3_residual_pipe:$O/cfg3_residual_fetch:$O/cfg3_residual_write:262144:24328
