#!/bin/bash
# The round's profile set, on the GPU box (under gpurun): kernel-trace stats + the three PMC passes for the north-star and cfg3
# workloads, then profiles/r04_traffic.json.  Outputs under gpurun_out/r04/ (copy what is to be kept into profiles/).
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
for W in northstar cfg3; do
  B="python bench.py --workload $W --no-cpu-baseline --no-other-configs --steps 3 --warmup 1"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${W}_stats -- $B > $O/${W}_stats.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $O/${W}_pipe -- $B > $O/${W}_pipe.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${W}_fetch -- $B > $O/${W}_fetch.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${W}_write -- $B > $O/${W}_write.log 2>&1
  cp $O/${W}_stats/*/*kernel_stats.csv $O/r04_${W}_kernel_stats.csv
  python scratch/pmc_summary.py $O/${W}_pipe $O/${W}_fetch $O/${W}_write > $O/r04_${W}_pmc.txt
done
python scratch/make_traffic_json.py $O/r04_traffic.json \
  northstar/neohookean/grad/n1:$O/northstar_pipe:$O/northstar_fetch:$O/northstar_write:262144:59012 \
  cfg3/j2/grad/n1:$O/cfg3_pipe:$O/cfg3_fetch:$O/cfg3_write:262144:319240
