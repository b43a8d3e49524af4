#!/bin/bash
# the loop-back rehearsals of profiles/rNN_secondary_configs.txt (one rank of an N-rank job alone on one GPU), on the GPU box
set -e
O=gpurun_out/rehearse; mkdir -p $O
for spec in "2 northstar" "4 northstar" "8 northstar" "8 cfg3" "4 cfg4" "8 cfg5"; do
  set -- $spec
  timeout -k 10 300 python bench.py --rehearse-rccl $1 --workload $2 --steps 10 --warmup 3 > $O/r$1_$2.json 2> $O/r$1_$2.err
  python - "$1" "$2" "$O/r$1_$2.json" <<'PY'
import json, sys
n, w, f = sys.argv[1:]
d = json.loads(open(f).read().strip().splitlines()[-1])
fp = d.get("footprint", {})
print("--rehearse-rccl %s --workload %s: %.3f ms per step for this rank (%s elements), set-up %.1f s, host peak %.2f GiB, device %.2f GiB in use" % (
    n, w, d["ms_per_step"], d.get("elements_this_rank", d.get("config", {}).get("elements_this_rank", "?")), fp.get("setup_s", float("nan")), fp.get("host_peak_rss_gib", float("nan")), fp.get("device_gib_in_use_after_the_run", float("nan"))), flush=True)
PY
done
