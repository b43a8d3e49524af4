#!/bin/bash
# engine / memory clocks and power while the north-star assembly (then cfg3) runs back to back: is the 2.4 GHz of bench.py's
# `binding` object what the chip holds under this load?   usage: bash scratch/clock_under_load.sh > gpurun_out/clock.txt
cd $(dirname $0)/..
for W in northstar cfg3; do
  python bench.py --workload $W --steps $([ $W = northstar ] && echo 5000 || echo 1000) --warmup 3 --no-cpu-baseline --no-other-configs > /tmp/clk_$W.json 2>/dev/null &
  PID=$!
  k=0
  while kill -0 $PID 2>/dev/null; do
    k=$((k + 1))
    echo "== $W t=$((k * 3))s $(/opt/rocm/bin/rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -i 'sclk\|mclk\|Power (W)\|GPU use' | sed 's/.*: //' | tr '\n' ' ')"
    sleep 3
  done
  wait $PID
  python - <<P
import json
d = json.loads(open("/tmp/clk_$W.json").read().strip().splitlines()[-1])
print("== $W", d["ms_per_step"], "ms per step over", d["steps"], "steps")
P
done
