#!/bin/bash
# usage: scratch/ablate.sh lib1.so lib2.so ... : run the bench with each library in place of the built one
cp mimi_amd/lib/libmimi_hip.so /tmp/lib_save.so
for l in "$@"; do
  cp "$l" mimi_amd/lib/libmimi_hip.so
  echo "== $l"
  timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done
cp /tmp/lib_save.so mimi_amd/lib/libmimi_hip.so
