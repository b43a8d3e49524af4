#!/bin/bash
# usage: scratch/ab_lib.sh NAME [GIT_REV|-] [extra hipcc flags...]  ->  scratch/lib_NAME.so: domain.hip of the working tree (or of
# GIT_REV) compiled beside the other objects of mimi_amd/lib/obj, for same-box A/B timing
# (MIMI_HIP_LIBRARY=scratch/lib_NAME.so python bench.py ...)
set -e
name=$1; rev=$2; shift; shift || true
src=/root/repo/mimi_amd/csrc
if [ -n "$rev" ] && [ "$rev" != "-" ]; then
  rm -rf /tmp/ab_$name && mkdir -p /tmp/ab_$name
  (git archive $rev mimi_amd/csrc include) | tar -x -C /tmp/ab_$name
  src=/tmp/ab_$name/mimi_amd/csrc
fi
obj=/root/repo/mimi_amd/lib/obj
(cd $src && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result "$@" -c domain.hip -o /tmp/domain_$name.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/scratch/lib_$name.so /tmp/domain_$name.o $obj/tensor_p3.o $obj/contact.o $obj/krylov.o $obj/exchange.o
echo scratch/lib_$name.so
