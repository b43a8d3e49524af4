#!/bin/bash
# usage: scratch/mklib.sh NAME [extra hipcc flags...]  -> scratch/lib_NAME.so
name=$1; shift
cd /root/repo/mimi_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -Wno-unused-result "$@" -o /root/repo/scratch/lib_$name.so domain.hip contact.hip krylov.hip 2>&1 | grep -i "error" -A5 | head -10
