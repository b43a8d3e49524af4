#!/bin/bash
# usage: scratch/p3_variants.sh NAME [-D...]...  -> scratch/lib_p3_NAME.so (tensor_p3.hip rebuilt with the flags, other objects reused)
name=$1; shift
cd /root/repo/mimi_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result "$@" -c tensor_p3.hip -o /tmp/tp3_$name.o 2>&1 | grep -i " error" -A5 | head
hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/scratch/lib_p3_$name.so ../lib/obj/domain.o /tmp/tp3_$name.o ../lib/obj/contact.o ../lib/obj/krylov.o ../lib/obj/exchange.o
