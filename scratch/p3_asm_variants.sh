#!/bin/bash
# usage: scratch/p3_asm_variants.sh NAME [generator options, e.g. --drop=store,load --per_s3=3]  -> scratch/lib_p3_NAME.so
# tensor_p3.hip rebuilt around a column loop generated with the options (timing experiments; --drop variants compute nothing useful)
set -e
name=$1; shift
python /root/repo/mimi_amd/csrc/gen_tp3_contract.py /tmp/p3asm_$name.inc "$@"
(cd /root/repo/mimi_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result "-DT3_LOOP_INC=\"/tmp/p3asm_$name.inc\"" -c tensor_p3.hip -o /tmp/p3asm_$name.o 2>&1 | grep -i " error" -A5 | head)
obj=/root/repo/mimi_amd/lib/obj
hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/scratch/lib_p3_$name.so $obj/domain.o /tmp/p3asm_$name.o $obj/contact.o $obj/krylov.o $obj/exchange.o
echo scratch/lib_p3_$name.so
