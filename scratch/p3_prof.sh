#!/bin/bash
# instrumented build of the degree-3 kernels (scratch/tp3_prof.hip = tensor_p3.hip + s_memtime stamps, made by the
# python snippet in scratch/README) -> scratch/lib_p3_prof.so ; run with MIMI_HIP_LIBRARY=scratch/lib_p3_prof.so python scratch/p3_ablate.py
cd /root/repo/mimi_amd/csrc && cp ../../scratch/tp3_prof.hip ./tp3_prof_tmp.hip && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -c tp3_prof_tmp.hip -o /tmp/tp3_prof.o 2>&1 | grep -i " error" -A5 | head; rm -f tp3_prof_tmp.hip
hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/scratch/lib_p3_prof.so ../lib/obj/domain.o /tmp/tp3_prof.o ../lib/obj/contact.o ../lib/obj/krylov.o ../lib/obj/exchange.o
