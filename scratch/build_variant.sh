#!/bin/bash
# another build of the library for same-box A/B runs (MIMI_HIP_LIBRARY=scratch/lib_NAME.so):
#   bash scratch/build_variant.sh NAME [CSRC_DIR] -- extra hipcc flags
set -e
NAME=$1; shift
ROOT=$(cd $(dirname $0)/.. && pwd)
CSRC=$ROOT/mimi_amd/csrc
if [ "$1" != "--" ] && [ -n "$1" ]; then CSRC=$1; shift; fi
[ "$1" = "--" ] && shift
OUT=$ROOT/scratch/lib_$NAME.so
TMP=$(mktemp -d)
cd $CSRC
for s in domain tensor_p3 contact krylov exchange; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result "$@" -c $s.hip -o $TMP/$s.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -shared -munsafe-fp-atomics $TMP/*.o -o $OUT
rm -rf $TMP
ls -la $OUT
