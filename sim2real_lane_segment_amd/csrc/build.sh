#!/bin/bash
# Builds librln.so (gfx950 only) in-tree. Usage: build.sh [extra hipcc flags]
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -I../../include"
mkdir -p build
pids=()
for f in igemm pointwise dense3 pw1 ct3 fc3 net; do
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ igemm.h -nt build/$f.o ] || [ pointwise.h -nt build/$f.o ] || [ common.h -nt build/$f.o ] || [ dense3.h -nt build/$f.o ] || [ pw1.h -nt build/$f.o ] || [ ct3.h -nt build/$f.o ] || [ fc3.h -nt build/$f.o ] || [ split16.h -nt build/$f.o ] || [ storage.h -nt build/$f.o ] || [ ../../include/rln.h -nt build/$f.o ]; then
    $HIPCC $FLAGS "$@" -c $f.hip -o build/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o librln.so build/igemm.o build/pointwise.o build/dense3.o build/pw1.o build/ct3.o build/fc3.o build/net.o
echo "built $(pwd)/librln.so"
