#!/bin/bash
# Builds liblattigo_ring_hip.so (gfx950 code object + host ABI) in-tree with hipcc.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../../include -I. -Wall -Wno-unused-function"
mkdir -p build
pids=()
for f in lr_ntt.hip lr_ewise.hip lr_bext.hip; do
  $HIPCC $FLAGS -c $f -o build/${f%.hip}.o &
  pids+=($!)
done
$HIPCC $FLAGS -x hip -c lr_abi.cpp -o build/lr_abi.o &
pids+=($!)
$HIPCC $FLAGS -x hip -c lr_precompute.cpp -o build/lr_precompute.o &
pids+=($!)
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../liblattigo_ring_hip.so build/lr_ntt.o build/lr_ewise.o build/lr_bext.o build/lr_abi.o build/lr_precompute.o
echo "built $(cd .. && pwd)/liblattigo_ring_hip.so"
