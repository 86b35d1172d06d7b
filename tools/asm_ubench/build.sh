#!/bin/bash
# Builds the instruction probes (gen.py <set>) and their runner into tools/asm_ubench/build/<set>/ (cross-compiles here; run on the GPU
# box: tools/asm_ubench/build/<set>/run tools/asm_ubench/build/<set>).   build.sh [default|energy|pairing|fp64]
set -e
cd "$(dirname "$0")"
SET=${1:-default}
OUT=build/$SET
LLVM=${LLVM:-/opt/rocm/lib/llvm/bin}
mkdir -p $OUT
python3 gen.py $OUT $( [ "$SET" = default ] || echo $SET )
for f in $OUT/*.s; do
  $LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $f -o ${f%.s}.o
  $LLVM/ld.lld -shared ${f%.s}.o -o ${f%.s}.hsaco
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 run.cpp -o $OUT/run
echo "built $OUT"
