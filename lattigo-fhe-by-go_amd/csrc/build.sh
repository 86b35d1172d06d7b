#!/bin/bash
# Builds liblattigo_ring_hip.so (gfx950 code object + host ABI) in-tree with hipcc.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../../include -I. -Wall -Wno-unused-function"
mkdir -p build
# hand-scheduled assembly kernels: generate -> assemble -> embed
LLVM=${LLVM:-/opt/rocm/lib/llvm/bin}
for n in 14 15; do
  python3 asmgen/gen_ntt.py $n build/ntt_fwd$n.s
  python3 asmgen/gen_intt.py $n build/ntt_inv$n.s
  for k in fwd inv; do
    $LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c build/ntt_$k$n.s -o build/ntt_$k$n.o
    $LLVM/ld.lld -shared build/ntt_$k$n.o -o build/ntt_$k$n.hsaco
  done
done
python3 - <<'PY'
out = []
for k in ("fwd", "inv"):
    for n in (14, 15):
        data = open("build/ntt_%s%d.hsaco" % (k, n), "rb").read()
        out.append('extern "C" const unsigned char lr_hsaco_%s%d[] __attribute__((aligned(4096))) = {' % (k, n))
        out.append(",".join(str(b) for b in data))
        out.append("};")
        out.append('extern "C" const unsigned long lr_hsaco_%s%d_size = %d;' % (k, n, len(data)))
open("build/lr_asm_blob.cpp", "w").write("\n".join(out) + "\n")
PY
pids=()
for f in lr_ntt.hip lr_ewise.hip lr_bext.hip; do
  $HIPCC $FLAGS -c $f -o build/${f%.hip}.o &
  pids+=($!)
done
$HIPCC $FLAGS -x hip -c lr_abi.cpp -o build/lr_abi.o &
pids+=($!)
$HIPCC $FLAGS -x hip -c lr_precompute.cpp -o build/lr_precompute.o &
pids+=($!)
$HIPCC $FLAGS -x hip -c lr_asm.cpp -o build/lr_asm.o &
pids+=($!)
g++ -O1 -fPIC -c build/lr_asm_blob.cpp -o build/lr_asm_blob.o &
pids+=($!)
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../liblattigo_ring_hip.so build/lr_ntt.o build/lr_ewise.o build/lr_bext.o build/lr_abi.o build/lr_precompute.o build/lr_asm.o build/lr_asm_blob.o
echo "built $(cd .. && pwd)/liblattigo_ring_hip.so"
