#!/bin/bash
# Builds liblattigo_ring_hip.so (gfx950 code object + host ABI) in-tree with hipcc.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../../include -I. -Wall -Wno-unused-function"
# LR_BUILD_DIAG=1: the diagnostics build -- adds the clock-stamping ("t") and persistent ("p") code objects of the negative-result
# experiments and lets lr_options::ntt_timeline / ntt_persist select them; the default build ships neither
[ -n "$LR_BUILD_DIAG" ] && FLAGS="$FLAGS -DLR_BUILD_DIAG=1"
mkdir -p build
# the C ABI, one translation unit per handle family (lr_host.hpp is what they share)
ABI_UNITS="lr_abi_core lr_abi_ring lr_abi_bext lr_abi_ckks lr_abi_batcher lr_abi_bfv lr_abi_bfv_batcher lr_abi_peer"
link() {
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o ../liblattigo_ring_hip.so build/lr_ntt.o build/lr_ewise.o build/lr_bext.o $(for u in $ABI_UNITS; do echo build/$u.o; done) build/lr_precompute.o build/lr_asm.o build/lr_asm_blob.o
  echo "built $(cd .. && pwd)/liblattigo_ring_hip.so"
}
# developer shortcut: `build.sh lr_abi_ckks.cpp lr_ewise.hip` recompiles only the named sources and relinks (everything else must
# have been built before); without arguments the whole library is built from scratch, which is what __graft_entry__.build() runs
if [ $# -gt 0 ]; then
  spids=()
  for f in "$@"; do
    case $f in
      *.hip) $HIPCC $FLAGS -c $f -o build/${f%.hip}.o & spids+=($!) ;;
      *.cpp) $HIPCC $FLAGS -x hip -c $f -o build/${f%.cpp}.o & spids+=($!) ;;
    esac
  done
  for p in "${spids[@]}"; do wait $p; done     # (a failed compile stops here: set -e)
  link
  exit 0
fi
# hand-scheduled assembly kernels: generate -> assemble -> embed
LLVM=${LLVM:-/opt/rocm/lib/llvm/bin}
# forward: modes 0 (q < 2^61), 1 (q <= 2^60), 2 (q < 2^57); inverse: modes 0 and 1
gen_one() {  # kind degree mode [threads]; fewer than 1024 threads = the several-workgroups-per-CU plans ("x" kernels)
  local tag=$2; [ "${4:-1024}" != 1024 ] && tag=${2}x
  python3 asmgen/gen_$( [ "$1" = fwd ] && echo ntt || echo intt ).py $2 build/ntt_$1${tag}_m$3.s $3 ${4:-1024}
  $LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c build/ntt_$1${tag}_m$3.s -o build/ntt_$1${tag}_m$3.o
  $LLVM/ld.lld -shared build/ntt_$1${tag}_m$3.o -o build/ntt_$1${tag}_m$3.hsaco
}
gpids=()
for n in 14 15; do
  for m in 0 1 2; do gen_one fwd $n $m & gpids+=($!); done
  for m in 0 1; do gen_one inv $n $m & gpids+=($!); done
done
# measured best plan per degree and direction: 2^14: 512 threads; 2^13: 256 forward / 512 inverse; 2^12: 256
for m in 0 1 2; do gen_one fwd 14 $m 512 & gpids+=($!); gen_one fwd 13 $m 256 & gpids+=($!); gen_one fwd 12 $m 256 & gpids+=($!); done
for m in 0 1; do gen_one inv 14 $m 512 & gpids+=($!); gen_one inv 13 $m 512 & gpids+=($!); gen_one inv 12 $m 256 & gpids+=($!); done
# N = 2^16: 2^15 sub-block kernels (forward with the top stage fused into the loads = s, plain = p; inverse: lazy = s, with the
# last stage by the second finisher of each pair = f)
gen_sub() {  # kind tag mode [plain]
  python3 asmgen/gen_$( [ "$1" = fwd ] && echo ntt || echo intt ).py 16 build/ntt_$1$2_m$3.s $3 1024 $4
  $LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c build/ntt_$1$2_m$3.s -o build/ntt_$1$2_m$3.o
  $LLVM/ld.lld -shared build/ntt_$1$2_m$3.o -o build/ntt_$1$2_m$3.hsaco
}
for m in 0 1 2; do gen_sub fwd 16s $m & gpids+=($!); gen_sub fwd 16p $m plain & gpids+=($!); done
for m in 0 1; do gen_sub inv 16s $m & gpids+=($!); gen_sub inv 16f $m fused & gpids+=($!); done
# dual kernels (mode 3): FP64 body for the limbs below 2^46, integer mode-2 body for the others
gen_one fwd 14 3 & gpids+=($!); gen_one fwd 15 3 & gpids+=($!)
gen_one fwd 14 3 512 & gpids+=($!); gen_one fwd 13 3 256 & gpids+=($!); gen_one fwd 12 3 256 & gpids+=($!)
gen_sub fwd 16s 3 & gpids+=($!); gen_sub fwd 16p 3 plain & gpids+=($!)
gen_one inv 14 3 & gpids+=($!); gen_one inv 15 3 & gpids+=($!)
gen_one inv 14 3 512 & gpids+=($!); gen_one inv 13 3 512 & gpids+=($!); gen_one inv 12 3 256 & gpids+=($!)
gen_sub inv 16s 3 & gpids+=($!); gen_sub inv 16f 3 fused & gpids+=($!)
# mode 4: the dual forward kernels with the subtract-multiply-add epilogue on the FP64 body (ModDown inside the key switch)
gen_one fwd 14 4 & gpids+=($!); gen_one fwd 15 4 & gpids+=($!)
gen_one fwd 14 4 512 & gpids+=($!); gen_one fwd 13 4 256 & gpids+=($!); gen_one fwd 12 4 256 & gpids+=($!)
gen_sub fwd 16s 4 & gpids+=($!); gen_sub fwd 16p 4 plain & gpids+=($!)
# mode 5: the integer kernels of mode 1 (q <= 2^60) with the subtract-multiply-add epilogue in integer arithmetic (rescale / ModDown on the
# reference's 60-bit rings)
gen_one fwd 14 5 & gpids+=($!); gen_one fwd 15 5 & gpids+=($!)
gen_one fwd 14 5 512 & gpids+=($!); gen_one fwd 13 5 256 & gpids+=($!); gen_one fwd 12 5 256 & gpids+=($!)
gen_sub fwd 16s 5 & gpids+=($!); gen_sub fwd 16p 5 plain & gpids+=($!)
# diagnostics: the 2^15 kernels with per-phase clock stamps (LR_NTT_TIMELINE=1, tools/timeline.py): integer and dual, both directions
gen_tl() {  # kind mode
  python3 asmgen/gen_$( [ "$1" = fwd ] && echo ntt || echo intt ).py 15 build/ntt_${1}15_m${2}t.s $2 1024 timeline
  $LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c build/ntt_${1}15_m${2}t.s -o build/ntt_${1}15_m${2}t.o
  $LLVM/ld.lld -shared build/ntt_${1}15_m${2}t.o -o build/ntt_${1}15_m${2}t.hsaco
}
if [ -n "$LR_BUILD_DIAG" ]; then for k in fwd inv; do for m in 1 3; do gen_tl $k $m & gpids+=($!); done; done; fi
# persistent forward 2^15 kernels (several polys per workgroup, next poly's loads prefetched) and their timeline builds
gen_p() {  # mode flavour suffix
  python3 asmgen/gen_ntt.py 15 build/ntt_fwd15p_m$1$3.s $1 1024 $2
  $LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c build/ntt_fwd15p_m$1$3.s -o build/ntt_fwd15p_m$1$3.o
  $LLVM/ld.lld -shared build/ntt_fwd15p_m$1$3.o -o build/ntt_fwd15p_m$1$3.hsaco
}
if [ -n "$LR_BUILD_DIAG" ]; then
  for m in 0 1 2 3; do gen_p $m persist "" & gpids+=($!); done
  for m in 1 3; do gen_p $m persist-timeline t & gpids+=($!); done
fi
# N = 2^15 as two 2^14 sub-blocks ("h": plain forward sub-blocks after the stage over bit 14, lazy inverse ones before it): launches too
# small to fill the chip with one workgroup per transform
gen_h() {  # kind mode
  python3 asmgen/gen_$( [ "$1" = fwd ] && echo ntt || echo intt ).py 15 build/ntt_${1}15h_m$2.s $2 1024 halves
  $LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c build/ntt_${1}15h_m$2.s -o build/ntt_${1}15h_m$2.o
  $LLVM/ld.lld -shared build/ntt_${1}15h_m$2.o -o build/ntt_${1}15h_m$2.hsaco
}
for m in 0 1 2 3 4 5; do gen_h fwd $m & gpids+=($!); done
for m in 0 1 3; do gen_h inv $m & gpids+=($!); done
for p in "${gpids[@]}"; do wait $p; done
python3 - <<'PY'
out = ['struct lr_asm_blob { const char *name; const unsigned char *data; unsigned long size; };']
names = [("fwd", n, m) for n in (14, 15) for m in (0, 1, 2)] + [("inv", n, m) for n in (14, 15) for m in (0, 1)]
names += [("fwd", n, m) for n in ("12x", "13x", "14x", "16s", "16p") for m in (0, 1, 2)] + [("inv", n, m) for n in ("12x", "13x", "14x", "16s", "16f") for m in (0, 1)]
names += [("fwd", n, 3) for n in (14, 15, "12x", "13x", "14x", "16s", "16p")] + [("inv", n, 3) for n in (14, 15, "12x", "13x", "14x", "16s", "16f")]
names += [("fwd", n, 4) for n in (14, 15, "12x", "13x", "14x", "16s", "16p")]
names += [("fwd", n, 5) for n in (14, 15, "12x", "13x", "14x", "16s", "16p")]
import os
names = [(k, n, str(m)) for k, n, m in names]
if os.environ.get("LR_BUILD_DIAG"):
    names += [(k, 15, m) for k in ("fwd", "inv") for m in ("1t", "3t")]
    names += [("fwd", "15p", m) for m in ("0", "1", "2", "3", "1t", "3t")]
names += [("fwd", "15h", m) for m in "012345"] + [("inv", "15h", m) for m in "013"]
for k, n, m in names:
    data = open("build/ntt_%s%s_m%s.hsaco" % (k, n, m), "rb").read()
    out.append('static const unsigned char blob_%s%s_m%s[] __attribute__((aligned(4096))) = {' % (k, n, m))
    out.append(",".join(str(b) for b in data))
    out.append("};")
out.append('extern "C" const lr_asm_blob lr_asm_blobs[] = {')
for k, n, m in names:
    out.append('  {"lr_ntt_%s%s_m%s", blob_%s%s_m%s, sizeof(blob_%s%s_m%s)},' % (k, n, m, k, n, m, k, n, m))
out.append("};")
out.append('extern "C" const int lr_asm_blob_count = %d;' % len(names))
open("build/lr_asm_blob.cpp", "w").write("\n".join(out) + "\n")
PY
pids=()
for f in lr_ntt.hip lr_ewise.hip lr_bext.hip; do
  $HIPCC $FLAGS -c $f -o build/${f%.hip}.o &
  pids+=($!)
done
for u in $ABI_UNITS; do
  $HIPCC $FLAGS -x hip -c $u.cpp -o build/$u.o &
  pids+=($!)
done
$HIPCC $FLAGS -x hip -c lr_precompute.cpp -o build/lr_precompute.o &
pids+=($!)
$HIPCC $FLAGS -x hip -c lr_asm.cpp -o build/lr_asm.o &
pids+=($!)
g++ -O1 -fPIC -c build/lr_asm_blob.cpp -o build/lr_asm_blob.o &
pids+=($!)
for p in "${pids[@]}"; do wait $p; done
link
