#!/bin/bash
# Experimental build whose DEVICE assembly is rewritten between compiler and assembler (development aid):
#   tools/build_asm_rewrite.sh <name> <sed script file | rewriter.py> [-DRAYHS_EXP_... ...]
# -> build_exp/<name>/librayhs_hip.so.  The pipeline is hipcc's own (hipcc -### shows it), cut open at the .s file:
# device -S, sed, assemble, lld -shared, clang-offload-bundler, host compile with -fcuda-include-gpubinary.
set -e
name=$1; script=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/rayhs_amd/csrc
obj=$root/build_exp/o_$name; bin=$root/build_exp/$name
mkdir -p $obj $bin
LLVM=/opt/rocm/lib/llvm/bin
for v in exact fast; do
  flags="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -DRAYHS_DEV_KNOBS $* -DRAYHS_KERNEL_SUFFIX=$v"
  [ $v = exact ] && flags="$flags -ffp-contract=off"
  /opt/rocm/bin/hipcc $flags --cuda-device-only -S $src/trace_kernel.hip -o $obj/dev_$v.s
  case $script in   # a sed script rewrites both variants; a python rewriter (tools/bbprof_rewrite.py) the exact kernels only
    *.py) if [ $v = exact ]; then python3 $script $obj/dev_$v.s $obj/dev_${v}_mod.s $bin/bbmap.json; else cp $obj/dev_$v.s $obj/dev_${v}_mod.s; fi;;
    *) sed -E -f $script $obj/dev_$v.s > $obj/dev_${v}_mod.s;;
  esac
  $LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $obj/dev_${v}_mod.s -o $obj/dev_$v.o
  $LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $obj/dev_$v.out $obj/dev_$v.o
  $LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$obj/dev_$v.out -output=$obj/dev_$v.hipfb
  /opt/rocm/bin/hipcc $flags --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $obj/dev_$v.hipfb -c $src/trace_kernel.hip -o $obj/trace_$v.o
done
# the rest of the library as usual (the trace objects above are newer than their sources: make leaves them alone)
make -C $src -j4 -s BUILD=$obj BIN=$bin KDEFS="-DRAYHS_DEV_KNOBS $*" $bin/librayhs_hip.so
echo "built build_exp/$name: $(grep -c . $script) rewrite rule(s), $(diff $obj/dev_exact.s $obj/dev_exact_mod.s | grep -c '^>') lines changed in the exact kernels"
