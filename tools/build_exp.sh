#!/bin/bash
# Experimental build of the library (development aid):  tools/build_exp.sh <name> [-DRAYHS_EXP_... ...]
# -> build_exp/<name>/librayhs_hip.so, always with -DRAYHS_DEV_KNOBS (the development environment knobs exist only there);
# time it on the GPU box with  tools/gpu_batch.sh <tag> <name> ...  (RAYHS_LIB=build_exp/<name>/librayhs_hip.so tools/kbench.py)
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$root/build_exp/o_$name" "$root/build_exp/$name"
make -C "$root/rayhs_amd/csrc" -j4 -s BUILD="$root/build_exp/o_$name" BIN="$root/build_exp/$name" KDEFS="-DRAYHS_DEV_KNOBS $*" "$root/build_exp/$name/librayhs_hip.so"
echo "built build_exp/$name ($*)"
