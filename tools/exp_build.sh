#!/bin/bash
# Kernel-ablation build: tools/exp_build.sh NAME FILE.hip -DMACRO... -> gpurun_exp_NAME.so (git-ignored),
# selected at run time with CRIMAC_LIB=$PWD/gpurun_exp_NAME.so.  Needs an up-to-date normal build.
cd "$(dirname "$0")/.." || exit 1
name=$1; file=$2; shift 2
pkg=crimac_classifiers_unet_amd
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -munsafe-fp-atomics -Wno-unused-value "$@" \
  -c $pkg/csrc/$file -o /tmp/exp_$name.o || exit 1
objs=$(ls $pkg/build/*.o | grep -v "/${file%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_exp_$name.so /tmp/exp_$name.o $objs || exit 1
echo gpurun_exp_$name.so
