#!/bin/bash
# Build a variant of libndjir_hip.so with extra -D flags for kernel experiments (timing only):
#   tools/build_variant.sh <name> <file.hip> [-DFLAG ...]   ->  ndjir_amd/_lib/variants/<name>.so   (use: NDJIR_HIP_LIB=<path>)
# Only <file.hip> is recompiled with the flags; every other object comes from the regular build (run make first).
set -e
cd "$(dirname "$0")/../ndjir_amd/csrc"
name=$1; src=$2; shift; shift
out=../_lib/variants; mkdir -p $out/obj_$name
cp ../_lib/obj/*.o $out/obj_$name/
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -fno-slp-vectorize "$@" -c $src -o $out/obj_$name/${src%.hip}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/$name.so $out/obj_$name/*.o
rm -rf $out/obj_$name
echo built $out/$name.so
