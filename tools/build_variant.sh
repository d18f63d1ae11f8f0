#!/bin/bash
# Build a variant of libndjir_hip.so with extra -D flags for kernel experiments:
#   tools/build_variant.sh <name> [-DFLAG ...]   ->  ndjir_amd/_lib/variants/<name>.so
set -e
cd "$(dirname "$0")/../ndjir_amd/csrc"
name=$1; shift
out=../_lib/variants; mkdir -p $out/obj_$name
for f in *.hip; do
  o=$out/obj_$name/${f%.hip}.o
  if [ "$f" = "mlp.hip" ] || [ "$f" = "mlp6.hip" ] || [ "$f" = "wgrad.hip" ] || [ ! -f ../_lib/obj/${f%.hip}.o ]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -fno-slp-vectorize "$@" -c $f -o $o
  else
    cp ../_lib/obj/${f%.hip}.o $o
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/$name.so $out/obj_$name/*.o
rm -rf $out/obj_$name
echo built $out/$name.so
