#!/bin/bash
# ISA of ONE instantiation of the 128 / 64-point-tile chain kernel (cross-compiled, no GPU): bash tools/isa_one.sh MODE RPW NWAVES TM [out.s] [extra flags]
# e.g. bash tools/isa_one.sh 1 4 8 128 /tmp/k.s -> k_chainw<1, 4, 8, 128>;  1 2 4 64 -> the two-workgroups-per-CU kernel.
# Prints register / spill counts and the scratch traffic of the kernel.
M=$1; R=$2; W=$3; T=$4; OUT=${5:-/tmp/k.s}; shift 5
HERE=$(cd "$(dirname "$0")/.." && pwd)
cat > /tmp/isa_one.hip <<EOT
#define NDJIR_NO_LAUNCHER
#include "$HERE/ndjir_amd/csrc/mlp3w.hip"
template __global__ void ndjir::x3w::k_chainw<$M, $R, $W, $T>(ndjir::ChainArgs);
EOT
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -fno-slp-vectorize -I"$HERE/ndjir_amd/csrc" -S --cuda-device-only /tmp/isa_one.hip -o "$OUT" -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | grep -E "VGPRs:|Spill|ScratchSize" | sed 's/.*remark: //'
echo "scratch ops: $(grep -c 'scratch_' "$OUT")   s_waitcnt vmcnt(0): $(grep -c 'vmcnt(0)' "$OUT")   lines: $(wc -l < "$OUT")"
