#!/bin/bash
# ISA of ONE instantiation of the pipelined 128-point-tile chain kernel (cross-compiled, no GPU): bash tools/isa_chainp.sh MODE [out.s] [extra flags]
# Prints register / spill counts and the scratch traffic of the kernel.
M=$1; OUT=${2:-/tmp/kp.s}; shift 2
HERE=$(cd "$(dirname "$0")/.." && pwd)
cat > /tmp/isa_chainp.hip <<EOT
#define NDJIR_NO_LAUNCHER
#include "$HERE/ndjir_amd/csrc/mlp3p.hip"
template __global__ void ndjir::x3p::k_chainp<$M>(ndjir::ChainArgs);
EOT
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -fno-slp-vectorize -I"$HERE/ndjir_amd/csrc" -S --cuda-device-only /tmp/isa_chainp.hip -o "$OUT" -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | grep -E "error|VGPRs:|AGPRs:|Spill|ScratchSize" | sed 's/.*remark: //'
echo "scratch ops: $(grep -c 'scratch_' "$OUT")   s_waitcnt vmcnt(0): $(grep -c 'vmcnt(0)' "$OUT")   lines: $(wc -l < "$OUT")  mfma: $(grep -c v_mfma "$OUT")"
