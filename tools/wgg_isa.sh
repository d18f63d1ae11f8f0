#!/bin/bash
# ISA + register use of k_wgrad_group with only one main-tile variant compiled in:  tools/wgg_isa.sh <LAY> [extra flags] -> /tmp/wgg.s
cd /root/repo/ndjir_amd/csrc
lay=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -fno-slp-vectorize -DWGG_ONLY=$lay "$@" -S --cuda-device-only wgrad.hip -o /tmp/wgrad_only.s 2>/dev/null
awk '/^_ZN5ndjir13k_wgrad_groupEPKNS_8WggTableE:/,/^\.Lfunc_end.*k_wgrad_group/' /tmp/wgrad_only.s > /tmp/wgg.s
grep -A30 "^\s*\.amdhsa_kernel _ZN5ndjir13k_wgrad_groupEPKNS_8WggTableE" /tmp/wgrad_only.s | grep "next_free_vgpr\|next_free_sgpr\|private_segment_fixed_size" 
echo "scratch ops: $(grep -c scratch_ /tmp/wgg.s)  mfma: $(grep -c v_mfma /tmp/wgg.s)  waterfall(readfirstlane): $(grep -c v_readfirstlane /tmp/wgg.s)"
