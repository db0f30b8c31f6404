#!/bin/bash
# Register / scratch table of every kernel in a gfx950 code object or in libics_hip.so (offload bundles are unpacked into a scratch dir):
#   scripts/isa_table.sh [file]        default: image-cases-studies_amd/libics_hip.so
# Columns: kernel, scratch bytes, VGPRs, VGPR spills, SGPRs, LDS (static).  tests/test_isa.py asserts on the same notes.
set -e
F=${1:-image-cases-studies_amd/libics_hip.so}
T=$(mktemp -d)
cp "$F" $T/in.bin
cd $T
if head -c 24 in.bin | grep -q "__CLANG_OFFLOAD_BUNDLE__"; then   # hipcc -c --cuda-device-only writes a bare bundle
  /opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=in.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=x.gfx950
elif /opt/rocm/lib/llvm/bin/llvm-readelf -h in.bin 2>/dev/null | grep -q "AMDGPU"; then cp in.bin x.gfx950
else /opt/rocm/lib/llvm/bin/llvm-objdump --offloading in.bin > /dev/null; fi
for f in *gfx950; do
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes "$f" | grep -E "\.name:|\.vgpr_count|vgpr_spill|private_segment_fixed|\.sgpr_count:" | paste - - - - - |
    awk '{print $2, "scratch=" $4, "sgpr=" $6, "vgpr=" $8, "spill=" $10}'
done | c++filt | sed 's/(anonymous namespace):://; s/^void //' | sort
rm -rf $T
