#!/bin/bash
# One PSF size of the matrix-core convolution, device code only, with the register table: scripts/isa_one.sh 9 [-DICS_EPI_TB...]
K=$1; shift
cd "$(dirname "$0")/../image-cases-studies_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -pragma-unroll-threshold=200000 -DICS_MFMA_PART=0 -DICS_MFMA_ONLY_K=$K "$@" --cuda-device-only -c ics_conv_mfma.hip -o /tmp/isa_one_$K.co && ../../scripts/isa_table.sh /tmp/isa_one_$K.co
