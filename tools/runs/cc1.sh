#!/bin/bash
# compile ONE csrc file with the build's flags, keep the ISA, print the register table: tools/cc1.sh mogan_pgemm
cd /root/repo/multiple-objects-gan_amd
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-variable -Wno-unused-value $MOGAN_CFLAGS -c csrc/$1.hip -o build/$1.o -save-temps=obj 2>&1 | grep -v "^$" | head -30
grep -E "\.(vgpr|agpr)_count|vgpr_spill_count|\.name:" build/$1-hip-amdgcn-amd-amdhsa-gfx950.s | paste - - - - | sed 's/_ZN12_GLOBAL__N_1//'
