#!/bin/bash
# builds tools/lab/libmogan_lab<N>.so = the product library with mogan_dconv.hip compiled under -DLAB=N
set -e
cd "$(dirname "$0")/../.."
P=multiple-objects-gan_amd
python $P/build.py > /dev/null
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DLAB=$v -c $P/csrc/mogan_dconv.hip -o /tmp/dconv_lab$v.o &
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DLAB=$v -c $P/csrc/mogan_gemm.hip -o /tmp/gemm_lab$v.o 2>/dev/null &
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/lab/libmogan_lab$v.so /tmp/dconv_lab$v.o $P/build/mogan_elem.o /tmp/gemm_lab$v.o $P/build/mogan_norm.o $P/build/mogan_stn_attn.o
done
