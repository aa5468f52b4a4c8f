#!/bin/bash
# tools/lab/libmogan_f32.so = the product library with every MFMA kernel in its native fp32-MFMA form (-DMOGAN_X6=0);
# select it with MOGAN_LIB=$PWD/tools/lab/libmogan_f32.so for A/B runs against the split-bf16 default.
set -e
cd "$(dirname "$0")/../.."
P=multiple-objects-gan_amd
python $P/build.py > /dev/null
mkdir -p /tmp/mogan_f32
for f in $P/csrc/*.hip; do
  b=$(basename $f .hip)
  if grep -q mogan_mma.h $f; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DMOGAN_X6=0 -c $f -o /tmp/mogan_f32/$b.o 2>/dev/null &
  else
    cp $P/build/$b.o /tmp/mogan_f32/$b.o
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/lab/libmogan_f32.so /tmp/mogan_f32/*.o
echo tools/lab/libmogan_f32.so
