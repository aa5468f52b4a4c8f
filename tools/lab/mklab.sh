#!/bin/bash
# tools/lab/mklab.sh <name> <source.hip> [-DFLAG ...]: libmogan_hip.so with ONE translation unit rebuilt with extra flags ->
# multiple-objects-gan_amd/build/lab_<name>.so (select it with MOGAN_LIB; the other objects come from the product build)
set -e
R=/root/repo/multiple-objects-gan_amd; name=$1; src=$2; shift 2
base=$(basename $src .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" -c $R/csrc/$base.hip -o /tmp/lab_$name.o 2> /tmp/lab_$name.err || { grep -i "error" /tmp/lab_$name.err; exit 1; }
objs=$(ls $R/build/*.o | grep -v "/$base.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/lab_$name.so $objs /tmp/lab_$name.o
echo $R/build/lab_$name.so
