#!/bin/bash
# lab variants of libmogan_hip.so that differ in csrc/mogan_wino.hip only: $1 = name, rest = extra compiler flags
# -> tools/lab/libmogan_<name>.so (select with MOGAN_LIB)
set -e
R=$(cd $(dirname $0)/../.. && pwd); P=$R/multiple-objects-gan_amd
NAME=$1; shift
O=/tmp/wino_lab_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -Wno-unused-variable -Wno-unused-value "$@" -c $P/csrc/mogan_wino.hip -o $O
OBJS=$(ls $P/build/*.o | grep -v mogan_wino.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/lab/libmogan_$NAME.so $OBJS $O
echo $R/tools/lab/libmogan_$NAME.so
