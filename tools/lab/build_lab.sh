#!/bin/bash
# lab variant of libmogan_hip.so that differs in ONE source: $1 = name, $2 = source stem (e.g. mogan_pgemm), rest = extra flags
# -> tools/lab/libmogan_<name>.so (select with MOGAN_LIB)
set -e
R=$(cd $(dirname $0)/../.. && pwd); P=$R/multiple-objects-gan_amd
NAME=$1; SRC=$2; shift; shift
O=/tmp/lab_${NAME}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -Wno-unused-variable -Wno-unused-value "$@" -c $P/csrc/$SRC.hip -o $O
OBJS=$(ls $P/build/mogan_*.o | grep -v "$SRC.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/lab/libmogan_$NAME.so $OBJS $O
echo $R/tools/lab/libmogan_$NAME.so
