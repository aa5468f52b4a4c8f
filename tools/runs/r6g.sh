#!/bin/bash
# PMC counters of the Winograd forward kernels in isolation (third form vs fourth form): where do the cycles go
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6g; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $O/avail.txt 2>&1
pass() { # name, lib-env, counters...
  n=$1; shift; e=$1; shift
  env $e rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/p_$n -o w -- python $R/tools/pmc_wino.py > /tmp/p_$n.log 2>&1
  python $R/tools/pmc_agg.py /tmp/p_$n > $O/$n.txt 2>&1
}
for k in old new; do
  if [ $k = old ]; then E="MOGAN_WINO4=0"; else E="MOGAN_WINO4=1"; fi
  E="$E MOGAN_LIB=$R/tools/lab/libmogan_w4s1.so"
  pass ${k}_a "$E" SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY
  pass ${k}_b "$E" SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM
  pass ${k}_c "$E" SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS
  pass ${k}_d "$E" SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL TA_TA_BUSY_sum TA_BUSY_avr
  pass ${k}_e "$E" TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCC_HIT_sum TCC_MISS_sum
done
ls /tmp/p_old_a > $O/ls.txt 2>&1; tail -5 /tmp/p_old_a.log >> $O/ls.txt
