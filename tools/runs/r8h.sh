#!/bin/bash
# PMC counters of wino_wgrad2_kernel (512 blocks) and wino_wgrad_kernel on 96 -> 192 at 128x128
R=/root/repo; O=$R/gpurun_out/r8h; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export MOGAN_WG2_BLOCKS=512
pass() { n=$1; shift; e=$1; shift
  env $e rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/p_$n -o w -- python $R/tools/pmc_wgrad.py > /tmp/p_$n.log 2>&1
  python $R/tools/pmc_agg.py /tmp/p_$n | grep wgrad >> $O/$n.txt 2>&1; }
for k in 1 0; do
  E="MOGAN_WG2=$k"
  pass k$k "$E" SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY
  pass k$k "$E" SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM
  pass k$k "$E" SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS
  pass k$k "$E" SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL TA_TA_BUSY_sum GRBM_GUI_ACTIVE
  pass k$k "$E" SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVES SQ_CYCLES SQ_THREAD_CYCLES_VALU SQ_IFETCH
done
