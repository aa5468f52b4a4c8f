#!/bin/bash
# PMC passes over one deep layer of the packed-weight path: tools/pmc_pk.sh <cfg> <split> [layer] [dgrad]
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
run() { rm -rf /tmp/pp; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pp -o m -- python $R/tools/pmc_pk.py $ARGS > /tmp/pp.log 2>&1
  python - <<'PY'
import csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
names = {}
for r in csv.DictReader(open('/tmp/pp/m_kernel_trace.csv')):
    names[r['Dispatch_Id']] = (r['Kernel_Name'], int(r['End_Timestamp']) - int(r['Start_Timestamp']))
seen = set()
for r in csv.DictReader(open('/tmp/pp/m_counter_collection.csv')):
    k = names.get(r['Dispatch_Id'], (r.get('Kernel_Name', '?'), 0))
    if 'pgemm' not in k[0]: continue
    acc[k[0][:40]][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Dispatch_Id'] not in seen:
        seen.add(r['Dispatch_Id']); n[k[0][:40]] += 1; acc[k[0][:40]]['ns'] += k[1]
for k, v in acc.items():
    print(k, 'launches', n[k], {c: round(x / n[k], 1) for c, x in v.items()})
PY
}
ARGS="$*"
run SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
run TCC_HIT_sum TCC_MISS_sum SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
run FETCH_SIZE
