#!/bin/bash
R=/root/repo; O=$R/gpurun_out/r7a; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for L in "1536 8 3072 4 2" "768 16 1536 4 2" "3072 4 1536 3 1"; do
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pk -o w -- python $R/tools/pmc_pk_layer.py $L > /tmp/pk.log 2>&1
  echo "== $L" >> $O/pk.txt; grep "packed path" /tmp/pk.log >> $O/pk.txt
  python - >> $O/pk.txt <<'PY'
import csv,glob,collections
f=glob.glob('/tmp/pk/*counter_collection.csv')[0]
a=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name']
    if 'pgemm_kernel' in n or 'apack' in n or 'splitk' in n or 'wpack' in n:
        a[n.split('(')[0][-40:]].append(float(r['Counter_Value']))
for k,v in a.items(): print(k, len(v), [round(x/1e3,1) for x in v[-4:]], 'MB')
PY
  rm -rf /tmp/pk
done
