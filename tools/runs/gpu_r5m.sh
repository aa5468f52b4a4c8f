R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5m; mkdir -p $O; cd $R
run() { q=$1; ord=$2
  v=$(env MOGAN_STREAM_ORDER=$ord GPU_MAX_HW_QUEUES=$q python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.1f,%.2f' % (d['value'], d['ms_per_step']))")
  echo "q=$q,$ord,$v" | tee -a $O/queues.csv; }
: > $O/queues.csv
D="s2,s3,s1,wm,s0,gc,w2,w1,w0,cG,cD"
run 4 $D
for q in 5 6 8; do run $q $D; run $q "s2,s3,s1,s0,wm,gc"; run $q "s2,wm,s3,gc,s1,s0"; done
run 3 $D; run 3 "s2,s3,wm,s1,gc,s0"
