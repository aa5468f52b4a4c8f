mkdir -p gpurun_out/r4d
L=gpurun_out/r4d/lab.log; rm -f $L
for v in "" noslp sched3 sched4 sched4ns; do
  echo "== variant '$v'" >> $L
  if [ -z "$v" ]; then python tools/time_wino.py 2>&1 | grep "96->192 128\|96-> 96 128\|96->192 64\|96-> 96 64\|192-> 96\|768->768" >> $L
  else MOGAN_LIB=$PWD/tools/lab/libmogan_$v.so python tools/time_wino.py 2>&1 | grep "96->192 128\|96-> 96 128\|96->192 64\|96-> 96 64\|192-> 96\|768->768" >> $L; fi
done
cat $L
bash tools/prof_stats.sh r4d/ks_single
python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | cut -c1-200
