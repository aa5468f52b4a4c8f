mkdir -p gpurun_out/r4q
out=gpurun_out/r4q/touch.csv; : > $out
run() { v=$(env "$@" python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.1f,%.2f' % (d['value'], d['ms_per_step']))"); echo "$*,$v" | tee -a $out; }
run A=default
run MOGAN_TOUCH_ORDER=c,b,a,e,w
run MOGAN_TOUCH_ORDER=c,b,a,w,e
run MOGAN_TOUCH_ORDER=a,b,c,e,w
run MOGAN_TOUCH_ORDER=e,c,b,a,w
run MOGAN_TOUCH_ORDER=c,e,b,a,w
run MOGAN_TOUCH_ORDER=x,c,b,a,e,w
run MOGAN_TOUCH_ORDER=c,b,a,x,e,w
run MOGAN_TOUCH_ORDER=w,c,b,a,e
run A=default
