# secondary workloads only -> gpurun_out/<round>_family_bench_n1.jsonl (copy to profiles/ afterwards)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; RN=${ROUND:-r04}; cd $R
: > $O/${RN}_family_bench_n1.jsonl
for w in mnist clevr coco_s1 coco_s2; do python bench.py --workload $w --steps 20 --warmup 5 2>/dev/null | tail -1 >> $O/${RN}_family_bench_n1.jsonl; done
python - <<'PY'
import json, os
for l in open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", os.environ.get("ROUND", "r04") + "_family_bench_n1.jsonl")):
    x = json.loads(l)
    print(x["config"].get("workload")[:40], round(x["value"], 1))
PY
