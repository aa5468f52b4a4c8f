#!/usr/bin/env python
"""Per-config numbers for the secondary BASELINE.json configs in one go: runs `bench.py --workload <name>` for each of
mnist / clevr / coco_s1 / coco_s2 and collects the JSON lines (see bench.py for the workload definitions).

    python tools/bench_family.py > profiles/r01_family_bench_n1.jsonl
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
names = sys.argv[1:] or ["mnist", "clevr", "coco_s1", "coco_s2"]
for n in names:
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", n, "--steps", "20", "--warmup", "5"],
                         capture_output=True, text=True)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    print(lines[-1] if lines else '{"workload": "%s", "error": %r}' % (n, out.stderr[-400:]), flush=True)
