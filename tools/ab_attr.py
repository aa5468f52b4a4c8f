"""A/B runs without environment switches: set module attributes of the package, then run a script in this process.

    python tools/ab_attr.py mogan_amd.hip.ops:WINO_PREP=False mogan_amd.attngan.miscc.losses:D_PAIR=True -- bench.py --steps 30

(the lost / lab settings that rounds 1-5 exposed as MOGAN_* environment variables are module attributes since round 6)"""
import ast
import importlib
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mogan_loader  # noqa: E402

mogan_loader.load()
i = sys.argv.index("--")
for spec in sys.argv[1:i]:
    target, value = spec.split("=", 1)
    mod, name = target.split(":")
    obj = importlib.import_module(mod)
    parts = name.split(".")                               # (module:Class.ATTR reaches a class attribute)
    for part in parts[:-1]:
        obj = getattr(obj, part)
    setattr(obj, parts[-1], ast.literal_eval(value))
script = sys.argv[i + 1]
sys.argv = sys.argv[i + 1:]
runpy.run_path(os.path.join(ROOT, script) if not os.path.isabs(script) else script, run_name="__main__")
