"""Diagnostic: coco_s2 generator gradients vs the reference fixture (tests/test_stackgan_gpu.py::test_networks), per parameter:
checksum error relative to the abs-sum.  Run with the default library and with MOGAN_LIB=multiple-objects-gan_amd/libmogan_hip_f32.so."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_stackgan_gpu as t
from helpers import probe
case = sys.argv[1] if len(sys.argv) > 1 else "coco_s2"
g = t.golden("stackgan_%s_nets" % case)
tree, stage, B, cfg, model, G, D = t.build(case)
b = t.to_device(t.synthetic.make_batch(tree, B, stage=stage, seed=21, text_dim=12), t.DEV)
b["z"] = b["z"].clone().requires_grad_(True)
fake, mu, logvar, ll, s1 = t.run_g(G, tree, stage, b)
loss = (fake * t.T("G.gimg", fake.shape).to(t.DEV)).sum()
if mu is not None:
    loss = loss + (mu * t.T("G.gmu", mu.shape).to(t.DEV)).sum() + (logvar * t.T("G.glv", logvar.shape).to(t.DEV)).sum()
loss.backward()
print("library:", os.environ.get("MOGAN_LIB", "default"))
rows = []
for k, p in G.named_parameters():
    key = "gg_" + k.replace(".", "__")
    if key in g.files:
        got, want = probe(p.grad), np.asarray(g[key], np.float64)
        rows.append((float(np.abs(got[:3] - want[:3]).max() / (abs(want[1]) + 1e-30)), k))
for e, k in sorted(rows, reverse=True)[:12]:
    print("%.3e  %s" % (e, k))
