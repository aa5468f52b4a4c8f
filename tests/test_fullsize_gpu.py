"""-m gpu: the hot path at BASELINE.json's FULL sizes (coco_train.yml widths, B = 16 per GPU), where the CPU oracle takes
half a minute per step and the golden fixtures do not reach.  Checked through size-independent properties:

  * every convolution family of the step at its real layer shapes: the three kernels of a layer (forward, data gradient,
    weight gradient -- Winograd, direct halo-tile, streaming or implicit GEMM, whichever the dispatch picks) must be each
    other's adjoints,  <conv(x; w), dy> = <x, dgrad(dy; w)> = <w, wgrad(dy, x)>,  and the forward must be linear in x;
  * one full-width train step: finite state, |delta p| of Adam's first step bounded by lr, the EMA recurrence
    ema' = 0.999 ema + 0.001 p', BatchNorm counters, tanh range of the images, and agreement of two engines started from
    the same state (eager multi-stream vs single-stream) on the losses.

fp32 tolerance: the inner products are sums of 1e7..1e9 products accumulated in fp64 on the host side of the check; the
kernels' own fp32 accumulation differs between algorithms (Winograd vs direct) by ~1e-6 relative per output, so the
adjoint identities are required to hold to 2e-4 relative to the product of the norms' scale.
"""
import numpy as np
import pytest
import torch

from helpers import load_pkg

load_pkg()
from mogan_amd.hip import ops  # noqa: E402
from mogan_amd.attngan import synthetic  # noqa: E402
from mogan_amd.attngan.miscc.config import cfg, set_coco_train_defaults  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda"
B = 16

# (name, Cin, H, W, Cout, k, stride, pad, up) at coco_train.yml widths (GF 48 -> ngf 96 after GLU, DF 96)
LAYERS = [
    ("ResBlock conv1 128x128 (Winograd F(2,3))", 96, 128, 128, 192, 3, 1, 1, 0),
    ("ResBlock conv2 64x64 (Winograd F(2,3))", 96, 64, 64, 96, 3, 1, 1, 0),
    ("upBlock 128->256 (up-conv identity, F(2,2) dgrad/wgrad)", 96, 128, 128, 96, 3, 1, 1, 1),
    ("GET_IMAGE_G 256x256 (streaming kernels)", 48, 256, 256, 3, 3, 1, 1, 0),
    ("D first conv 256->128 (streaming kernels)", 3, 256, 256, 96, 4, 2, 1, 0),
    ("D256 down 128->64 (F(2,2))", 96, 128, 128, 192, 4, 2, 1, 0),
    ("D256 down 64->32 (F(2,2))", 192, 64, 64, 384, 4, 2, 1, 0),
    ("D256 down 32->16 (F(2,2))", 384, 32, 32, 768, 4, 2, 1, 0),
    ("D256 down 16->8 (implicit GEMM, tuned)", 768, 16, 16, 1536, 4, 2, 1, 0),
    ("D256 down 8->4 (implicit GEMM, split-K)", 1536, 8, 8, 3072, 4, 2, 1, 0),
    ("D256 3x3 at 4x4 (implicit GEMM)", 3072, 4, 4, 1536, 3, 1, 1, 0),
    ("Inception Conv2d_4a 73->71 valid (Winograd, ragged)", 80, 73, 73, 192, 3, 1, 0, 0),
    ("Inception 1x7 at 17x17", 160, 17, 17, 192, (1, 7), 1, (0, 3), 0),
]


def _dot(a, b):
    return float((a.double().flatten() * b.double().flatten()).sum())


@pytest.mark.parametrize("layer", LAYERS, ids=[l[0] for l in LAYERS])
def test_conv_kernels_are_mutual_adjoints_at_full_size(layer):
    name, Cin, H, W, Cout, k, s, pad, up = layer
    kh, kw = (k, k) if isinstance(k, int) else k
    ph, pw = (pad, pad) if isinstance(pad, int) else pad
    gen = torch.Generator(device=DEV).manual_seed(hash(name) % 100003)
    x = torch.randn(B, Cin, H, W, device=DEV, generator=gen)
    x2 = torch.randn(B, Cin, H, W, device=DEV, generator=gen)
    w = torch.randn(Cout, Cin, kh, kw, device=DEV, generator=gen) * (1.0 / (Cin * kh * kw)) ** 0.5
    y = ops.conv2d_forward(x, w, s, ph, pw, up)
    dy = torch.randn(y.shape, device=DEV, generator=gen)
    dx = ops.conv2d_dgrad(dy, w, x.shape, s, ph, pw, up)     # (with a fused upsample: summed back to the source resolution)
    dw = ops.conv2d_wgrad(dy, x, w.shape, s, ph, pw, up)
    torch.cuda.synchronize()
    assert torch.isfinite(y).all() and torch.isfinite(dx).all() and torch.isfinite(dw).all()
    assert dx.shape == x.shape
    xin = x
    lhs = _dot(y, dy)
    scale = float(y.double().norm() * dy.double().norm())
    assert abs(lhs - _dot(xin, dx)) <= 2e-4 * scale, "forward / data-gradient adjoint"
    assert abs(lhs - _dot(w, dw)) <= 2e-4 * scale, "forward / weight-gradient adjoint"
    # linearity in x
    y2 = ops.conv2d_forward(x2, w, s, ph, pw, up)
    y12 = ops.conv2d_forward(1.5 * x - 0.25 * x2, w, s, ph, pw, up)
    err = float((y12 - (1.5 * y - 0.25 * y2)).abs().max())
    assert err <= 2e-5 * max(1.0, float(y.abs().max())), "linearity"


def _nonfinite_report(eng):
    out = []
    for n, o in zip(["G", "D64", "D128", "D256"], [eng.optG] + eng.optDs):
        for what, t in (("p", o.p), ("g", o.g), ("m", o.m), ("v", o.v)):
            bad = (~torch.isfinite(t)).nonzero().flatten()
            if bad.numel():
                out.append("%s.%s: %d of %d non-finite, first index %d, last %d" % (n, what, bad.numel(), t.numel(),
                                                                                   int(bad[0]), int(bad[-1])))
    return "; ".join(out) or "all finite"


def test_full_width_train_step_properties():
    from mogan_amd.attngan.trainer import TrainEngine, build_networks
    set_coco_train_defaults()
    cfg.TRAIN.GENERATOR_LR = cfg.TRAIN.DISCRIMINATOR_LR = 2e-4
    lr = 2e-4
    te, ie, G, Ds = build_networks(device=DEV, seed=4321)
    eng = TrainEngine(te, ie, G, Ds, use_graph=False)
    cpu = synthetic.make_batch(B, words_num=cfg.TEXT.WORDS_NUM, nef=cfg.TEXT.EMBEDDING_DIM, seed=11, text="tokens")
    bt = synthetic.to_device(cpu, DEV)
    bt["cap_lens_cpu"] = cpu["cap_lens"].clone()
    bt["cap_lens"] = bt["cap_lens"].to(torch.int32)
    p0 = [o.p.clone() for o in [eng.optG] + eng.optDs]
    ema0 = eng.optG.ema.clone()
    nbt0 = G.state_dict()["h_net1.fc.1.num_batches_tracked"].clone()
    state0 = eng._snapshot()
    logs = eng.step(dict(bt))
    torch.cuda.synchronize()
    for k in ("errD0", "errD1", "errD2", "errG", "kl", "w_loss", "s_loss"):
        v = float(logs[k])
        assert np.isfinite(v), k
    assert 0.0 < float(logs["errD2"]) < 20.0 and float(logs["kl"]) >= 0.0
    assert float(logs["fake64"].abs().max()) <= 1.0 + 1e-6                         # tanh range
    for o, before in zip([eng.optG] + eng.optDs, p0):
        d = (o.p - before).abs()
        assert torch.isfinite(o.p).all(), _nonfinite_report(eng)
        # Adam, first step, bias-corrected: |delta| = lr * |g| / (|g| + eps') <= lr
        assert float(d.max()) <= lr * (1 + 1e-3)
        assert float((d > 0.5 * lr).float().mean()) > 0.5                          # most weights did move by ~lr
    # EMA of the generator (trainer.py:341-342)
    want = 0.999 * ema0.double() + 0.001 * eng.optG.p.double()
    assert float((eng.optG.ema.double() - want).abs().max()) <= 1e-7
    assert int(G.state_dict()["h_net1.fc.1.num_batches_tracked"]) == int(nbt0) + 1
    # a second engine state: the same step on ONE stream must give the same losses (fp32 summation order of the
    # kernels is fixed; only the order of independent launches differs)
    eng._restore(state0)
    eng.multi_stream = False
    logs1 = eng.step(dict(bt))
    torch.cuda.synchronize()
    for k in ("errD0", "errD1", "errD2", "errG"):
        np.testing.assert_allclose(float(logs1[k]), float(logs[k]), rtol=2e-5, err_msg=k)


def _bench_module():
    import importlib
    return importlib.import_module("bench")          # ROOT is on sys.path (conftest); importing runs no benchmark


def test_headline_workload_one_step_against_the_oracle():
    """BASELINE config 5's per-GPU shard -- full coco_train.yml widths, B = 16 -- through the default launch mode (generator
    eager, discriminator branches replayed as hipGraphs, `inputs_ready` early start): after a few training steps (weights off
    their initial values, as after bench.py's timed region) ONE step of the HIP engine against ONE step of the CPU oracle from
    the same weights, z and eps.  This is bench.py's `parity` leg (bench.cpu_baseline_leg), here under -m gpu.
    Stated fp32 tolerance: every loss 1e-4 relative (errG carries the DAMSM terms weighted by 50), generated images 1e-3
    max-abs after the ~100-layer generator."""
    from mogan_amd.attngan.trainer import TrainEngine, build_networks
    bench = _bench_module()
    set_coco_train_defaults()
    cfg.TRAIN.BATCH_SIZE = B
    te, ie, G, Ds = build_networks(device=DEV, seed=1234)
    eng = TrainEngine(te, ie, G, Ds)
    assert eng.branch_graphs
    batch, bt_cpu = bench.make_device_batch(B, seed=0, device=torch.device(DEV))
    torch.cuda.synchronize()
    batch["inputs_ready"] = torch.cuda.Event()
    batch["inputs_ready"].record()
    gen = torch.Generator(device=DEV).manual_seed(1000)
    for _ in range(4):
        b = dict(batch)
        b["z"] = torch.randn(B, cfg.GAN.Z_DIM, device=DEV, generator=gen)
        b["eps"] = torch.randn(B, cfg.GAN.CONDITION_DIM, device=DEV, generator=gen)
        eng.prefetch_text(batch["captions"], batch["cap_lens_cpu"])
        eng.step(b)
    torch.cuda.synchronize()
    dev_batch = dict(batch)
    dev_batch.update(eng.encode_batch_for_cpu(batch))
    base, parity = bench.cpu_baseline_leg(eng, bt_cpu, dev_batch, B, torch.device(DEV), timed_steps=0)
    print("parity", {k: v for k, v in parity.items() if k not in ("hip", "oracle", "what")})
    for k in ("errD0", "errD1", "errD2", "errG", "w_loss", "s_loss"):
        assert parity[k + "_rel"] <= 1e-4, (k, parity[k + "_rel"], parity["hip"], parity["oracle"])
    assert parity["kl_rel"] <= 1e-4
    assert parity["img64_max_abs"] <= 1e-3 and parity["img256_max_abs"] <= 1e-3
    assert parity["ok"]


@pytest.mark.parametrize("name", ["clevr", "coco_s2"])
def test_secondary_workloads_one_step_against_the_oracle_at_their_stated_size(name):
    """BASELINE configs 2 and 3 at the size they are quoted on (CLEVR B = 32; MS-COCO StackGAN stage II 256x256 B = 24, yml
    widths): `bench.py --workload <name>` -- a few timed steps, then its `parity` object: one HIP step against one step of
    oracle/stackgan_oracle.py from the benchmarked weights on the same batch / z / eps.  Losses 1e-4 relative, images 1e-3."""
    import argparse
    bench = _bench_module()
    args = argparse.Namespace(family_batch=None, graph=False, no_graph=False, warmup=2, steps=3, no_roofline=True,
                              no_cpu_baseline=False)
    real = bench.family_cpu_baseline
    bench.family_cpu_baseline = lambda *a, **k: real(*a, **dict(k, timed_steps=0))     # parity step only
    try:
        out = bench.run_family(name, args, torch.device(DEV))
    finally:
        bench.family_cpu_baseline = real
    par = out["parity"]
    print(name, {k: v for k, v in par.items() if k not in ("what",)})
    assert out["config"]["batch_per_gpu"] == {"clevr": 32, "coco_s2": 24}[name]
    assert par["ok"], par
