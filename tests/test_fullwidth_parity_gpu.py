"""-m gpu: VALUE parity at coco_train.yml widths (GF_DIM 48 -> 96 channels, DF_DIM 96), i.e. on the kernels bench.py times
(Winograd F(2,3) / F(2,2), direct halo-tile, up-conv identity, tuned implicit GEMM) -- not only on the reduced-width
fixtures, where every convolution falls to the generic kernel.

  * blocks at B = 16 against fixtures the REFERENCE produced at full width (tests/golden/make_golden_fullwidth.py):
    ResBlock(96) 64x64, upBlock(96,48) 128->256, downBlock(384,768) 32->16 -- forward, dx, every dW, BN running stats;
  * G_NET forward and discriminator_loss through D_NET256 + backward at B = 4 against the reference fixture (sampled
    elements + sums of every tensor) AND, full tensor for full tensor, against the CPU oracle run on this host;
  * RNN_ENCODER against the reference fixture;
  * every GEMM-backed launch geometry of one full-width B = 16 train step: the tuned (tile, split-K) table entry against
    the heuristic's choice at that exact shape;
  * 50 fresh engines in ONE process (stream pool cycling, a graph capture per engine), every first step finite and equal.

Stated fp32 tolerances (SURVEY.md section 8(c) envelope; measured values are printed by each test):
  forward tensors (images, attention, block outputs)   max-abs <= 1e-4 (values O(1))
  loss                                                 rel <= 1e-5
  dx / dW of a single block                            rel-L2 <= 1e-4 per tensor (BN gamma/beta: 1e-3)
  D_NET256 gradients (8 LeakyReLU layers deep)         rel-L2 <= 2e-5 per tensor against the fp64 oracle evaluated with the
                                                       same LeakyReLU sign decisions; <= 5e-3 against the fp32 fixture
                                                       (kink allowance, see the test)
  BN running statistics                                max-abs <= 1e-5
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import GOLDEN, big_probe_close, det_array, det_fill_state, load_pkg, max_abs, rel_l2

load_pkg()
from mogan_amd.attngan import synthetic  # noqa: E402
from mogan_amd.attngan.miscc.config import cfg, set_coco_train_defaults  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda"


def T(name, shape, scale=1.0, shift=0.0):
    return torch.from_numpy(det_array(name, shape, scale, shift))


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(autouse=True)
def full_cfg():
    set_coco_train_defaults()
    cfg.TRAIN.GENERATOR_LR = cfg.TRAIN.DISCRIMINATOR_LR = 2e-4
    cfg.STN_ALIGN_CORNERS, cfg.ATT_MASK_MODE, cfg.ADAM_EPS_MODE = False, 0, 0
    yield


BLOCKS = {"res": (lambda m: m.ResBlock(96), (16, 96, 64, 64), (16, 96, 64, 64)),
          "up": (lambda m: m.upBlock(96, 48), (16, 96, 128, 128), (16, 48, 256, 256)),
          "down": (lambda m: m.downBlock(384, 768), (16, 384, 32, 32), (16, 768, 16, 16))}


@pytest.mark.parametrize("tag", list(BLOCKS))
def test_full_width_block_vs_reference_fixture(tag):
    """model.py:67-81 (ResBlock), 48-55 (upBlock), 594-602 (downBlock) at the widths and batch of the benchmark."""
    from mogan_amd.attngan import model
    g = golden("fw_blocks")
    make, xs, gs = BLOCKS[tag]
    mod = make(model)
    det_fill_state(mod, "fw.%s." % tag)
    mod = mod.to(DEV).train()
    x = T("fw.%s.x" % tag, xs).to(DEV).requires_grad_(True)
    y = mod(x)
    y.backward(T("fw.%s.g" % tag, gs).to(DEV))
    torch.cuda.synchronize()
    big_probe_close(y, g[tag + "_y"], tol_abs=1e-4, what=tag + " y")
    # downBlock ends in a LeakyReLU: one pre-activation within rounding of zero whose sign comes out differently than in
    # the reference's fp32 run moves dx / dW by ~2e-4 (tools/diag_block_kinks.py: one such element of 3.1 M with the
    # split-bf16 MFMA form, none with the native fp32 form).  The fixture comparison therefore carries a kink allowance,
    # and the values are pinned to 2e-6 against fp64 with THIS run's sign decisions imposed (below).
    kink = 10.0 if tag == "down" else 1.0
    big_probe_close(x.grad, g[tag + "_dx"], tol_rel_l2=1e-4 * kink, what=tag + " dx")
    for k, p in mod.named_parameters():
        tol = 1e-3 if p.dim() == 1 else 1e-4 * kink
        big_probe_close(p.grad, g["%s_d_%s" % (tag, k.replace(".", "__"))], tol_rel_l2=tol, what="%s d%s" % (tag, k))
    if tag == "down":
        sd = {k: v.detach().cpu().double() for k, v in mod.state_dict().items()}
        xd = x.detach().cpu().double().requires_grad_(True)
        wd = sd["0.weight"].clone().requires_grad_(True)
        # (the running statistics in sd are the updated ones: batch_norm below runs in training mode and ignores them)
        t = F.batch_norm(F.conv2d(xd, wd, None, 2, 1), None, None, sd["1.weight"], sd["1.bias"], True, 0.1, 1e-5)
        mask = y.detach().cpu() > 0
        flips = int((mask != (t.detach() > 0)).sum())
        assert flips <= 8, "%d LeakyReLU decisions differ from fp64" % flips
        torch.where(mask, t, 0.2 * t).backward(T("fw.%s.g" % tag, gs).double())
        assert rel_l2(x.grad, xd.grad) < 2e-6 and rel_l2(mod[0].weight.grad, wd.grad) < 2e-6, \
            (rel_l2(x.grad, xd.grad), rel_l2(mod[0].weight.grad, wd.grad))
    for k, v in mod.state_dict().items():
        if "running" in k:
            big_probe_close(v, g["%s_s_%s" % (tag, k.replace(".", "__"))], tol_abs=1e-5, what="%s %s" % (tag, k))


def test_full_width_gnet_forward_and_dnet256_loss_backward():
    """model.py:478-528, 738-760 and miscc/losses.py:136-174 at full width, B = 4: against the reference's fixture
    (sampled) and against the oracle (every element)."""
    from mogan_amd.attngan import model
    from mogan_amd.attngan.miscc import losses as L
    from mogan_amd.hip import ops
    from oracle import attngan_oracle as O
    g = golden("fw_nets")
    B = 4
    cpu = synthetic.make_batch(B, words_num=12, nef=256, seed=21)
    bt = synthetic.to_device(cpu, DEV)
    G = model.G_NET()
    sdg = det_fill_state(G, "G.")
    G = G.to(DEV).train()
    with torch.no_grad():
        imgs, atts, mu, logvar = G(bt["z"], bt["sent_emb"], bt["words_embs"], bt["mask"], bt["tmi"], bt["label_one_hot"],
                                   eps=bt["eps"])
    D = model.D_NET256()
    sdd = det_fill_state(D, "D2.")
    D = D.to(DEV).train()
    ops.ACT_TRACE = []                       # the LeakyReLU outputs of this pass, in launch order (hip/ops.py)
    try:
        errD = L.discriminator_loss(D, bt["imgs"][2], imgs[2], bt["sent_emb"], None, None, None)
    finally:
        trace, ops.ACT_TRACE = ops.ACT_TRACE, None
    errD.backward()
    torch.cuda.synchronize()
    # (1) the reference fixture (fp32 torch-CPU).  Forward tensors and the loss: tight.  Gradients: a LeakyReLU whose
    # pre-activation lies within fp32 noise of 0 may decide differently here than there (measured on this very case: 1 of
    # 3.1 M decisions behind BN3 differs from an fp64 run, |t| = 3e-7, and moves the three tensors below it by 1e-4..2e-3
    # while everything above agrees to 2e-6) -- so against the fixed fp32 fixture the gradients get a kink allowance,
    # and the tight gradient check is (2), where the oracle is handed the decisions made here.
    for k, t in (("img64", imgs[0]), ("img128", imgs[1]), ("img256", imgs[2]), ("att64", atts[0]), ("att128", atts[1]),
                 ("mu", mu), ("logvar", logvar)):
        big_probe_close(t, g[k], tol_abs=1e-4, what=k)
    for k, v in G.state_dict().items():
        key = "g_s_" + k.replace(".", "__")
        if key in g.files:
            big_probe_close(v, g[key], tol_abs=1e-5, what=k)
    want = float(g["errD2"][0])
    assert abs(float(errD.detach()) - want) <= 1e-5 * abs(want), ("errD2", float(errD.detach()), want)
    for k, p in D.named_parameters():
        big_probe_close(p.grad, g["d2_g_" + k.replace(".", "__")], tol_rel_l2=5e-3, what="D256 d" + k)
    for k, v in D.state_dict().items():
        if "running" in k:
            big_probe_close(v, g["d2_s_" + k.replace(".", "__")], tol_abs=1e-5, what="D256 " + k)
    # (2) the oracle in fp64 with this pass's LeakyReLU decisions, every element of every tensor
    dt = torch.float64
    ocfg = O.Cfg()
    og, od = O.from_state_dict(sdg, dtype=dt, requires_grad=False), O.from_state_dict(sdd, dtype=dt)
    c64 = {k: (v.to(dt) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in cpu.items()}
    c64["imgs"] = [t.to(dt) for t in cpu["imgs"]]
    with torch.no_grad():
        oimgs, oatts, omu, olv, _ = O.g_net(og, ocfg, c64["z"], c64["sent_emb"], c64["words_embs"], c64["mask"],
                                            c64["tmi"], c64["label_one_hot"], c64["eps"])
    rep = {}
    for k, a, b in [("img%d" % (64 << i), imgs[i], oimgs[i]) for i in range(3)] + \
                   [("att%d" % (64 << i), atts[i], oatts[i]) for i in range(2)]:
        rep[k] = max_abs(a, b)
        assert rep[k] <= 1e-4, "%s differs from the fp64 oracle by %.3e" % (k, rep[k])
    O.LRELU_MASKS = [(t > 0).cpu() for a, t in trace if a == ops.ACT_LRELU]
    O.LRELU_FLIPS = []
    try:
        # the same fake image as the HIP pass (so that the decisions belong to the same function)
        oerr = O.discriminator_loss(2, od, c64["imgs"][2], imgs[2].detach().cpu().to(dt), c64["sent_emb"], c64, ocfg)
        assert len(O.LRELU_MASKS) == 0, "%d LeakyReLU launches were not consumed by the oracle" % len(O.LRELU_MASKS)
        flips = O.LRELU_FLIPS
    finally:
        O.LRELU_MASKS = O.LRELU_FLIPS = None
    # the imposed decisions may differ from the oracle's own only at pre-activations within fp32 rounding of zero: a handful
    # of the ~2e7 decisions of the two passes, each at |x| <= 1e-5 (a wrong mask would otherwise be inherited silently)
    nflip, ndec = sum(f[0] for f in flips), sum(f[1] for f in flips)
    assert nflip <= 16 and max(f[2] for f in flips) <= 1e-5, \
        "%d of %d imposed LeakyReLU decisions differ from the fp64 oracle's (largest |x| %.3e)" % (
            nflip, ndec, max(f[2] for f in flips))
    oerr.backward()
    assert abs(float(errD.detach()) - float(oerr.detach())) <= 1e-5 * abs(float(oerr.detach()))
    worst = (0.0, "")
    for k, p in D.named_parameters():
        r = rel_l2(p.grad, od[k].grad)
        worst = max(worst, (r, k))
        assert r <= 2e-5, "D256 d%s: rel-L2 %.3e vs the fp64 oracle (same LeakyReLU decisions)" % (k, r)
    print("full-width parity: max-abs vs fp64 oracle %s; errD2 %.7f (reference %.7f, oracle %.7f); worst D256 grad rel-L2 "
          "%.2e (%s), %d LeakyReLU launches matched" % ({k: "%.1e" % v for k, v in rep.items()}, float(errD.detach()), want,
                                                       float(oerr.detach()), worst[0], worst[1], len(trace)))


def test_full_width_gnet_backward_every_gradient_against_the_fp64_oracle():
    """model.py:478-528 at coco_train.yml widths, B = 4: the generator's BACKWARD -- the Winograd F(2,3) data-gradient /
    weight-gradient chains of the ResBlocks at 64x64 and 128x128, the up-conv identity kernels, the grouped BatchNorm backward of
    the object pathway, the attention backward at 64x64 / 128x128 -- element for element against the fp64 oracle (which
    tests/test_oracle_golden.py pins to the reference's fixtures), for a loss that is a fixed random linear functional of all
    three images, mu and logvar, with the LeakyReLU decisions of the checked pass handed to the oracle (BBOX_NET; see above).
    Stated tolerance: every G gradient tensor rel-L2 <= 1e-2 -- the stacked BN+GLU generator is
    ill-conditioned in fp32, torch-fp32 itself is at 1.7e-3..3.2e-3 from fp64 there; the measured value is printed."""
    from mogan_amd.attngan import model
    from mogan_amd.hip import ops
    from oracle import attngan_oracle as O
    B, dt = 4, torch.float64
    cpu = synthetic.make_batch(B, words_num=12, nef=256, seed=23)
    bt = synthetic.to_device(cpu, DEV)
    G = model.G_NET()
    sdg = det_fill_state(G, "G.")
    G = G.to(DEV).train()
    ops.ACT_TRACE = []
    try:
        imgs, atts, mu, logvar = G(bt["z"], bt["sent_emb"], bt["words_embs"], bt["mask"], bt["tmi"], bt["label_one_hot"],
                                   eps=bt["eps"])
    finally:
        trace, ops.ACT_TRACE = ops.ACT_TRACE, None
    # cotangents scaled so that the three image terms weigh alike (sum over 3 * H * W * B elements each)
    ups = [T("FW.gimg%d" % i, im.shape, 1.0 / float(im[0].numel()) ** 0.5) for i, im in enumerate(imgs)]
    ups += [T("FW.gmu", mu.shape, 0.1), T("FW.glv", logvar.shape, 0.1)]
    with ops.wgrad_overlap():
        sum((t * u.to(DEV)).sum() for t, u in zip(list(imgs) + [mu, logvar], ups)).backward()
    torch.cuda.synchronize()
    og = O.from_state_dict(sdg, dtype=dt)
    c64 = {k: (v.to(dt) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in cpu.items()}
    O.LRELU_MASKS = [(t > 0).cpu() for a, t in trace if a == ops.ACT_LRELU]
    O.LRELU_FLIPS = []
    try:
        oimgs, oatts, omu, olv, _ = O.g_net(og, O.Cfg(), c64["z"], c64["sent_emb"], c64["words_embs"], c64["mask"],
                                            c64["tmi"], c64["label_one_hot"], c64["eps"])
        assert not O.LRELU_MASKS
        flips = O.LRELU_FLIPS
    finally:
        O.LRELU_MASKS = O.LRELU_FLIPS = None
    assert sum(f[0] for f in flips) <= 4, flips
    for k, a, b in [("img%d" % (64 << i), imgs[i], oimgs[i]) for i in range(3)]:
        assert max_abs(a, b) <= 1e-4, (k, max_abs(a, b))
    sum((t * u.to(dt)).sum() for t, u in zip(list(oimgs) + [omu, olv], ups)).backward()
    worst, rows = (0.0, ""), []
    for k, p in G.named_parameters():
        assert p.grad is not None and og[k].grad is not None, k
        r = rel_l2(p.grad, og[k].grad)
        rows.append((r, k))
        worst = max(worst, (r, k))
    rows.sort(reverse=True)
    print("full-width G_NET backward vs fp64 oracle, B = %d: worst rel-L2 %.2e (%s); top 5: %s" % (
        B, worst[0], worst[1], ["%s %.1e" % (k, r) for r, k in rows[:5]]))
    for r, k in rows:
        assert r <= 1e-2, "G d%s: rel-L2 %.3e vs the fp64 oracle" % (k, r)


def test_rnn_encoder_vs_reference_fixture():
    """model.py:120-204: embedding + packed bidirectional LSTM (stock nn.LSTM on the device)."""
    import sys
    sys.path.insert(0, GOLDEN)
    from mogan_amd.attngan import model
    g = golden("rnn")
    rng = np.random.RandomState(77)                      # = make_golden_fullwidth.rnn_inputs (no reference import here)
    lens = np.array([12, 11, 9, 9, 6, 5])
    cap = np.zeros((6, 12), dtype=np.int64)
    for b, n in enumerate(lens):
        cap[b, :n] = rng.randint(1, 500, size=n)
    cfg.RNN_TYPE = 'LSTM'
    enc = model.RNN_ENCODER(500, nhidden=256)
    sd = det_fill_state(enc, "RNN.")
    assert sorted(sd.keys()) == list(g["keys"])
    enc = enc.to(DEV).eval()
    with torch.no_grad():
        words, sent = enc(torch.from_numpy(cap).to(DEV), torch.from_numpy(lens), enc.init_hidden(6))
    assert tuple(words.shape) == g["words"].shape and tuple(sent.shape) == g["sent"].shape
    np.testing.assert_allclose(words.cpu().numpy(), g["words"], atol=2e-5, rtol=1e-4)
    np.testing.assert_allclose(sent.cpu().numpy(), g["sent"], atol=2e-5, rtol=1e-4)


# ------------------------------------------------------------------------------------------- tuned table
_GEMM_ENTRIES = ("mogan_conv2d_fwd", "mogan_conv2d_affine_fwd", "mogan_conv2d_dgrad", "mogan_conv2d_wgrad",
                 "mogan_upconv3x3_fwd", "mogan_upconv3x3_dgrad", "mogan_upconv3x3_wgrad", "mogan_bmm")


# positions of the integer (shape) arguments of each entry point (include/mogan_hip.h)
_INT_ARGS = {"mogan_conv2d_fwd": (3, 14), "mogan_conv2d_affine_fwd": (5, 16), "mogan_conv2d_dgrad": (3, 14),
             "mogan_conv2d_wgrad": (3, 15), "mogan_upconv3x3_fwd": (3, 8), "mogan_upconv3x3_dgrad": (3, 8),
             "mogan_upconv3x3_wgrad": (3, 9), "mogan_bmm": (3, 17)}


def _record_geometries(step_fn):
    """Run step_fn with hip.ops.call wrapped: the set of (entry point, integer arguments) of its GEMM-backed launches."""
    from mogan_amd.hip import ops
    seen, orig = {}, ops.call

    def spy(name, *args):
        if name in _GEMM_ENTRIES:
            lo, hi = _INT_ARGS[name]
            seen.setdefault((name, tuple(int(a) for a in args[lo:hi])), None)
        return orig(name, *args)

    ops.call = spy
    try:
        step_fn()
        torch.cuda.synchronize()
    finally:
        ops.call = orig
    return list(seen)


def _replay(name, ints, gen):
    """Re-issue one recorded launch on fresh random operands; returns the output tensor."""
    from mogan_amd.hip import ops
    from mogan_amd.hip.lib import call, ptr, stream_ptr, workspace
    rn = lambda *s: torch.randn(*s, device=DEV, generator=gen)
    wsp, wsn = workspace(torch.device(DEV, torch.cuda.current_device()))
    if name == "mogan_bmm":
        Z, M, N, K = ints[:4]
        sa, sb, so = ints[4:7], ints[7:10], ints[10:13]
        acc = ints[13]
        size = lambda dims, st: 1 + sum((d - 1) * s for d, s in zip(dims, st))
        a, b = rn(size((Z, M, K), sa)), rn(size((Z, K, N), sb))
        out = torch.zeros(size((Z, M, N), so), device=DEV)
        call(name, ptr(a), ptr(b), ptr(out), Z, M, N, K, *sa, *sb, *so, acc, wsp, wsn, stream_ptr())
        return out
    if name.startswith("mogan_upconv3x3"):
        B, Cin, Hs, Ws, Cout = ints[:5]
        x, w, y = rn(B, Cin, Hs, Ws), rn(Cout, Cin, 3, 3) * 0.05, rn(B, Cout, 2 * Hs, 2 * Ws)
        if name.endswith("_fwd"):
            return ops.conv2d_forward(x, w, 1, 1, 1, 1)
        if name.endswith("_dgrad"):
            return ops.conv2d_dgrad(y, w, x.shape, 1, 1, 1, 1)
        return ops.conv2d_wgrad(y, x, w.shape, 1, 1, 1, 1)
    B, Cin, Hs, Ws, Cout, KH, KW, stride, ph, pw = ints[:10]
    up = ints[10] if name != "mogan_conv2d_affine_fwd" else 0
    x, w = rn(B, Cin, Hs, Ws), rn(Cout, Cin, KH, KW) * (1.0 / (Cin * KH * KW)) ** 0.5
    if name == "mogan_conv2d_fwd":
        return ops.conv2d_forward(x, w, stride, ph, pw, up)
    if name == "mogan_conv2d_affine_fwd":
        return ops.conv2d_affine_relu(x, w, rn(Cout).abs() + 0.5, rn(Cout), stride, (ph, pw))
    OH, OW = ops.conv_out_hw(Hs, Ws, KH, KW, stride, ph, pw, up)
    dy = rn(B, Cout, OH, OW)
    if name == "mogan_conv2d_dgrad":
        return ops.conv2d_dgrad(dy, w, x.shape, stride, ph, pw, up)
    return ops.conv2d_wgrad(dy, x, w.shape, stride, ph, pw, up)


def _bench_engine(B, fast_init=True):
    from mogan_amd.attngan.trainer import TrainEngine, build_networks
    if fast_init:
        os.environ["MOGAN_FAST_INIT"] = "1"
    try:
        te, ie, G, Ds = build_networks(device=DEV, seed=4321)
    finally:
        os.environ.pop("MOGAN_FAST_INIT", None)
    eng = TrainEngine(te, ie, G, Ds, use_graph=False)
    cpu = synthetic.make_batch(B, words_num=cfg.TEXT.WORDS_NUM, nef=cfg.TEXT.EMBEDDING_DIM, seed=11, text="tokens")
    bt = synthetic.to_device(cpu, DEV)
    bt["cap_lens_cpu"] = cpu["cap_lens"].clone()
    bt["cap_lens"] = bt["cap_lens"].to(torch.int32)
    return eng, bt


def test_tuned_table_entries_match_the_heuristic_at_every_bench_shape():
    """The 500+ (tile config, split-K) entries of hip/tuned_gemm_gfx950.csv are keyed on the exact GEMM dims of the
    benchmark's layers.  For every GEMM-backed launch geometry of one full-width B = 16 step (eager, single stream, the
    Inception trunk launched eagerly so that its launches are seen) the result with the table registered must equal the
    result of the heuristic's choice: same products, only the K-split summation order differs (rel-L2 <= 1e-5)."""
    from mogan_amd.attngan import inception
    from mogan_amd.hip import lib
    eng, bt = _bench_engine(16)
    eng.multi_stream, eng.graph_encoder = False, False
    # the frozen encoder module by module for the recording: its explicit forward/backward (inception.FrozenTrunk) goes to the
    # same implicit-GEMM kernel with the same (M, N, K) keys for every convolution it does not group, through entry points
    # this replay does not decode
    fast, inception.FAST_TRUNK = inception.FAST_TRUNK, False
    try:
        geos = _record_geometries(lambda: eng.step(dict(bt)))
    finally:
        inception.FAST_TRUNK = fast
    del eng
    torch.cuda.empty_cache()
    assert len(geos) > 150, len(geos)
    L = lib.load()
    worst, n_diff = (0.0, None), 0
    try:
        for name, ints in geos:
            outs = []
            for tuned in (True, False):
                L.mogan_gemm_tune_clear()
                if tuned:
                    assert lib._register_tuned(L) > 0
                outs.append(_replay(name, ints, torch.Generator(device=DEV).manual_seed(7)))
            torch.cuda.synchronize()
            assert torch.isfinite(outs[0]).all(), (name, ints)
            r = rel_l2(outs[0], outs[1])
            n_diff += int(r > 0)
            if r > worst[0]:
                worst = (r, (name, ints))
            assert r <= 1e-5, "tuned vs heuristic differ by rel-L2 %.3e at %s %s" % (r, name, ints)
    finally:
        L.mogan_gemm_tune_clear()
        lib._register_tuned(L)
    print("tuned table: %d launch geometries, %d with a different summation order, worst rel-L2 %.2e at %s"
          % (len(geos), n_diff, worst[0], worst[1]))


def test_fifty_fresh_engines_in_one_process():
    """Round 1 saw (once in seven suite runs) garbage generator gradients in the first full-width step of a long-lived
    process: scratch buffers were keyed by the raw stream handle, and torch's 32-entry stream pool hands the capture
    stream's handle to a later engine's branch stream.  The fix (hip/lib.py: capture-session workspaces) is exercised
    here without any retry: 50 engines built one after the other in this process (each creates 9 streams -> the pool
    wraps every 4th engine; each captures its own encoder graph pair), every first step must be finite and give the
    same losses as the first engine's (same seed, same batch)."""
    n = int(os.environ.get("MOGAN_STRESS_ENGINES", "50"))
    ref = None
    for it in range(n):
        eng, bt = _bench_engine(4)
        logs = eng.step(dict(bt))
        torch.cuda.synchronize()
        vals = {k: float(logs[k]) for k in ("errD0", "errD1", "errD2", "errG", "kl", "w_loss", "s_loss")}
        for o, nm in zip([eng.optG] + eng.optDs, ("G", "D64", "D128", "D256")):
            assert torch.isfinite(o.g).all() and torch.isfinite(o.p).all(), "engine %d: non-finite %s bucket" % (it, nm)
        assert all(np.isfinite(v) for v in vals.values()), (it, vals)
        if ref is None:
            ref = vals
        for k, v in vals.items():
            assert abs(v - ref[k]) <= 1e-4 * abs(ref[k]) + 1e-6, "engine %d: %s %.7f vs %.7f" % (it, k, v, ref[k])
        del eng, logs
        torch.cuda.empty_cache()
    print("stress: %d fresh engines, losses %s" % (n, {k: round(v, 5) for k, v in ref.items()}))


# (Cin, H, Cout, k, stride, pad): the convolutions that carry the step's FLOPs, at their real spatial sizes
FULL_SIZE_CONVS = [(3, 256, 96, 4, 2, 1), (96, 128, 192, 4, 2, 1), (192, 64, 384, 4, 2, 1), (384, 32, 768, 4, 2, 1),
                   (768, 16, 1536, 4, 2, 1), (96, 64, 192, 3, 1, 1), (96, 128, 96, 3, 1, 1), (48, 256, 3, 3, 1, 1)]


@pytest.mark.parametrize("B", [4, 16])
@pytest.mark.parametrize("layer", FULL_SIZE_CONVS, ids=lambda l: "%d-%d@%d-k%ds%d" % (l[0], l[2], l[1], l[3], l[4]))
def test_full_size_convolution_values_vs_fp64(layer, B):
    """Forward, data gradient and weight gradient of one convolution at its benchmark shape (both the B = 4 shard of
    BASELINE config 4 and B = 16) against torch-CPU fp64 -- every element, rel-L2 <= 5e-6 (measured 1e-7..1e-6): the value
    check that the adjoint identities of test_fullsize_gpu.py cannot give (a consistently wrong operator satisfies them)."""
    import torch.nn.functional as F
    from mogan_amd.hip import ops
    Cin, H, Cout, k, s, p = layer
    if B == 16 and Cin * Cout * k * k * (H // s) ** 2 > 3e9:
        pytest.skip("fp64 CPU reference of this layer at B=16 takes > 20 s; B=4 covers the same kernels")
    g = torch.Generator().manual_seed(Cin * 7 + H + B)
    x = torch.randn(B, Cin, H, H, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) * (1.0 / (Cin * k * k)) ** 0.5
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yd = F.conv2d(xd, wd, None, s, p)
    dy = torch.randn(yd.shape, generator=g)
    yd.backward(dy.double())
    xg, wg, dyg = x.to(DEV), w.to(DEV), dy.to(DEV)
    y = ops.conv2d_forward(xg, wg, s, p, p, 0)
    dx = ops.conv2d_dgrad(dyg, wg, xg.shape, s, p, p, 0)
    dw = ops.conv2d_wgrad(dyg, xg, wg.shape, s, p, p, 0)
    torch.cuda.synchronize()
    for what, a, b in (("forward", y, yd), ("data gradient", dx, xd.grad), ("weight gradient", dw, wd.grad)):
        r = rel_l2(a, b)
        assert r <= 5e-6, "%s of %s at B=%d: rel-L2 %.3e vs fp64" % (what, layer, B, r)
