"""-m gpu: the product networks / losses / train step (HIP kernels through the C ABI) against the
golden vectors captured from the reference's python (tests/golden/*.npz), on the same
deterministic weights and inputs.

Stated fp32 tolerances (SURVEY.md §8(c) measured envelope of the reference itself):
  generated tensors, attention maps, D features     max-abs <= 1e-4 (values O(1))
  scalar losses                                     rel <= 1e-5 (DAMSM-weighted G loss: 1e-4)
  D weight gradients                                checksum rel <= 1e-3 (BN gamma/beta: 5e-3, sums with cancellation)
  G weight gradients                                checksum rel <= 1e-2 (ill-conditioned through the
                                                    stacked BN+GLU generator even for torch-fp32 itself)
  post-Adam parameters / EMA / BN running stats     checksum rel <= 1e-4 of the tensor's abs-sum
"""
import os

import numpy as np
import pytest
import torch

from helpers import AdamDeltaCheck, GOLDEN, det_array, det_fill_state, load_pkg, probe, probe_close
from standin import StandInEncoder

load_pkg()
from mogan_amd.attngan import synthetic  # noqa: E402
from mogan_amd.attngan.miscc.config import cfg  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda"


def T(name, shape, scale=1.0, shift=0.0):
    return torch.from_numpy(det_array(name, shape, scale, shift))


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def close(got, want, atol=1e-4, rtol=1e-4, what=""):
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
    np.testing.assert_allclose(got, want, rtol=rtol, atol=atol, err_msg=what)


@pytest.fixture(autouse=True)
def small_cfg():
    cfg.GAN.GF_DIM, cfg.GAN.DF_DIM, cfg.GAN.R_NUM, cfg.GAN.Z_DIM = 4, 4, 2, 100
    cfg.TEXT.EMBEDDING_DIM, cfg.TEXT.WORDS_NUM = 16, 5
    cfg.TREE.BRANCH_NUM = 3
    cfg.TRAIN.SMOOTH.GAMMA1, cfg.TRAIN.SMOOTH.GAMMA2 = 4.0, 5.0
    cfg.TRAIN.SMOOTH.GAMMA3, cfg.TRAIN.SMOOTH.LAMBDA = 10.0, 50.0
    cfg.TRAIN.GENERATOR_LR = cfg.TRAIN.DISCRIMINATOR_LR = 2e-4
    cfg.STN_ALIGN_CORNERS, cfg.ATT_MASK_MODE, cfg.ADAM_EPS_MODE = False, 0, 0
    yield


def test_blocks():
    from mogan_amd.attngan import model
    g = golden("blocks")
    cfg.GAN.R_NUM = 2
    for tag, make, gshape in (("up", lambda: model.upBlock(8, 4), (4, 4, 16, 16)),
                              ("res", lambda: model.ResBlock(8), (4, 8, 8, 8)),
                              ("lrelu3", lambda: model.Block3x3_leakRelu(8, 6), (4, 6, 8, 8)),
                              ("down", lambda: model.downBlock(8, 6), (4, 6, 4, 4))):
        mod = make()
        det_fill_state(mod, "blocks.%s." % tag)
        mod = mod.to(DEV).train()
        x = T("blocks.x", (4, 8, 8, 8)).to(DEV).requires_grad_(True)
        y = mod(x)
        y.backward(T("blocks.%s.g" % tag, gshape).to(DEV))
        close(y, g[tag + "_y"], 2e-5, what=tag)
        close(x.grad, g[tag + "_dx"], 5e-5, 1e-3, what=tag + " dx")
        for k, p in mod.named_parameters():
            close(p.grad, g["%s_d_%s" % (tag, k.replace(".", "__"))], 1e-4, 1e-3, what=tag + k)
        for k, v in mod.state_dict().items():
            if "running" in k:
                close(v, g["%s_s_%s" % (tag, k.replace(".", "__"))], 1e-5, what=tag + k)


def test_global_attention_reference_mask_indexing():
    from mogan_amd.attngan.GlobalAttention import GlobalAttentionGeneral, func_attention
    g = golden("attn")
    for B in (3, 4):
        att = GlobalAttentionGeneral(6, 10)
        det_fill_state(att, "attn.")
        att = att.to(DEV)
        h = T("attn.h%d" % B, (B, 6, 4, 4)).to(DEV).requires_grad_(True)
        ctx = T("attn.ctx%d" % B, (B, 10, 5)).to(DEV).requires_grad_(True)
        p = "b%d_" % B
        att.applyMask(torch.from_numpy(g[p + "mask"]).to(DEV))
        wc, a = att(h, ctx)
        ((wc * T("attn.gw%d" % B, wc.shape).to(DEV)).sum() + (a * T("attn.ga%d" % B, a.shape).to(DEV)).sum()).backward()
        close(wc, g[p + "wc"], 2e-5); close(a, g[p + "attn"], 2e-5)
        close(h.grad, g[p + "dh"], 5e-5, 1e-3); close(ctx.grad, g[p + "dctx"], 5e-5, 1e-3)
        close(att.conv_context.weight.grad, g[p + "dw"], 5e-5, 1e-3)
    q = T("fattn.q", (2, 8, 4)).to(DEV).requires_grad_(True)
    c = T("fattn.c", (2, 8, 3, 3)).to(DEV).requires_grad_(True)
    wc, a = func_attention(q, c, 4.0)
    (wc * T("fattn.gw", wc.shape).to(DEV)).sum().backward()
    close(wc, g["f_wc"], 2e-5); close(a, g["f_attn"], 2e-5)
    close(q.grad, g["f_dq"], 5e-5, 1e-3); close(c.grad, g["f_dc"], 5e-5, 1e-3)


def test_g_net_end_to_end():
    from mogan_amd.attngan import model
    g = golden("gnet")
    bt = synthetic.to_device(synthetic.make_batch(3, words_num=5, nef=16, seed=11), DEV)
    G = model.G_NET()
    det_fill_state(G, "G.")
    G = G.to(DEV).train()
    z = bt["z"].clone().requires_grad_(True)
    sent = bt["sent_emb"].clone().requires_grad_(True)
    words = bt["words_embs"].clone().requires_grad_(True)
    imgs, atts, mu, logvar = G(z, sent, words, bt["mask"], bt["tmi"], bt["label_one_hot"], eps=bt["eps"])
    loss = sum((im * T("G.gimg%d" % i, im.shape).to(DEV)).sum() for i, im in enumerate(imgs))
    loss = loss + (mu * T("G.gmu", mu.shape).to(DEV)).sum() + (logvar * T("G.glv", logvar.shape).to(DEV)).sum()
    loss.backward()
    close(mu, g["mu"], 1e-5); close(logvar, g["logvar"], 1e-5)
    close(imgs[0], g["img64"]); close(imgs[1][:, :, ::2, ::2], g["img128"]); close(imgs[2][:, :, ::4, ::4], g["img256"])
    close(atts[0][:, :, ::4, ::4], g["att64"]); close(atts[1][:, :, ::8, ::8], g["att128"])
    probe_close(probe(imgs[2]), g["img256_p"], 1e-4, what="img256")
    probe_close(probe(atts[1]), g["att128_p"], 1e-4, what="att128")
    close(z.grad, g["dz"], 1e-3, 1e-2); close(sent.grad, g["dsent"], 1e-3, 1e-2)
    for k, p in G.named_parameters():
        probe_close(probe(p.grad), g["g_" + k.replace(".", "__")], 1e-2, what="grad " + k)
    for k, v in G.state_dict().items():
        if "running" in k:
            probe_close(probe(v), g["s_" + k.replace(".", "__")], 1e-4, what=k)


def test_d_nets():
    from mogan_amd.attngan import model
    g = golden("dnets")
    B = 3
    bt = synthetic.to_device(synthetic.make_batch(B, words_num=5, nef=16, seed=11), DEV)
    for i, cls in enumerate((model.D_NET64, model.D_NET128, model.D_NET256)):
        D = cls()
        det_fill_state(D, "D%d." % i)
        D = D.to(DEV).train()
        x = bt["imgs"][i].clone().requires_grad_(True)
        f = D(x, bt["label_one_hot"], bt["tm"], bt["tmi"]) if i == 0 else D(x)
        c = D.COND_DNET(f, bt["sent_emb"])
        u = D.UNCOND_DNET(f)
        cw = D.COND_DNET(f[:B - 1], bt["sent_emb"][1:B])
        dev = lambda n, s: T(n, s).to(DEV)
        loss = (f * dev("D%d.gf" % i, f.shape)).sum() + (c * dev("D%d.gc" % i, c.shape)).sum() \
            + (u * dev("D%d.gu" % i, u.shape)).sum() + (cw * dev("D%d.gcw" % i, cw.shape)).sum()
        loss.backward()
        p = "d%d_" % i
        close(f, g[p + "feat"]); close(c, g[p + "cond"], 1e-5); close(u, g[p + "uncond"], 1e-5)
        close(cw, g[p + "wrong"], 1e-5)
        probe_close(probe(x.grad), g[p + "dx_p"], 1e-3, what="dx")
        for k, v in D.named_parameters():   # 1-D (BN) grads are sums with heavy cancellation: looser
            probe_close(probe(v.grad), g[p + "g_" + k.replace(".", "__")], 1e-3 if v.dim() > 1 else 5e-3,
                        what="D%d %s" % (i, k))
        for k, v in D.state_dict().items():
            if "running" in k:
                probe_close(probe(v), g[p + "s_" + k.replace(".", "__")], 1e-4, what=k)


def _build_all():
    from mogan_amd.attngan import model
    G = model.G_NET()
    det_fill_state(G, "G.")
    Ds = []
    for i, cls in enumerate((model.D_NET64, model.D_NET128, model.D_NET256)):
        D = cls()
        det_fill_state(D, "D%d." % i)
        Ds.append(D.to(DEV).train())
    enc = StandInEncoder(16)
    det_fill_state(enc, "ENC.")
    for p in enc.parameters():
        p.requires_grad = False
    return G.to(DEV).train(), Ds, enc.to(DEV).eval()


def _noise_masks():
    """helpers.oracle_gradient_noise for the two-step trajectory of test_two_train_steps (CPU oracle in fp32 and fp64, cached)."""
    import test_oracle_golden as tog
    from helpers import oracle_gradient_noise
    ocfg = tog.SMALL
    bts = [synthetic.make_batch(4, words_num=5, nef=16, seed=100 + s_) for s_ in range(2)]
    return oracle_gradient_noise(ocfg, _oracle_build, bts)


def _oracle_build(dt):
    import test_oracle_golden as tog
    return tog._build_all(tog.SMALL, dt)


def test_losses():
    from mogan_amd.attngan.miscc import losses as L
    g = golden("losses")
    B = 4
    bt = synthetic.to_device(synthetic.make_batch(B, words_num=5, nef=16, seed=5), DEV)
    G, Ds, enc = _build_all()
    ones, zeros, match = torch.ones(B, device=DEV), torch.zeros(B, device=DEV), torch.arange(B, device=DEV)
    fakes = [T("L.fake%d" % i, im.shape, 0.5).to(DEV).requires_grad_(True) for i, im in enumerate(bt["imgs"])]
    for i, D in enumerate(Ds):
        kw = dict(local_labels=bt["label_one_hot"], transf_matrices=bt["tm"],
                  transf_matrices_inv=bt["tmi"]) if i == 0 else {}
        errD = L.discriminator_loss(D, bt["imgs"][i], fakes[i], bt["sent_emb"], ones, zeros, None, **kw)
        errD.backward()
        np.testing.assert_allclose(float(errD), float(g["errD%d" % i]), rtol=1e-5)
        for k, v in D.named_parameters():
            probe_close(probe(v.grad), g["d%d_g_%s" % (i, k.replace(".", "__"))], 1e-3 if v.dim() > 1 else 5e-3, what=k)
        D.zero_grad()
    errG, logs = L.generator_loss(Ds, enc, fakes, ones, bt["words_embs"], bt["sent_emb"], match, bt["cap_lens"],
                                  bt["class_ids"], None, local_labels=bt["label_one_hot"],
                                  transf_matrices=bt["tm"], transf_matrices_inv=bt["tmi"])
    errG.backward()
    np.testing.assert_allclose(float(errG), float(g["errG"]), rtol=1e-4)
    assert "w_loss" in logs and "g_loss2" in logs
    for i, f in enumerate(fakes):
        probe_close(probe(f.grad), g["dfake%d_p" % i], 1e-3, what="dfake%d" % i)
    feat = T("L.feat", (B, 16, 17, 17)).to(DEV).requires_grad_(True)
    code = T("L.code", (B, 16)).to(DEV).requires_grad_(True)
    w0, w1, _ = L.words_loss(feat, bt["words_embs"], match, bt["cap_lens"], bt["class_ids"], B)
    s0, s1 = L.sent_loss(code, bt["sent_emb"], match, bt["class_ids"], B)
    (w0 + 2 * w1 + 3 * s0 + 4 * s1).backward()
    for got, key in ((w0, "w0"), (w1, "w1"), (s0, "s0"), (s1, "s1")):
        np.testing.assert_allclose(float(got), float(g[key]), rtol=2e-5, err_msg=key)
    close(feat.grad, g["dfeat"], 1e-5, 1e-3); close(code.grad, g["dcode"], 1e-5, 1e-3)
    with torch.no_grad():
        _, _, att = L.words_loss(feat, bt["words_embs"], None, bt["cap_lens"], bt["class_ids"], B)
    close(att[0], g["watt0"], 1e-5); close(att[3], g["watt3"], 1e-5)
    mu = T("L.mu", (B, 100), 0.5).to(DEV).requires_grad_(True)
    lv = T("L.lv", (B, 100), 0.5).to(DEV).requires_grad_(True)
    kl = L.KL_loss(mu, lv)
    kl.backward()
    np.testing.assert_allclose(float(kl), float(g["kl"]), rtol=1e-5)
    close(mu.grad, g["dmu"], 1e-7, 1e-4); close(lv.grad, g["dlv"], 1e-7, 1e-4)


@pytest.mark.parametrize("which", [1, 2])
@pytest.mark.parametrize("force_deep", [False, True])
def test_discriminator_loss_paired_pass_equals_the_two_calls(which, force_deep):
    """discriminator_loss runs D(real) and D(fake.detach()) (two calls in the reference, miscc/losses.py:146,152) as ONE pass over
    [real; fake] with BatchNorm statistics per half (losses.D_PAIR, groups = 2 down the network): loss, every gradient, the
    running statistics (updated real first, then fake) and the call counters against the two-call path of the same kernels;
    with the packed-weight deep blocks forced on (grouped tail kernels) and off (grouped BatchNorm kernels); the activation
    trace comes out in the reference's call order."""
    from mogan_amd.attngan import model
    from mogan_amd.attngan.miscc import losses as L
    from mogan_amd.hip import ops
    cfg.GAN.DF_DIM = 32
    B, S = 4, 64 << which
    cls = (model.D_NET64, model.D_NET128, model.D_NET256)[which]
    real = T("pp.real%d" % which, (B, 3, S, S), 0.5).to(DEV)
    fake = T("pp.fake%d" % which, (B, 3, S, S), 0.5, 0.1).to(DEV).requires_grad_(True)
    sent = T("pp.sent", (B, 16)).to(DEV)
    ones, zeros = torch.ones(B, device=DEV), torch.zeros(B, device=DEV)
    res = {}
    ops.pk_debug_force(1 if force_deep else 0, -1, 0)
    was = L.D_PAIR
    try:
        for pair in (True, False):
            D = cls()
            det_fill_state(D, "PP%d." % which)
            D = D.to(DEV).train()
            if force_deep:
                for p_ in D.parameters():
                    if p_.dim() == 4:
                        ops.attach_packs(p_)
            L.D_PAIR = pair
            deep0 = ops.DEEP_STATS["fwd"]
            ops.ACT_TRACE = []
            try:
                err = L.discriminator_loss(D, real, fake, sent, ones, zeros, None)
            finally:
                trace, ops.ACT_TRACE = ops.ACT_TRACE, None
            err.backward()
            torch.cuda.synchronize()
            res[pair] = (float(err), {k: v.grad.clone() for k, v in D.named_parameters()},
                         {k: v.clone() for k, v in D.state_dict().items() if "running" in k or "tracked" in k},
                         [(a, t.clone()) for a, t in trace], ops.DEEP_STATS["fwd"] - deep0)
    finally:
        L.D_PAIR = was
        ops.pk_debug_force(0, -1, 0)
    assert fake.grad is None                              # (the D update detaches the fake image)
    (e1, g1, s1, t1, n1), (e2, g2, s2, t2, n2) = res[True], res[False]
    if force_deep:
        assert n2 > 0 and n1 > 0 and n1 < n2, "deep blocks: %d launches paired, %d in two calls" % (n1, n2)
    assert abs(e1 - e2) <= 2e-6 * abs(e2)
    for k in g2:
        d = float((g1[k] - g2[k]).norm() / (g2[k].norm() + 1e-30))
        assert d <= (2e-5 if g2[k].dim() > 1 else 2e-4), "%s: %.3e" % (k, d)
    for k in s2:
        if "tracked" in k:
            assert int(s1[k]) == int(s2[k]), k
        else:
            assert float((s1[k] - s2[k]).abs().max()) <= 1e-6 * max(1.0, float(s2[k].abs().max())), k
    assert len(t1) == len(t2)
    for (a1, x1), (a2, x2) in zip(t1, t2):
        assert a1 == a2 and x1.shape == x2.shape
        assert float((x1 - x2).abs().max()) <= 1e-5


@pytest.mark.parametrize("which", [0, 1, 2])
def test_discriminator_loss_in_two_halves_equals_the_whole(which):
    """miscc/losses.discriminator_loss_real + _fake (the real-image terms evaluated and back-propagated before the fake images
    exist, then the fake-image terms) against discriminator_loss with one backward: the loss value, every parameter gradient
    (accumulated in place by the two backward passes), and the BatchNorm running statistics -- the conditional head's
    wrong-pair call runs early but updates the running buffers in the reference's call order (real, fake, wrong;
    mogan_bn_running_update) -- and the call counters."""
    from mogan_amd.attngan import model
    from mogan_amd.attngan.miscc import losses as L
    B = 4
    cls = (model.D_NET64, model.D_NET128, model.D_NET256)[which]
    bt = synthetic.to_device(synthetic.make_batch(B, words_num=5, nef=16, seed=17), DEV)
    real = bt["imgs"][which]
    fake = T("h2.fake%d" % which, real.shape, 0.5, 0.1).to(DEV).requires_grad_(True)
    kw = dict(local_labels=bt["label_one_hot"], transf_matrices=bt["tm"], transf_matrices_inv=bt["tmi"]) if which == 0 else {}
    ones, zeros = torch.ones(B, device=DEV), torch.zeros(B, device=DEV)
    res = {}
    for split in (True, False):
        D = cls()
        det_fill_state(D, "H2%d." % which)
        D = D.to(DEV).train()
        for p_ in D.parameters():
            p_.grad = torch.zeros_like(p_)                # (dense .grad buffers: the kernels accumulate in place)
        if split:
            errR, pend = L.discriminator_loss_real(D, real, bt["sent_emb"], **kw)
            errR.backward()
            assert len(pend) == 1                         # the wrong-pair call of the conditional head's BatchNorm
            errF = L.discriminator_loss_fake(D, fake, bt["sent_emb"], pend, **kw)
            errF.backward()
            err = float(errR) + float(errF)
        else:
            e = L.discriminator_loss(D, real, fake, bt["sent_emb"], ones, zeros, None, **kw)
            e.backward()
            err = float(e)
        torch.cuda.synchronize()
        res[split] = (err, {k: v.grad.clone() for k, v in D.named_parameters()},
                      {k: v.clone() for k, v in D.state_dict().items() if "running" in k or "tracked" in k})
    assert fake.grad is None
    (e1, g1, s1), (e2, g2, s2) = res[True], res[False]
    assert abs(e1 - e2) <= 2e-6 * abs(e2)
    for k in g2:
        d = float((g1[k] - g2[k]).norm() / (g2[k].norm() + 1e-30))
        assert d <= 2e-6, "%s: %.3e" % (k, d)
    for k in s2:
        if "tracked" in k:
            assert int(s1[k]) == int(s2[k]), k
        else:
            assert float((s1[k] - s2[k]).abs().max()) <= 2e-7 * max(1.0, float(s2[k].abs().max())), k


def test_g_net_eval_mode():
    """netG.eval() forward of the sampling path (trainer.py:398,431-437): BN folded into per-channel affines."""
    g = golden("gnet_eval")
    from mogan_amd.attngan.model import G_NET
    G = G_NET()
    det_fill_state(G, "G.")
    G = G.to(DEV).eval()
    bt = synthetic.to_device(synthetic.make_batch(3, words_num=5, nef=16, seed=11), DEV)
    with torch.no_grad():
        imgs, atts, mu, logvar = G(bt["z"], bt["sent_emb"], bt["words_embs"], bt["mask"], bt["tmi"], bt["label_one_hot"],
                                   bt["eps"])
    close(imgs[0], g["img64"], 1e-3, 1e-4)
    close(imgs[2][:, :, ::4, ::4], g["img256"], 1e-3, 1e-4)
    probe_close(probe(atts[1]), g["att128_p"], 1e-4, what="att128")
    close(mu, g["mu"], 1e-4, 1e-5)
    assert all(int(v) == 0 for k, v in G.state_dict().items() if k.endswith("num_batches_tracked"))


@pytest.mark.parametrize("mode", ["eager", "graph", "branch_graphs", "branch_graphs_and_g", "branch_graphs_and_g_fwd",
                                  "branch_graphs_inputs_ready"])
def test_two_train_steps(mode):
    """the op order of the step (fake images generated once, each D updated
    before generator_loss forwards through it), Adam, EMA, BN running statistics -- eager, as one
    replayed hipGraph, with the discriminator branches as hipGraphs beside an eager generator, with the generator's forward
    replayed as well and its backward eager on the captured tape (branch_graphs_and_g_fwd: the default, MOGAN_G_GRAPHS=2), with
    forward / backward + Adam both replayed (MOGAN_G_GRAPHS=1), and -- branch_graphs_inputs_ready --
    the way bench.py and condGANTrainer.train() drive the engine: every batch carries an `inputs_ready` event, so D_i(real) of
    step 1 is replayed on its branch stream while the main stream is still in step 0's generator backward (no host
    synchronisation between the two steps; the branch results the main stream reads live outside the graphs' pool)."""
    from mogan_amd.attngan.trainer import TrainEngine
    g = golden("step")
    G, Ds, enc = _build_all()
    eng = TrainEngine(None, enc, G, Ds, use_graph=mode == "graph", branch_graphs=mode.startswith("branch_graphs"))
    assert eng.branch_graphs == mode.startswith("branch_graphs")
    eng.g_graphs = mode in ("branch_graphs_and_g", "branch_graphs_and_g_fwd")
    eng.g_fwd_only = mode == "branch_graphs_and_g_fwd"      # only the generator's forward replayed, its backward eager on the captured tape
    early = mode == "branch_graphs_inputs_ready"
    nets = [("G", G)] + [("D%d" % i, D) for i, D in enumerate(Ds)]
    init = {n: {k: probe(v) for k, v in net.state_dict().items()} for n, net in nets}
    bts = [synthetic.to_device(synthetic.make_batch(4, words_num=5, nef=16, seed=100 + step), DEV) for step in range(2)]
    if early:
        torch.cuda.synchronize()
        for bt in bts:
            bt["inputs_ready"] = torch.cuda.Event()
            bt["inputs_ready"].record()
    all_logs = []
    for step in range(2):
        logs = eng.step(bts[step])
        all_logs.append({k: v.clone() for k, v in logs.items()})
        if early and step == 0:
            continue                                   # straight into step 1: its D_i(real) replays overlap step 0's tail
        torch.cuda.synchronize()
        for st_, lg in enumerate(all_logs):
            # step 1 gets twice step 0's tolerance (round 5: ten times): the fp32 and the fp64 run of the oracle's own trajectory
            # differ by <= 4.2e-6 relative in every loss of step 1 (helpers.oracle_gradient_noise: the arithmetic's envelope)
            for k in ("errD0", "errD1", "errD2", "kl"):
                np.testing.assert_allclose(float(lg[k]), float(g["s%d_" % st_ + k]), rtol=1e-4 * (1 + st_), err_msg=k)
            np.testing.assert_allclose(float(lg["errG"]), float(g["s%d_errG" % st_]), rtol=2e-4 * (1 + st_))
            close(lg["fake64"], g["s%d_fake64" % st_], 2e-4 * (1 + st_), 1e-3)
        p = "s%d_" % step
        # Adam's first steps move every element by ~lr*sign(g): where |g| is at the fp32 noise floor the
        # sign is not reproducible, which shows on the tiny 1-D tensors (a 12-element BN
        # bias of magnitude 0.1 moves by up to 2*lr per step) -> looser checksum tolerance for those
        def tol_of(v):
            big = v.dim() > 1
            return (1e-4 if big else 1e-3) if step == 0 else (1e-3 if big else 5e-3)
        for k, v in G.state_dict().items():
            if v.is_floating_point():
                probe_close(probe(v), g[p + "G_" + k.replace(".", "__")], tol_of(v), what="G " + k)
        for i, D in enumerate(Ds):
            for k, v in D.state_dict().items():
                if v.is_floating_point():
                    probe_close(probe(v), g["%sD%d_%s" % (p, i, k.replace(".", "__"))], tol_of(v),
                                what="D%d %s" % (i, k))
        for (k, _), a in zip(G.named_parameters(), eng.optG.ema_params()):
            probe_close(probe(a), g[p + "ema_" + k.replace(".", "__")], 1e-4, what="ema " + k)
        # the Adam update itself, judged on parameter deltas (see helpers.AdamDeltaCheck): every sampled element whose fp64-oracle
        # gradient lies above the measured fp32 noise floor in all steps so far, 99 % of those within lr/4 of the reference's delta
        # (round 5 allowed a blanket 15 % / 5 % of all elements)
        masks, _ = _noise_masks()
        for n, net in nets:
            deltas = AdamDeltaCheck(lr=2e-4)
            for k, v in net.named_parameters():
                deltas.add(init[n][k], probe(v), g["%s%s_%s" % (p, n, k.replace(".", "__"))], judged=masks[(step, n, k)])
            deltas.check(0.01, what="%s step %d" % (n, step), min_judged_frac=0.5)
    if mode.startswith("branch_graphs"):
        assert len(eng._bg.get("G", {})) == (1 if eng.g_graphs else 0)      # one generator graph pair, replayed twice


def test_parked_weight_gradients_on_the_generator_tape(monkeypatch):
    """with the generator's forward replayed (MOGAN_G_GRAPHS=2, the default) autograd runs the eager backward
    on the CAPTURE stream; a weight gradient parked there for merging (ops._wgrad_accumulate: K <= MOGAN_MERGE_WGRAD_K and a weight
    tensor >= MOGAN_MERGE_WGRAD_W -- at coco_train.yml widths INIT_STAGE_G.upsample1, 5.3 M weights, K = 1024) was flushed by
    stream key on the MAIN stream, i.e. never: the layer alternated between no and stale gradients.  The threshold is lowered so
    that the reduced-width generator parks too; three steps of the replayed-forward engine must equal the eager-generator engine
    parameter for parameter, and nothing may stay parked behind a step."""
    from mogan_amd.attngan.trainer import TrainEngine
    from mogan_amd.hip import ops
    monkeypatch.setattr(ops, "_MERGE_W", 1)
    bts = [synthetic.to_device(synthetic.make_batch(4, words_num=5, nef=16, seed=300 + s_), DEV) for s_ in range(3)]
    runs = []
    for fwd_graph in (False, True):
        G, Ds, enc = _build_all()
        eng = TrainEngine(None, enc, G, Ds, use_graph=False, branch_graphs=True)
        eng.g_graphs, eng.g_fwd_only = fwd_graph, fwd_graph
        traj = []
        for bt in bts:
            eng.step(dict(bt))
            torch.cuda.synchronize()
            assert not ops._wgrad_pending, "parked weight gradients left behind a step: %d" % len(ops._wgrad_pending)
            assert len(ops._wgrad_frames) == 1 and not ops._wgrad_frames[0].keep
            traj.append({k: v.detach().clone() for k, v in G.named_parameters()})
        runs.append(traj)
        if fwd_graph:
            assert len(eng._bg.get("G", {})) == 1
    lr = 2e-4
    for st_, (a, b) in enumerate(zip(*runs)):
        for k in a:
            # same kernels, same operands, same order of the sums: the two engines differ in WHERE the launches are queued only
            # (atomics in the transformer's backward leave last-bit noise, and Adam turns a noise-level gradient's sign into +-lr:
            # a missing or stale gradient moves EVERY element of its tensor by ~lr)
            off = float(((a[k] - b[k]).abs() > 0.25 * lr).float().mean())
            assert off <= 0.02, "step %d, %s: %.1f %% of the parameters differ by > lr/4 between the eager and the " \
                                "replayed-forward generator" % (st_, k, 100 * off)


@pytest.mark.parametrize("B", [4, 20])
def test_object_pathways_batched_equal_looped(B):
    """The reference's loops over the objects (model.py:105-114, 395-407, 662-672) run as ONE batch of 3*B samples with per-object
    BatchNorm statistics (model.BATCH_OBJECTS); this must give what the literal loops give -- outputs, every parameter gradient, the
    BatchNorm running statistics and call counters -- for B = 4 (every grouped BatchNorm in one launch) and B = 20 (16x16 maps:
    20*256 values per channel and object exceed the one-launch limit, the grouped call falls back to one call per object)."""
    from helpers import rel_l2
    from mogan_amd.attngan import model
    bt = synthetic.to_device(synthetic.make_batch(B, words_num=5, nef=16, seed=77), DEV)
    results = []
    for batched in (True, False):
        model.BATCH_OBJECTS = batched
        try:
            G, Ds, _ = _build_all()
            fake, _, mu, logvar = G(bt["z"], bt["sent_emb"], bt["words_embs"], bt["mask"], bt["tmi"], bt["label_one_hot"], bt["eps"])
            feat = Ds[0](fake[0], bt["label_one_hot"], bt["tm"], bt["tmi"])
            loss = (feat * feat).mean() + sum((f * f).mean() for f in fake) + (mu * logvar).mean()
            loss.backward()
            torch.cuda.synchronize()
            res = {"fake%d" % i: f.detach() for i, f in enumerate(fake)}
            res["feat"] = feat.detach()
            for n, net in (("G", G), ("D", Ds[0])):
                for k, v in net.named_parameters():
                    if v.grad is not None:                       # (the logits heads of D_NET64 are not part of this loss)
                        res["%s.grad.%s" % (n, k)] = v.grad.detach().clone()
                for k, v in net.named_buffers():
                    res["%s.buf.%s" % (n, k)] = v.detach().clone().float()
            results.append(res)
        finally:
            model.BATCH_OBJECTS = True
    a, b = results
    assert a.keys() == b.keys()
    worst = ("", 0.0)
    for k in a:
        if "num_batches_tracked" in k:
            assert torch.equal(a[k], b[k]), k
            continue
        e = rel_l2(a[k], b[k])
        if e > worst[1]:
            worst = (k, e)
        assert e <= 1e-5, (k, e)                      # measured: <= 6e-7 (other tile shapes and split-K factors in the batched convolutions)
    print("worst:", worst)


@pytest.mark.parametrize("global_loss", [False, True])
def test_two_train_steps_through_rccl(global_loss, monkeypatch):
    """The N>1 code path on one GPU: process group "nccl" (= RCCL) with world_size 1, flat-bucket all-reduces on the
    branch streams, and (global_loss) the gathered-batch loss mode of attngan/parallel.py -- with one rank both must
    reproduce the reference's single-GPU trajectory."""
    import torch.distributed as dist
    from mogan_amd.attngan.trainer import TrainEngine
    monkeypatch.setenv("MOGAN_FORCE_DIST", "1")
    monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % (29300 + os.getpid() % 500), rank=0, world_size=1)
    try:
        cfg.TRAIN.GLOBAL_BATCH_LOSS = global_loss
        g = golden("step")
        G, Ds, enc = _build_all()
        eng = TrainEngine(None, enc, G, Ds, distributed=True, use_graph=False)
        assert eng.distributed and eng.world == 1
        for step in range(2):
            bt = synthetic.to_device(synthetic.make_batch(4, words_num=5, nef=16, seed=100 + step), DEV)
            logs = eng.step(bt)
            torch.cuda.synchronize()
            p = "s%d_" % step
            for k in ("errD0", "errD1", "errD2", "kl"):
                np.testing.assert_allclose(float(logs[k]), float(g[p + k]), rtol=1e-4 * (1 + 9 * step), err_msg=k)
            np.testing.assert_allclose(float(logs["errG"]), float(g[p + "errG"]), rtol=2e-4 * (1 + 9 * step))
            close(logs["fake64"], g[p + "fake64"], 2e-4 * (1 + 9 * step), 1e-3)
    finally:
        cfg.TRAIN.GLOBAL_BATCH_LOSS = False
        dist.destroy_process_group()


def test_every_gradient_tensor_in_full_against_the_fp64_oracle():
    """The reference fixtures hold 35-number summaries of the big gradient tensors; a localized error in a 10^5-element
    gradient would be invisible there.  Here EVERY parameter gradient of the three discriminators (discriminator_loss +
    backward) and of the generator (a loss on all three images, mu and logvar) is compared element for element with the fp64
    oracle -- which test_oracle_golden.py pins to those same fixtures -- evaluated with this pass's LeakyReLU decisions
    (hip/ops.ACT_TRACE -> oracle.LRELU_MASKS, see test_fullwidth_parity_gpu.py).  Stated tolerances: D gradients rel-L2
    <= 1e-4 per tensor (measured <= 1.1e-5), G gradients <= 2e-3 (SURVEY section 8(c) allows 1e-2: ill-conditioned through the
    stacked BN+GLU generator even for torch-fp32 itself; measured against fp64: 3.2e-5)."""
    from helpers import rel_l2
    from mogan_amd.attngan import model
    from mogan_amd.attngan.miscc import losses as L
    from mogan_amd.hip import ops
    from oracle import attngan_oracle as O
    ocfg = O.Cfg(gf_dim=4, df_dim=4, emb_dim=16, r_num=2, words_num=5)
    B, dt = 4, torch.float64
    cpu = synthetic.make_batch(B, words_num=5, nef=16, seed=31)
    bt = synthetic.to_device(cpu, DEV)
    c64 = {k: (v.to(dt) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in cpu.items()}
    c64["imgs"] = [t.to(dt) for t in cpu["imgs"]]
    fakes = [T("FT.fake%d" % i, im.shape, 0.5) for i, im in enumerate(cpu["imgs"])]
    worst = {}
    for i, cls in enumerate((model.D_NET64, model.D_NET128, model.D_NET256)):
        D = cls()
        sd = det_fill_state(D, "D%d." % i)
        D = D.to(DEV).train()
        kw = dict(local_labels=bt["label_one_hot"], transf_matrices=bt["tm"], transf_matrices_inv=bt["tmi"]) if i == 0 else {}
        ops.ACT_TRACE = []
        try:
            err = L.discriminator_loss(D, bt["imgs"][i], fakes[i].to(DEV), bt["sent_emb"], None, None, None, **kw)
        finally:
            trace, ops.ACT_TRACE = ops.ACT_TRACE, None
        err.backward()
        od = O.from_state_dict(sd, dtype=dt)
        O.LRELU_MASKS = [(t > 0).cpu() for a, t in trace if a == ops.ACT_LRELU]
        try:
            oerr = O.discriminator_loss(i, od, c64["imgs"][i], fakes[i].to(dt), c64["sent_emb"], c64, ocfg)
            assert not O.LRELU_MASKS
        finally:
            O.LRELU_MASKS = None
        oerr.backward()
        assert abs(float(err.detach()) - float(oerr.detach())) <= 1e-5 * abs(float(oerr.detach()))
        for k, p in D.named_parameters():
            r = rel_l2(p.grad, od[k].grad)
            worst["D%d" % i] = max(worst.get("D%d" % i, 0.0), r)
            assert r <= 1e-4, "D%d %s: rel-L2 %.3e" % (i, k, r)
    G = model.G_NET()
    sdg = det_fill_state(G, "G.")
    G = G.to(DEV).train()
    ops.ACT_TRACE = []
    try:
        imgs, atts, mu, logvar = G(bt["z"], bt["sent_emb"], bt["words_embs"], bt["mask"], bt["tmi"], bt["label_one_hot"],
                                   eps=bt["eps"])
    finally:
        trace, ops.ACT_TRACE = ops.ACT_TRACE, None
    ups = [T("FT.gimg%d" % i, im.shape) for i, im in enumerate(imgs)] + [T("FT.gmu", mu.shape), T("FT.glv", logvar.shape)]
    sum((t * u.to(DEV)).sum() for t, u in zip(list(imgs) + [mu, logvar], ups)).backward()
    og = O.from_state_dict(sdg, dtype=dt)
    O.LRELU_MASKS = [(t > 0).cpu() for a, t in trace if a == ops.ACT_LRELU]
    try:
        oimgs, _, omu, olv, _ = O.g_net(og, ocfg, c64["z"], c64["sent_emb"], c64["words_embs"], c64["mask"], c64["tmi"],
                                        c64["label_one_hot"], c64["eps"])
        assert not O.LRELU_MASKS
    finally:
        O.LRELU_MASKS = None
    sum((t * u.to(dt)).sum() for t, u in zip(list(oimgs) + [omu, olv], ups)).backward()
    for k, p in G.named_parameters():
        r = rel_l2(p.grad, og[k].grad)
        worst["G"] = max(worst.get("G", 0.0), r)
        assert r <= 2e-3, "G %s: rel-L2 %.3e" % (k, r)
    print("full-tensor gradient parity, worst rel-L2 per network:", {k: "%.1e" % v for k, v in worst.items()})
