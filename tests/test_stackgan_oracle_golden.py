"""Pins oracle/stackgan_oracle.py against the golden vectors captured from the reference's own python for the
three sibling trees (tests/golden/make_golden_stackgan.py): coco-stackgan stage I / II, clevr, multi-mnist.
CPU only.  The oracle calls the same torch-CPU ops as the reference in (nearly) the same order, so fp32
agreement is ~1e-6; tolerances leave ~10x headroom over what was observed."""
import numpy as np
import pytest
import torch

from helpers import AdamDeltaCheck, load_pkg, probe, probe_close
from oracle import stackgan_oracle as S
from stackgan_cases import CASES, T, det_state, golden, oracle_cfg, specs, sub

load_pkg()
from mogan_amd.stackgan import synthetic  # noqa: E402

FAST = [c for c in CASES if c != "coco_s2"]


def close(got, want, rtol=2e-5, atol=2e-6, what=""):
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
    np.testing.assert_allclose(got, want, rtol=rtol, atol=atol, err_msg=what)


def _freeze_stage1(net):
    for k, v in net.items():
        if k.startswith("STAGE1_G.") and v.is_floating_point():
            v.requires_grad_(False)


def _nets(case):
    cfg, tree, stage, B = oracle_cfg(case)
    gs, ds = specs(cfg)
    G = S.from_state_dict(det_state(gs, "G."))
    D = S.from_state_dict(det_state(ds, "D."))
    if stage == 2:
        _freeze_stage1(G)
    return cfg, tree, stage, B, G, D, gs, ds


@pytest.mark.parametrize("case", list(CASES))
def test_state_dict_layout(case):
    """the oracle's key->shape specs are the reference modules' state_dict keys, in order."""
    g = golden("stackgan_%s_nets" % case)
    cfg, tree, stage, B = oracle_cfg(case)
    gs, ds = specs(cfg)
    assert list(gs) == [str(k) for k in g["g_keys"]]
    assert list(ds) == [str(k) for k in g["d_keys"]]


@pytest.mark.parametrize("case", list(CASES))
def test_theta_of_tree(case):
    """bbox -> theta of each tree (S/miscc/utils.py:19-52; M computes in float64) on the synthetic boxes."""
    g = golden("stackgan_%s_nets" % case)
    tree, stage, B, _ = CASES[case]
    bt = synthetic.make_batch(tree, B, stage=stage, seed=21, text_dim=12)
    tm, tmi = (bt["tm_s2"], bt["tmi_s2"]) if stage == 2 else (bt["tm"], bt["tmi"])
    np.testing.assert_array_equal(tm.reshape(-1, 2, 3).numpy(), g["ref_tm"])
    np.testing.assert_array_equal(tmi.reshape(-1, 2, 3).numpy(), g["ref_tmi"])


def _check_nets(case):
    g = golden("stackgan_%s_nets" % case)
    cfg, tree, stage, B, G, D, _, _ = _nets(case)
    bt = synthetic.make_batch(tree, B, stage=stage, seed=21, text_dim=12)
    z = bt["z"].clone().requires_grad_(True)
    st = S.TrainState(G, D, cfg)
    fake, mu, logvar = S.generate(st, dict(bt, z=z))
    loss = (fake * T("G.gimg", fake.shape)).sum()
    if mu is not None:
        loss = loss + (mu * T("G.gmu", mu.shape)).sum() + (logvar * T("G.glv", logvar.shape)).sum()
    loss.backward()
    close(sub(fake), g["fake_sub"], rtol=1e-4, atol=1e-5)
    probe_close(probe(fake), g["fake_p"], 1e-5, what="fake")
    if "dz" in g.files:
        close(z.grad, g["dz"], rtol=2e-3, atol=1e-5)
    else:
        assert z.grad is None              # stage II: z only feeds the detached stage-I image
    if mu is not None:
        close(mu, g["mu"]); close(logvar, g["logvar"])
    for k, v in S.parameters(G):
        if v.grad is not None:
            probe_close(probe(v.grad), g["gg_" + k.replace(".", "__")], 2e-4, what="G grad " + k)
    for k, v in G.items():
        if "running" in k:
            probe_close(probe(v), g["gs_" + k.replace(".", "__")], 1e-5, what=k)
    # discriminator
    tm, tmi = (bt["tm_s2"], bt["tmi_s2"]) if stage == 2 else (bt["tm"], bt["tmi"])
    x = bt["real_imgs"].clone().requires_grad_(True)
    f = S.d_features(D, cfg, x, bt["label_one_hot"], tm, tmi)
    cond = T("D.cond", (B, 128), 0.5) if tree == "coco" else bt["label_one_hot"].sum(1)
    c = S.cond_logits(D, cfg, f, cond)
    cw = S.cond_logits(D, cfg, f[:B - 1], cond[1:])
    loss = (f * T("D.gf", f.shape)).sum() + (c * T("D.gc", c.shape)).sum() + (cw * T("D.gcw", cw.shape)).sum()
    if stage == 2:
        u = S.uncond_logits(D, f)
        loss = loss + (u * T("D.gu", u.shape)).sum()
        close(u, g["d_uncond"], rtol=1e-4)
    loss.backward()
    close(f, g["d_feat"], rtol=1e-4, atol=1e-5)
    close(c, g["d_cond"], rtol=1e-4); close(cw, g["d_wrong"], rtol=1e-4)
    probe_close(probe(x.grad), g["d_dx_p"], 1e-4, what="dx")
    for k, v in S.parameters(D):
        probe_close(probe(v.grad), g["dg_" + k.replace(".", "__")], 1e-4, what="D grad " + k)
    for k, v in D.items():
        if "running" in k:
            probe_close(probe(v), g["ds_" + k.replace(".", "__")], 1e-5, what=k)


def _check_steps(case):
    g = golden("stackgan_%s_step" % case)
    cfg, tree, stage, B, G, D, _, _ = _nets(case)
    st = S.TrainState(G, D, cfg)
    init = {n: {k: probe(v) for k, v in S.parameters(net)} for n, net in (("G", G), ("D", D))}
    for step in range(2):
        bt = synthetic.make_batch(tree, B, stage=stage, seed=300 + step, text_dim=12)
        logs = S.train_step(st, bt)
        p = "s%d_" % step
        for k in ("errD", "errD_real", "errD_wrong", "errD_fake", "errG") + (("kl",) if cfg.text else ()):
            np.testing.assert_allclose(logs[k], float(g[p + k]), rtol=2e-5 * (1 + 20 * step), err_msg=k)
        close(sub(logs["fake"]), g[p + "fake_sub"], rtol=1e-3 * (1 + 10 * step), atol=1e-4 * (1 + 10 * step))
        tol = 2e-5 if step == 0 else 2e-4
        for name, net in (("G", G), ("D", D)):
            for k, v in net.items():
                if v.is_floating_point():
                    probe_close(probe(v), g["%s%s_%s" % (p, name, k.replace(".", "__"))], tol, what=name + " " + k)
            deltas = AdamDeltaCheck(lr=2e-4)          # the Adam update itself, on parameter deltas
            for k, v in S.parameters(net):
                if not (tree == "mnist" and k.startswith("label.")):      # never receives a gradient (M/model.py:163)
                    deltas.add(init[name][k], probe(v), g["%s%s_%s" % (p, name, k.replace(".", "__"))])
            deltas.check(0.03, what="%s %s step %d" % (case, name, step))


@pytest.mark.parametrize("case", FAST)
def test_networks(case):
    _check_nets(case)


@pytest.mark.parametrize("case", FAST)
def test_two_train_steps(case):
    _check_steps(case)


def test_stage2_networks_and_steps():
    """coco stage II at the generator's full width (GF_DIM 192 is forced by S/model.py:340)."""
    _check_nets("coco_s2")
    _check_steps("coco_s2")
