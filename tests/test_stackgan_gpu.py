"""HIP path of the StackGAN-family trees (coco-stackgan stage I/II, clevr, multi-mnist) against the golden
vectors captured from the reference (tests/golden/stackgan_*.npz) -- forward, gradients, BN running statistics,
losses and the two-step train trajectory (eager and hipGraph).  Tolerances follow SURVEY.md §8(c): generated
tensors max-abs <= 1e-4-ish, scalar losses rel 1e-5 (first step), D grads 1e-4, G grads 1e-2 (fp32 through the
stacked BN generator is ill-conditioned), post-Adam parameters through abs-sum-relative checksums."""
import os

import numpy as np
import pytest
import torch

from helpers import AdamDeltaCheck, checksum_close, det_fill_state, load_pkg, probe, probe_close
from stackgan_cases import CASES, T, golden, sub

pytestmark = pytest.mark.gpu
load_pkg()
from mogan_amd.stackgan import synthetic  # noqa: E402
from mogan_amd.stackgan.engine import StackGANEngine  # noqa: E402
from mogan_amd.attngan.synthetic import to_device  # noqa: E402

DEV = "cuda"
# absolute tolerance on generated pixels.  coco_s2 runs the full-width (68 M parameter) stage-II generator at
# B=2: an fp64 run of the oracle differs from the fp32 reference itself by max-abs 9.1e-5 (rms 1.7e-5), the HIP
# path by 1.35e-4 -- both fp32 paths sit ~1e-4 from the exact result.
NOISE = {"coco_s1": 2e-4, "clevr": 2e-4, "mnist": 2e-4, "coco_s2": 5e-4}


def tree_modules(tree):
    if tree == "coco":
        from mogan_amd.stackgan.coco import model
        from mogan_amd.stackgan.coco.miscc.config import cfg
    elif tree == "clevr":
        from mogan_amd.stackgan.clevr import model
        from mogan_amd.stackgan.clevr.miscc.config import cfg
    else:
        from mogan_amd.stackgan.multi_mnist import model
        from mogan_amd.stackgan.multi_mnist.miscc.config import cfg
    return model, cfg


def build(case, device=DEV):
    tree, stage, B, kw = CASES[case]
    model, cfg = tree_modules(tree)
    cfg.GAN.GF_DIM, cfg.GAN.DF_DIM, cfg.GAN.CONDITION_DIM = kw["gf_dim"], kw["df_dim"], kw["cond_dim"]
    cfg.GAN.R_NUM = kw.get("r_num", 2)
    cfg.USE_BBOX_LAYOUT = True
    if tree == "coco":
        cfg.STAGE, cfg.TEXT.DIMENSION = stage, kw["text_dim"]
    if stage == 2:
        G, D = model.STAGE2_G(model.STAGE1_G()), model.STAGE2_D()
    else:
        G, D = model.STAGE1_G(), model.STAGE1_D()
    det_fill_state(G, "G.")
    det_fill_state(D, "D.")
    return tree, stage, B, cfg, model, G.to(device).train(), D.to(device).train()


def close(got, want, rtol, atol=0.0, what=""):
    np.testing.assert_allclose(got.detach().cpu().numpy(), want, rtol=rtol, atol=atol, err_msg=what)


def run_g(G, tree, stage, b):
    if tree == "coco" and stage == 2:
        s1, fake, mu, logvar, ll = G(b["txt_embedding"], b["z"], b["tmi"], b["tm_s2"], b["tmi_s2"], b["label_one_hot"],
                                     eps=b["eps"], eps_s1=b["eps_s1"])
        return fake, mu, logvar, ll, s1
    if tree == "coco":
        _, fake, mu, logvar, ll = G(b["txt_embedding"], b["z"], b["tmi"], b["label_one_hot"], eps=b["eps"])
        return fake, mu, logvar, ll, None
    out = G(b["z"], b["tmi"], b["label_one_hot"])
    return (out[1] if isinstance(out, tuple) else out), None, None, None, None


@pytest.mark.parametrize("case", list(CASES))
def test_networks(case):
    g = golden("stackgan_%s_nets" % case)
    tree, stage, B, cfg, model, G, D = build(case)
    assert list(G.state_dict().keys()) == [str(k) for k in g["g_keys"]]
    assert list(D.state_dict().keys()) == [str(k) for k in g["d_keys"]]
    b = to_device(synthetic.make_batch(tree, B, stage=stage, seed=21, text_dim=12), DEV)
    b["z"] = b["z"].clone().requires_grad_(True)
    fake, mu, logvar, ll, s1 = run_g(G, tree, stage, b)
    loss = (fake * T("G.gimg", fake.shape).to(DEV)).sum()
    if mu is not None:
        loss = loss + (mu * T("G.gmu", mu.shape).to(DEV)).sum() + (logvar * T("G.glv", logvar.shape).to(DEV)).sum()
    loss.backward()
    close(sub(fake), g["fake_sub"], rtol=1e-3, atol=NOISE[case], what="fake")
    probe_close(probe(fake), g["fake_p"], 1e-4, what="fake probe")
    if s1 is not None:
        close(sub(s1), g["s1_sub"], rtol=1e-3, atol=1e-4, what="stage-I image")
    if "dz" in g.files:
        assert float((b["z"].grad.cpu() - torch.from_numpy(g["dz"])).norm() / np.linalg.norm(g["dz"])) < 1e-2
    if mu is not None:
        close(mu, g["mu"], rtol=1e-4, atol=1e-5)
        close(logvar, g["logvar"], rtol=1e-4, atol=1e-5)
        close(ll, g["local_labels"], rtol=1e-3, atol=1e-4)
    for k, p in G.named_parameters():
        key = "gg_" + k.replace(".", "__")
        if key in g.files:
            # ReLU decisions of pre-activations within rounding of zero: both MFMA forms of the library sit 1e-3 .. 1e-2 from
            # the reference's fp32 run on these gradients (tools/diag_stackgan_ggrads.py); the label layer (a Linear over
            # B*K rows -> BatchNorm -> ReLU) is the most sensitive: 3.2e-3 (native fp32 form) / 1.1e-2 .. 1.3e-2 (split-bf16)
            probe_close(probe(p.grad), g[key], 3e-2 if k.startswith("label.") else 1e-2, what="G grad " + k)
    for k, v in G.state_dict().items():
        if "running" in k:
            probe_close(probe(v), g["gs_" + k.replace(".", "__")], 1e-4, what=k)
    # discriminator
    tm, tmi = (b["tm_s2"], b["tmi_s2"]) if stage == 2 else (b["tm"], b["tmi"])
    x = b["real_imgs"].clone().requires_grad_(True)
    f = D(x, b["label_one_hot"], tm, tmi)
    cond = T("D.cond", (B, 128), 0.5).to(DEV) if tree == "coco" else b["label_one_hot"].sum(1)
    c = D.get_cond_logits(f, cond)
    cw = D.get_cond_logits(f[:B - 1], cond[1:])
    loss = (f * T("D.gf", f.shape).to(DEV)).sum() + (c * T("D.gc", c.shape).to(DEV)).sum() \
        + (cw * T("D.gcw", cw.shape).to(DEV)).sum()
    if D.get_uncond_logits is not None:
        u = D.get_uncond_logits(f)
        loss = loss + (u * T("D.gu", u.shape).to(DEV)).sum()
        close(u, g["d_uncond"], rtol=1e-3, atol=1e-4)
    loss.backward()
    close(f, g["d_feat"], rtol=1e-3, atol=1e-4)
    close(c, g["d_cond"], rtol=1e-3, atol=1e-4)
    close(cw, g["d_wrong"], rtol=1e-3, atol=1e-4)
    probe_close(probe(x.grad), g["d_dx_p"], 1e-3, what="dx")
    for k, p in D.named_parameters():
        probe_close(probe(p.grad), g["dg_" + k.replace(".", "__")], 5e-3 if p.dim() == 1 else 1e-3,
                    what="D grad " + k)
    for k, v in D.state_dict().items():
        if "running" in k:
            probe_close(probe(v), g["ds_" + k.replace(".", "__")], 1e-4, what=k)


@pytest.mark.parametrize("use_graph", [False, True])
@pytest.mark.parametrize("case", list(CASES))
def test_two_train_steps(case, use_graph):
    """S/trainer.py:188-231 op order (D updated before the G loss goes through it), Adam, BN buffers."""
    g = golden("stackgan_%s_step" % case)
    tree, stage, B, cfg, model, G, D = build(case)
    variant = model.VARIANT
    init = {name: {k: probe(v) for k, v in net.state_dict().items() if v.is_floating_point()}
            for name, net in (("G", G), ("D", D))}
    trainable = {name: {k for k, p_ in net.named_parameters() if p_.requires_grad} for name, net in (("G", G), ("D", D))}
    eng = StackGANEngine(G, D, cfg, variant, stage=stage, use_graph=use_graph)
    for step in range(2):
        b = to_device(synthetic.make_batch(tree, B, stage=stage, seed=300 + step, text_dim=12), DEV)
        logs = eng.step(b)
        p = "s%d_" % step
        chaotic = case == "coco_s2" and step > 0
        for k in ("errD", "errD_real", "errD_wrong", "errD_fake", "errG") + (("kl",) if variant.text else ()):
            np.testing.assert_allclose(float(logs[k]), float(g[p + k]), rtol=3e-2 if chaotic else 2e-4 * (1 + 20 * step),
                                       err_msg=k)
        if chaotic:
            # Second step of the full-width stage-II generator at B=2: the first Adam step moved 68 M weights by
            # +-lr each and the sign of near-zero gradients is fp32 noise, so the *reference's own* fp32 trajectory
            # is already max-abs 6.3e-2 / rms 8.9e-3 away from an fp64 run of the same step (losses: errG 0.7 %,
            # errD_fake 1.9 %; measured with oracle/stackgan_oracle.py in float64).  Judge at that noise level.
            d = sub(logs["fake"]).cpu().numpy() - g[p + "fake_sub"]
            assert np.sqrt((d ** 2).mean()) < 2.5e-2 and np.abs(d).max() < 0.25, (np.sqrt((d ** 2).mean()), np.abs(d).max())
        else:
            close(sub(logs["fake"]), g[p + "fake_sub"], rtol=2e-3 * (1 + 10 * step), atol=NOISE[case] * (1 + 10 * step))
        for name, net in (("G", G), ("D", D)):
            deltas = AdamDeltaCheck(lr=2e-4)
            for k, v in net.state_dict().items():
                if v.is_floating_point():
                    want = g["%s%s_%s" % (p, name, k.replace(".", "__"))]
                    # checksums of the tensor; the sampled elements are judged through their Adam deltas below
                    checksum_close(probe(v), want, (2e-3 if v.dim() == 1 else 5e-4) * (1 + 4 * step),
                                   what=name + " " + k)
                    if k in trainable[name] and not (tree == "mnist" and k.startswith("label.")):
                        deltas.add(init[name][k], probe(v), want)
                    elif "running" not in k:
                        np.testing.assert_array_equal(probe(v), init[name][k], err_msg="frozen " + k)
                elif k.endswith("num_batches_tracked"):
                    want = g["%s%s_%s" % (p, name, k.replace(".", "__"))]
                    assert float(v) == float(want[0]), k
            # G gradients through the stacked BN generator are ill-conditioned in fp32 (rel-L2
            # 1.7e-3..3.2e-3 between fp32 and fp64 runs of the reference itself) -> more sign flips than in D
            # (the chaotic second step of coco_s2, see above, doubles that)
            deltas.check((0.30 if chaotic else 0.15) if name == "G" else (0.10 if chaotic else 0.05),
                         what="%s %s step %d" % (case, name, step))


def _run_main(pkg, yml_text, tmp_path, name, extra=()):
    import importlib
    entry = importlib.import_module("mogan_amd.stackgan.%s.main" % pkg)
    yml = tmp_path / (name + ".yml")
    yml.write_text(yml_text)
    out = tmp_path / name
    entry.main(["--cfg", str(yml), "--synthetic", "8", "--manualSeed", "3", "--output_dir", str(out)] + list(extra))
    import glob
    ckpts = sorted(glob.glob(str(out / "Model" / "checkpoint_*.pth")))
    assert ckpts, "no checkpoint written"
    return ckpts[-1]


def test_family_train_loops_and_checkpoints(tmp_path):
    """`main.py --cfg ... ` -> GANTrainer.train() of each tree on synthetic items: LR-decay epoch, save_model in the
    reference's checkpoint layout (S/miscc/utils.py:162-176), and stage II loading the stage-I generator from a
    stage-I checkpoint (S/trainer.py:76-108)."""
    common = "GPU_ID: '0'\nZ_DIM: 100\nWORKERS: 0\nUSE_BBOX_LAYOUT: True\n"
    train = "TRAIN: {FLAG: True, BATCH_SIZE: 4, MAX_EPOCH: 2, LR_DECAY_EPOCH: 1, SNAPSHOT_INTERVAL: 1}\n"
    ck = _run_main("clevr", common + train + "GAN: {CONDITION_DIM: 16, DF_DIM: 4, GF_DIM: 4}\n", tmp_path, "clevr")
    sd = torch.load(ck, map_location="cpu", weights_only=False)
    assert set(sd) == {"epoch", "netG", "optimG", "netD", "optimD"} and sd["netD"] == {} and sd["epoch"] == 1
    assert all(torch.isfinite(v).all() for v in sd["netG"].values() if v.is_floating_point())
    # S/trainer.py:237-260: scalar summaries + sample grids on iteration 0 of each epoch (i % 500 == 0)
    import json
    from PIL import Image
    rows = [json.loads(l) for l in open(str(tmp_path / "clevr" / "Log" / "scalars.jsonl"))]
    assert {r["tag"] for r in rows} == {"D_loss", "D_loss_real", "D_loss_wrong", "D_loss_fake", "G_loss"}
    assert sorted({r["step"] for r in rows}) == [1, 3] and all(np.isfinite(r["value"]) for r in rows)
    for f in ("real_samples.png", "fake_samples_epoch_000.png", "fake_samples_epoch_001.png"):
        assert Image.open(str(tmp_path / "clevr" / "Image" / f)).size == (4 * 66 + 2, 66 + 2)
    _run_main("multi_mnist", common + train + "GAN: {CONDITION_DIM: 128, DF_DIM: 4, GF_DIM: 4}\n", tmp_path, "mnist")
    s1 = _run_main("coco", common + "STAGE: 1\nIMSIZE: 64\n" + train.replace("}", ", COEFF: {KL: 2.0}}")
                   + "GAN: {CONDITION_DIM: 128, DF_DIM: 4, GF_DIM: 192}\nTEXT: {DIMENSION: 16}\n", tmp_path, "s1",
                   extra=["--max_epoch", "1"])
    s2 = _run_main("coco", common + "STAGE: 2\nIMSIZE: 256\nSTAGE1_G: '%s'\n" % s1
                   + "TRAIN: {FLAG: True, BATCH_SIZE: 2, MAX_EPOCH: 1, LR_DECAY_EPOCH: 1, SNAPSHOT_INTERVAL: 1, COEFF: {KL: 2.0}}\n"
                   + "GAN: {CONDITION_DIM: 128, DF_DIM: 4, GF_DIM: 192, R_NUM: 1}\nTEXT: {DIMENSION: 16}\n", tmp_path, "s2",
                   extra=["--synthetic", "4"])
    g1 = torch.load(s1, map_location="cpu", weights_only=False)["netG"]
    g2 = torch.load(s2, map_location="cpu", weights_only=False)["netG"]
    # the frozen stage-I generator inside STAGE2_G still holds the stage-I checkpoint's weights
    assert torch.equal(g2["STAGE1_G.fc.0.weight"], g1["fc.0.weight"])
    assert not torch.equal(g2["STAGE1_G.fc.1.running_mean"], g1["fc.1.running_mean"])     # its BN buffers keep running


def test_family_real_data_training_and_sampling(tmp_path):
    """`main.py --cfg ... --data_dir ...` on tiny real-data trees (the file formats of the three reference TextDatasets,
    tests/stackgan_data_cases.py): one epoch through DataLoader -> prepare_batch -> the engine, then TRAIN.FLAG: False ->
    `GANTrainer.sample` from the written checkpoint (S/trainer.py:287-419, C/trainer.py:198-295, M/trainer.py:208-343): the
    PNG grids exist, have the reference's layout (10 columns; clevr / mnist: a second row with the label text) and hold
    finite, non-constant samples."""
    import glob
    import importlib
    from PIL import Image
    import stackgan_data_cases as C
    from mogan_amd.stackgan import t7

    def run(pkg, yml_text, name, extra=()):
        entry = importlib.import_module("mogan_amd.stackgan.%s.main" % pkg)
        yml = tmp_path / (name + ".yml")
        yml.write_text(yml_text)
        out = tmp_path / name
        entry.main(["--cfg", str(yml), "--manualSeed", "3", "--output_dir", str(out)] + list(extra))
        return sorted(glob.glob(str(out / "Model" / "checkpoint_*.pth")))

    def check_grid(path, rows, imsize, channels_equal):
        im = np.asarray(Image.open(path).convert("RGB"), dtype=np.float32)
        assert im.shape[:2] == (rows * (imsize + 2) + 2, 10 * (imsize + 2) + 2), (path, im.shape)
        tiles = [im[2:2 + imsize, 2 + k * (imsize + 2): 2 + k * (imsize + 2) + imsize] for k in range(10)]
        # tile 0 = the real image (random pixels), tiles 1..9 = the samples (a one-epoch generator of width 4: near-grey)
        assert tiles[0].std() > 20.0 and all(t.std() < tiles[0].std() for t in tiles[1:]), path
        if rows == 2:                                                                       # the text strip: mostly white
            strip = im[2 + imsize + 2: 2 + imsize + 2 + imsize]
            assert (strip > 250).mean() > 0.8 and (strip < 200).any()     # (black text sits at mid-grey: global min / max)

    common = "GPU_ID: '0'\nZ_DIM: 100\nWORKERS: 0\nUSE_BBOX_LAYOUT: True\n"
    train = "TRAIN: {FLAG: True, BATCH_SIZE: 3, MAX_EPOCH: 1, LR_DECAY_EPOCH: 1, SNAPSHOT_INTERVAL: 1}\n"
    off = "TRAIN: {FLAG: False, BATCH_SIZE: 1}\n"
    # ---- clevr
    gan = "GAN: {CONDITION_DIM: 16, DF_DIM: 4, GF_DIM: 4}\n"
    d = C.build_clevr_tree(str(tmp_path))
    ck = run("clevr", common + train + gan, "clevr", ["--data_dir", d])[-1]
    run("clevr", common + off + gan + "NET_G: '%s'\n" % ck, "clevr_sample", ["--data_dir", d])
    files = sorted(glob.glob(ck[:-4] + "_samples_4_objects/vis_*.png"))
    assert len(files) == C.N_ITEMS                                     # the test split's six scenes (num_samples = 25 > 6)
    for f in files:
        check_grid(f, 2, 64, False)
    # ---- multi-mnist
    gan = "GAN: {CONDITION_DIM: 128, DF_DIM: 4, GF_DIM: 4}\n"
    d = C.build_mnist_tree(str(tmp_path))
    ck = run("multi_mnist", common + train + gan, "mnist", ["--data_dir", d])[-1]
    run("multi_mnist", common + off + gan + "NET_G: '%s'\n" % ck, "mnist_sample", ["--data_dir", d])
    files = sorted(glob.glob(ck[:-4] + "_samples_3_digits/vis_*.png"))
    assert len(files) == 25
    check_grid(files[0], 2, 64, True)
    check_grid(files[-1], 2, 64, True)
    # ---- coco stage I, stage II
    d, img_dir, raw = C.build_coco_tree(str(tmp_path))
    caps = ["caption number %d / of the test split" % i for i in range(C.N_ITEMS)]
    t7.save(os.path.join(d, "test", "val_captions.t7"),
            {"raw_txt": caps, "fea_txt": [raw["emb"][i, :1].copy() for i in range(C.N_ITEMS)]})
    coco = common + "IMG_DIR: '%s'\nTEXT: {DIMENSION: 16}\n" % img_dir
    gan1 = "GAN: {CONDITION_DIM: 128, DF_DIM: 4, GF_DIM: 192}\n"
    gan2 = "GAN: {CONDITION_DIM: 128, DF_DIM: 4, GF_DIM: 192, R_NUM: 1}\n"
    tr1 = train.replace("}", ", COEFF: {KL: 2.0}}")
    s1 = run("coco", coco + "STAGE: 1\nIMSIZE: 64\n" + tr1 + gan1, "s1", ["--data_dir", d])[-1]
    run("coco", coco + "STAGE: 1\nIMSIZE: 64\n" + off + gan1 + "NET_G: '%s'\n" % s1, "s1_sample", ["--data_dir", d])
    files = sorted(glob.glob(s1[:-4] + "_visualize_bbox/*.png"))
    assert 1 <= len(files) <= C.N_ITEMS and all("caption number" in os.path.basename(f) for f in files)
    for f in files:
        check_grid(f, 1, 64, False)
    tr2 = tr1.replace("BATCH_SIZE: 3", "BATCH_SIZE: 2")
    # (NET_G: '' -- the tree's cfg is one global per process, and the sampling run above has set it)
    s2 = run("coco", coco + "STAGE: 2\nIMSIZE: 256\nNET_G: ''\nSTAGE1_G: '%s'\n" % s1 + tr2 + gan2, "s2", ["--data_dir", d])[-1]
    run("coco", coco + "STAGE: 2\nIMSIZE: 256\n" + off + gan2 + "NET_G: '%s'\n" % s2, "s2_sample", ["--data_dir", d])
    files = sorted(glob.glob(s2[:-4] + "_visualize_bbox/*.png"))
    assert 1 <= len(files) <= C.N_ITEMS
    check_grid(files[0], 1, 256, False)
