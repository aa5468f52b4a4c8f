"""-m gpu: (1) the HIP Inception-v3 trunk (CNN_ENCODER) against its torch-CPU restatement
(oracle/inception_oracle.py; the Inception arithmetic is unpinned by the reference),
forward and input-gradient; (2) the drop-in entry point: main.py -> condGANTrainer.train() on synthetic
batches for two iterations, checkpoint written in the reference's format."""
import glob
import os

import numpy as np
import pytest
import torch

from helpers import ROOT, load_pkg

load_pkg()
from mogan_amd.attngan.miscc.config import cfg  # noqa: E402
from oracle import inception_oracle as IO  # noqa: E402

pytestmark = pytest.mark.gpu


def test_cnn_encoder_vs_cpu_restatement():
    from mogan_amd.attngan import model
    cfg.TRAIN.FLAG, cfg.TEXT.EMBEDDING_DIM = True, 32
    torch.manual_seed(3)
    enc = model.CNN_ENCODER(32)
    # non-trivial eval-mode BN statistics
    for m in enc.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.05); m.running_var.uniform_(0.8, 1.2); m.weight.data.uniform_(0.9, 1.1)
            m.bias.data.normal_(0, 0.05)
    enc.eval()
    sd = {k: v.detach().clone().double() for k, v in enc.state_dict().items()}
    x = (torch.rand(2, 3, 256, 256) * 2 - 1).requires_grad_(True)
    gf, gc = torch.randn(2, 32, 17, 17), torch.randn(2, 32)
    f_ref, c_ref = IO.cnn_encoder(sd, x.double())
    ((f_ref * gf.double()).sum() + (c_ref * gc.double()).sum()).backward()
    enc = enc.cuda()
    xd = x.detach().cuda().requires_grad_(True)
    f, c = enc(xd)
    ((f * gf.cuda()).sum() + (c * gc.cuda()).sum()).backward()
    rel = lambda a, b: float((a.detach().cpu().double() - b.detach()).norm() / b.detach().norm())
    assert tuple(f.shape) == (2, 32, 17, 17) and tuple(c.shape) == (2, 32)
    assert rel(f, f_ref) < 2e-5, rel(f, f_ref)
    assert rel(c, c_ref) < 2e-5, rel(c, c_ref)
    # input gradient: 94 convs deep with ReLU / max-pool kinks -- torch-CPU fp32 vs fp64 of this very restatement
    # already differ by 9.9e-3 rel-L2 on these inputs (decisions that flip at the fp32 noise floor), so that is the scale
    assert rel(xd.grad, x.grad) < 3e-2, rel(xd.grad, x.grad)


def test_main_trains_on_synthetic_and_checkpoints(tmp_path):
    from mogan_amd.attngan import main as entry
    yml = tmp_path / "tiny.yml"
    yml.write_text("CONFIG_NAME: 'tiny'\nDATASET_NAME: 'coco'\nWORKERS: 0\nTREE: {BRANCH_NUM: 3}\n"
                   "GAN: {DF_DIM: 8, GF_DIM: 8, Z_DIM: 100, R_NUM: 1}\n"
                   "TEXT: {EMBEDDING_DIM: 32, CAPTIONS_PER_IMAGE: 5, WORDS_NUM: 6}\n"
                   "TRAIN: {FLAG: True, BATCH_SIZE: 4, MAX_EPOCH: 1, SNAPSHOT_INTERVAL: 1, NET_E: ''}\n")
    out = tmp_path / "out"
    entry.main(["--cfg", str(yml), "--synthetic", "8", "--manualSeed", "7", "--output_dir", str(out)])
    ckpts = sorted(glob.glob(os.path.join(str(out), "Model", "checkpoint_*.pth")))
    assert ckpts, "no checkpoint written"
    sd = torch.load(ckpts[-1], map_location="cpu", weights_only=False)
    assert set(sd) == {"epoch", "netG", "optimG", "netD", "optimD"} and len(sd["netD"]) == 3
    assert "h_net1.upsample1.1.weight" in sd["netG"] and "COND_DNET.jointConv.0.weight" in sd["netD"][2]
    assert all(torch.isfinite(v).all() for v in sd["netG"].values() if v.is_floating_point())
    nbt = sd["netG"]["h_net1.fc.1.num_batches_tracked"]
    assert int(nbt) == 2                                   # two iterations, one BN call each


def test_sampling_from_a_checkpoint(tmp_path):
    """trainer.py:387-470: train one epoch, then load the checkpoint's (EMA) generator in eval mode and write one
    256x256 image per caption."""
    from PIL import Image
    from mogan_amd.attngan import main as entry
    from mogan_amd.attngan.datasets import SyntheticTextDataset
    from mogan_amd.attngan.miscc.config import cfg
    from mogan_amd.attngan.trainer import condGANTrainer
    yml = tmp_path / "tiny.yml"
    yml.write_text("CONFIG_NAME: 'tiny'\nDATASET_NAME: 'coco'\nWORKERS: 0\nTREE: {BRANCH_NUM: 3}\n"
                   "GAN: {DF_DIM: 8, GF_DIM: 8, Z_DIM: 100, R_NUM: 1}\n"
                   "TEXT: {EMBEDDING_DIM: 32, CAPTIONS_PER_IMAGE: 5, WORDS_NUM: 6}\n"
                   "TRAIN: {FLAG: True, BATCH_SIZE: 4, MAX_EPOCH: 1, SNAPSHOT_INTERVAL: 1, NET_E: ''}\n")
    out = tmp_path / "out"
    entry.main(["--cfg", str(yml), "--synthetic", "8", "--manualSeed", "7", "--output_dir", str(out)])
    ckpt = sorted(glob.glob(os.path.join(str(out), "Model", "checkpoint_*.pth")))[-1]
    cfg.TRAIN.NET_G, cfg.TRAIN.NET_E = ckpt, ''
    ds = SyntheticTextDataset(length=4, n_words=100)
    dl = torch.utils.data.DataLoader(ds, batch_size=4, drop_last=True, shuffle=False)
    algo = condGANTrainer(str(out), dl, 100, ds.ixtoword, resume=False)
    written = algo.sampling("valid")
    cfg.TRAIN.NET_G = ''
    assert len(written) == 4 and all(os.path.isfile(p) for p in written)
    im = np.asarray(Image.open(written[0]))
    assert im.shape == (256, 256, 3) and im.dtype == np.uint8 and im.std() > 0


def test_bench_two_ranks_control_flow():
    """bench.py launched as the driver launches it for N>1 (torch.distributed.run, 2 ranks) -- on this 1-GPU box both
    ranks share cuda:0 and the process group is gloo (MOGAN_ONE_GPU / MOGAN_DIST_BACKEND), which exercises everything
    but RCCL itself: matched collectives on every rank in the warm-up, the timed steps AND the roofline leg, the MAX
    over ranks, one JSON line from rank 0."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, MOGAN_ONE_GPU="1", MOGAN_DIST_BACKEND="gloo", MOGAN_FAST_INIT="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(29600 + os.getpid() % 300), os.path.join(ROOT, "bench.py"), "--gpus", "2",
           "--steps", "2", "--warmup", "1"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 32 and d["scaling"] == "weak"
    assert "roofline" in d and "cpu_baseline" not in d and d["value"] > 0


@pytest.mark.parametrize("parallel", [False, True])
def test_graphed_encoder_matches_eager(parallel, monkeypatch):
    """The train engine replays the frozen encoder as a hipGraph pair (optionally with the Mixed-block branches captured
    on parallel streams, inception._parallel): same kernels, so outputs and the image gradient must be identical to
    the eager, single-stream evaluation."""
    from mogan_amd.attngan import inception
    from mogan_amd.attngan.model import CNN_ENCODER
    monkeypatch.setattr(inception, "PARALLEL_BRANCHES", parallel)
    cfg.TRAIN.FLAG = True
    torch.manual_seed(3)
    enc = CNN_ENCODER(32).to("cuda").eval()
    for p in enc.parameters():
        p.requires_grad = False
    x = (torch.rand(2, 3, 256, 256, device="cuda") * 2 - 1)
    gf, gc = torch.randn(2, 32, 17, 17, device="cuda"), torch.randn(2, 32, device="cuda")
    xe = x.clone().requires_grad_(True)
    f, c = enc(xe)
    torch.autograd.backward((f, c), (gf, gc))
    graphed = torch.cuda.make_graphed_callables(enc, (torch.zeros_like(x).requires_grad_(True),))
    for _ in range(2):                                   # replay twice: static buffers are reused
        xg = x.clone().requires_grad_(True)
        f2, c2 = graphed(xg)
        torch.autograd.backward((f2, c2), (gf, gc))
        torch.cuda.synchronize()
        assert torch.equal(f2, f) and torch.equal(c2, c)
        assert torch.allclose(xg.grad, xe.grad, rtol=0, atol=0) or float((xg.grad - xe.grad).abs().max()) < 1e-6 * float(xe.grad.abs().max())
