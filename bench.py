#!/usr/bin/env python
"""bench.py -- images/sec of one coco-attngan G+D train step (256x256) on N MI355X of one node.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N --steps K --warmup W          (re-executes itself under torch.distributed.run, N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (code/coco/attngan/trainer.py:281-342) over one synthetic
minibatch of B=16 per GPU: text-encode, G forward, 3 x (D zero_grad, discriminator_loss, backward,
Adam), generator_loss incl. Inception + DAMSM words/sentence losses + KL, backward, Adam, EMA.
fp32 everywhere, random-init networks of the full coco_train.yml widths, inputs resident in HBM.
W untimed warm-up steps, then exactly K steps bracketed by barrier + torch.cuda.synchronize();
elapsed = MAX over ranks; rank 0 prints ONE JSON line.  Extra objects on that line:
  roofline      dominant MFMA kernel family by time (round 4: the implicit GEMM runs as gemm_kernel, the weight-heavy layers as
                pgemm_kernel, the frozen Inception trunk as pgemm_group_kernel; all listed under `families`, their sum under
                `implicit_gemm_layers`): algorithmic fp32 flops per launch (2*M*N*K
                of the true GEMM dims) / average launch duration measured live with HIP events on the launch
                stream (mogan_prof_*).  `peak` = the matrix-pipe peak of the form the library computes fp32
                products in (mogan_mfma_form): split-bf16 = 2500 TFLOP/s dense bf16 / 6 partial products per
                fp32 product = 416.7; native v_mfma_f32_32x32x2_f32 = 157.3 (also given as `frac_f32_mfma`).
  cpu_baseline  the CPU oracle (oracle/attngan_oracle.py, a torch-CPU port of the reference step)
                timed on this host's cores on the same workload (rank 0, N=1 only): 1 warm-up + 2 timed steps.
  parity        that warm-up step against ONE HIP step from the same weights with the same z / eps: relative
                difference of every loss, max-abs difference of the generated 64x64 / 256x256 images.
"""
import argparse
import json
import os
import sys
import time


def _self_launch():
    """`python bench.py --gpus N` with N > 1 and no launcher environment: become the launcher.  The process is replaced by
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py
    <same arguments>` (one rank per GPU, what train.sh:25-29 / trainer.py:296 hand to nn.parallel.data_parallel in the
    reference), so the scaling run is ONE command either way; rank 0 still prints the one JSON line."""
    if "RANK" in os.environ or "WORLD_SIZE" in os.environ:
        return
    n = 1
    for i, a in enumerate(sys.argv):
        if a == "--gpus" and i + 1 < len(sys.argv):
            n = int(sys.argv[i + 1])
        elif a.startswith("--gpus="):
            n = int(a.split("=", 1)[1])
    if n <= 1:
        return
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


if __name__ == "__main__":
    _self_launch()

# The eager multi-stream AttnGAN step runs on a fixed hardware-queue arrangement (mogan_amd/hip/lib.py, "hardware queues":
# 4 queues; no idle streams ahead of the engine's in a single process, 3 in a process group); must be in the
# environment before the HIP runtime starts.  The captured (--graph) step and the secondary workloads keep the runtime's default.
_EAGER_ATTNGAN = "--graph" not in sys.argv and not any(a.startswith("--workload") for a in sys.argv)
if _EAGER_ATTNGAN:
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import mogan_loader  # noqa: E402

mogan_loader.load()
from mogan_amd.attngan import synthetic  # noqa: E402
from mogan_amd.attngan.miscc.config import cfg, set_coco_train_defaults  # noqa: E402
from mogan_amd.attngan.trainer import TrainEngine, build_networks  # noqa: E402
from mogan_amd.hip import lib  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3       # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32
PEAK_BF16_MFMA_TFLOPS = 2500.0     # same guide: dense bf16 (v_mfma_f32_32x32x16_bf16), no sparsity


def dtype_note():
    if int(lib.load().mogan_mfma_form()) == 6:
        return ("fp32 tensors and accumulators; every fp32 product is formed on the bf16 matrix pipe from the exact 3-piece "
                "bf16 split of both operands (6 partial products, dropped terms <= 2^-23 |ab|): error against fp64 equal to "
                "the native fp32-MFMA build's (tools/diag_x6_precision.py, DESIGN.md section 4)")
    return "fp32 everywhere (native v_mfma_f32_32x32x2_f32)"


def mfma_peak():
    """(peak TFLOP/s of fp32-equivalent flops, description) of the MFMA form libmogan_hip.so was built with"""
    form = int(lib.load().mogan_mfma_form())
    if form == 6:
        return PEAK_BF16_MFMA_TFLOPS / 6.0, "split-bf16: 3 bf16 pieces per fp32 operand, 6 v_mfma_f32_32x32x16_bf16 partial " \
                                            "products per fp32 product (2500 TFLOP/s dense bf16 / 6)"
    return PEAK_F32_MFMA_TFLOPS, "native v_mfma_f32_32x32x2_f32"
MODES = ("conv_fwd", "conv_dgrad", "conv_wgrad", "bmm", "dconv_fwd", "dconv_dgrad", "dconv_wgrad")
TILES = ("128x128", "96x128", "128x32", "32x128", "64x64", "128x64", "64x128")


def make_device_batch(B, seed, device):
    bt = synthetic.make_batch(B, words_num=cfg.TEXT.WORDS_NUM, nef=cfg.TEXT.EMBEDDING_DIM, seed=seed,
                              text="tokens")
    cap_lens_cpu = bt["cap_lens"].clone()
    b = synthetic.to_device(bt, device)
    b["cap_lens_cpu"] = cap_lens_cpu
    b["cap_lens"] = b["cap_lens"].to(torch.int32)
    return b, bt


def load_traffic():
    """HBM bytes per launch per kernel family from the newest committed PMC pass (profiles/r<NN>_pmc_traffic.json:
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over this same command); launch-weighted over the instantiations."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not found:
        return {}, None
    path = found[-1]
    import hashlib
    # the counters cannot be read inside a timed run (rocprofv3 --pmc serialises the kernels): `traffic` is a lookup in the
    # newest committed counter pass -- the line says which file, so a reader can tell a stale pass from a current one
    source = {"file": os.path.relpath(path, ROOT), "sha256_16": hashlib.sha256(open(path, "rb").read()).hexdigest()[:16],
              "measured_in_this_run": False}
    acc = {}
    for name, v in json.load(open(path))["kernels"].items():
        fam = name.split("<")[0]
        n = v.get("launches") or 0
        a = acc.setdefault(fam, [0.0, 0])
        # fetch x2: the guide's gfx950 correction for 16-byte coalesced reads (an upper bound here, see tools/pmc_traffic.py)
        a[0] += (2.0 * (v.get("fetch_kb_per_launch") or 0.0) + (v.get("write_kb_per_launch") or 0.0)) * 1024.0 * n
        a[1] += n
    return {f: (b / n if n else None) for f, (b, n) in acc.items()}, source


def roofline_leg(engine, run_step, steps=2):
    """Eager steps with every MFMA-kernel launch bracketed by HIP events on its stream.  The branches are kept on
    ONE stream for this leg (no side streams), so a launch's duration is that kernel alone on the GPU."""
    import ctypes
    from mogan_amd.hip import ops
    g, ms_, wg = engine.use_graph, getattr(engine, "multi_stream", False), ops._WGRAD_ENV
    ge = getattr(engine, "graph_encoder", False)
    bgr = getattr(engine, "branch_graphs", False)
    if hasattr(engine, "branch_graphs"):
        engine.branch_graphs = False                  # replayed graphs bypass the per-launch hooks as well
    engine.use_graph = False
    if hasattr(engine, "multi_stream"):
        engine.multi_stream = False
    if hasattr(engine, "graph_encoder"):
        engine.graph_encoder = False            # a replayed graph bypasses the per-launch hooks: launch the encoder eagerly
    ops._WGRAD_ENV = False
    run_step()                                        # eager warm-up (allocator)
    torch.cuda.synchronize()
    lib.call("mogan_prof_enable", 1)
    t0 = time.perf_counter()
    for _ in range(steps):
        run_step()
    torch.cuda.synchronize()
    wall_ms = (time.perf_counter() - t0) * 1e3 / steps
    buf = (ctypes.c_double * (5 * 64))()
    n = lib.load().mogan_prof_collect(ctypes.cast(buf, ctypes.c_void_p), 64)
    if os.environ.get("MOGAN_LAYERS_CSV"):          # per-launch list of one more step (input of tools/tune_gemm.py)
        run_step(); torch.cuda.synchronize()
        lib.call("mogan_prof_dump", os.environ["MOGAN_LAYERS_CSV"].encode())
    lib.call("mogan_prof_enable", 0)
    engine.use_graph = g
    if hasattr(engine, "multi_stream"):
        engine.multi_stream = ms_
    if hasattr(engine, "graph_encoder"):
        engine.graph_encoder = ge
    if hasattr(engine, "branch_graphs"):
        engine.branch_graphs = bgr
    ops._WGRAD_ENV = wg
    rows = []
    for i in range(n):
        m, c, launches, flops, ms = buf[5 * i:5 * i + 5]
        if int(m) >= 10:        # grouped launches of the same GEMM: the frozen Inception trunk on pixel panels (attngan/inception.py)
            name = "pgemm_group_kernel<%s,%s>" % ({10: "fwd", 11: "dgrad"}[int(m)], ("128x64", "64x128")[int(c)])
        elif int(m) >= 7:       # packed-weight path of the deep discriminator layers (csrc/mogan_pgemm.hip)
            name = "pgemm_kernel<%s,%s>" % ({7: "fwd", 8: "dgrad", 9: "wgrad"}[int(m)], ("128x64", "128x128", "256x64")[int(c)])
        elif m < 4:
            name = "gemm_kernel<%s,%s>" % (MODES[int(m)], TILES[int(c)])
        elif int(m) == 6:
            name = "wino_wgrad_kernel" if int(c) == 1 else "dconv_wgrad_kernel"
        elif int(c) == 3:       # fused Winograd F(2x2,3x3), 16 waves per block (round 6): flops = the 16/36 of the direct multiplies it executes
            name = "wino5_fwd_kernel<%s>" % MODES[int(m)][6:]
        elif int(c) == 1:       # the 8-wave form (ragged tile grids, fused affine epilogue: the frozen encoder's stem)
            name = "wino3_fwd_kernel<%s>" % MODES[int(m)][6:]
        elif int(c) == 2:       # the pre-split direct kernel (csrc/mogan_dconv2.hip, round 5)
            name = "dconv2_fwd_kernel<%s>" % MODES[int(m)][6:]
        else:
            name = "dconv_fwd_kernel<%s>" % MODES[int(m)][6:]
        rows.append(dict(kernel=name,
                         launches_per_step=launches / steps, gflop_per_step=flops / steps / 1e9,
                         ms_per_step=ms / steps, tflops=(flops / 1e12) / (ms / 1e3) if ms > 0 else 0.0))
    rows.sort(key=lambda r: -r["ms_per_step"])
    return rows, wall_ms


def cpu_baseline_leg(engine, bt_cpu, dev_batch, B, device, timed_steps=2, budget_s=150.0):
    """The oracle (torch-CPU port of the reference step) on this host, same workload: one warm-up step + `timed_steps`
    timed ones.  The warm-up step doubles as the full-width PARITY check of the benchmarked computation: the HIP engine
    and the oracle start from the same weights (the engine's, after the timed region; Adam moments reset on both
    sides), get the same z / eps, and run one whole train step each -- returned as `parity`: relative difference of
    every loss, max-abs difference of the 64x64 and 256x256 generated images (stated fp32 tolerances: losses 1e-4 --
    errG carries the DAMSM terms weighted by 50 --, images 1e-3 after the ~100-layer generator; asserted finite)."""
    from oracle import attngan_oracle as O
    from oracle import inception_oracle as IO
    ocfg = O.Cfg(gf_dim=cfg.GAN.GF_DIM, df_dim=cfg.GAN.DF_DIM, emb_dim=cfg.TEXT.EMBEDDING_DIM,
                 r_num=cfg.GAN.R_NUM, words_num=cfg.TEXT.WORDS_NUM)
    cpu = lambda sd: {k: v.detach().cpu().clone() for k, v in sd.items()}
    torch.cuda.synchronize()
    for o in [engine.optG] + engine.optDs:           # both sides start Adam from step 0
        o.m.zero_(); o.v.zero_(); o.state.zero_()
    st = O.TrainState(O.from_state_dict(cpu(engine.netG.state_dict())),
                      [O.from_state_dict(cpu(d.state_dict())) for d in engine.netsD], ocfg)
    enc_sd = cpu(engine.image_encoder.state_dict())
    batch = dict(bt_cpu)
    batch["words_embs"] = dev_batch["words_embs"].cpu()
    batch["sent_emb"] = dev_batch["sent_emb"].cpu()
    batch["mask"] = dev_batch["mask"].cpu()
    enc = lambda x: IO.cnn_encoder(enc_sd, x)
    gen = torch.Generator().manual_seed(4242)
    batch["z"] = torch.randn(B, cfg.GAN.Z_DIM, generator=gen)
    batch["eps"] = torch.randn(B, cfg.GAN.CONDITION_DIM, generator=gen)
    # the HIP step on exactly these inputs
    hb = {k: v for k, v in dev_batch.items() if k != "inputs_ready"}
    hb["z"], hb["eps"] = batch["z"].to(device), batch["eps"].to(device)
    hlogs = engine.step(hb)
    torch.cuda.synchronize()
    times = []
    t_all = time.perf_counter()
    ologs = None
    for i in range(1 + timed_steps):
        if i > 0:
            batch["z"] = torch.randn(B, cfg.GAN.Z_DIM, generator=gen)
            batch["eps"] = torch.randn(B, cfg.GAN.CONDITION_DIM, generator=gen)
        t0 = time.perf_counter()
        logs = O.train_step(st, batch, enc)
        times.append(time.perf_counter() - t0)
        if i == 0:
            ologs = logs
        if time.perf_counter() - t_all > budget_s and len(times) >= 2:
            break
    parity = {}
    for k in ("errD0", "errD1", "errD2", "errG", "kl", "w_loss", "s_loss"):
        h, o = float(hlogs[k]), float(ologs[k])
        assert h == h and abs(h) != float("inf"), "non-finite %s in the HIP step" % k
        parity[k + "_rel"] = abs(h - o) / (abs(o) + 1e-30)
    for k, name in (("fake64", "img64_max_abs"), ("fake_last", "img256_max_abs")):
        assert torch.isfinite(hlogs[k]).all(), "non-finite %s in the HIP step" % k
        parity[name] = float((hlogs[k].detach().cpu() - ologs[k]).abs().max())
    parity["hip"] = {k: float(v) for k, v in hlogs.items() if torch.is_tensor(v) and v.dim() == 0}
    parity["oracle"] = {k: float(v) for k, v in ologs.items() if not torch.is_tensor(v)}
    parity["ok"] = bool(all(parity[k + "_rel"] <= 1e-4 for k in ("errD0", "errD1", "errD2", "errG", "w_loss", "s_loss"))
                        and parity["img64_max_abs"] <= 1e-3 and parity["img256_max_abs"] <= 1e-3)
    parity["what"] = ("one full-width B=%d train step from the benchmarked engine's weights, same z/eps: HIP vs the CPU oracle "
                      "(losses: relative difference; images: max-abs)" % B)
    timed = times[1:] if len(times) > 1 else times
    sec = sorted(timed)[len(timed) // 2] if len(timed) % 2 else sum(timed) / len(timed)
    base = dict(value=B / sec, unit="images/s", cores=torch.get_num_threads(), kind="port",
                sample="%d timed step(s) of the same B=%d workload after %d warm-up step(s) "
                       "(oracle/attngan_oracle.py + inception_oracle.py, torch-CPU fp32, %.2f s/step; per-step %s)"
                       % (len(timed), B, len(times) - len(timed), sec, ["%.1f" % t for t in times]))
    return base, parity


# ------------------------------------------------------------------ secondary BASELINE configs (StackGAN family)
WORKLOADS = {
    "mnist": ("mnist", 1, 64, "multi_mnist", "mnist_train.yml", "Multi-MNIST StackGAN 1x64x64, 3 digits"),
    "clevr": ("clevr", 1, 32, "clevr", "clevr_train.yml", "CLEVR StackGAN 64x64, 4 objects, 13-dim labels"),
    "coco_s1": ("coco", 1, 128, "coco", "coco_s1_train.yml", "MS-COCO StackGAN stage-I 64x64"),
    "coco_s2": ("coco", 2, 24, "coco", "coco_s2_train.yml",
                "MS-COCO StackGAN stage-II 256x256, char-CNN-RNN embeddings (synthetic)"),
}


def family_build(name, device, batch=None):
    import importlib
    tree, stage, B, pkg, yml, desc = WORKLOADS[name]
    B = batch or B
    model = importlib.import_module("mogan_amd.stackgan.%s.model" % pkg)
    config = importlib.import_module("mogan_amd.stackgan.%s.miscc.config" % pkg)
    config.cfg_from_file(os.path.join(os.path.dirname(model.__file__), "cfg", yml))
    cfg = config.cfg
    torch.manual_seed(1234)
    if stage == 2:
        G, D = model.STAGE2_G(model.STAGE1_G()), model.STAGE2_D()
    else:
        G, D = model.STAGE1_G(), model.STAGE1_D()
    from mogan_amd.stackgan.trainer_base import weights_init
    G.apply(weights_init)
    D.apply(weights_init)
    return tree, stage, B, cfg, model, G.to(device), D.to(device), desc


def family_cpu_baseline(engine, cfg, tree, stage, B, bt_cpu, dev_batch=None, timed_steps=3, budget_s=25.0):
    from oracle import stackgan_oracle as S
    ocfg = S.SCfg(tree, stage=stage, gf_dim=cfg.GAN.GF_DIM, df_dim=cfg.GAN.DF_DIM, cond_dim=cfg.GAN.CONDITION_DIM,
                  text_dim=cfg.TEXT.DIMENSION if tree == "coco" else 0, r_num=cfg.GAN.R_NUM)
    cpu = lambda sd: {k: v.detach().cpu().clone() for k, v in sd.items()}
    G = S.from_state_dict(cpu(engine.netG.state_dict()))
    if stage == 2:
        for k, v in G.items():
            if k.startswith("STAGE1_G.") and v.is_floating_point():
                v.requires_grad_(False)
    st = S.TrainState(G, S.from_state_dict(cpu(engine.netD.state_dict())), ocfg)
    # PARITY of the benchmarked computation at its stated size: the HIP engine and the oracle start from the same weights (the
    # engine's, after the timed region; Adam moments reset on both sides), get the same batch incl. z / eps, and run one whole
    # train step each (the oracle's is also its warm-up step)
    hlogs = None
    if dev_batch is not None:
        torch.cuda.synchronize()
        for o in (engine.optG, engine.optD):
            o.m.zero_(); o.v.zero_(); o.state.zero_()
        hlogs = {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in engine.step(dict(dev_batch)).items()}
        torch.cuda.synchronize()
    times, t_all, ologs = [], time.perf_counter(), None
    for i in range(1 + timed_steps):
        t0 = time.perf_counter()
        logs = S.train_step(st, bt_cpu)
        times.append(time.perf_counter() - t0)
        if i == 0:
            ologs = logs
        if time.perf_counter() - t_all > budget_s and len(times) >= 2:
            break
    timed = times[1:] if len(times) > 1 else times
    sec = sum(timed) / len(timed)
    base = dict(value=B / sec, unit="images/s", cores=torch.get_num_threads(), kind="port",
                sample="%d timed step(s) of the same B=%d workload after %d warm-up step(s) "
                       "(oracle/stackgan_oracle.py, torch-CPU fp32, %.2f s/step)" % (len(timed), B,
                                                                                      len(times) - len(timed), sec))
    if hlogs is None:
        return base, None
    parity = {}
    keys = [k for k in ("errD", "errD_real", "errD_wrong", "errD_fake", "errG", "kl") if k in ologs and k in hlogs]
    for k in keys:
        h, o = float(hlogs[k]), float(ologs[k])
        assert h == h and abs(h) != float("inf"), "non-finite %s in the HIP step" % k
        parity[k + "_rel"] = abs(h - o) / (abs(o) + 1e-30)
    assert torch.isfinite(hlogs["fake"]).all(), "non-finite generated images in the HIP step"
    parity["img_max_abs"] = float((hlogs["fake"].cpu() - ologs["fake"]).abs().max())
    parity["hip"] = {k: float(hlogs[k]) for k in keys}
    parity["oracle"] = {k: float(ologs[k]) for k in keys}
    parity["ok"] = bool(all(parity[k + "_rel"] <= 1e-4 for k in keys) and parity["img_max_abs"] <= 1e-3)
    parity["what"] = ("one B=%d train step at the yml widths from the benchmarked engine's weights, same batch / z / eps: HIP vs "
                      "the CPU oracle (losses: relative difference; generated images: max-abs)" % B)
    return base, parity


def run_family(name, args, device):
    tree, stage, B, cfg, model, G, D, desc = family_build(name, device, args.family_batch)
    from mogan_amd.stackgan.engine import StackGANEngine
    from mogan_amd.stackgan import synthetic as fsyn
    # the 64x64 trees are launch-bound (one hipGraph replay wins); stage II at 256x256 is GPU-bound (eager + wgrad side stream)
    use_graph = args.graph or (name != "coco_s2" and not args.no_graph)
    engine = StackGANEngine(G, D, cfg, model.VARIANT, stage=stage, use_graph=use_graph)
    cd = cfg.GAN.CONDITION_DIM
    bt_cpu = fsyn.make_batch(tree, B, stage=stage, seed=0, cond_dim=cd,
                                  text_dim=cfg.TEXT.DIMENSION if tree == "coco" else 0)
    batch = synthetic.to_device(bt_cpu, device)
    gen = torch.Generator(device=device).manual_seed(1000)

    def run_step():
        b = dict(batch)
        b["z"] = torch.randn(B, cfg.Z_DIM, device=device, generator=gen)
        if tree == "coco":
            b["eps"] = torch.randn(B, cd, device=device, generator=gen)
            if stage == 2:
                b["eps_s1"] = torch.randn(B, cd, device=device, generator=gen)
        return engine.step(b)

    for _ in range(args.warmup):
        run_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        logs = run_step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ms = elapsed / args.steps * 1e3
    out = {"metric": "images/sec per G+D train step, %s" % name, "value": B * args.steps / elapsed,
           "unit": "images/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "dtype_note": dtype_note(), "data": "synthetic",
           "config": {"workload": desc + ", G+D train step, widths GF %d / DF %d, fp32" % (cfg.GAN.GF_DIM, cfg.GAN.DF_DIM),
                      "batch_per_gpu": B, "global_batch": B, "parallelism": "dp1",
                      "launch": "hipGraph" if engine.use_graph else "eager"},
           "losses": {k: float(v) for k, v in logs.items() if torch.is_tensor(v) and v.dim() == 0}}
    if not args.no_roofline:
        rows, eager_ms = roofline_leg(engine, run_step)
        tot_ms = sum(r["ms_per_step"] for r in rows)
        tot_gf = sum(r["gflop_per_step"] for r in rows)
        dom = rows[0] if rows else None
        peak, peak_note = mfma_peak()
        out["roofline"] = {"bound": "mfma", "kernel": dom and dom["kernel"], "achieved": dom and dom["tflops"],
                           "peak": peak, "peak_note": peak_note, "unit": "TFLOP/s",
                           "frac": dom and dom["tflops"] / peak, "frac_f32_mfma": dom and dom["tflops"] / PEAK_F32_MFMA_TFLOPS,
                           "traffic": None,
                           "all_gemm": {"gflop_per_step": tot_gf, "gflop_per_image": tot_gf / B, "ms_per_step": tot_ms,
                                        "achieved": tot_gf / tot_ms if tot_ms else 0.0,
                                        "frac": (tot_gf / tot_ms) / peak if tot_ms else 0.0,
                                        "share_of_step_ms": tot_ms / ms},
                           "eager_ms_per_step": eager_ms,
                           "kernels": [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()}
                                       for r in rows[:8]]}
    if not args.no_cpu_baseline:
        out["cpu_baseline"], out["parity"] = family_cpu_baseline(engine, cfg, tree, stage, B, bt_cpu, dev_batch=batch)
    print(json.dumps(out), flush=True)
    del engine, G, D
    torch.cuda.empty_cache()
    return out



def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16, help="minibatch per GPU (BASELINE config: 16)")
    ap.add_argument("--graph", action="store_true", help="replay the step as one captured hipGraph (N=1 only) instead "
                    "of eager multi-stream launches (the default: measured faster once the independent branches "
                    "and the weight gradients run on side streams)")
    ap.add_argument("--no-graph", action="store_true", help="(default; kept for older command lines)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=2, help="timed steps of the CPU oracle leg after its warm-up step (25-30 s "
                    "each on the GPU box's host; default 2 keeps the default run within minutes, BASELINE.md section 3 asks "
                    "for 5: --cpu-steps 5)")
    ap.add_argument("--no-text-prefetch", action="store_true", help="text-encode every batch at the start of its own step")
    ap.add_argument("--no-inputs-ready", action="store_true", help="A/B: D_i(real) of a step waits for the main stream (the "
                    "tail of the previous step) instead of for the batch's inputs_ready event")
    ap.add_argument("--dp-mode", default="both", choices=["both", "branch_graphs", "eager"],
                    help="N > 1: discriminator branches as hipGraphs, eager, or (default) both measured in one run")
    ap.add_argument("--no-comm-stats", action="store_true", help="N > 1: no timing events around the collectives")
    ap.add_argument("--debug-losses", action="store_true", help="print the losses of every step (adds a host sync)")
    ap.add_argument("--workload", default="attngan", choices=["attngan"] + list(WORKLOADS),
                    help="attngan = the headline metric (default); the others are the secondary BASELINE configs "
                         "(SURVEY.md §8(d): per-config numbers), single GPU")
    ap.add_argument("--family-batch", type=int, default=None, help="override the batch of a secondary workload")
    args = ap.parse_args()
    if args.workload != "attngan":
        device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
        torch.cuda.set_device(device)
        run_family(args.workload, args, device)
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    force_dist = bool(os.environ.get("MOGAN_FORCE_DIST")) and "RANK" in os.environ
    if os.environ.get("MOGAN_ONE_GPU"):                 # control-flow test of the N>1 path on a 1-GPU box (with gloo)
        local = 0
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)                       # before the process group: RCCL binds to the current device
    if _EAGER_ATTNGAN:
        lib.reserve_hw_queues()                         # before any other stream exists (RCCL's included)
        from mogan_amd.attngan import trainer as _tr
        _tr.create_engine_streams()                     # the step's streams, bound to their queues before RCCL's exist
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL's stream on a priority queue of its own: a long all-reduce then cannot hold back the branch streams that would
        # otherwise share its (in-order) hardware queue.  Neutral at world = 1 (299 vs 300 img/s); DESIGN.md section 9 (5)
        os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")
        backend = os.environ.get("MOGAN_DIST_BACKEND", "nccl")      # "nccl" IS RCCL on ROCm
        dist.init_process_group(backend, rank=rank, world_size=world, **({"device_id": device} if backend == "nccl" else {}))
    assert world == args.gpus, "WORLD_SIZE=%d but --gpus %d (plain `python bench.py --gpus N` launches its own ranks)" % (
        world, args.gpus)

    # wall clock per phase of this run -> stderr and `phase_s` of the line (the driver's bench run took 491 s
    # around a 0.73 s timed region).  On the builder's boxes the whole default run is 102 s: initialisation 2 s (the orthogonal
    # init of D_NET256's largest layer is 1.1 s), warm-up + timed region 4 s, roofline leg 0.2 s, CPU oracle + parity 94 s
    # (profiles/r06_ab.txt); what a run spends beyond that is outside this process (first import of torch on a fresh box) or the
    # host's CPU leg -- the line says which
    phase_s, _t_phase = {}, [time.perf_counter()]

    def lap(name):
        now = time.perf_counter()
        phase_s[name] = round(phase_s.get(name, 0.0) + now - _t_phase[0], 2)
        _t_phase[0] = now
        if rank == 0:
            print("[bench] %-28s %7.1f s" % (name, phase_s[name]), file=sys.stderr, flush=True)

    lap("import + device + process group")
    set_coco_train_defaults()
    B = args.batch
    cfg.TRAIN.BATCH_SIZE = B
    text_encoder, image_encoder, netG, netsD = build_networks(device=device, seed=1234)   # identical replicas
    torch.cuda.synchronize()
    lap("build_networks (init)")
    use_graph = (world == 1) and args.graph and not args.no_graph
    dist_on = world > 1 or force_dist
    batch, bt_cpu = make_device_batch(B, seed=rank, device=device)
    gen = torch.Generator(device=device).manual_seed(1000 + rank)
    # the synthetic minibatch is resident in HBM before the timed region: tell the engine, so that the D(real)
    # forwards of step n+1 need not queue behind the tail of step n on the main stream (condGANTrainer.train() records the
    # same event behind its look-ahead batch's host-to-device copies / feeder kernels)
    torch.cuda.synchronize()
    if not args.no_inputs_ready:
        batch["inputs_ready"] = torch.cuda.Event()
        batch["inputs_ready"].record()

    def make_runner(engine):
        def run_step():
            b = dict(batch)
            b["z"] = torch.randn(B, cfg.GAN.Z_DIM, device=device, generator=gen)         # trainer.py:294
            b["eps"] = torch.randn(B, cfg.GAN.CONDITION_DIM, device=device, generator=gen)  # model.py:336
            # the train loop hands the NEXT batch's captions to the engine with every step (one text encoding per step, as in
            # trainer.py:281-289, software-pipelined one step ahead; condGANTrainer.train does the same)
            if not args.no_text_prefetch and not engine.use_graph:
                engine.prefetch_text(batch["captions"], batch["cap_lens_cpu"])
            return engine.step(b)
        return run_step

    def timed_region(engine, run_step):
        """W warm-up steps, then exactly K steps between barrier + synchronize on both sides; elapsed = MAX over ranks"""
        for _ in range(args.warmup):
            run_step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        engine.comm.enabled = dist_on and not args.no_comm_stats
        t0 = time.perf_counter()
        for _ in range(args.steps):
            logs = run_step()
            if args.debug_losses:
                print("step", {k: round(float(v), 4) for k, v in logs.items() if v.dim() == 0}, file=sys.stderr, flush=True)
        host_elapsed = time.perf_counter() - t0         # all launches of the K steps queued (diagnostic: host- or GPU-bound?)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        comm = engine.comm.report(args.steps) if engine.comm.enabled else None
        engine.comm.enabled = False
        return elapsed, host_elapsed, logs, comm

    # Data parallel: the step has two launch modes for the discriminator branches (trainer.TrainEngine) -- replayed hipGraphs with
    # each bucket's all-reduce issued between two replays ("branch_graphs"), or eager launches whose weight-gradient kernels
    # release D_NET256's bucket chunk by chunk while its backward is still running ("eager").  Which one wins depends on how long
    # the collectives take with real peers, so one invocation measures BOTH (same networks, a fresh engine each) and reports the
    # better one as `value`, both under `modes`.
    modes = [None]
    if dist_on:
        modes = [args.dp_mode] if args.dp_mode != "both" else ["branch_graphs", "eager"]
    results, engine = {}, None
    for mode in modes:
        if engine is not None:
            engine.close()
            del engine, run_step
            import gc
            gc.collect()
            torch.cuda.empty_cache()
        if mode is not None:
            os.environ["MOGAN_BRANCH_GRAPHS_DP"] = "1" if mode == "branch_graphs" else "0"
        engine = TrainEngine(text_encoder, image_encoder, netG, netsD, distributed=dist_on, use_graph=use_graph and not force_dist)
        run_step = make_runner(engine)
        lap("engine construction")
        elapsed, host_elapsed, logs, comm = timed_region(engine, run_step)
        lap("warm-up + timed region")
        results[mode] = dict(elapsed=elapsed, host_elapsed=host_elapsed, logs=logs, comm=comm,
                             launch="hipGraph" if engine.use_graph else "%s, %d streams + wgrad side streams" % (
                                 "generator eager + discriminator branches as hipGraphs" if engine.branch_graphs else "eager",
                                 1 + (len(engine.side) if engine.multi_stream else 0)))
    best = min(results, key=lambda m: results[m]["elapsed"])
    elapsed, host_elapsed, logs = (results[best][k] for k in ("elapsed", "host_elapsed", "logs"))

    for k, v in logs.items():                           # a benchmark of a diverged computation is not a benchmark
        if torch.is_tensor(v) and v.dim() == 0:
            assert bool(torch.isfinite(v)), "non-finite %s after the timed steps" % k
    ms = elapsed / args.steps * 1e3
    out = {
        "metric": "images/sec per G+D train step, 256x256 coco-attngan",
        "value": world * B * args.steps / elapsed, "unit": "images/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "host_enqueue_ms_per_step": host_elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "dtype_note": dtype_note(), "data": "synthetic",
        "config": {"workload": "MS-COCO AttnGAN 256x256 G+D train step: G_NET + D_NET64/128/256 + "
                               "GlobalAttentionGeneral + Inception/DAMSM losses (random-init), coco_train.yml "
                               "widths (GF 48, DF 96, T 12), fp32", "batch_per_gpu": B, "global_batch": world * B,
                   "parallelism": "dp%d" % world, "launch": results[best]["launch"]},
        "losses": {k: float(v) for k, v in logs.items() if torch.is_tensor(v) and v.dim() == 0},
    }
    if dist_on:
        # per launch mode: throughput and, per gradient bucket, what its all-reduce cost and how much of it the consumer saw
        out["dp"] = {"backend": dist.get_backend(), "world_size_seen": dist.get_world_size(), "mode_reported": best,
                     "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                     "modes": {m: {"value": world * B * args.steps / r["elapsed"], "ms_per_step": r["elapsed"] / args.steps * 1e3,
                                   "host_enqueue_ms_per_step": r["host_elapsed"] / args.steps * 1e3, "launch": r["launch"],
                                   "buckets": r["comm"]} for m, r in results.items()}}
    rows = None
    if not args.no_roofline:
        # EVERY rank runs these extra steps (they contain the gradient all-reduces: a rank-0-only leg would leave
        # the collectives unmatched and hang for N>1); rank 0 reports
        rows, eager_ms = roofline_leg(engine, run_step)
        if world > 1:
            dist.barrier()
        lap("roofline leg")
    if rank == 0 and rows is not None:
        fams = {}
        for r in rows:                                   # kernel families = the three MFMA kernels of csrc/
            f = r["kernel"].split("<")[0]
            a = fams.setdefault(f, dict(kernel=f, launches_per_step=0.0, gflop_per_step=0.0, ms_per_step=0.0))
            for k in ("launches_per_step", "gflop_per_step", "ms_per_step"):
                a[k] += r[k]
        traffic, traffic_source = load_traffic()
        peak, peak_note = mfma_peak()
        for f, a in fams.items():
            a["tflops"] = a["gflop_per_step"] / a["ms_per_step"] if a["ms_per_step"] else 0.0
            a["frac"] = a["tflops"] / peak
            a["traffic_bytes_per_launch"] = traffic.get(f)
        dom = max(fams.values(), key=lambda a: a["ms_per_step"])
        # the implicit-GEMM layer set of rounds 1-2 (everything the direct / Winograd kernels do not take) is served by two
        # kernels since round 3: gemm_kernel (gathers and splits fp32 operands) and pgemm_kernel (packed weights); their sum is
        # the figure comparable with the earlier rounds' gemm_kernel family
        ig = [fams[k] for k in ("gemm_kernel", "pgemm_kernel", "pgemm_group_kernel") if k in fams]
        ig_ms, ig_gf = sum(a["ms_per_step"] for a in ig), sum(a["gflop_per_step"] for a in ig)
        tot_ms = sum(r["ms_per_step"] for r in rows)
        tot_gf = sum(r["gflop_per_step"] for r in rows)
        out["roofline"] = {
            "bound": "mfma", "kernel": dom["kernel"], "achieved": dom["tflops"], "peak": peak, "peak_note": peak_note,
            "unit": "TFLOP/s", "frac": dom["frac"], "frac_f32_mfma": dom["tflops"] / PEAK_F32_MFMA_TFLOPS,
            "traffic": dom["traffic_bytes_per_launch"], "traffic_source": traffic_source,
            "launches_per_step": dom["launches_per_step"],
            "avg_launch_ms": dom["ms_per_step"] / dom["launches_per_step"],
            "gflop_per_launch": dom["gflop_per_step"] / dom["launches_per_step"],
            "implicit_gemm_layers": {"kernels": "gemm_kernel + pgemm_kernel + pgemm_group_kernel", "gflop_per_step": ig_gf, "ms_per_step": ig_ms,
                                     "achieved": ig_gf / ig_ms if ig_ms else 0.0, "frac": (ig_gf / ig_ms) / peak if ig_ms else 0.0},
            "families": [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in a.items()}
                         for a in sorted(fams.values(), key=lambda a: -a["ms_per_step"])],
            "all_gemm": {"gflop_per_step": tot_gf, "gflop_per_image": tot_gf / B, "ms_per_step": tot_ms,
                         "achieved": tot_gf / tot_ms if tot_ms else 0.0,
                         "frac": (tot_gf / tot_ms) / peak if tot_ms else 0.0,
                         "share_of_step_ms": tot_ms / ms},
            "eager_ms_per_step": eager_ms,
            "kernels": [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()} for r in rows[:10]],
        }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        dev_batch = dict(batch)
        dev_batch.update(engine.encode_batch_for_cpu(batch))
        out["cpu_baseline"], out["parity"] = cpu_baseline_leg(engine, bt_cpu, dev_batch, B, device, timed_steps=args.cpu_steps,
                                                              budget_s=max(150.0, 45.0 * (1 + args.cpu_steps)))
        out["cpu_baseline"]["sample"] += "; --cpu-steps %d%s" % (args.cpu_steps, "" if args.cpu_steps >= 5 else
                                                               " (the default run's bounded sample; --cpu-steps 5 = BASELINE.md's five)")
        lap("cpu_baseline + parity")
    if world > 1 or force_dist:
        dist.destroy_process_group()
    if os.environ.get("MOGAN_CHAIN_EVENTS") and rank == 0:       # diagnostic: where the main stream (generator chain) spends the step
        out["chain_ms"] = engine.chain_report()
    out["phase_s"] = phase_s
    if rank == 0:                      # the JSON line is the last thing on stdout (RCCL prints its banner there too, through C stdio:
        sys.stdout.flush()             # flush that buffer first, or the banner lands behind the line at exit)
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
