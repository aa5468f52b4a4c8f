#!/usr/bin/env python
"""Per-config numbers for the secondary BASELINE.json configs ("also report per-config
numbers for configs 2, 3"): images/sec of one G+D train step of the StackGAN-family trees on ONE MI355X, same
timing contract as bench.py (W untimed warm-up steps, K timed steps between synchronizes, inputs resident in
HBM, fp32, random-init networks at the reference yml widths), one JSON line per workload with the same
`roofline` (fp32-MFMA kernel families, HIP-event timed) and `cpu_baseline` (oracle/stackgan_oracle.py on the
host cores, bounded sample) objects.

    python tools/bench_family.py --workload clevr coco_s2 --steps 20 --warmup 5

workloads: mnist   (config 0: Multi-MNIST 1x64x64, 3 digits, batch 64)
           clevr   (config 1: CLEVR 64x64, 4 objects, 13-dim labels, batch 32; the reference's clevr tree has
                    no 128x128 stage -- code/clevr/model.py defines STAGE1_G/STAGE1_D only)
           coco_s1 (coco-stackgan stage I 64x64, batch 128 of coco_s1_train.yml)
           coco_s2 (config 2: coco-stackgan stage II 256x256, char-CNN-RNN embeddings, batch 24)
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (loads the package, roofline_leg)
from mogan_amd.attngan.synthetic import to_device  # noqa: E402
from mogan_amd.stackgan import synthetic  # noqa: E402
from mogan_amd.stackgan.engine import StackGANEngine  # noqa: E402
from mogan_amd.stackgan.trainer_base import weights_init  # noqa: E402

WORKLOADS = {
    "mnist": ("mnist", 1, 64, "multi_mnist", "mnist_train.yml", "Multi-MNIST StackGAN 1x64x64, 3 digits"),
    "clevr": ("clevr", 1, 32, "clevr", "clevr_train.yml", "CLEVR StackGAN 64x64, 4 objects, 13-dim labels"),
    "coco_s1": ("coco", 1, 128, "coco", "coco_s1_train.yml", "MS-COCO StackGAN stage-I 64x64"),
    "coco_s2": ("coco", 2, 24, "coco", "coco_s2_train.yml",
                "MS-COCO StackGAN stage-II 256x256, char-CNN-RNN embeddings (synthetic)"),
}


def build(name, device, batch=None):
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
    G.apply(weights_init)
    D.apply(weights_init)
    return tree, stage, B, cfg, model, G.to(device), D.to(device), desc


def cpu_baseline(engine, cfg, tree, stage, B, bt_cpu):
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
    times, t_all = [], time.perf_counter()
    for i in range(4):
        t0 = time.perf_counter()
        S.train_step(st, bt_cpu)
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - t_all > 25.0:
            break
    timed = times[1:] if len(times) > 1 else times
    sec = sum(timed) / len(timed)
    return dict(value=B / sec, unit="images/s", cores=torch.get_num_threads(), kind="port",
                sample="%d timed step(s) of the same B=%d workload after %d warm-up step(s) "
                       "(oracle/stackgan_oracle.py, torch-CPU fp32, %.2f s/step)" % (len(timed), B,
                                                                                      len(times) - len(timed), sec))


def run(name, args, device):
    tree, stage, B, cfg, model, G, D, desc = build(name, device, args.batch)
    engine = StackGANEngine(G, D, cfg, model.VARIANT, stage=stage, use_graph=not args.no_graph)
    cd = cfg.GAN.CONDITION_DIM
    bt_cpu = synthetic.make_batch(tree, B, stage=stage, seed=0, cond_dim=cd,
                                  text_dim=cfg.TEXT.DIMENSION if tree == "coco" else 0)
    batch = to_device(bt_cpu, device)
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
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": desc + ", G+D train step, widths GF %d / DF %d, fp32" % (cfg.GAN.GF_DIM, cfg.GAN.DF_DIM),
                      "batch_per_gpu": B, "global_batch": B, "parallelism": "dp1",
                      "launch": "hipGraph" if engine.use_graph else "eager"},
           "losses": {k: float(v) for k, v in logs.items() if torch.is_tensor(v) and v.dim() == 0}}
    if not args.no_roofline:
        rows, eager_ms = bench.roofline_leg(engine, run_step)
        tot_ms = sum(r["ms_per_step"] for r in rows)
        tot_gf = sum(r["gflop_per_step"] for r in rows)
        dom = rows[0] if rows else None
        out["roofline"] = {"bound": "mfma", "kernel": dom and dom["kernel"], "achieved": dom and dom["tflops"],
                           "peak": bench.PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                           "frac": dom and dom["tflops"] / bench.PEAK_F32_MFMA_TFLOPS, "traffic": None,
                           "all_gemm": {"gflop_per_step": tot_gf, "gflop_per_image": tot_gf / B, "ms_per_step": tot_ms,
                                        "achieved": tot_gf / tot_ms if tot_ms else 0.0,
                                        "frac": (tot_gf / tot_ms) / bench.PEAK_F32_MFMA_TFLOPS if tot_ms else 0.0,
                                        "share_of_step_ms": tot_ms / ms},
                           "eager_ms_per_step": eager_ms,
                           "kernels": [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()}
                                       for r in rows[:8]]}
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(engine, cfg, tree, stage, B, bt_cpu)
    print(json.dumps(out), flush=True)
    del engine, G, D
    torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", nargs="+", default=["clevr", "coco_s2"], choices=list(WORKLOADS))
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    for name in args.workload:
        run(name, args, device)


if __name__ == "__main__":
    main()
