"""One rank of tests/test_dp_engine_gpu.py (launched by torch.distributed.run, 2 ranks).  On the 1-GPU test box both ranks share
cuda:0 and the process group is gloo; everything else is the N>1 product path: TrainEngine(distributed=True) with the
replica broadcast at construction, per-network flat-bucket all-reduces, 1/world folded into the fused Adam."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import det_fill_state, load_pkg  # noqa: E402
from standin import StandInEncoder  # noqa: E402

load_pkg()
from mogan_amd.attngan import model, synthetic  # noqa: E402
from mogan_amd.attngan.miscc.config import cfg  # noqa: E402
from mogan_amd.attngan.trainer import TrainEngine  # noqa: E402


def small_cfg():
    cfg.GAN.GF_DIM, cfg.GAN.DF_DIM, cfg.GAN.R_NUM, cfg.GAN.Z_DIM = 4, 4, 2, 100
    cfg.TEXT.EMBEDDING_DIM, cfg.TEXT.WORDS_NUM, cfg.TREE.BRANCH_NUM = 16, 5, 3
    cfg.TRAIN.GENERATOR_LR = cfg.TRAIN.DISCRIMINATOR_LR = 2e-4
    cfg.TRAIN.SMOOTH.GAMMA1, cfg.TRAIN.SMOOTH.GAMMA2 = 4.0, 5.0           # cfg/coco_train.yml (= oracle.Cfg defaults)
    cfg.TRAIN.SMOOTH.GAMMA3, cfg.TRAIN.SMOOTH.LAMBDA = 10.0, 50.0
    cfg.STN_ALIGN_CORNERS, cfg.ATT_MASK_MODE, cfg.ADAM_EPS_MODE = False, 0, 0


def full_width(out_dir, rank, world):
    """DP_FULL=1: the benchmark's networks (coco_train.yml widths, the real Inception encoder) on a local batch of 4 -- BASELINE
    config 4's literal shard --, two steps: checksums of every bucket for the replica-equality check, the reducers' chunking."""
    from mogan_amd.attngan.miscc.config import set_coco_train_defaults
    from mogan_amd.attngan.trainer import build_networks
    set_coco_train_defaults()
    cfg.TRAIN.BATCH_SIZE = 4
    os.environ.setdefault("MOGAN_FAST_INIT", "1")
    te, ie, G, Ds = build_networks(device="cuda", seed=1234 + rank)          # different weights per rank on purpose
    eng = TrainEngine(te, ie, G, Ds, distributed=True, use_graph=False)
    logs = None
    for step in range(2):
        bt = synthetic.make_batch(4, words_num=cfg.TEXT.WORDS_NUM, nef=cfg.TEXT.EMBEDDING_DIM, seed=50 + 10 * step + rank,
                                  text="tokens")
        lens = bt["cap_lens"].clone()
        b = synthetic.to_device(bt, "cuda")
        b["cap_lens_cpu"], b["cap_lens"] = lens, b["cap_lens"].to(torch.int32)
        logs = eng.step(b)
    torch.cuda.synchronize()
    sums = {}
    for name, o in [("G", eng.optG)] + [("D%d" % i, o) for i, o in enumerate(eng.optDs)]:
        sums[name] = (float(o.p.double().sum()), float(o.p.double().abs().sum()), float(o.m.double().abs().sum()),
                      bool(torch.isfinite(o.p).all()))
    sums["ema"] = (float(eng.optG.ema.double().sum()),)
    torch.save({"sums": sums, "logs": {k: float(v) for k, v in logs.items() if v.dim() == 0},
                "reducers": {("G" if o is eng.optG else "D%d" % eng.optDs.index(o)): (len(r.chunks), r.early)
                             for o in [eng.optG] + eng.optDs for r in [eng.reducers.get(id(o))] if r is not None}},
               os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def main():
    out_dir = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group(os.environ.get("MOGAN_DIST_BACKEND", "gloo"), rank=rank, world_size=world)
    if os.environ.get("DP_FULL"):
        return full_width(out_dir, rank, world)
    small_cfg()
    dev = "cuda"
    # rank 0 holds the weights the oracle knows; every other rank starts from DIFFERENT ones on purpose -- the engine's
    # replica broadcast must replace them
    tag = "" if rank == 0 else "other%d." % rank
    G = model.G_NET()
    det_fill_state(G, tag + "G.")
    Ds = []
    for i, cls in enumerate((model.D_NET64, model.D_NET128, model.D_NET256)):
        D = cls()
        det_fill_state(D, tag + "D%d." % i)
        Ds.append(D.to(dev).train())
    enc = StandInEncoder(16)
    det_fill_state(enc, "ENC.")
    for p in enc.parameters():
        p.requires_grad = False
    eng = TrainEngine(None, enc.to(dev).eval(), G.to(dev).train(), Ds, distributed=True, use_graph=False)
    assert eng.distributed and eng.world == world
    logs = None
    for step in range(int(os.environ.get("DP_STEPS", "1"))):
        bt = synthetic.to_device(synthetic.make_batch(4, words_num=5, nef=16, seed=100 + 10 * step + rank), dev)
        logs = eng.step(bt)
    torch.cuda.synchronize()
    sd = {"G": {k: v.cpu() for k, v in G.state_dict().items()},
          "D": [{k: v.cpu() for k, v in D.state_dict().items()} for D in Ds],
          "ema": eng.optG.ema.cpu(), "logs": {k: float(v) for k, v in logs.items() if v.dim() == 0},
          "reducers": [(len(r.chunks), r.early) for r in eng.reducers.values()]}
    torch.save(sd, os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
