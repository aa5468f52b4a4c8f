"""Read-before-write hunt: one full-width AttnGAN train step (B=16, eager multi-stream) on a FRESH engine after the caching
allocator's free blocks and the per-stream workspaces of libmogan_hip were filled with NaN.  A kernel that reads scratch or
output memory it has not written (hidden in a fresh process, where device memory is zero) shows up as non-finite state.
    python tools/poison_step.py [rounds]"""
import os, sys, math, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mogan_loader; mogan_loader.load()
from mogan_amd.hip import lib
from mogan_amd.attngan import synthetic
from mogan_amd.attngan.miscc.config import cfg, set_coco_train_defaults
from mogan_amd.attngan.trainer import TrainEngine, build_networks

DEV = "cuda"


def poison(gb=24):
    keep = []
    for sz in [1 << 30] * gb + [64 << 20] * 32 + [2 << 20] * 256 + [512 << 10] * 256 + [4096] * 2048:
        keep.append(torch.full((sz // 4,), float("nan"), device=DEV))
    for buf in lib._ws.values():
        buf.view(torch.float32).fill_(float("nan"))
    torch.cuda.synchronize()
    del keep


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    set_coco_train_defaults()
    B = 16
    bad = 0
    for r in range(rounds):
        te, ie, G, Ds = build_networks(device=DEV, seed=100 + r)
        eng = TrainEngine(te, ie, G, Ds, use_graph=False)
        cpu = synthetic.make_batch(B, words_num=cfg.TEXT.WORDS_NUM, nef=cfg.TEXT.EMBEDDING_DIM, seed=r, text="tokens")
        bt = synthetic.to_device(cpu, DEV)
        bt["cap_lens_cpu"] = cpu["cap_lens"].clone()
        bt["cap_lens"] = bt["cap_lens"].to(torch.int32)
        for step in range(int(os.environ.get("POISON_STEPS", "3"))):
            poison()
            logs = eng.step(dict(bt))
            torch.cuda.synchronize()
            fin = {n: bool(torch.isfinite(o.p).all()) for n, o in zip(["G", "D64", "D128", "D256"], [eng.optG] + eng.optDs)}
            lf = all(math.isfinite(float(logs[k])) for k in ("errD0", "errD1", "errD2", "errG"))
            ok = all(fin.values()) and lf
            bad += not ok
            print("round %d step %d: %s %s losses_finite=%s" % (r, step, "ok" if ok else "NON-FINITE", fin, lf), flush=True)
            if not ok:
                for n, o in zip(["G", "D64", "D128", "D256"], [eng.optG] + eng.optDs):
                    nanmask = ~torch.isfinite(o.p)
                    if nanmask.any():
                        idx = nanmask.nonzero().flatten()
                        print("   %s: %d non-finite of %d, first %d last %d" % (n, idx.numel(), o.p.numel(), int(idx[0]), int(idx[-1])))
                break
        del eng, te, ie, G, Ds
    print("BAD" if bad else "CLEAN")


main()
