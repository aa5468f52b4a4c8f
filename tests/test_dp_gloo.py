"""N>1 path on CPU: world_size 2 over gloo (127.0.0.1).  Checks the data-parallel exchange of the
trainer -- flat gradient buckets whose views are the module's .grad, one sum all-reduce per network,
1/world folded into the optimizer step -- against the single-process result on the concatenated batch.
(The Adam arithmetic itself is a HIP kernel and is covered by the -m gpu tests; here the oracle's
python Adam stands in for it.)"""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from helpers import load_pkg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _net():
    torch.manual_seed(7)
    return nn.Sequential(nn.Linear(6, 5), nn.Tanh(), nn.Linear(5, 3))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    load_pkg()
    from mogan_amd.attngan.trainer import FlatAdam, allreduce_flat
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        net = _net()
        flat = FlatAdam(net, lr=2e-4, with_ema=True)
        for p, o in zip(flat.params, flat.offsets):                       # params/grads are views of the buckets
            assert p.data_ptr() == flat.p[o:o + p.numel()].data_ptr()
            assert p.grad.data_ptr() == flat.g[o:o + p.numel()].data_ptr()
        torch.manual_seed(100)
        x = torch.randn(8, 6)[rank * 4:(rank + 1) * 4]                    # this rank's shard of the global batch
        flat.zero_grad()
        net(x).pow(2).mean().backward()                                   # mean-reduced loss, like BCE/KL
        assert allreduce_flat(flat.g) is None
        # numpy, not torch tensors: a tensor crosses the queue as a shared fd the exiting worker may close first
        q.put((rank, (flat.g / world).numpy().copy(), [o for o in flat.offsets]))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_flat_bucket_allreduce_world2():
    load_pkg()
    world, port = 2, 29571 + (os.getpid() % 200)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=90) for _ in range(world)], key=lambda t: t[0])
    res = [(r, torch.from_numpy(g), o) for r, g, o in res]
    [p.join(30) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    # single process on the concatenated batch = mean of the shard gradients
    from mogan_amd.attngan.trainer import FlatAdam
    net = _net()
    flat = FlatAdam(net, lr=2e-4)
    torch.manual_seed(100)
    net(torch.randn(8, 6)).pow(2).mean().backward()
    for rank, g, offs in res:
        assert offs == flat.offsets
        torch.testing.assert_close(g, flat.g, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(res[0][1], res[1][1], rtol=0, atol=0)      # replicas see identical reduced grads


# ------------------------------------------------------------------ global-batch loss mode (attngan/parallel.py)
def _toy():
    torch.manual_seed(3)
    backbone = nn.Sequential(nn.Linear(6, 5), nn.Tanh())          # per-sample feature extractor (replicated)
    head = nn.Linear(5, 4)                                        # logits head evaluated on the gathered batch
    return backbone, head


def _toy_loss(feat, head):
    """batch-coupled like DAMSM: a B x B similarity matrix with cross-entropy against the diagonal, plus the
    "wrong pair" shift of the D losses."""
    h = head(feat)
    sim = h @ h.t()
    ce = nn.functional.cross_entropy(sim, torch.arange(h.shape[0]))
    wrong = (h[:-1] * h[1:]).sum(1).mean()
    return ce + 0.1 * wrong


def _gb_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    load_pkg()
    from mogan_amd.attngan import parallel
    from mogan_amd.attngan.miscc.config import cfg
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg.TRAIN.GLOBAL_BATCH_LOSS = True
        assert parallel.enabled()
        backbone, head = _toy()
        torch.manual_seed(100)
        x = torch.randn(8, 6)[rank * 4:(rank + 1) * 4]
        feat = parallel.gather_cat(backbone(x))
        assert feat.shape[0] == 8
        ids = parallel.gather_ids([10 * rank + i for i in range(4)])
        assert list(ids) == [0, 1, 2, 3, 10, 11, 12, 13]
        loss = _toy_loss(feat, head)
        loss.backward()
        gb = torch.cat([p.grad.reshape(-1) for p in backbone.parameters()])
        gh = torch.cat([p.grad.reshape(-1) for p in head.parameters()])
        dist.all_reduce(gb)
        dist.all_reduce(gh)                                         # what the flat-bucket exchange does: sum, then / world
        q.put((rank, float(loss), (gb / world).numpy().copy(), (gh / world).numpy().copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_global_batch_loss_mode_world2():
    """Gathered features + local-slice backward (x world) + the usual averaged all-reduce == the single-process
    gradients of the loss on the whole batch, for the per-sample backbone and for the replicated head."""
    load_pkg()
    world, port = 2, 29771 + (os.getpid() % 200)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gb_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=90) for _ in range(world)], key=lambda t: t[0])
    res = [(r, l, torch.from_numpy(a), torch.from_numpy(b)) for r, l, a, b in res]
    [p.join(30) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    backbone, head = _toy()
    torch.manual_seed(100)
    loss = _toy_loss(backbone(torch.randn(8, 6)), head)
    loss.backward()
    gb = torch.cat([p.grad.reshape(-1) for p in backbone.parameters()])
    gh = torch.cat([p.grad.reshape(-1) for p in head.parameters()])
    for rank, l, rb, rh in res:
        assert abs(l - float(loss)) < 1e-6
        torch.testing.assert_close(rb, gb, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(rh, gh, rtol=1e-5, atol=1e-6)
