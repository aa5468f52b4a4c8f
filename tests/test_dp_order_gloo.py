"""Multi-GPU readiness without hardware: the overlap machinery of the N > 1 step -- trainer.ChunkedReducer,
which starts a gradient chunk's all-reduce the moment its last contribution has been queued -- on 4 and 8 gloo ranks on CPU.
The reducer's stream / event calls are replaced by host stand-ins (gloo collectives are synchronous), everything else is the
product code: chunking of the flat bucket at parameter boundaries, the contribution counts learned in the first step, the chunk
launches driven by hip/ops.GRAD_HOOKS, finish().  Checked on every rank:
  * the ORDER of the collectives (chunk index sequence) is the same on all ranks in every step although the ranks queue their
    weight gradients with different delays -- it is a function of the model structure only (a mismatch would deadlock RCCL);
  * the reduced bucket equals the sum of the ranks' local gradients;
  * from the second step on chunks leave before the backward has finished (`early` > 0);
  * a contribution that arrives after its chunk has left raises on that rank (the documented error path) instead of letting
    the replicas drift apart."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from helpers import load_pkg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _HostStream:
    cuda_stream = 0

    def wait_event(self, ev):
        pass

    def wait_stream(self, s):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class _HostEvent:
    def record(self, stream=None):
        pass


def _net():
    torch.manual_seed(11)
    return nn.Sequential(*[nn.Linear(64, 64) for _ in range(6)])


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    load_pkg()
    from mogan_amd.attngan import trainer
    from mogan_amd.hip import ops
    # host stand-ins for the stream / event calls of the reducer (no GPU here)
    host = _HostStream()
    torch.cuda.current_stream = lambda *a, **k: host
    torch.cuda.Event = _HostEvent
    torch.cuda.stream = lambda s: s
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        net = _net()
        flat = trainer.FlatAdam(net, lr=2e-4)
        red = trainer.ChunkedReducer(flat, chunk_bytes=2 * 64 * 64 * 4, comm_stream=host, name="toy")
        assert len(red.chunks) >= 3
        order = []
        launch = red._launch

        def spy(ci):
            order.append(ci)
            launch(ci)
        red._launch = spy
        offs = {id(p_): o_ for p_, o_ in zip(flat.params, flat.offsets)}
        params = list(net.parameters())[::-1]                 # a backward pass produces the deep layers' gradients first
        out = []
        for step in range(4):
            flat.zero_grad()
            assert red.active
            order.clear()
            gen = torch.Generator().manual_seed(1000 * step + rank)
            local = torch.zeros_like(flat.g)
            late_error = None
            for pi, p in enumerate(params):
                # every rank "computes" its own gradient; contributions per parameter: two for weights (real + fake pass of
                # a discriminator update), one for biases -- and in step 3 rank 1 alone queues a third one for the deepest
                # weight AFTER its chunk has left: the error path
                n = 2 if p.dim() == 2 else 1
                for _ in range(n):
                    g = torch.randn(p.shape, generator=gen)
                    p.grad.add_(g)
                    o = offs[id(p)]
                    local[o:o + p.numel()] += g.reshape(-1)
                    ops._grad_hit(p.grad)
            if step == 3 and rank == 1:
                ops._grad_hit(params[0].grad)                # (nothing is added: only the bookkeeping sees it)
            try:
                red.finish()
            except RuntimeError as e:
                late_error = str(e)
            tot = local.clone()
            if late_error is None and not (step == 3):
                dist.all_reduce(tot)
            out.append({"order": list(order), "early": red.early, "late": late_error,
                        "ok": bool(step == 3 or torch.allclose(flat.g, tot, rtol=1e-5, atol=1e-5))})
            if step == 3:
                break
        q.put((rank, out, len(red.chunks)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [4, 8])
def test_chunked_reducer_collective_order(world):
    load_pkg()
    port = 29871 + (os.getpid() % 200) + world
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    nchunks = res[0][2]
    for step in range(3):
        orders = [r[1][step]["order"] for r in res]
        assert all(o == orders[0] for o in orders), "step %d: collective order differs across ranks: %r" % (step, orders)
        assert sorted(orders[0]) == list(range(nchunks))          # every chunk exactly once
        assert all(r[1][step]["ok"] for r in res), "step %d: reduced bucket != sum of the local gradients" % step
        assert all(r[1][step]["late"] is None for r in res)
        if step == 0:
            assert all(r[1][0]["early"] == 0 for r in res)        # calibration step: everything leaves in finish()
        else:
            assert all(r[1][step]["early"] >= nchunks - 1 for r in res)
        assert orders[0] == list(range(nchunks))                   # deep chunks first, as cut
    # step 3: the rank whose backward queued a contribution behind its chunk's all-reduce raises; the others do not
    for rank, out, _ in res:
        if rank == 1:
            assert out[3]["late"] is not None and "arrived after their chunk" in out[3]["late"]
        else:
            assert out[3]["late"] is None
