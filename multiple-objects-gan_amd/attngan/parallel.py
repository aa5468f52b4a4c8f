"""Optional "global-batch" loss mode (last row): reproduce the reference's multi-GPU semantics
exactly.  There, nn.parallel.data_parallel runs the D feature extractors / G on per-GPU shards (per-replica BN
statistics) but GATHERS the features on GPU 0, where the logits heads (incl. their BatchNorm), the "wrong pair"
shift and the DAMSM contrastive matrices see the WHOLE batch (code/coco/attngan/miscc/losses.py:146-169,193-221).

Here every rank all-gathers the (small) tensors that cross that boundary -- D features (B,768,4,4), Inception
region features (B,256,17,17) and codes (B,256), sentence/word embeddings -- and evaluates heads + losses on the
global batch redundantly (identical replicas, deterministic kernels => identical values on every rank).  Backward
needs no communication: each rank already holds d(loss)/d(gathered tensor) for all samples and keeps the slice of
its own samples.  Because the flat-bucket all-reduce later averages gradients over ranks, that slice is scaled by
world_size (for the replicated head parameters, whose gradients are identical on all ranks, the average is the
value itself).

Default (cfg.TRAIN.GLOBAL_BATCH_LOSS = False): each rank evaluates the reference single-GPU step on its shard.
"""
import os

import torch
import torch.distributed as dist

from .miscc.config import cfg


def enabled():
    # MOGAN_FORCE_DIST: exercise the path on one GPU (world_size 1), like TrainEngine does for the all-reduce
    return bool(cfg.TRAIN.get("GLOBAL_BATCH_LOSS", False)) and dist.is_available() and dist.is_initialized() \
        and (dist.get_world_size() > 1 or bool(os.environ.get("MOGAN_FORCE_DIST")))


def _all_gather(x):
    world = dist.get_world_size()
    parts = [torch.empty_like(x) for _ in range(world)]
    dist.all_gather(parts, x.contiguous())
    return torch.cat(parts, 0)


class GatherCat(torch.autograd.Function):
    """cat over ranks along dim 0; backward = own slice * world (see the module docstring)."""

    @staticmethod
    def forward(ctx, x):
        ctx.n = x.shape[0]
        return _all_gather(x)

    @staticmethod
    def backward(ctx, g):
        r, world = dist.get_rank(), dist.get_world_size()
        return g[r * ctx.n:(r + 1) * ctx.n] * float(world)


def gather_cat(x):
    """autograd-aware all-gather (no-op unless the global-batch mode is on)."""
    return GatherCat.apply(x) if enabled() else x


def gather_const(x):
    """all-gather of a tensor that carries no gradient (conditions, caption lengths)."""
    return _all_gather(x.detach()) if enabled() else x


def gather_ids(ids):
    """class ids (numpy / list) of all ranks, or None."""
    if ids is None or not enabled():
        return ids
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, list(map(int, ids)))
    import numpy as np
    return np.asarray([v for part in out for v in part])
