"""coco-attngan trainer (mirror of code/coco/attngan/trainer.py, train path only).

`condGANTrainer` keeps the reference constructor / build_models / define_optimizers /
prepare_labels / train / save_model surface.  The body of the reference's minibatch loop
(trainer.py:269-342) lives in `TrainEngine.step`, which is what bench.py times:

    text-encode -> G forward (once) -> for i in 0..2: zero_grad(D_i), discriminator_loss, backward,
    Adam(D_i) -> zero_grad(G), generator_loss (through the *updated* Ds, incl. DAMSM) + KL, backward,
    Adam(G) -> EMA(0.999)

MI355X-first differences (results identical to the reference's single-GPU step on the local batch):
  * parameters, gradients and Adam moments of each network live in flat fp32 buckets
    (`FlatAdam`): one fused Adam(+EMA) launch and one RCCL all-reduce per network instead of the
    reference's per-call replicate/broadcast of nn.parallel.data_parallel (trainer.py:296,
    miscc/losses.py:146,152,193: ~2.7 GB of parameter broadcast per step);
  * one process per GPU; D_i's gradient all-reduce runs on a side stream and only D_i's own Adam
    waits for it, so it overlaps D_{i+1}'s forward/backward (D256, the 643 MB bucket, is needed last);
  * the G step does not compute weight gradients of the Ds (the reference computes and then discards
    them at the next zero_grad: trainer.py:304,329-333);
  * the whole device part of the step can be captured into one hipGraph (`use_graph=True`) -- the
    step is a few thousand short launches at 4x4..16x16 resolution that are otherwise launch-bound.
"""
import glob
import logging
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from ..hip import ops
from .miscc.config import cfg
from .miscc.losses import (KL_loss, discriminator_loss, discriminator_loss_fake, discriminator_loss_real,
                           generator_d_branch, generator_damsm_branch, generator_loss, generator_total, split_d_loss)
from .miscc.utils import copy_G_params, load_params, mkdir_p, weights_init
from .model_base import BNCallCounter
from .model import CNN_ENCODER, D_NET64, D_NET128, D_NET256, G_NET, RNN_ENCODER


class FlatAdam:
    """Flat fp32 parameter / gradient / moment buckets of one network + the fused Adam(+EMA) step
    (torch.optim.Adam(betas=(0.5,0.999)) of trainer.py:137-148 and the EMA of trainer.py:341-342)."""

    ALIGN = 64   # elements

    def __init__(self, module, lr, with_ema=False, eps_mode=None):
        self.module, self.lr = module, float(lr)
        self.eps_mode = int(cfg.ADAM_EPS_MODE if eps_mode is None else eps_mode)
        self.params = [p for p in module.parameters() if p.requires_grad]
        dev = self.params[0].device
        offs, total = [], 0
        for p in self.params:
            offs.append(total)
            total += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.offsets, self.numel = offs, total
        self.p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.g = torch.zeros(total, dtype=torch.float32, device=dev)
        self.m = torch.zeros(total, dtype=torch.float32, device=dev)
        self.v = torch.zeros(total, dtype=torch.float32, device=dev)
        for p, o in zip(self.params, offs):
            self.p[o:o + p.numel()].copy_(p.data.reshape(-1))
            p.data = self.p[o:o + p.numel()].view_as(p)
            p.grad = self.g[o:o + p.numel()].view_as(p)
        self.ema = self.p.clone() if with_ema else None
        self.state = torch.zeros(3, dtype=torch.float32, device=dev)     # device-resident step counter
        # packed copies of the convolution weights (hip/ops.WeightPacks, csrc/mogan_pgemm.hip): created lazily by the layers
        # that qualify, re-packed here after every change of the bucket
        self._pk_cell = [0]
        self.packs = [ops.attach_packs(p, self._pk_cell) for p in self.params if p.dim() == 4] if dev.type == "cuda" else []
        self._hook = None
        if isinstance(module, torch.nn.Module):
            # through a weak reference: the module must not keep a dropped optimizer (and its four flat buckets, 2.5 GB for
            # D_NET256) alive -- nor let it go on re-packing the shared weight copies at every later load_state_dict
            import weakref
            ref = weakref.ref(self)

            def _touch(m, keys, _ref=ref):
                me = _ref()
                if me is not None:
                    me.touch()
            self._hook = module.register_load_state_dict_post_hook(_touch)

    def close(self):
        """Detach from the module (a second optimizer over the same network follows): the load_state_dict hook goes, and this
        object no longer re-packs the weight copies it shares with its successor."""
        if self._hook is not None:
            self._hook.remove()
            self._hook = None
        self.repack_on_touch = False
        self.packs = []

    repack_on_touch = False     # set by an engine that holds captured graphs: those read the packed copies without asking

    def touch(self):
        """the bucket's parameters were written (load_state_dict, a broadcast, a restore): packed copies are rebuilt at
        their next use -- or right now, on the current stream, when replayed hipGraphs consume them (a replay never calls
        WeightPacks.pointer(), so a lazily rebuilt copy would be a stale one until the graph's own post-Adam pack node ran)"""
        self._pk_cell[0] += 1
        if self.repack_on_touch:
            ops.repack_all(self.packs)

    def repack(self):
        """the same, but the copies in use are rebuilt NOW on the current stream -- after the optimizer step (one pack per
        weight version) and wherever a replayed hipGraph will use the copies without a pack node of its own"""
        self._pk_cell[0] += 1
        ops.repack_all(self.packs)           # (packed panels pack by pack; every Winograd filter image of the bucket in ONE launch)

    on_zero = None          # set by ChunkedReducer: a new accumulation round of this bucket begins

    def zero_grad(self):
        self.g.zero_()
        if self.on_zero is not None:
            self.on_zero()

    def step(self, grad_scale=1.0):
        ops.adam_step(self.p, self.g, self.m, self.v, self.ema, self.lr, 0.5, 0.999, 1e-8,
                      dev_state=self.state, eps_mode=self.eps_mode, grad_scale=grad_scale)
        self.repack()

    def ema_params(self):
        return [self.ema[o:o + p.numel()].view_as(p) for p, o in zip(self.params, self.offsets)]

    def state_dict(self):
        """torch.optim.Adam's layout (what the reference saves as optimG / optimD, trainer.py:188-196): per-parameter
        step / exp_avg / exp_avg_sq (views of the flat moment buckets, cloned by torch.save) + one param group."""
        step = float(self.state[0].item())
        state = {i: {"step": torch.tensor(step), "exp_avg": self.m[o:o + p.numel()].view_as(p),
                     "exp_avg_sq": self.v[o:o + p.numel()].view_as(p)}
                 for i, (p, o) in enumerate(zip(self.params, self.offsets))}
        group = {"lr": self.lr, "betas": (0.5, 0.999), "eps": 1e-8, "weight_decay": 0, "amsgrad": False,
                 "params": list(range(len(self.params)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        """Accepts torch.optim.Adam's {'state', 'param_groups'} (reference checkpoints, and what state_dict() emits) and
        the flat {'step', 'exp_avg', 'exp_avg_sq'} layout of round-1 checkpoints."""
        if "state" in sd:
            # torch.optim.Adam fills `state` lazily (a parameter that never received a gradient has no entry) and keys it by
            # the ids of param_groups[*]["params"], in the order of the parameter list the optimizer was built over -- the
            # reference builds it over ALL netG.parameters() (trainer.py:137-148), this bucket over the trainable ones.
            st = sd["state"]
            ids = [i for g in sd.get("param_groups") or [] for i in g.get("params", [])]
            if not ids:
                ids = list(range(len(self.params)))
            if len(ids) == len(self.params):
                targets = list(zip(self.params, self.offsets))
            else:
                allp = list(self.module.parameters())
                if len(ids) != len(allp):
                    raise ValueError("optimizer state covers %d parameters, the network has %d (%d trainable)"
                                     % (len(ids), len(allp), len(self.params)))
                off = {id(p): o for p, o in zip(self.params, self.offsets)}
                targets = [(p, off.get(id(p))) for p in allp]
            step = 0.0
            for key, (p, o) in zip(ids, targets):
                e = st.get(key, st.get(str(key)))
                if e is None or o is None:
                    continue                                  # no moments saved / not trainable here: zeros stay
                if e["exp_avg"].numel() != p.numel() or e["exp_avg_sq"].numel() != p.numel():
                    raise ValueError("optimizer state entry %s has %d elements, the parameter %d"
                                     % (key, e["exp_avg"].numel(), p.numel()))
                self.m[o:o + p.numel()].copy_(e["exp_avg"].reshape(-1))
                self.v[o:o + p.numel()].copy_(e["exp_avg_sq"].reshape(-1))
                step = max(step, float(e["step"]))
            self.state[0] = step
            if sd.get("param_groups"):
                self.lr = float(sd["param_groups"][0].get("lr", self.lr))
            return
        self.state[0] = float(sd["step"])
        self.m.copy_(sd["exp_avg"])
        self.v.copy_(sd["exp_avg_sq"])

    def broadcast(self, src=0):
        """Replica synchronisation: rank `src`'s parameters, EMA shadow, moments and step counter to every rank."""
        for t in (self.p, self.m, self.v, self.state) + ((self.ema,) if self.ema is not None else ()):
            dist.broadcast(t, src)
        self.touch()


class CommStats:
    """Per-bucket communication timing of the data-parallel step (bench.py --gpus N reports it; off by default: it creates
    timing events around every collective and around every wait of a consumer).  Two figures per gradient bucket and step:
      allreduce_ms  time the bucket's collectives took on the stream they execute on (peers' arrival included),
      exposed_ms    time the consumer -- the stream that runs the bucket's Adam -- was held at its wait for them
    (trainer.py:296, miscc/losses.py:146-193 are the reference's per-call gathers these collectives replace)."""

    def __init__(self):
        self.enabled, self.rec, self.meta = False, [], {}

    def begin(self, name, kind, stream=None, nbytes=0):
        if not self.enabled:
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(stream) if stream is not None else ev.record()
        m = self.meta.setdefault(name, {"bytes": 0, "collectives": 0})
        if kind == "allreduce":
            m["bytes"] += int(nbytes)
            m["collectives"] += 1
        return (name, kind, ev, stream)

    def end(self, tok):
        if tok is None:
            return
        name, kind, ev0, stream = tok
        ev1 = torch.cuda.Event(enable_timing=True)
        ev1.record(stream) if stream is not None else ev1.record()
        self.rec.append((name, kind, ev0, ev1))

    def report(self, steps):
        """{bucket: {bytes, collectives, allreduce_ms, exposed_ms}} per step, averaged over `steps` recorded steps"""
        torch.cuda.synchronize()
        out = {}
        for name, m in self.meta.items():
            out[name] = {"bytes_per_step": m["bytes"] // max(steps, 1), "collectives_per_step": m["collectives"] / max(steps, 1),
                         "allreduce_ms": 0.0, "exposed_ms": 0.0}
        for name, kind, e0, e1 in self.rec:
            o = out.setdefault(name, {"bytes_per_step": 0, "collectives_per_step": 0, "allreduce_ms": 0.0, "exposed_ms": 0.0})
            o["allreduce_ms" if kind == "allreduce" else "exposed_ms"] += e0.elapsed_time(e1) / max(steps, 1)
        self.rec, self.meta = [], {}
        return out


def allreduce_flat(flat_g, comm_stream=None, stats=None, name=None):
    """Sum all-reduce of one flat gradient bucket over the default process group (RCCL on GPUs, gloo in
    the CPU tests).  With a side stream the collective is ordered after the work already queued on the
    current stream and an event is returned for the consumer (the bucket's Adam) to wait on."""
    if comm_stream is None:
        # issued on the caller's stream: the process group orders the collective behind the work queued on it and the stream
        # continues behind the collective -- its duration IS the consumer's wait
        tok = stats.begin(name, "allreduce", nbytes=flat_g.numel() * 4) if stats is not None else None
        dist.all_reduce(flat_g)
        if tok is not None:
            stats.end(tok)
            stats.rec.append((name, "exposed") + stats.rec[-1][2:])
        return None
    ev = torch.cuda.Event()
    ev.record()
    comm_stream.wait_event(ev)
    with torch.cuda.stream(comm_stream):
        dist.all_reduce(flat_g)
        done = torch.cuda.Event()
        done.record()
    return done


class ChunkedReducer:
    """Overlap of a big gradient bucket's all-reduce with the backward pass that fills it (replaces
    the per-call gather of nn.parallel.data_parallel, trainer.py:296, miscc/losses.py:146-193).

    The flat bucket is cut at parameter boundaries into chunks of >= chunk_bytes.  The kernels that write a parameter's
    gradient report in through hip/ops.GRAD_HOOKS; how many contributions each parameter receives per backward (two for a
    D update: real and fake; three for the conditional head; one where both are merged into one launch) is learned in the
    first step.  From the second step on, the moment the last contribution of a chunk has been QUEUED, events are recorded
    on the stream that runs the backward and on its weight-gradient side stream, the communication stream waits for both
    and starts the chunk's all-reduce -- the deep layers of D_NET256 (produced first, 3/4 of its 643 MB) travel while the
    shallow layers are still being differentiated.  finish() reduces whatever is left (everything, in the first step) and
    makes the caller's stream wait for all chunks.  Same sums as one all-reduce of the whole bucket; the order of the
    collectives is a function of the model's structure only, hence identical on every rank."""

    def __init__(self, flat, chunk_bytes, comm_stream, stats=None, name=None):
        self.flat, self.comm, self.stats, self.name = flat, comm_stream, stats, name
        base = flat.g.data_ptr()
        self.base = base
        bounds, start = [], flat.numel
        # from the end of the bucket (the heads / deep layers, whose gradients are complete first) towards the front
        cuts = [flat.numel]
        for p, o in list(zip(flat.params, flat.offsets))[::-1]:
            if (cuts[-1] - o) * 4 >= chunk_bytes:
                cuts.append(o)
        if cuts[-1] != 0:
            cuts.append(0)
        if len(cuts) > 2 and (cuts[-2] - cuts[-1]) * 4 < chunk_bytes // 4:     # a crumb at the front joins its neighbour
            del cuts[-2]
        self.chunks = [(cuts[i + 1], cuts[i]) for i in range(len(cuts) - 1)]    # (lo, hi) element ranges, deep first
        self.chunk_of, self.members = {}, [[] for _ in self.chunks]
        for p, o in zip(flat.params, flat.offsets):
            for ci, (lo, hi) in enumerate(self.chunks):
                if lo <= o < hi:
                    self.chunk_of[base + 4 * o] = ci
                    self.members[ci].append(base + 4 * o)
        self.expected, self.counts, self.done, self.launched, self.early = None, {}, [], set(), 0
        self.active, self.late = False, []
        # registered through a weak reference: the process-global hook list must not keep a dropped engine (and its flat
        # buckets, 2.5 GB for D_NET256) alive; close() / garbage collection of the reducer removes the entry
        import weakref
        ref = weakref.ref(self)

        def _hit(ptr, stream, _ref=ref):
            me = _ref()
            if me is not None:
                me.hit(ptr, stream)
        self._hook = (base, base + 4 * flat.numel, _hit)
        ops.GRAD_HOOKS.append(self._hook)
        weakref.finalize(self, ChunkedReducer._unhook, self._hook)
        flat.on_zero = self.begin

    @staticmethod
    def _unhook(hook):
        try:
            ops.GRAD_HOOKS.remove(hook)
        except ValueError:
            pass

    def close(self):
        """Detach from hip/ops.GRAD_HOOKS and from the bucket (idempotent)."""
        ChunkedReducer._unhook(self._hook)
        if getattr(self.flat, "on_zero", None) == self.begin:
            self.flat.on_zero = None

    def begin(self):
        self.counts, self.done, self.launched, self.active, self.early = {}, [], set(), True, 0
        self.late = []

    def hit(self, ptr, stream):
        if not self.active:
            return
        self.counts[ptr] = self.counts.get(ptr, 0) + 1
        if self.expected is None:
            return
        ci = self.chunk_of.get(ptr)
        if ci is None:
            return
        if ci in self.launched:
            self.late.append(ptr)            # a contribution queued AFTER the chunk's all-reduce: see finish()
            return
        if all(self.counts.get(q, 0) >= self.expected.get(q, 1 << 30) for q in self.members[ci]):
            self._launch(ci)
            self.early += 1                  # (diagnostic: chunks that left before the backward pass had finished)

    def _launch(self, ci):
        cur = torch.cuda.current_stream()
        side = ops._wgrad_streams.get(cur.cuda_stream)
        for st in (cur, side):
            if st is not None:
                ev = torch.cuda.Event()
                ev.record(st)
                self.comm.wait_event(ev)
        lo, hi = self.chunks[ci]
        with torch.cuda.stream(self.comm):
            tok = self.stats.begin(self.name, "allreduce", nbytes=(hi - lo) * 4) if self.stats is not None else None
            dist.all_reduce(self.flat.g[lo:hi])
            if tok is not None:
                self.stats.end(tok)
            ev = torch.cuda.Event()
            ev.record()
        self.done.append(ev)
        self.launched.add(ci)

    def _wait_all(self):
        """the current stream (the bucket's consumer) waits for every chunk; the time it is held there is the exposed part"""
        cur = torch.cuda.current_stream()
        tok = self.stats.begin(self.name, "exposed") if self.stats is not None else None
        for ev in self.done:
            cur.wait_event(ev)
        if tok is not None:
            self.stats.end(tok)

    def reduce_now(self):
        """Branch graphs: the backward that fills the bucket is one replayed hipGraph -- no hook fires, nothing can leave early.
        The whole bucket, chunk by chunk (the first chunks are summed while the later ones still travel), behind everything
        queued on the current stream; the current stream then waits for all of them.  The learned schedule of the eager path
        (`expected`) is left alone."""
        self.counts, self.done, self.launched, self.late, self.active, self.early = {}, [], set(), [], False, 0
        for ci in range(len(self.chunks)):
            self._launch(ci)
        self._wait_all()
        return len(self.chunks)

    def finish(self):
        """after the backward (and the weight-gradient join): reduce the chunks not yet on their way, then make the current
        stream wait for all of them"""
        self.active = False
        if self.late:
            # the call pattern of this backward had MORE contributions for a parameter than the learned schedule: its chunk
            # was summed over the ranks before the last weight gradient was queued, the bucket now holds a mix of reduced
            # and local terms and the replicas would drift apart silently.  There is no cheap repair (the peers' late terms
            # are gone into their own mixes), so this is an error; construct the engine with MOGAN_DP_CHUNK_MB=0 (whole-
            # bucket all-reduce after the backward) for models whose backward is not the same every step.
            n = len(self.late)
            self.expected, self.late = None, []
            raise RuntimeError("ChunkedReducer: %d gradient contribution(s) arrived after their chunk's all-reduce had been "
                               "issued (the backward's call pattern changed between steps); gradients of this step are "
                               "inconsistent across ranks.  Set MOGAN_DP_CHUNK_MB=0 for this model." % n)
        if self.expected is None:
            self.expected = dict(self.counts)                  # calibration step
        elif any(self.counts.get(q, 0) != n for q, n in self.expected.items()):
            self.expected = dict(self.counts)                  # FEWER contributions than learned (no chunk left early on
            #                                                    stale counts): re-learn, the rest is reduced below
        for ci in range(len(self.chunks)):
            if ci not in self.launched:
                self._launch(ci)
        self._wait_all()
        return len(self.chunks)


# HIP assigns a stream its hardware queue when the stream is created, round-robin over GPU_MAX_HW_QUEUES, and which streams
# share a queue decides +-8 % of the step (hip/lib.py, "hardware queues").  The engines of one process therefore take their
# streams from this table: the FIRST engine creates them in the one order the queue defaults were measured with, every later
# engine (bench.py's second data-parallel launch mode, a test's second engine) runs on the very same streams and so on the
# same queue layout -- instead of on whatever the next entries of torch's 32-stream pool happen to map to.
_ENGINE_STREAMS = {}


def _engine_stream(key):
    key = (torch.cuda.current_device(),) + tuple(key)
    s = _ENGINE_STREAMS.get(key)
    if s is None:
        s = _ENGINE_STREAMS[key] = torch.cuda.Stream()
    return s


# first-use order of the step's streams (tokens: s<i> = branch stream of D_i / s3 = the Inception-DAMSM branch, w<i> / wm = the
# weight-gradient streams of the branches / of the main stream, gc = the generator graph's capture stream, cG / cD = communication
# streams, x = an unused stream).
# Measured (profiles/r05_queue_table.csv; 4 hardware queues, no reserved streams, img/s single process / member of a 1-rank
# RCCL group): this order 437 / 435 (two repeats each), "s2,s3,wm,s1,s0" 438 / 435, "s0,s1,s2,s3,w2,w1,w0,wm" 411 / 409,
# "s2,s1,s0,s3,w2,w1,w0,wm" 399 / 398, "s3,s2,s1,x,s0,wm" 407; streams bound lazily at their first use in the step (rounds 1-4):
# 432-434 / 401.  The layout is a function of this order alone, and the two kinds of process agree to 1 %.  gc = the stream the
# generator's eager backward runs on when its forward is a replayed graph (MOGAN_G_GRAPHS=2): behind s0 447.9 img/s (eager
# generator on that box: 441.1), in front of everything 443.7, between wm and s0 422.6, behind s2 414.1.
ENGINE_STREAM_ORDER = "s2,s3,s1,wm,s0,gc,w2,w1,w0,cG,cD"
_KEEP_STREAMS = []
_STREAMS_CREATED = set()


def create_engine_streams(n_discriminators=3, touch=True):
    """The streams a TrainEngine of this process will run on, created NOW, in the engine's canonical order (branch streams, their
    weight-gradient streams in the order the branches run, the generator's weight-gradient stream, the communication streams of
    the chunked buckets) and, with `touch`, each used once so that the runtime binds it to its hardware queue in that order.
    Entry points call this right after torch.cuda.set_device -- BEFORE torch.distributed creates the process group -- so the
    stream -> hardware-queue layout of the step does not depend on whether (and when) RCCL adds its own stream: one layout, one
    queue default for every kind of process (hip/lib.py: hw_queue_defaults; profiles/r05_queue_table.csv)."""
    dev = torch.cuda.current_device()
    if dev in _STREAMS_CREATED:
        return []
    _STREAMS_CREATED.add(dev)
    side = [_engine_stream(("side", i)) for i in range(n_discriminators + 1)]
    table = {"s%d" % i: (lambda i=i: side[i]) for i in range(n_discriminators + 1)}
    table.update({"w%d" % i: (lambda i=i: ops.precreate_wgrad_stream(side[i])) for i in range(n_discriminators)})
    table["wm"] = lambda: ops.precreate_wgrad_stream(torch.cuda.current_stream())
    table["cG"] = lambda: _engine_stream(("comm", "G"))
    table["cD"] = lambda: _engine_stream(("comm", "D256"))
    table["gc"] = lambda: _engine_stream(("gcap",))      # capture stream of the generator's forward graph = the stream its eager backward runs on
    table["x"] = lambda: torch.cuda.Stream()              # a stream nobody uses: takes a slot of the round-robin
    order = os.environ.get("MOGAN_STREAM_ORDER") or ENGINE_STREAM_ORDER
    made = [table[k]() for k in order.split(",") if k in table]
    _KEEP_STREAMS.extend(made)
    if touch:
        t = torch.zeros(64, device="cuda")
        cur = torch.cuda.current_stream()
        for s in made:
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                t.add_(1.0)
            cur.wait_stream(s)
        torch.cuda.synchronize()
    return made


class TrainEngine:
    """Device-side state of one rank: networks, flat optimizers, DP communicator, optional hipGraph."""

    G_GRAPH_VARIANTS = 4       # generator forward graphs kept per engine, one per shape of the text tensors (see _g_graph_for)

    def __init__(self, text_encoder, image_encoder, netG, netsD, distributed=False, use_graph=False, branch_graphs=None):
        self.text_encoder, self.image_encoder, self.netG, self.netsD = text_encoder, image_encoder, netG, netsD
        self.optG = FlatAdam(netG, cfg.TRAIN.GENERATOR_LR, with_ema=True)
        self.optDs = [FlatAdam(d, cfg.TRAIN.DISCRIMINATOR_LR) for d in netsD]
        self.bn_counter = BNCallCounter([netG] + list(netsD))
        self.distributed = bool(distributed) and dist.is_available() and dist.is_initialized() \
            and (dist.get_world_size() > 1 or bool(os.environ.get("MOGAN_FORCE_DIST")))   # env: exercise the RCCL path at N=1
        self.world = dist.get_world_size() if self.distributed else 1
        self.use_graph = use_graph
        self._graph = None
        self._static = None
        self.last = {}
        # independent branches of the step (the three D updates; in the G step the three D forwards and the
        # Inception/DAMSM branch) run on side streams so that their many small launches overlap; captured, they
        # become parallel branches of the hipGraph.  MOGAN_STREAMS=0 keeps everything on one stream.
        self.graph_encoder = os.environ.get("MOGAN_GRAPH_ENCODER", "1") != "0" and not use_graph
        self.early_damsm_bwd = True      # the DAMSM / Inception data gradient does not wait for errG_total.backward()
        self._enc_graphs = {}
        self.multi_stream = os.environ.get("MOGAN_STREAMS", "1") != "0"
        # Branch graphs (single process, multi-stream, not the whole-step graph): every discriminator branch -- D_i(real);
        # D_i(fake), loss, backward incl. the weight gradients on their side stream, Adam, re-pack, the generator-step forward
        # through the updated D_i and its gradient with respect to the fake image -- is captured ONCE as two hipGraphs per
        # discriminator and replayed on the branch's stream; the generator itself (forward, backward, Adam) stays eager on the
        # main stream.  ~900 of the step's ~2000 launches leave the host path (34 -> ~20 ms of python per step), the stream
        # structure of the eager step is unchanged.  MOGAN_BRANCH_GRAPHS=0 / branch_graphs=False: everything eager.
        if branch_graphs is None:
            branch_graphs = os.environ.get("MOGAN_BRANCH_GRAPHS", "1") != "0"
        # Under data parallelism the second graph of a branch is cut once more, between the backward and Adam: the bucket's
        # all-reduce is issued eagerly between the two replays (collectives are not captured); MOGAN_BRANCH_GRAPHS_DP=0 keeps
        # the eager step with its hook-driven ChunkedReducer for the discriminators as well.
        if self.distributed and os.environ.get("MOGAN_BRANCH_GRAPHS_DP", "1") == "0":
            branch_graphs = False
        self.branch_graphs = bool(branch_graphs) and self.multi_stream and not use_graph
        # The generator: MOGAN_G_GRAPHS = 2 (default since round 5) replays its FORWARD as a hipGraph (one per shape of the text
        # tensors, at most TrainEngine.G_GRAPH_VARIANTS = 4; further shapes run eagerly, with a logged warning) and keeps the backward eager -- on the autograd
        # tape recorded during the capture (retain_graph; the static inputs are refreshed through .data so that the tape's saved
        # tensors keep their version), so the weight gradients still overlap the data-gradient chain on their side stream and the
        # data-parallel reducers still see every gradient as it is queued.  Host enqueue per step 29.8 -> 20.8 ms at B = 16 (12.7 ->
        # 8.0 at B = 4, 55.6 -> 37.0 at B = 32), throughput +0.5 % at B = 16 / 32, equal at B = 4 / 8 (profiles/r05_ab.txt).
        # 1 (round 3, single process only): forward AND backward + Adam + EMA as two hipGraphs -- host enqueue 17.6 ms, but the
        # weight gradients then run in line: 407.7 vs 426.0 img/s.  0: eager.
        gmode = os.environ.get("MOGAN_G_GRAPHS", "2")
        self.g_fwd_only = gmode == "2"
        self.g_graphs = self.branch_graphs and gmode != "0" and (self.g_fwd_only or not self.distributed)
        self._bg = None
        self._g_refused = set()
        if torch.cuda.is_available() and self.multi_stream and not use_graph:
            create_engine_streams(len(netsD))       # (a no-op when the entry point has done it before the process group came up)
        # the discriminator loss in two halves: the real-image terms evaluated and back-propagated ahead of the generator's
        # forward (miscc/losses.py: discriminator_loss_real / _fake) -- opt-in, losses.D_SPLIT; the default is one loss, one backward after the forward
        self.split_d = self.multi_stream and split_d_loss()
        self.side = [_engine_stream(("side", i)) for i in range(len(netsD) + 1)]
        # stream creation order fixes the stream -> hardware-queue map (see ops.precreate_wgrad_stream): branch streams,
        # then the weight-gradient streams in the order the branches run (D256, D128, D64, generator), communication
        # streams last -- measured: with the communication streams created in between, the generator's wgrad stream landed
        # on the main stream's queue and the RCCL path ran 12 % slower before a single byte was exchanged
        for i in range(len(netsD))[::-1]:
            ops.precreate_wgrad_stream(self.side[i])
        ops.precreate_wgrad_stream(torch.cuda.current_stream())
        self.comm_stream = None          # collectives are issued on the branch streams (see _allreduce_async)
        self.comm = CommStats()          # per-bucket all-reduce / exposed time (enabled by bench.py for N > 1)
        self._debug_no_ar = bool(os.environ.get("MOGAN_DEBUG_NO_ALLREDUCE"))   # diagnostic: cost of the collectives' ordering
        if self.distributed and self.world > 1:
            self.sync_replicas()
        # buckets above 2 x MOGAN_DP_CHUNK_MB (default 96 MB: only D_NET256's 643 MB bucket) are reduced in chunks while
        # their backward is still running (ChunkedReducer); 0 = one all-reduce per bucket after the backward
        self.reducers = {}
        chunk = int(float(os.environ.get("MOGAN_DP_CHUNK_MB", "96")) * (1 << 20))
        if self.distributed and chunk > 0 and not self._debug_no_ar:
            for o in [self.optG] + self.optDs:
                c = chunk
                if o is self.optG:
                    # the generator's all-reduce is the one nothing else hides (it sits between errG.backward() and the
                    # generator's Adam on the main stream): its bucket travels in thirds while the backward is still running
                    # -- h_net3 / img_net3 at the end of the bucket are differentiated first --, only the last chunk is exposed
                    c = min(chunk, max(8 << 20, o.numel * 4 // 3))
                if o.numel * 4 >= 2 * c:
                    self.reducers[id(o)] = ChunkedReducer(o, c, _engine_stream(("comm", self._bucket_name(o))), self.comm,
                                                          self._bucket_name(o))

    def _bucket_name(self, flat):
        if flat is self.optG:
            return "G"
        return "D%d" % (64 << self.optDs.index(flat))

    def close(self):
        """Detach the engine from the process-global hooks (a second engine over the same networks follows: bench.py's two
        data-parallel launch modes)."""
        for r in self.reducers.values():
            r.close()
        self.reducers = {}
        # the optimizers' buckets and the captured graphs (with their private memory pools) go with the engine: a successor
        # builds its own over the same networks (its FlatAdam re-points the parameters at its buckets)
        for o in [self.optG] + list(self.optDs):
            if o is not None:
                o.close()
        self._bg, self._graph, self._enc_graphs = None, None, {}

    def repack_all(self):
        """Weights were written behind the optimizers' backs (load_params / invalidate_all_packs, a checkpoint restore) while
        this engine holds captured graphs: rebuild every packed copy in use now and order the branch streams behind it."""
        cur = torch.cuda.current_stream()
        for o in [self.optG] + self.optDs:
            o.repack()
        for s in self.side:
            s.wait_stream(cur)

    def sync_replicas(self, src=0):
        """Every rank starts from rank `src`'s weights, EMA shadow, optimizer state and BatchNorm buffers: the step only
        exchanges gradients, so replicas that differ at step 0 (per-rank init seeds, a checkpoint only rank 0 could read)
        would stay different forever.  One broadcast per flat bucket + one per buffer, once."""
        for o in [self.optG] + self.optDs:
            o.broadcast(src)
        for net in [self.netG] + list(self.netsD) + [self.text_encoder, self.image_encoder]:
            if not isinstance(net, torch.nn.Module):
                continue
            for b in net.buffers():
                if b.dim() > 0:
                    dist.broadcast(b, src)
            for p in net.parameters():
                if not p.requires_grad:                       # frozen encoders: not in any bucket.  On the parameter itself
                    with torch.no_grad():                     # (not .data): the in-place version counter must move, the
                        dist.broadcast(p, src)                # encoder's derived-weight caches are keyed on it
        dist.broadcast(self.bn_counter.flat, src)

    # -- data parallel: sum all-reduce of a flat gradient bucket over RCCL on a side stream ----------
    def _allreduce_async(self, flat):
        if not self.distributed or self._debug_no_ar:
            return None
        red = self.reducers.get(id(flat))
        if red is not None:                      # most of it is already on its way (started during the backward)
            red.finish()
            return None
        # issued on the branch's own stream: the process group's internal stream orders the collective behind the work
        # queued on it and the branch continues (Adam) behind the collective; a dedicated communication stream only adds
        # a stream whose event wait blocks a hardware queue (measured slower)
        return allreduce_flat(flat.g, None, self.comm, self._bucket_name(flat))

    def _opt_step(self, flat, pending):
        if pending is not None:
            torch.cuda.current_stream().wait_event(pending)
        flat.step(grad_scale=1.0 / self.world)

    def _encoder(self, fake_img):
        """The frozen, eval-mode image encoder as a replayed hipGraph pair (forward / data gradient).  Inception-v3 is
        ~600 short launches per direction whose host cost (17 ms forward, alone) exceeds their GPU time (5 ms): the
        branch was host-bound and on the critical path of the step.  It has no state to update (eval BN, no weight
        gradients), its shapes are static and it lives on one stream, so torch.cuda.make_graphed_callables applies."""
        enc = self.image_encoder
        if not self.graph_encoder or not isinstance(enc, torch.nn.Module) or enc.training \
                or any(p.requires_grad for p in enc.parameters()):
            return enc
        key = tuple(fake_img.shape)
        g = self._enc_graphs.get(key)
        if g is None:
            sample = torch.zeros(key, dtype=torch.float32, device=fake_img.device, requires_grad=True)
            # a plain function, not the module: make_graphed_callables would otherwise patch enc.forward in place
            from ..hip import lib
            with lib.capture_guard():
                g = torch.cuda.make_graphed_callables(lambda x: enc(x), (sample,))
            self._enc_graphs[key] = g
        return g

    def _d_kw(self, i, b):
        return dict(local_labels=b["label_one_hot"], transf_matrices=b["tm"], transf_matrices_inv=b["tmi"]) if i == 0 else {}

    def _d_real(self, i, b, sent_emb=None):
        """zero_grad + the real-image half of D_i's update: independent of the generator, so it runs beside the G forward (and the
        tail of the previous step).  Split form (miscc/losses.split_d_loss; opt-in, off by default): the real-image terms of the loss are
        evaluated AND back-propagated here -- returns ("split", R, pending running-statistics updates); otherwise only D_i(real)
        is evaluated and its features are returned (None with the paired pass)."""
        from .miscc.losses import _call_d, paired
        self.optDs[i].zero_grad()
        if paired(self.netsD[i]):                # D_i(real) rides with D_i(fake) as one [real; fake] pass (losses.D_PAIR)
            return None
        if self.split_d:
            errR, pend = discriminator_loss_real(self.netsD[i], b["imgs"][i], b["sent_emb"] if sent_emb is None else sent_emb,
                                                 **self._d_kw(i, b))
            with ops.wgrad_overlap():
                errR.backward()
            return ("split", errR.detach(), pend)
        if i == 0:
            return _call_d(self.netsD[i], b["imgs"][i], b["label_one_hot"], b["tm"], b["tmi"])
        return _call_d(self.netsD[i], b["imgs"][i], None, None, None)

    def _d_fake(self, i, b, fake_imgs, early, sent_emb=None):
        """the fake-image half (split form): F, its backward; returns errD_i = R + F (detached)"""
        _, errR, pend = early
        errF = discriminator_loss_fake(self.netsD[i], fake_imgs[i], b["sent_emb"] if sent_emb is None else sent_emb, pend,
                                       **self._d_kw(i, b))
        with ops.wgrad_overlap():
            errF.backward()
        return ops.scalar_sum([errR, errF.detach()])

    def _d_loss(self, i, b, fake_imgs, real_labels, fake_labels, real_features=None):
        kw = dict(local_labels=b["label_one_hot"], transf_matrices=b["tm"],
                  transf_matrices_inv=b["tmi"]) if i == 0 else {}
        return discriminator_loss(self.netsD[i], b["imgs"][i], fake_imgs[i], b["sent_emb"], real_labels,
                                  fake_labels, None, real_features=real_features, **kw)

    def _d_update(self, i, b, fake_imgs, real_labels, fake_labels, real_features=None):
        """[zero_grad,] loss, backward, (all-reduce,) Adam of D_i on the current stream."""
        if real_features is None:
            self.optDs[i].zero_grad()
        errD = self._d_loss(i, b, fake_imgs, real_labels, fake_labels, real_features)
        with ops.wgrad_overlap():
            errD.backward()
        self._opt_step(self.optDs[i], self._allreduce_async(self.optDs[i]))
        return errD.detach()

    # -- the reference loop body -------------------------------------------------------------------
    def _phase(self, name):
        """MOGAN_PHASE_TIMES=1: device-synchronised wall time per phase of the step (diagnostic; serialises the phases).
        MOGAN_CHAIN_EVENTS=1: an event on the MAIN stream at every phase boundary instead (no synchronisation, the step runs
        as usual); chain_report() turns them into the time the main stream -- the generator's dependency chain -- spent in
        each phase, waits for the side branches included."""
        if os.environ.get("MOGAN_CHAIN_EVENTS"):
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self._chain = getattr(self, "_chain", [])
            self._chain.append((name, ev))
            return
        if not getattr(self, "phase_times", False):       # (diagnostic attribute, tools/phase_times.py: host-synchronised phase times)
            return
        torch.cuda.synchronize()
        now = time.perf_counter()
        if getattr(self, "_ph_last", None) is not None:
            self._ph = getattr(self, "_ph", {})
            self._ph[self._ph_name] = self._ph.get(self._ph_name, 0.0) + (now - self._ph_last) * 1e3
        self._ph_last, self._ph_name = now, name

    def chain_report(self, skip=5):
        """mean milliseconds between consecutive MOGAN_CHAIN_EVENTS events over the recorded steps (after `skip` steps)"""
        ch = getattr(self, "_chain", [])
        torch.cuda.synchronize()
        names = []
        for n, _ in ch:
            if n in names:
                break
            names.append(n)
        k = len(names)
        steps = len(ch) // k
        acc = {}
        for st in range(skip, steps):
            for i in range(k):
                a = ch[st * k + i]
                nxt = ch[st * k + i + 1] if st * k + i + 1 < len(ch) else None
                if nxt is None:
                    continue
                acc.setdefault(a[0], []).append(a[1].elapsed_time(nxt[1]))
        return {n: round(sum(v) / len(v), 3) for n, v in acc.items()}

    def device_step(self, b):
        """trainer.py:291-342 given the text embeddings; `b` holds device tensors:
        imgs[3], z, eps, words_embs, sent_emb, mask, cap_lens, tm, tmi, label_one_hot."""
        netG, netsD = self.netG, self.netsD
        B = b["z"].shape[0]
        from ..hip import lib
        # split-K block target of this engine's streams (per stream, include/mogan_hip.h): 384 when the branches run side by
        # side, 768 when one stream has the GPU to itself
        target = int(os.environ.get("MOGAN_SPLIT_TARGET", 384 if self.multi_stream else 768))
        if getattr(self, "_split_target", None) != target:
            self._split_target = target
            from ..hip import ops as _ops
            streams = [torch.cuda.current_stream()] + list(self.side)
            streams += [_ops._wgrad_streams[s.cuda_stream] for s in streams if s.cuda_stream in _ops._wgrad_streams]
            # streams this engine does not own (torch's internal capture stream of the encoder graph) follow the default
            lib.call("mogan_gemm_set_split_target", target)
        real_labels = b["z"].new_ones(B)
        fake_labels = b["z"].new_zeros(B)
        match_labels = b["match_labels"]
        real_feat = {}
        if self.branch_graphs and (self._bg is None or self._bg["B"] == B):     # (a ragged last batch runs eagerly)
            return self._branch_graph_step(b, real_labels, fake_labels, match_labels)
        if self.multi_stream:
            # D_i(real) depends neither on the generator nor on the text encoder: it runs beside them.  With an
            # `inputs_ready` event (recorded by the caller once the batch tensors are on the device) the D branches
            # do not even wait for the main stream, i.e. they also overlap the tail of the previous step (G backward,
            # Adam): only their own stream order (D_i's previous Adam) and the input batch matter.
            cur0 = torch.cuda.current_stream()
            ready = b.get("inputs_ready")
            text_ev = self._text_first(b, ready) if self.split_d else None     # (the real half needs the sentence embedding)
            for i in range(len(netsD))[::-1]:
                if ready is not None:
                    self.side[i].wait_event(ready)
                    if text_ev is not None:
                        self.side[i].wait_event(text_ev)
                else:
                    self.side[i].wait_stream(cur0)
                with torch.cuda.stream(self.side[i]):
                    real_feat[i] = self._d_real(i, b)
            if ready is not None:
                cur0.wait_event(ready)
        self._phase("text+Gfwd")
        if "words_embs" not in b:        # trainer.py:281-289 (eager path: after the fork above)
            b["words_embs"], b["sent_emb"], b["mask"] = self._text_for(b)
        fake_imgs, _, mu, logvar = netG(b["z"], b["sent_emb"], b["words_embs"], b["mask"], b["tmi"],
                                        b["label_one_hot"], b.get("eps"))
        out = {}
        # The three D updates are independent of each other (own parameters, own fake image), so they run
        # largest-first: D256's 643 MB gradient all-reduce then overlaps the D128 and D64 forward/backward and
        # each optimizer step waits only for its own bucket.  Same results as the reference order 0,1,2.
        order = list(range(len(netsD)))[::-1]
        cur = torch.cuda.current_stream()
        nD = len(netsD)
        if self.multi_stream:
            # One branch per discriminator: its update (zero_grad, loss, backward, all-reduce, Adam) and then -- on the
            # same stream, hence behind its own Adam and independent of the other Ds -- the G-step forward through it.
            # The Inception/DAMSM branch only needs the fake image, so it starts right after D256's branch was queued
            # and overlaps the D updates (and, for N>1, D256's 643 MB gradient all-reduce).  The branches meet again
            # where the generator loss is summed; autograd replays each branch's backward on its own stream.
            parts = {}

            def d_head(i):          # D_i: loss on (real, fake), backward -- up to the point where the gradient is complete
                s = self.side[i]
                s.wait_stream(cur)
                with torch.cuda.stream(s):
                    early = real_feat.get(i)
                    if isinstance(early, tuple):
                        out["errD%d" % i] = self._d_fake(i, b, fake_imgs, early)
                    else:
                        errD = self._d_loss(i, b, fake_imgs, real_labels, fake_labels, early)
                        with ops.wgrad_overlap():
                            errD.backward()
                        out["errD%d" % i] = errD.detach()

            def d_tail(i):          # all-reduce, Adam, then the G-step forward through the updated D_i
                with torch.cuda.stream(self.side[i]):
                    self._opt_step(self.optDs[i], self._allreduce_async(self.optDs[i]))
                    for p in netsD[i].parameters():          # G step: no weight gradients of the Ds
                        p.requires_grad_(False)
                    kw = dict(local_labels=b["label_one_hot"], transf_matrices=b["tm"],
                              transf_matrices_inv=b["tmi"]) if i == 0 else {}
                    parts["g_loss%d" % i] = generator_d_branch(netsD[i], fake_imgs[i], b["sent_emb"], **kw)

            # host order: the largest D first (its work starts early), the Inception/DAMSM branch right behind it; the
            # tails smallest-first -- the collectives of one process group execute in issue order, and D64's / D128's
            # all-reduce must not queue behind the event of D256's longer backward
            self._phase("D heads + Inception")
            d_head(order[0])
            s = self.side[nD]
            s.wait_stream(cur)
            damsm_grad = None
            with torch.cuda.stream(s):
                # The DAMSM terms depend on the fake image and the frozen encoders only -- not on the Ds -- so their
                # backward (Inception data gradient, ~6 ms of short launches) does not have to wait for errG_total:
                # it runs here, beside the D updates, on a detached leaf; its image gradient joins the generator's
                # backward below as a second root (d errG / d img256 = D256 path + this; two terms, same sum).
                img = fake_imgs[nD - 1]
                if self.early_damsm_bwd:
                    img = img.detach().requires_grad_(True)
                w_loss, s_loss = generator_damsm_branch(
                    self._encoder(img), img, b["words_embs"], b["sent_emb"], match_labels, b["cap_lens"],
                    b.get("class_ids"), B)
                if self.early_damsm_bwd:
                    damsm_grad, = torch.autograd.grad(ops.scalar_sum([w_loss, s_loss]), img)
                    w_loss, s_loss = w_loss.detach(), s_loss.detach()
                parts["w_loss"], parts["s_loss"] = w_loss, s_loss
            for i in order[1:]:
                d_head(i)
            self._phase("D tails + G-step D fwd")
            for i in order[::-1]:
                d_tail(i)
            self._text_in_window()
            for s in self.side:
                cur.wait_stream(s)
            self._phase("G backward")
            self.optG.zero_grad()
            errG_total = generator_total(parts, nD)
        else:
            prev = None
            for i in order:
                self.optDs[i].zero_grad()
                errD = self._d_loss(i, b, fake_imgs, real_labels, fake_labels)
                with ops.wgrad_overlap():
                    errD.backward()
                pending = self._allreduce_async(self.optDs[i])
                if prev is not None:                         # the previous D's all-reduce hid behind this D's work
                    self._opt_step(self.optDs[prev[0]], prev[1])
                prev = (i, pending)
                out["errD%d" % i] = errD.detach()
            self._opt_step(self.optDs[prev[0]], prev[1])
            # G update: gradients flow through the (updated) Ds to the fake images only
            self.optG.zero_grad()
            for d in netsD:
                for p in d.parameters():
                    p.requires_grad_(False)
            errG_total, parts = generator_loss(netsD, self.image_encoder, fake_imgs, real_labels, b["words_embs"],
                                               b["sent_emb"], match_labels, b["cap_lens"], b.get("class_ids"), None,
                                               local_labels=b["label_one_hot"], transf_matrices=b["tm"],
                                               transf_matrices_inv=b["tmi"], return_logs=False)
        kl_loss = KL_loss(mu, logvar)
        errG_total = ops.scalar_sum([errG_total, kl_loss])          # trainer.py:330
        with ops.wgrad_overlap():
            if self.multi_stream and damsm_grad is not None:
                torch.autograd.backward([errG_total, fake_imgs[nD - 1]], [None, damsm_grad])
            else:
                errG_total.backward()
        for d in netsD:
            for p in d.parameters():
                p.requires_grad_(True)
        self._phase("G adam")
        self._opt_step(self.optG, self._allreduce_async(self.optG))       # Adam + EMA in one launch
        self.bn_counter.flush()                                           # all num_batches_tracked, one launch
        out.update(errG=errG_total.detach(), kl=kl_loss.detach(), fake64=fake_imgs[0].detach(),
                   fake_last=fake_imgs[-1].detach())
        out.update({k: v.detach() for k, v in parts.items()})
        self._phase("end")
        return out

    # -- branch graphs -----------------------------------------------------------------------------------------------
    _BG_KEYS = ("sent_emb", "label_one_hot", "tm", "tmi")

    def _bg_capture(self, b, fake_imgs):
        """Capture the discriminator branches on their streams (see __init__).  Static inputs: the real images, the fake
        images, the sentence embedding and (D_NET64) labels / boxes; static outputs: errD_i, g_loss_i and d g_loss_i / d fake_i.
        The forward of D_i(real) and the rest live in two graphs of one memory pool so that D_i(real) can be replayed early,
        beside the generator's forward."""
        from ..hip import lib as _lib
        from .miscc.losses import _call_d, paired
        netsD, nD = self.netsD, len(self.netsD)
        st = {k: b[k].clone() for k in self._BG_KEYS}
        st["imgs"] = [t.clone() for t in b["imgs"]]
        st["fake"] = [t.detach().clone() for t in fake_imgs]
        st["sent"] = [b["sent_emb"].clone() for _ in self.netsD]       # split form: a copy per branch, written on ITS stream
        split = self.split_d
        B = b["z"].shape[0]
        real_labels, fake_labels = b["z"].new_ones(B), b["z"].new_zeros(B)
        bg = {"static": st, "gR": [], "gU": [], "gA": [], "out": [], "calls": [], "B": B}
        torch.cuda.synchronize()
        counter = self.bn_counter
        # no collective may be captured: the reducers' hooks (armed by zero_grad) stay silent while the backward is recorded
        hooks = [(o, o.on_zero) for o in self.optDs if getattr(o, "on_zero", None) is not None]
        for o, _ in hooks:
            o.on_zero = None
            self.reducers[id(o)].active = False
        for i in range(nD):
            s = self.side[i]
            # (weight gradients stay in line inside the branch graphs: a forked graph replays slowly, 47.1 vs 42.1 ms per step, round 3)
            kw = dict(local_labels=st["label_one_hot"], transf_matrices=st["tm"], transf_matrices_inv=st["tmi"]) if i == 0 else {}
            pool = torch.cuda.graph_pool_handle()
            calls0 = list(counter.calls)
            gR, gU = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            gA = torch.cuda.CUDAGraph() if self.distributed else None
            # The branch's results (errD_i, g_loss_i, d g_loss_i / d fake_i) are read by the MAIN stream during the generator's
            # backward of step k, while gR of step k+1 may already replay on the branch stream (the `inputs_ready` early start):
            # gR shares the private memory pool of gU / gA, so results living INSIDE that pool could be overwritten by gR's
            # temporaries.  They are therefore copied, as the last nodes of the graph, into buffers allocated here, outside the
            # pool; gR never touches those, and the next gU / gA writes them only behind `s.wait_stream(main)` of step k+1.
            res = (torch.empty((), dtype=torch.float32, device=st["fake"][i].device),
                   torch.empty((), dtype=torch.float32, device=st["fake"][i].device), torch.empty_like(st["fake"][i]))
            with _lib.capture_guard():
                with torch.cuda.graph(gR, pool=pool, stream=s):
                    feat = self._d_real(i, st, st["sent"][i])          # (zero_grad; D_i(real) or the whole real half)
                def tail(errD):
                    # Adam (+ re-pack), then the generator-step forward through the updated D_i and its image gradient
                    self._opt_step(self.optDs[i], None)
                    for p in netsD[i].parameters():
                        p.requires_grad_(False)
                    leaf = st["fake"][i].detach().requires_grad_(True)
                    g_loss = generator_d_branch(netsD[i], leaf, st["sent"][i] if split else st["sent_emb"], **kw)
                    g_img, = torch.autograd.grad(g_loss, leaf)
                    for p in netsD[i].parameters():
                        p.requires_grad_(True)
                    for dst, src in zip(res, (errD.detach(), g_loss.detach(), g_img)):
                        dst.copy_(src)
                    return res

                with torch.cuda.graph(gU, pool=pool, stream=s):
                    if isinstance(feat, tuple):
                        errD = self._d_fake(i, st, st["fake"], feat, st["sent"][i])
                    else:
                        errD = discriminator_loss(netsD[i], st["imgs"][i], st["fake"][i], st["sent_emb"], real_labels,
                                                  fake_labels, None, real_features=feat, **kw)
                        with ops.wgrad_overlap():
                            errD.backward()
                    if gA is None:
                        out = tail(errD)
                if gA is not None:
                    with torch.cuda.graph(gA, pool=pool, stream=s):      # replayed behind the bucket's all-reduce
                        out = tail(errD.detach())
            del feat, errD
            bg["gR"].append(gR); bg["gU"].append(gU); bg["gA"].append(gA); bg["out"].append(out)
            bg["calls"].append([a - c for a, c in zip(counter.calls, calls0)])     # BatchNorm calls the replays stand for
            counter.calls = calls0
        for o, h in hooks:
            o.on_zero = h
        torch.cuda.synchronize()
        return bg

    def _g_capture(self, b):
        """Generator forward, and generator backward + Adam/EMA, as two hipGraphs (one memory pool; replayed on the main
        stream).  Static inputs: z, eps, the text tensors, labels / boxes; the backward graph reads the discriminator graphs'
        static outputs (g_loss_i, d g_loss_i / d fake_i) and static copies of the DAMSM branch's results."""
        from ..hip import lib as _lib
        bg, st, nD = self._bg, self._bg["static"], len(self.netsD)
        B, dev = bg["B"], b["z"].device
        gs = {k: b[k].clone() for k in ("z", "words_embs", "mask", "tmi", "label_one_hot")}
        gs["eps"] = b["eps"].clone() if b.get("eps") is not None else torch.randn(B, cfg.GAN.CONDITION_DIM, device=dev)
        gs["damsm_grad"] = torch.zeros_like(st["fake"][nD - 1])
        gs["w_loss"], gs["s_loss"] = torch.zeros((), device=dev), torch.zeros((), device=dev)
        if getattr(self, "_g_cap_stream", None) is None:
            # the capture stream: one per process (not a fresh pool stream per engine: torch's pool has 32 and cycles).  With the
            # forward-only graph the eager backward's nodes RUN on this stream (autograd executes a node where its forward was
            # recorded): its weight gradients use the main stream's weight-gradient side stream, as an eager generator's do
            self._g_cap_stream = _engine_stream(("gcap",))
            main_w = ops.precreate_wgrad_stream(torch.cuda.current_stream())
            ops._wgrad_streams.setdefault(self._g_cap_stream.cuda_stream, main_w)
        counter, cap = self.bn_counter, self._g_cap_stream
        calls0 = list(counter.calls)
        pool = torch.cuda.graph_pool_handle()
        gF, gB = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        # autograd graphs of earlier (eager) iterations must be gone: an AccumulateGrad node that survives from them belongs to the
        # main stream, and the engine would then pull that stream into the capture (a forked graph -- those replay slowly)
        import gc
        gc.collect()
        torch.cuda.synchronize()
        with _lib.capture_guard():
            with torch.cuda.graph(gF, pool=pool, stream=cap):
                fake_imgs, _, mu, logvar = self.netG(gs["z"], st["sent_emb"], gs["words_embs"], gs["mask"], gs["tmi"],
                                                     gs["label_one_hot"], gs["eps"])
            if self.g_fwd_only:
                calls = [a - c for a, c in zip(counter.calls, calls0)]
                counter.calls = calls0
                torch.cuda.synchronize()
                return {"gF": gF, "gB": None, "gs": gs, "fake": [t.detach() for t in fake_imgs], "live": (list(fake_imgs), mu, logvar),
                        "calls": calls}
            with torch.cuda.graph(gB, pool=pool, stream=cap):
                self.optG.zero_grad()
                parts = {"g_loss%d" % i: bg["out"][i][1] for i in range(nD)}
                parts["w_loss"], parts["s_loss"] = gs["w_loss"], gs["s_loss"]
                kl_loss = KL_loss(mu, logvar)
                errG_total = ops.scalar_sum([generator_total(parts, nD), kl_loss.detach()])
                grads = [bg["out"][i][2] for i in range(nD)]
                grads[nD - 1] = ops.add(grads[nD - 1], gs["damsm_grad"])
                with ops.wgrad_overlap():
                    torch.autograd.backward(list(fake_imgs) + [kl_loss], grads + [None])
                self._opt_step(self.optG, None)
                gout = (errG_total.detach(), kl_loss.detach())
        calls = [a - c for a, c in zip(counter.calls, calls0)]
        counter.calls = calls0
        torch.cuda.synchronize()
        return {"gF": gF, "gB": gB, "gs": gs, "fake": [t.detach() for t in fake_imgs], "gout": gout, "calls": calls}

    def _g_graph_for(self, b):
        """the generator graph pair for this batch's text shapes (captured at first use), or None: eager generator"""
        if not self.g_graphs:
            return None
        key = (tuple(b["words_embs"].shape), tuple(b["mask"].shape))
        table = self._bg.setdefault("G", {})
        g = table.get(key)
        if g is None:
            if len(table) < self.G_GRAPH_VARIANTS:
                # (a variant keeps the autograd tape of one generator forward -- all activations of a B = 16 256x256 pass,
                # ~1.7 GB at coco_train.yml widths -- alive for the life of the engine, and its capture synchronises the device)
                logging.getLogger("mogan").info("generator forward graph: capturing variant %d of at most %d for text shapes %r",
                                                len(table) + 1, self.G_GRAPH_VARIANTS, key)
                g = table[key] = self._g_capture(b)
            elif key not in self._g_refused:
                self._g_refused.add(key)
                logging.getLogger("mogan").warning("generator forward graph: text shapes %r run eagerly (%d variants captured already; "
                                                   "pad the captions to fewer lengths or raise TrainEngine.G_GRAPH_VARIANTS)",
                                                   key, len(table))
        return g

    def _branch_graph_step(self, b, real_labels, fake_labels, match_labels):
        """device_step with the discriminator branches replayed as hipGraphs (same streams, same order, same results)."""
        netG, netsD, nD = self.netG, self.netsD, len(self.netsD)
        B = b["z"].shape[0]
        cur = torch.cuda.current_stream()
        if self._bg is None:
            # warm-up (allocator, workspaces, packed weight copies) must not train: two eager steps on this batch, undone
            snap = self._snapshot()
            calls = list(self.bn_counter.calls)
            self.branch_graphs = False
            try:
                wb = {k: v for k, v in b.items() if k != "inputs_ready"}
                # the warm-up draws random numbers (CA_NET's eps when the batch carries none): the generators are put back
                # afterwards, so a run with branch graphs sees the same random stream as one without from step 1 on
                with torch.random.fork_rng(devices=[b["z"].device]):
                    for _ in range(2):
                        self.device_step(dict(wb))
                    with torch.no_grad():
                        if "words_embs" not in b:
                            b["words_embs"], b["sent_emb"], b["mask"] = self.encode_text(b["captions"], b["cap_lens_cpu"])
                        fk, _, _, _ = netG(b["z"], b["sent_emb"], b["words_embs"], b["mask"], b["tmi"], b["label_one_hot"], b.get("eps"))
                self._bg = self._bg_capture(b, fk)
                for o in [self.optG] + self.optDs:
                    o.repack_on_touch = True
            finally:
                self.branch_graphs = True
            self._restore(snap)
            self.bn_counter.calls = calls
            torch.cuda.synchronize()
        bg, st = self._bg, self._bg["static"]
        ready = b.get("inputs_ready")
        text_ev = self._text_first(b, ready) if self.split_d else None
        # D_i(real) -- split form: the whole real half of D_i's update --: beside the text encoder and the generator's forward
        # (and the tail of the previous step)
        for i in range(nD)[::-1]:
            s = self.side[i]
            if ready is not None:
                s.wait_event(ready)
                if text_ev is not None:
                    s.wait_event(text_ev)
            else:
                s.wait_stream(cur)
            with torch.cuda.stream(s):
                st["imgs"][i].copy_(b["imgs"][i])
                if i == 0:
                    for k in ("label_one_hot", "tm", "tmi"):
                        st[k].copy_(b[k])
                if self.split_d:
                    st["sent"][i].copy_(b["sent_emb"])
                bg["gR"][i].replay()
        if ready is not None:
            cur.wait_event(ready)
        self._phase("text+Gfwd")
        if "words_embs" not in b:
            b["words_embs"], b["sent_emb"], b["mask"] = self._text_for(b)
        # (.data: the copy must not move the version counter of a tensor the forward-only generator graph's autograd tape has saved)
        st["sent_emb"].data.copy_(b["sent_emb"])           # (main stream; the branches wait for it below)
        gg = self._g_graph_for(b)
        if gg is not None:
            gs = gg["gs"]
            for k in ("z", "words_embs", "mask", "tmi", "label_one_hot"):
                gs[k].data.copy_(b[k])
            if b.get("eps") is not None:
                gs["eps"].data.copy_(b["eps"])
            else:
                gs["eps"].data.normal_()                    # model.py:333-338: drawn per forward
            gg["gF"].replay()
            fake_imgs, mu, logvar = gg["fake"], None, None
            if gg["gB"] is None:                          # forward-only graph: the live outputs carry the captured autograd graph
                fake_imgs, mu, logvar = gg["live"]
                for j, n in enumerate(gg["calls"]):
                    self.bn_counter.calls[j] += n
        else:
            fake_imgs, _, mu, logvar = netG(b["z"], b["sent_emb"], b["words_embs"], b["mask"], b["tmi"], b["label_one_hot"], b.get("eps"))
        out, parts = {}, {}
        self._phase("D heads + Inception")

        def branch(i):
            s = self.side[i]
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                st["fake"][i].copy_(fake_imgs[i].detach())
                bg["gU"][i].replay()
                if bg["gA"][i] is not None:                # data parallel: sum the bucket over the ranks, then Adam and the rest
                    red = self.reducers.get(id(self.optDs[i]))
                    if red is not None:
                        red.reduce_now()
                    elif not self._debug_no_ar:
                        allreduce_flat(self.optDs[i].g, None, self.comm, self._bucket_name(self.optDs[i]))
                    bg["gA"][i].replay()
            for j, n in enumerate(bg["calls"][i]):
                self.bn_counter.calls[j] += n

        branch(nD - 1)
        s = self.side[nD]
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            img = fake_imgs[nD - 1].detach().requires_grad_(True)
            w_loss, s_loss = generator_damsm_branch(self._encoder(img), img, b["words_embs"], b["sent_emb"], match_labels,
                                                    b["cap_lens"], b.get("class_ids"), B)
            damsm_grad, = torch.autograd.grad(ops.scalar_sum([w_loss, s_loss]), img)
            parts["w_loss"], parts["s_loss"] = w_loss.detach(), s_loss.detach()
        for i in range(nD - 1)[::-1]:
            branch(i)
        self._phase("D tails + G-step D fwd")
        self._text_in_window()
        for s in self.side:
            cur.wait_stream(s)
        self._phase("G backward")
        for i in range(nD):
            errD, g_loss, _ = bg["out"][i]
            out["errD%d" % i] = errD.clone()
            parts["g_loss%d" % i] = g_loss.clone()
        if gg is not None and gg["gB"] is not None:
            gs = gg["gs"]
            gs["damsm_grad"].copy_(damsm_grad)
            gs["w_loss"].copy_(parts["w_loss"])
            gs["s_loss"].copy_(parts["s_loss"])
            gg["gB"].replay()
            for j, n in enumerate(gg["calls"]):
                self.bn_counter.calls[j] += n
            self._phase("G adam")
            self.bn_counter.flush()
            out.update(errG=gg["gout"][0].clone(), kl=gg["gout"][1].clone(), fake64=fake_imgs[0].clone(),
                       fake_last=fake_imgs[-1].clone())
            out.update({k: v.detach() for k, v in parts.items()})
            self._phase("end")
            return out
        self.optG.zero_grad()
        kl_loss = KL_loss(mu, logvar)
        errG_total = ops.scalar_sum([generator_total(parts, nD), kl_loss.detach()])          # the logged value
        grads = [bg["out"][i][2] for i in range(nD)]
        grads[nD - 1] = ops.add(grads[nD - 1], damsm_grad)       # d errG / d img256 = D256 path + DAMSM path
        with ops.wgrad_overlap():
            torch.autograd.backward(list(fake_imgs) + [kl_loss], grads + [None], retain_graph=gg is not None)
            if gg is not None:
                # the tape of a replayed forward was recorded on the capture stream and autograd runs its nodes THERE; the gradient
                # kernels write the parameters' .grad buffers directly (no AccumulateGrad node, hence no end-of-backward stream
                # synchronisation by the engine): the main stream -- Adam -- has to wait for that stream itself
                cur.wait_stream(self._g_cap_stream)
        self._phase("G adam")
        self._opt_step(self.optG, self._allreduce_async(self.optG))
        self.bn_counter.flush()
        keep = (lambda t: t.detach().clone()) if gg is not None else (lambda t: t.detach())     # (a replayed forward rewrites its outputs)
        out.update(errG=errG_total.detach(), kl=keep(kl_loss), fake64=keep(fake_imgs[0]), fake_last=keep(fake_imgs[-1]))
        out.update({k: v.detach() for k, v in parts.items()})
        self._phase("end")
        return out

    # -- text embeddings one step ahead --------------------------------------------------------------------------------
    def prefetch_text(self, captions, cap_lens_cpu):
        """trainer.py:281-289 for the NEXT batch: call this BEFORE step(current batch).  The frozen text encoder (Embedding +
        bi-LSTM, ~80 tiny launches, 0.7 ms) needs the captions only; the step runs it on the main stream right after the generator's
        forward, where that stream otherwise idles for ~25 ms waiting for the discriminator / Inception branches, instead of in front
        of the next generator forward.  (A separate stream did not help: its packets share an in-order hardware queue with the
        branch streams and finished just as late.)  step() picks the result up when it is given the very same captions tensor."""
        self._tx_next = (captions, cap_lens_cpu)

    def _text_first(self, b, ready=None):
        """split discriminator loss: the batch's text embeddings BEFORE the branches fork (their real halves read the sentence
        embedding); returns the event behind the encoder's launches, or None when the caller supplied the embeddings (they are
        then part of the batch, covered by its inputs_ready event / the stream order)"""
        if "words_embs" in b:
            return None
        if ready is not None:
            torch.cuda.current_stream().wait_event(ready)
        b["words_embs"], b["sent_emb"], b["mask"] = self._text_for(b)
        return self._text_ev

    def _text_in_window(self):
        """called by the step between the generator forward and the join with the side branches"""
        nxt, self._tx_next = getattr(self, "_tx_next", None), None
        if nxt is None:
            return
        w, s_, m = self.encode_text(nxt[0], nxt[1])
        ev = torch.cuda.Event()
        ev.record()
        self._tx_ready = (nxt[0], (w, s_, m), ev)

    def _text_for(self, b):
        """the batch's text embeddings: prefetched (see prefetch_text; same stream, so no event) or computed here"""
        hit, self._tx_ready = getattr(self, "_tx_ready", None), None
        if hit is not None and hit[0] is b["captions"]:
            self._text_ev = hit[2]
            return hit[1]
        out = self.encode_text(b["captions"], b["cap_lens_cpu"])
        self._text_ev = None
        if not torch.cuda.is_current_stream_capturing():
            self._text_ev = torch.cuda.Event()
            self._text_ev.record()
        return out

    def encode_text(self, captions, cap_lens):
        """trainer.py:281-289."""
        with torch.no_grad():
            hidden = self.text_encoder.init_hidden(captions.shape[0])
            words_embs, sent_emb = self.text_encoder(captions, cap_lens, hidden)
        mask = (captions == 0)
        if mask.size(1) > words_embs.size(2):
            mask = mask[:, :words_embs.size(2)]
        return words_embs.detach().contiguous(), sent_emb.detach().contiguous(), mask

    def encode_batch_for_cpu(self, b):
        w, s, m = self.encode_text(b["captions"], b["cap_lens_cpu"])
        return {"words_embs": w, "sent_emb": s, "mask": m}

    def step(self, batch):
        """One train iteration on a device batch (see synthetic.make_batch for the fields)."""
        b = dict(batch)
        if "words_embs" not in b and self.use_graph:      # the LSTM stays outside the capture
            b["words_embs"], b["sent_emb"], b["mask"] = self.encode_text(b["captions"], b["cap_lens_cpu"])
        if "match_labels" not in b:
            b["match_labels"] = torch.arange(b["z"].shape[0], device=b["z"].device)
        if not self.use_graph:
            self.last = self.device_step(b)
            return self.last
        return self._graph_step(b)

    def _state_tensors(self):
        ts = []
        for o in [self.optG] + self.optDs:
            ts += [o.p, o.m, o.v, o.state] + ([o.ema] if o.ema is not None else [])
        for net in [self.netG] + list(self.netsD):
            ts += [b for b in net.buffers() if b.dim() > 0]
        return ts + [self.bn_counter.flat]

    def _snapshot(self):
        return [t.clone() for t in self._state_tensors()]

    def _restore(self, snap):
        for t, s in zip(self._state_tensors(), snap):
            t.copy_(s)
        for o in [self.optG] + self.optDs:
            o.repack()

    # -- hipGraph capture of device_step ---------------------------------------------------------------
    _GRAPH_KEYS = ("z", "eps", "words_embs", "sent_emb", "mask", "cap_lens", "tm", "tmi", "label_one_hot",
                   "match_labels")

    def _graph_step(self, b):
        if self.distributed:
            raise RuntimeError("use_graph with RCCL all-reduce inside the capture is not supported; "
                               "run the eager step for N>1")
        if self._graph is None:
            st = {k: b[k].clone() for k in self._GRAPH_KEYS if k in b}
            st["imgs"] = [t.clone() for t in b["imgs"]]
            st["class_ids"] = b.get("class_ids")
            self._static = st
            snap = self._snapshot()                       # warm-up steps must not train
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                 # warm-up (allocator, workspaces, lazy init)
                for _ in range(2):
                    self.device_step(st)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self._graph = torch.cuda.CUDAGraph()
            from ..hip import lib as _lib
            with _lib.capture_guard(), torch.cuda.graph(self._graph):
                # the generator's weight gradients may fork their side stream from the capture stream (a depth-1 fork);
                # the D branches keep theirs in line: a fork of a fork crashes hipStreamEndCapture (ROCm 7.2)
                # (round 2: forking every branch AND weight-gradient stream from the capture stream itself before any work --
                # so that the later branch -> wgrad waits are only extra edges, not forks of forks -- segfaults in
                # hipStreamEndCapture just the same)
                ops.CAPTURE_WGRAD_OK.add(torch.cuda.current_stream().cuda_stream)
                ops.precreate_wgrad_stream(torch.cuda.current_stream())
                self._graph_out = self.device_step(st)
            for o in [self.optG] + self.optDs:
                o.repack_on_touch = True
            self._restore(snap)
        st = self._static
        for k in self._GRAPH_KEYS:
            if k in st:
                st[k].copy_(b[k])
        for dst, src in zip(st["imgs"], b["imgs"]):
            dst.copy_(src)
        self._graph.replay()
        self.last = self._graph_out
        return self.last


def build_networks(n_words=27297, device="cuda", image_encoder=None, seed=None):
    """Random-init networks as trainer.py:53-128 builds them (weights_init: orthogonal)."""
    if seed is not None:
        torch.manual_seed(seed)
    text_encoder = RNN_ENCODER(n_words, nhidden=cfg.TEXT.EMBEDDING_DIM)
    for p in text_encoder.parameters():
        p.requires_grad = False
    text_encoder.eval()
    if image_encoder is None:
        image_encoder = CNN_ENCODER(cfg.TEXT.EMBEDDING_DIM, pretrained=False)
    for p in image_encoder.parameters():
        p.requires_grad = False
    image_encoder.eval()
    netG = G_NET()
    netsD = []
    if cfg.TREE.BRANCH_NUM > 0:
        netsD.append(D_NET64())
    if cfg.TREE.BRANCH_NUM > 1:
        netsD.append(D_NET128())
    if cfg.TREE.BRANCH_NUM > 2:
        netsD.append(D_NET256())
    text_encoder, image_encoder, netG = text_encoder.to(device), image_encoder.to(device), netG.to(device)
    netsD = [d.to(device) for d in netsD]
    if os.environ.get("MOGAN_FAST_INIT"):     # profiling runs: rocSOLVER's QR (orthogonal_) crashes under rocprofv3 --pmc
        for net in [netG] + netsD:
            for p in net.parameters():
                if p.dim() > 1:
                    torch.nn.init.normal_(p, 0.0, (1.0 / p[0].numel()) ** 0.5)
    else:
        netG.apply(weights_init)          # on the device: orthogonal init of the 160M-parameter D256
        for d in netsD:
            d.apply(weights_init)
    return text_encoder, image_encoder, netG, netsD


def draw_bbox_lines(data_img, boxes, imsize):
    """trainer.py:556-566: white (value 1) one-pixel rectangles of the relative boxes (x, y, w, h) on every image of
    the row; pixel coordinates are int(imsize * v) of each value SEPARATELY, w and h are capped at imsize - 1, the first
    absent box (x <= -1) ends the loop."""
    for idx in range(boxes.shape[0]):
        x, y, w, h = tuple(int(imsize * float(v)) for v in boxes[idx])
        w = imsize - 1 if w > imsize - 1 else w
        h = imsize - 1 if h > imsize - 1 else h
        if x <= -1:
            break
        data_img[:, :, y, x:x + w] = 1
        data_img[:, :, y:y + h, x] = 1
        data_img[:, :, y + h, x:x + w] = 1
        data_img[:, :, y:y + h, x + w] = 1
    return data_img


def caption_sentence(cap, ixtoword):
    """trainer.py:569-576: the words of a zero-terminated caption joined by blanks, non-ASCII characters dropped."""
    words = []
    for ix in cap:
        if int(ix) == 0:
            break
        words.append(ixtoword[int(ix)].encode('ascii', 'ignore').decode('ascii'))
    return " ".join(words)


class condGANTrainer(object):
    """Same constructor and public methods as the reference class (trainer.py:29-366)."""

    def __init__(self, output_dir, data_loader, n_words, ixtoword, resume, distributed=False, use_graph=False):
        if cfg.TRAIN.FLAG:
            self.model_dir = os.path.join(output_dir, 'Model')
            self.image_dir = os.path.join(output_dir, 'Image')
            mkdir_p(self.model_dir)
            mkdir_p(self.image_dir)
        self.batch_size = cfg.TRAIN.BATCH_SIZE
        self.max_epoch = cfg.TRAIN.MAX_EPOCH
        self.snapshot_interval = cfg.TRAIN.SNAPSHOT_INTERVAL
        self.resume = resume
        self.gpus = [int(ix) for ix in str(cfg.GPU_ID).split(',')]
        self.n_words, self.ixtoword = n_words, ixtoword
        self.data_loader = data_loader
        self.num_batches = len(self.data_loader) if data_loader is not None else 0
        self.distributed, self.use_graph = distributed, use_graph
        self.log_images = os.environ.get("MOGAN_LOG_IMAGES", "1") != "0"
        # one process per GPU: the local device comes from LOCAL_RANK (torchrun), not from GPU_ID
        self.device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
        torch.cuda.set_device(self.device)

    def build_models(self):
        text_encoder, image_encoder, netG, netsD = build_networks(self.n_words, self.device)
        epoch = 0
        # trainer.py:53-67: the DAMSM encoders condition G and define w_loss / s_loss -- a wrong path must not silently
        # train against random encoders.  NET_E == '' is the explicit synthetic / benchmark setting (random-init encoders).
        if cfg.TRAIN.NET_E == '':
            print("WARNING: TRAIN.NET_E is empty -- random-init DAMSM encoders (synthetic / benchmark runs only)")
        else:
            img_path = cfg.TRAIN.NET_E.replace('text_encoder', 'image_encoder')
            for path in (cfg.TRAIN.NET_E, img_path):
                if not os.path.isfile(path):
                    raise FileNotFoundError("DAMSM encoder checkpoint not found: %s (cwd %s); set TRAIN.NET_E: '' to "
                                            "train against random-init encoders" % (path, os.getcwd()))
            text_encoder.load_state_dict(torch.load(cfg.TRAIN.NET_E, map_location='cpu'))
            image_encoder.load_state_dict(torch.load(img_path, map_location='cpu'))
            print('Load text / image encoder from:', cfg.TRAIN.NET_E, img_path)
        self._resume_sd = None
        if self.resume:
            ckpts = sorted(glob.glob(self.model_dir + "/" + '*.pth'))
            if not ckpts:
                raise FileNotFoundError("--resume %s: no checkpoint_*.pth under %s" % (self.resume, self.model_dir))
            sd = torch.load(ckpts[-1], map_location='cpu')
            netG.load_state_dict(sd["netG"])
            for i in range(len(netsD)):
                netsD[i].load_state_dict(sd["netD"][i])
            epoch = int(ckpts[-1][-8:-4]) + 1
            self._resume_sd = sd                               # optimizer state: define_optimizers
            print('Resume from:', ckpts[-1])
        return [text_encoder, image_encoder, netG, netsD, epoch]

    def define_optimizers(self, netG, netsD):
        """trainer.py:137-159: on resume the Adam moments and step counters come back too (the flat buckets are the
        engine's; both torch's optimizer layout and the round-1 flat layout are accepted)."""
        optG, optDs = self.engine.optG, self.engine.optDs
        sd = getattr(self, "_resume_sd", None)
        if sd is not None:
            if "optimG" in sd:
                optG.load_state_dict(sd["optimG"])
            for i, o in enumerate(optDs):
                if "optimD" in sd and i < len(sd["optimD"]):
                    o.load_state_dict(sd["optimD"][i])
            # the checkpoint's netG holds the EMA weights (save_model swaps them in): they are both the restart point of
            # the raw weights -- as in the reference -- and of the shadow copy
            self._resume_sd = None
            if self.distributed and dist.is_initialized() and dist.get_world_size() > 1:
                self.engine.sync_replicas()
        return optG, optDs

    def prepare_labels(self):
        B = self.batch_size
        return (torch.ones(B, device=self.device), torch.zeros(B, device=self.device),
                torch.arange(B, device=self.device))

    def save_model(self, netG, avg_param_G, netsD, optimG, optimsD, epoch, max_to_keep=5):
        """trainer.py:173-199: netG is saved with the EMA weights swapped in; newest 5 kept."""
        backup_para = copy_G_params(netG)
        load_params(netG, avg_param_G)
        checkpoint = {'epoch': epoch, 'netG': netG.state_dict(), 'optimG': optimG.state_dict(),
                      'netD': [d.state_dict() for d in netsD], 'optimD': [o.state_dict() for o in optimsD]}
        torch.save(checkpoint, "{}/checkpoint_{:04}.pth".format(self.model_dir, epoch))
        load_params(netG, backup_para)
        if max_to_keep is not None and max_to_keep > 0:
            ckpts = sorted(glob.glob(self.model_dir + "/" + '*.pth'))
            while len(ckpts) > max_to_keep:
                os.remove(ckpts[0])
                ckpts = ckpts[1:]

    def save_img_results(self, netG, noise, sent_emb, words_embs, mask, image_encoder, captions, cap_lens,
                         gen_iterations, transf_matrices_inv, label_one_hot, name='current'):
        """trainer.py:206-248: G_<name>_<it>_<i>.png = attention grids of the generator's two attention stages,
        D_<name>_<it>.png = the DAMSM word-region attention on the 256x256 fake image."""
        from PIL import Image
        from .miscc.losses import words_loss
        from .miscc.vis import build_super_images
        written = []
        with torch.no_grad():
            fake_imgs, attention_maps, _, _ = netG(noise, sent_emb, words_embs, mask, transf_matrices_inv, label_one_hot)
            for i in range(len(attention_maps)):
                if len(fake_imgs) > 1:
                    img, lr_img = fake_imgs[i + 1].detach().cpu(), fake_imgs[i].detach().cpu()
                else:
                    img, lr_img = fake_imgs[0].detach().cpu(), None
                attn_maps = attention_maps[i]
                res = build_super_images(img, captions, self.ixtoword, attn_maps, attn_maps.size(2), lr_imgs=lr_img,
                                         batch_size=captions.size(0))
                if res is not None:
                    fullpath = '%s/G_%s_%d_%d.png' % (self.image_dir, name, gen_iterations, i)
                    Image.fromarray(res[0]).save(fullpath)
                    written.append(fullpath)
            region_features, _ = image_encoder(fake_imgs[-1].detach())
            _, _, att_maps = words_loss(region_features.detach(), words_embs.detach(), None, cap_lens, None,
                                        captions.size(0))
            res = build_super_images(fake_imgs[-1].detach().cpu(), captions, self.ixtoword, att_maps,
                                     region_features.size(2), batch_size=captions.size(0))
            if res is not None:
                fullpath = '%s/D_%s_%d.png' % (self.image_dir, name, gen_iterations)
                Image.fromarray(res[0]).save(fullpath)
                written.append(fullpath)
        return written

    def sampling(self, split_dir, num_samples=30000):
        """trainer.py:387-470: load cfg.TRAIN.NET_G (EMA generator of a checkpoint) and cfg.TRAIN.NET_E (DAMSM text
        encoder), generate one 256x256 image per caption of the data loader with netG.eval() and save it as
        <NET_G minus .pth>/<split>/single/<key>_s<batch index>.png"""
        from PIL import Image
        from .datasets import prepare_data
        if cfg.TRAIN.NET_G == '':
            print('Error: the path for morels is not found!')
            return None
        if split_dir == 'test':
            split_dir = 'valid'
        netG, text_encoder = self._load_eval_models()
        save_dir = '%s/%s' % (cfg.TRAIN.NET_G[:cfg.TRAIN.NET_G.rfind('.pth')], split_dir)
        mkdir_p(save_dir)
        written = []
        for step, data in enumerate(self.data_loader, 0):
            if step >= num_samples:
                break
            imgs, captions, cap_lens, class_ids, keys, (tm, tmi), label_one_hot = prepare_data(data, self.device)
            B = captions.shape[0]
            with torch.no_grad():
                hidden = text_encoder.init_hidden(B)
                words_embs, sent_emb = text_encoder(captions, cap_lens.cpu(), hidden)
                mask = (captions == 0)
                if mask.size(1) > words_embs.size(2):
                    mask = mask[:, :words_embs.size(2)]
                noise = torch.randn(B, cfg.GAN.Z_DIM, device=self.device)
                fake_imgs, _, _, _ = netG(noise, sent_emb.contiguous(), words_embs.contiguous(), mask, tmi, label_one_hot)
            out = fake_imgs[-1].add(1.0).mul(127.5).clamp(0, 255).byte().permute(0, 2, 3, 1).cpu().numpy()
            for j in range(B):
                s_tmp = '%s/single/%s' % (save_dir, keys[j])
                mkdir_p(s_tmp[:s_tmp.rfind('/')] if '/' in keys[j] else '%s/single' % save_dir)
                fullpath = '%s_s%d.png' % (s_tmp, step)
                Image.fromarray(out[j]).save(fullpath)
                written.append(fullpath)
        return written

    def _load_eval_models(self):
        """The two checkpoints of the evaluation paths (trainer.py:397-417, 483-505): EMA generator from
        cfg.TRAIN.NET_G["netG"], DAMSM text encoder from cfg.TRAIN.NET_E, both in eval mode."""
        netG = G_NET()
        netG.apply(weights_init)
        sd = torch.load(cfg.TRAIN.NET_G, map_location='cpu')
        netG.load_state_dict(sd["netG"])
        print('Load G from: ', cfg.TRAIN.NET_G)
        netG = netG.to(self.device).eval()
        text_encoder = RNN_ENCODER(self.n_words, nhidden=cfg.TEXT.EMBEDDING_DIM)
        if cfg.TRAIN.NET_E != '':
            text_encoder.load_state_dict(torch.load(cfg.TRAIN.NET_E, map_location='cpu'))
            print('Load text encoder from:', cfg.TRAIN.NET_E)
        return netG, text_encoder.to(self.device).eval()

    def sample(self, split_dir, num_samples=25, draw_bbox=False):
        """trainer.py:474-579 (what main.py:158 runs for B_VALIDATION): for the first `num_samples` batches of an
        eval-mode loader take the FIRST sample of the sorted batch, generate nine 256x256 images for its caption /
        boxes / labels from nine noise vectors with netG.eval(), and save one row [real | 9 fakes] (boxes drawn as white
        lines when draw_bbox) as <NET_G minus .pth>_<split>/<caption>_<step>.png."""
        from .datasets import prepare_data
        from ..stackgan.logging_utils import save_image
        if cfg.TRAIN.NET_G == '':
            print('Error: the path for model NET_G is not found!')
            return None
        if split_dir == 'test':
            split_dir = 'valid'
        netG, text_encoder = self._load_eval_models()
        save_dir = '%s_%s' % (cfg.TRAIN.NET_G[:cfg.TRAIN.NET_G.rfind('.pth')], split_dir)
        mkdir_p(save_dir)
        imsize, nrep = cfg.TREE.BASE_SIZE << (cfg.TREE.BRANCH_NUM - 1), 9
        written, step = [], 0
        for step, data in enumerate(self.data_loader, 0):
            if step >= num_samples:
                break
            imgs, captions, cap_lens, class_ids, keys, (tm, tmi), label_one_hot, bbox = \
                prepare_data(data, self.device, eval=True)
            with torch.no_grad():
                hidden = text_encoder.init_hidden(captions.shape[0])
                words_embs, sent_emb = text_encoder(captions, cap_lens.cpu(), hidden)
                words_embs = words_embs[0:1].repeat(nrep, 1, 1).contiguous()
                sent_emb = sent_emb[0:1].repeat(nrep, 1).contiguous()
                mask = (captions == 0)[0:1]
                if mask.size(1) > words_embs.size(2):
                    mask = mask[:, :words_embs.size(2)]
                mask = mask.repeat(nrep, 1)
                noise = torch.randn(nrep, cfg.GAN.Z_DIM, device=self.device)
                fake_imgs, _, _, _ = netG(noise, sent_emb, words_embs, mask, tmi[0:1].repeat(nrep, 1, 1, 1).contiguous(),
                                          label_one_hot[0:1].repeat(nrep, 1, 1).contiguous())
            data_img = torch.zeros(1 + nrep, 3, imsize, imsize)
            data_img[0] = imgs[-1][0].cpu()
            data_img[1:] = fake_imgs[-1].float().cpu()
            if draw_bbox:
                draw_bbox_lines(data_img, np.asarray(bbox[0]), imsize)
            sentence = caption_sentence(captions[0].cpu().numpy(), self.ixtoword)
            fullpath = '{}/{}_{}.png'.format(save_dir, sentence, step)
            save_image(data_img, fullpath, nrow=1 + nrep, normalize=True)
            written.append(fullpath)
        print("Saved {} files to {}".format(len(written), save_dir))
        return written

    def train(self):
        from .datasets import prepare_data
        text_encoder, image_encoder, netG, netsD, start_epoch = self.build_models()
        self.engine = TrainEngine(text_encoder, image_encoder, netG, netsD, self.distributed, self.use_graph)
        optimizerG, optimizersD = self.define_optimizers(netG, netsD)
        if self.distributed and dist.is_initialized():
            # common seed up to here (identical replicas; sync_replicas makes that unconditional), from here on every rank
            # draws its own z / eps / augmentation stream: N ranks must not compute the same gradient N times
            torch.manual_seed(torch.initial_seed() + 1 + dist.get_rank())
        nz = cfg.GAN.Z_DIM
        gen_iterations = 0
        fixed_noise = None
        feeder = None
        if getattr(self.data_loader.dataset, "raw", False):      # TextDataset(raw=True): augmentation on the device
            from .datasets import prepare_data_raw
            from .feeder import DeviceFeeder
            feeder = DeviceFeeder(self.device, self.batch_size, sizes=tuple(cfg.TREE.BASE_SIZE << i
                                                                            for i in range(cfg.TREE.BRANCH_NUM)))
        for epoch in range(start_epoch, self.max_epoch):
            start_t = time.time()
            logs = {}
            sampler = getattr(self.data_loader, "sampler", None)
            if hasattr(sampler, "set_epoch"):
                sampler.set_epoch(epoch)                   # DistributedSampler: a new partition of the epoch per epoch
            def prepared(data):
                if feeder is not None:
                    imgs, captions, cap_lens, class_ids, keys, (tm, tmi), label_one_hot = prepare_data_raw(data, feeder)
                else:
                    imgs, captions, cap_lens, class_ids, keys, (tm, tmi), label_one_hot = prepare_data(data, self.device)
                d = dict(imgs=imgs, captions=captions, cap_lens=cap_lens, cap_lens_cpu=cap_lens.cpu(),
                         class_ids=class_ids, tm=tm, tmi=tmi, label_one_hot=label_one_hot)
                # everything the D_i(real) branches read (images, labels, boxes) is queued on this stream by now -- host copies
                # or the DeviceFeeder's kernels: with this event those branches of the batch's step start without waiting for
                # the tail of the previous step on the main stream (TrainEngine.device_step; what bench.py times)
                d["inputs_ready"] = torch.cuda.Event()
                d["inputs_ready"].record()
                return d

            # one batch of look-ahead: while step k runs, batch k+1 is already on the device and its captions go through the frozen
            # text encoder (TrainEngine.prefetch_text) -- trainer.py:276-289 software-pipelined, one text encoding per batch as there
            loader = iter(self.data_loader)
            first = next(loader, None)
            nxt = prepared(first) if first is not None else None
            while nxt is not None:
                batch = nxt
                batch["z"] = torch.randn(batch["captions"].shape[0], nz, device=self.device)
                imgs, captions, cap_lens, tmi, label_one_hot = (batch[k] for k in ("imgs", "captions", "cap_lens", "tmi",
                                                                                   "label_one_hot"))
                data = next(loader, None)
                nxt = prepared(data) if data is not None else None
                if nxt is not None and not self.use_graph:
                    self.engine.prefetch_text(nxt["captions"], nxt["cap_lens_cpu"])
                logs = self.engine.step(batch)
                if gen_iterations % 1000 == 0:        # trainer.py:320-351: the reference increments gen_iterations before
                    # this test, i.e. it first logs / saves at iteration 1000; here the very first iteration is included
                    # as well (an early sanity image)
                    print(' '.join('%s: %.2f' % (k, float(v)) for k, v in logs.items() if v.dim() == 0))
                    if self.log_images and captions.shape[0] >= 8 and (not self.distributed or dist.get_rank() == 0):
                        if fixed_noise is None or fixed_noise.shape[0] != captions.shape[0]:
                            fixed_noise = torch.randn(captions.shape[0], nz, device=self.device)
                        w, s_, m = self.engine.encode_text(captions, batch["cap_lens_cpu"])
                        backup_para = copy_G_params(netG)
                        load_params(netG, optimizerG.ema_params())
                        self.save_img_results(netG, fixed_noise, s_, w, m, image_encoder, captions, cap_lens, epoch,
                                              tmi, label_one_hot, name='average')
                        load_params(netG, backup_para)
                        self.engine.repack_all()     # (load_params marks every packed copy stale; graphs do not re-check)
                gen_iterations += 1
            end_t = time.time()
            if logs:
                errD_total = sum(float(logs["errD%d" % i]) for i in range(len(netsD)))
                print('''[%d/%d][%d]
                  Loss_D: %.2f Loss_G: %.2f Time: %.2fs''' % (epoch, self.max_epoch, self.num_batches,
                                                            errD_total, float(logs["errG"]), end_t - start_t))
            if epoch % cfg.TRAIN.SNAPSHOT_INTERVAL == 0 and (not self.distributed or dist.get_rank() == 0):
                self.save_model(netG, optimizerG.ema_params(), netsD, optimizerG, optimizersD, epoch)
        if not self.distributed or dist.get_rank() == 0:
            self.save_model(netG, optimizerG.ema_params(), netsD, optimizerG, optimizersD, self.max_epoch - 1)
