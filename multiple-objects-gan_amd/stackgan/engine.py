"""Device side of the StackGAN-family train step (one rank): the minibatch body of
code/coco/stackgan/trainer.py:188-231, code/clevr/trainer.py:127-157 and
code/multi-mnist/trainer.py:131-160 --

    G forward (once) -> zero_grad(D), discriminator loss (real, wrong, fake[, uncond]), backward, Adam(D)
                     -> zero_grad(G), generator loss through the *updated* D [+ KL * coeff], backward, Adam(G)

on flat fp32 parameter/gradient/moment buckets (one fused Adam launch and one RCCL all-reduce per
network), optionally captured into one hipGraph.  Same machinery as attngan/trainer.py's TrainEngine.
"""
import os

import torch
import torch.distributed as dist

from ..attngan.model_base import BNCallCounter
from ..attngan.trainer import FlatAdam, allreduce_flat
from ..hip import ops
from . import losses


class StackGANEngine:
    """`variant` is one of nets.COCO / CLEVR / MNIST; `stage` 1 or 2 (stage 2 exists for coco only).
    Batch fields (device tensors): real_imgs, z, tm, tmi, label_one_hot and, for coco, txt_embedding,
    eps [, eps_s1, tm_s2, tmi_s2 in stage 2]."""

    _KEYS = ("real_imgs", "z", "tm", "tmi", "label_one_hot", "txt_embedding", "eps", "eps_s1", "tm_s2", "tmi_s2")

    def __init__(self, netG, netD, cfg, variant, stage=1, distributed=False, use_graph=False):
        self.netG, self.netD, self.cfg, self.variant, self.stage = netG, netD, cfg, variant, stage
        eps_mode = int(cfg.get("ADAM_EPS_MODE", 0))
        self.optG = FlatAdam(netG, cfg.TRAIN.GENERATOR_LR, eps_mode=eps_mode)
        self.optD = FlatAdam(netD, cfg.TRAIN.DISCRIMINATOR_LR, eps_mode=eps_mode)
        self.bn_counter = BNCallCounter([netG, netD])
        self.distributed = bool(distributed) and dist.is_available() and dist.is_initialized() \
            and (dist.get_world_size() > 1 or bool(os.environ.get("MOGAN_FORCE_DIST")))
        self.world = dist.get_world_size() if self.distributed else 1
        self.use_graph = use_graph
        self._graph, self._static, self.last = None, None, {}
        self.side = torch.cuda.Stream()          # D(real) runs here, beside the generator forward

    def set_lr(self, generator_lr, discriminator_lr):
        """the reference halves both rates every LR_DECAY_EPOCH epochs (S/trainer.py:140-147)."""
        if self._graph is not None and (generator_lr != self.optG.lr or discriminator_lr != self.optD.lr):
            self._graph = None                      # lr is a launch argument baked into the capture
        self.optG.lr, self.optD.lr = float(generator_lr), float(discriminator_lr)

    # ------------------------------------------------------------------------------------------
    def _sync_step(self, flat):
        if self.distributed:            # on the stream the gradient was produced on (see attngan/trainer.py)
            allreduce_flat(flat.g, None)
        flat.step(grad_scale=1.0 / self.world)

    def generate(self, b):
        v = self.variant
        if v.text and self.stage == 2:
            _, fake, mu, logvar, _ = self.netG(b["txt_embedding"], b["z"], b["tmi"], b["tm_s2"], b["tmi_s2"],
                                               b["label_one_hot"], eps=b.get("eps"), eps_s1=b.get("eps_s1"))
        elif v.text:
            _, fake, mu, logvar, _ = self.netG(b["txt_embedding"], b["z"], b["tmi"], b["label_one_hot"],
                                               eps=b.get("eps"))
        else:
            out = self.netG(b["z"], b["tmi"], b["label_one_hot"])
            fake, mu, logvar = (out[1] if isinstance(out, tuple) else out), None, None
        return fake, mu, logvar

    def device_step(self, b):
        v, netG, netD = self.variant, self.netG, self.netD
        tm, tmi = (b["tm_s2"], b["tmi_s2"]) if self.stage == 2 else (b["tm"], b["tmi"])
        # Everything that touches D runs on ONE side stream (its weight gradients accumulate straight into the flat
        # .grad bucket, so they must stay ordered): D(real) does not depend on the generator and starts beside the G
        # forward; the D update and the G-step forward through the updated D follow on the same stream.
        cur, side = torch.cuda.current_stream(), self.side
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            self.optD.zero_grad()
            real_features = netD(b["real_imgs"], b["label_one_hot"].detach(), tm, tmi)
        fake_imgs, mu, logvar = self.generate(b)
        cond = mu if v.text else losses.label_condition(b["label_one_hot"], clamp=(v.name == "clevr"))
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            errD, errD_real, errD_wrong, errD_fake = losses.discriminator_loss(
                netD, b["real_imgs"], fake_imgs, b["label_one_hot"], tm, tmi, cond, real_features=real_features)
            with ops.wgrad_overlap():
                errD.backward()
            self._sync_step(self.optD)
            # G update through the updated D; D's own weight gradients are not needed (the reference computes
            # them and drops them at the next zero_grad)
            for p in netD.parameters():
                p.requires_grad_(False)
            errG = losses.generator_loss(netD, fake_imgs, b["label_one_hot"], tm, tmi, cond)
        cur.wait_stream(side)
        self.optG.zero_grad()
        out = dict(errD=errD.detach(), errD_real=errD_real, errD_wrong=errD_wrong, errD_fake=errD_fake,
                   errG=errG.detach(), fake=fake_imgs.detach())
        if v.text:
            kl = losses.KL_loss(mu, logvar)
            errG_total = errG + kl * float(self.cfg.TRAIN.COEFF.KL)
            out["kl"] = kl.detach()
        else:
            errG_total = errG
        with ops.wgrad_overlap():
            errG_total.backward()
        for p in netD.parameters():
            p.requires_grad_(True)
        self._sync_step(self.optG)
        self.bn_counter.flush()
        return out

    def step(self, batch):
        if not self.use_graph:
            self.last = self.device_step(batch)
            return self.last
        return self._graph_step(batch)

    # ------------------------------------------------------------------------------------------
    def _state_tensors(self):
        ts = []
        for o in (self.optG, self.optD):
            ts += [o.p, o.m, o.v, o.state]
        for net in (self.netG, self.netD):
            ts += [t for t in net.buffers() if t.dim() > 0]
            ts += [p.data for p in net.parameters() if not p.requires_grad]
        return ts + [self.bn_counter.flat]

    def _graph_step(self, b):
        if self.distributed:
            raise RuntimeError("use_graph with the RCCL all-reduce inside the capture is not supported; "
                               "run the eager step for N>1")
        if self._graph is None:
            st = {k: b[k].clone() for k in self._KEYS if b.get(k) is not None}
            self._static = st
            snap = [t.clone() for t in self._state_tensors()]          # warm-up steps must not train
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    self.device_step(st)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self._graph = torch.cuda.CUDAGraph()
            from ..hip import lib as _lib
            with _lib.capture_guard(), torch.cuda.graph(self._graph):
                self._graph_out = self.device_step(st)
            for t, s in zip(self._state_tensors(), snap):
                t.copy_(s)
            for o in (self.optG, self.optD):
                o.repack()               # the replayed graph uses the packed weight copies without a pack node of its own
        for k, dst in self._static.items():
            dst.copy_(b[k])
        self._graph.replay()
        self.last = self._graph_out
        return self.last
