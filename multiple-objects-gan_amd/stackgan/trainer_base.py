"""Train loop shared by the three StackGAN-family trees (code/coco/stackgan/trainer.py:29-306,
code/clevr/trainer.py:29-196, code/multi-mnist/trainer.py:28-206): network construction + weights_init,
optional checkpoint loading, halving both learning rates every LR_DECAY_EPOCH epochs, one
StackGANEngine.step per minibatch, `save_model` snapshots, and every 500 iterations the scalar summaries and the
sample grids of the reference (logging_utils.py).  The grid shows the fake batch of the step
itself (the reference runs the generator a second time on the same inputs for it); losses are printed per epoch with
the reference's format string."""
import glob
import os
import time

import torch
import torch.distributed as dist
import torch.nn as nn

from ..attngan.miscc.utils import mkdir_p
from ..attngan.synthetic import bbox_to_theta, one_hot_labels
from .engine import StackGANEngine
from .logging_utils import ScalarWriter, save_img_results
from .synthetic import _theta64


def weights_init(m):
    """S/miscc/utils.py:129-139: conv/linear weights ~ N(0, 0.02), BN gamma ~ N(1, 0.02), beta = 0."""
    classname = m.__class__.__name__
    if classname.find('Conv') != -1:
        m.weight.data.normal_(0.0, 0.02)
    elif classname.find('BatchNorm') != -1:
        m.weight.data.normal_(1.0, 0.02)
        m.bias.data.fill_(0)
    elif classname.find('Linear') != -1:
        m.weight.data.normal_(0.0, 0.02)
        if m.bias is not None:
            m.bias.data.fill_(0.0)


def save_model(netG, netD, optimG, optimD, epoch, model_dir, saveD=False, saveOptim=False, max_to_keep=5):
    """S/miscc/utils.py:162-176: same dict layout and file name; the newest `max_to_keep` files stay."""
    checkpoint = {'epoch': epoch,
                  'netG': netG.state_dict(),
                  'optimG': optimG.state_dict() if saveOptim else {},
                  'netD': netD.state_dict() if saveD else {},
                  'optimD': optimD.state_dict() if saveOptim else {}}
    torch.save(checkpoint, "{}/checkpoint_{:04}.pth".format(model_dir, epoch))
    print('Save G/D models')
    if max_to_keep is not None and max_to_keep > 0:
        ckpts = sorted(glob.glob(model_dir + "/" + '*.pth'))
        while len(ckpts) > max_to_keep:
            os.remove(ckpts[0])
            ckpts = ckpts[1:]


class GANTrainerBase(object):
    """Per-tree subclasses set `cfg`, `model` (the tree's model module) and `tree`."""
    cfg = None
    model = None
    tree = None

    def __init__(self, output_dir, distributed=False, use_graph=False):
        cfg = self.cfg
        if cfg.TRAIN.FLAG:
            self.model_dir = os.path.join(output_dir, 'Model')
            self.image_dir = os.path.join(output_dir, 'Image')
            self.log_dir = os.path.join(output_dir, 'Log')
            for d in (self.model_dir, self.image_dir, self.log_dir):
                mkdir_p(d)
        self.max_epoch = cfg.TRAIN.MAX_EPOCH
        self.snapshot_interval = cfg.TRAIN.SNAPSHOT_INTERVAL
        self.gpus = [int(ix) for ix in str(cfg.GPU_ID).split(',')]
        self.num_gpus = len(self.gpus)
        self.batch_size = cfg.TRAIN.BATCH_SIZE
        self.max_objects = self.model.VARIANT.max_objects
        self.distributed, self.use_graph = distributed, use_graph
        # one process per GPU: the device comes from LOCAL_RANK (torchrun), not from GPU_ID
        self.device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
        torch.cuda.set_device(self.device)

    # ----------------------------------------------------------------------------- networks
    def load_network_stageI(self):
        cfg, model = self.cfg, self.model
        netG, netD = model.STAGE1_G(), model.STAGE1_D()
        netG.apply(weights_init)
        netD.apply(weights_init)
        if cfg.NET_G != '':
            netG.load_state_dict(torch.load(cfg.NET_G, map_location='cpu')["netG"])
            print('Load from: ', cfg.NET_G)
        if cfg.NET_D != '':
            netD.load_state_dict(torch.load(cfg.NET_D, map_location='cpu'))
            print('Load from: ', cfg.NET_D)
        return netG.to(self.device), netD.to(self.device)

    def load_network_stageII(self):
        cfg, model = self.cfg, self.model
        netG = model.STAGE2_G(model.STAGE1_G())
        netG.apply(weights_init)
        if cfg.NET_G != '':
            netG.load_state_dict(torch.load(cfg.NET_G, map_location='cpu')["netG"])
            print('Load from: ', cfg.NET_G)
        elif cfg.STAGE1_G != '':
            netG.STAGE1_G.load_state_dict(torch.load(cfg.STAGE1_G, map_location='cpu')["netG"])
            print('Load from: ', cfg.STAGE1_G)
        else:
            print("Please give the Stage1_G path")
            return None
        netD = model.STAGE2_D()
        netD.apply(weights_init)
        if cfg.NET_D != '':
            netD.load_state_dict(torch.load(cfg.NET_D, map_location='cpu'))
            print('Load from: ', cfg.NET_D)
        return netG.to(self.device), netD.to(self.device)

    # ----------------------------------------------------------------------------- minibatch prologue
    def prepare_batch(self, data, stage):
        """What the reference does between `for i, data in enumerate(data_loader)` and the generator call."""
        dev, K, tree = self.device, self.max_objects, self.tree
        b = {"real_imgs": data[0].to(dev).float()}
        B = b["real_imgs"].shape[0]
        if tree == "clevr":                               # C/trainer.py:114-125: matrices come from the Dataset
            tm, tmi = data[1]
            b["tm"], b["tmi"] = tm.to(dev).float(), tmi.to(dev).float()
            b["label_one_hot"] = data[2].to(dev).float()
        elif tree == "mnist":                             # M/trainer.py:117-129: float64 boxes
            tm, tmi = _theta64(data[1].view(-1, 4))
            b["tm"], b["tmi"] = tm.view(B, K, 2, 3).to(dev), tmi.view(B, K, 2, 3).to(dev)
            b["label_one_hot"] = data[2].to(dev).float()
        else:                                             # S/trainer.py:156-186
            bbox, label = data[1], data[2]
            sets = list(bbox) if stage == 2 else [bbox]
            tm, tmi = bbox_to_theta(sets[0].view(-1, 4))
            b["tm"], b["tmi"] = tm.view(B, K, 2, 3).to(dev), tmi.view(B, K, 2, 3).to(dev)
            if stage == 2:
                tm2, tmi2 = bbox_to_theta(sets[1].view(-1, 4))
                b["tm_s2"], b["tmi_s2"] = tm2.view(B, K, 2, 3).to(dev), tmi2.view(B, K, 2, 3).to(dev)
                b["eps_s1"] = torch.randn(B, self.cfg.GAN.CONDITION_DIM, device=dev)
            b["label_one_hot"] = one_hot_labels(label.view(B, K)).to(dev)
            b["txt_embedding"] = data[3].to(dev).float()
            b["eps"] = torch.randn(B, self.cfg.GAN.CONDITION_DIM, device=dev)
        b["z"] = torch.randn(B, self.cfg.Z_DIM, device=dev)
        return b

    # ----------------------------------------------------------------------------- train
    def train(self, data_loader, stage=1):
        cfg = self.cfg
        nets = self.load_network_stageI() if stage == 1 else self.load_network_stageII()
        if nets is None:
            return
        netG, netD = nets
        engine = StackGANEngine(netG, netD, cfg, self.model.VARIANT, stage, self.distributed, self.use_graph)
        self.engine = engine
        generator_lr, discriminator_lr = cfg.TRAIN.GENERATOR_LR, cfg.TRAIN.DISCRIMINATOR_LR
        lr_decay_step = cfg.TRAIN.LR_DECAY_EPOCH
        rank0 = not self.distributed or dist.get_rank() == 0
        count, epoch, logs, writer = 0, 0, {}, None
        print("Start training...")
        for epoch in range(self.max_epoch):
            start_t = time.time()
            if epoch % lr_decay_step == 0 and epoch > 0:
                generator_lr *= 0.5
                discriminator_lr *= 0.5
                engine.set_lr(generator_lr, discriminator_lr)
            i = -1
            for i, data in enumerate(data_loader, 0):
                logs = engine.step(self.prepare_batch(data, stage))
                count += 1
                if i % 500 == 0 and rank0 and cfg.TRAIN.FLAG:          # S/trainer.py:237-260
                    if writer is None:
                        writer = ScalarWriter(self.log_dir)
                    for tag, key in (('D_loss', 'errD'), ('D_loss_real', 'errD_real'), ('D_loss_wrong', 'errD_wrong'),
                                     ('D_loss_fake', 'errD_fake'), ('G_loss', 'errG'), ('KL_loss', 'kl')):
                        if key in logs:
                            writer.add_scalar(tag, float(logs[key]), count)
                    save_img_results(data[0], logs["fake"], epoch, self.image_dir, getattr(cfg, "VIS_COUNT", 64))
            end_t = time.time()
            if logs and rank0:
                kl = ' Loss_KL: %.4f' % float(logs["kl"]) if "kl" in logs else ''
                print('''[%d/%d][%d/%d] Loss_D: %.4f Loss_G: %.4f%s
                     Loss_real: %.4f Loss_wrong:%.4f Loss_fake %.4f
                     Total Time: %.2fsec
                  ''' % (epoch, self.max_epoch, i, len(data_loader), float(logs["errD"]), float(logs["errG"]), kl,
                         float(logs["errD_real"]), float(logs["errD_wrong"]), float(logs["errD_fake"]),
                         end_t - start_t))
            if epoch % self.snapshot_interval == 0 and rank0:
                save_model(netG, netD, engine.optG, engine.optD, epoch, self.model_dir)
        if rank0:
            save_model(netG, netD, engine.optG, engine.optD, epoch, self.model_dir)
        if writer is not None:
            writer.close()

    # ----------------------------------------------------------------------------- sample
    def sample(self, data, num_samples=25, stage=1, draw_bbox=True, **kw):
        """The tree's `sample` (S/trainer.py:287, C/trainer.py:198, M/trainer.py:208): `data` is the test split's path (coco,
        mnist) or a data loader over it (clevr); see ..sampling."""
        from . import sampling
        if self.tree == "coco":
            return sampling.sample_coco(self, data, num_samples, stage, draw_bbox, **kw)
        if self.tree == "clevr":
            return sampling.sample_clevr(self, data, num_samples, draw_bbox, **kw)
        return sampling.sample_mnist(self, data, num_samples, draw_bbox, **kw)
