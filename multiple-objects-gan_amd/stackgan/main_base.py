"""Shared entry point of the three StackGAN-style trees (code/coco/stackgan/main.py, code/clevr/main.py,
code/multi-mnist/main.py): --cfg / --gpu / --data_dir / --manualSeed as in the reference -- real data through ..datasets' TextDatasets when
TRAIN.FLAG, `sample` from the checkpoint cfg.NET_G otherwise -- plus --synthetic N (train on N synthetic items: there are no
datasets on the GPU box), --max_epoch, --batch_size, --output_dir, --graph.  Each tree's
main.py binds its own cfg / trainer to `run`."""
import argparse
import datetime
import os
import pprint
import random

import torch

from . import datasets
from .datasets_synth import SyntheticDataset


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description='Train a GAN network')
    parser.add_argument('--cfg', dest='cfg_file', help='optional config file', default=None, type=str)
    parser.add_argument('--gpu', dest='gpu_id', type=str, default='0')
    parser.add_argument('--data_dir', dest='data_dir', type=str, default='')
    parser.add_argument('--manualSeed', type=int, help='manual seed')
    parser.add_argument('--synthetic', type=int, default=0)
    parser.add_argument('--max_epoch', type=int, default=None)
    parser.add_argument('--batch_size', type=int, default=None)
    parser.add_argument('--output_dir', type=str, default=None)
    parser.add_argument('--graph', action='store_true')
    return parser.parse_args(argv)


def run(tree, cfg, cfg_from_file, trainer_cls, argv=None):
    args = parse_args(argv)
    if args.cfg_file is not None:
        cfg_from_file(args.cfg_file)
    if args.gpu_id != -1:
        cfg.GPU_ID = args.gpu_id
    if args.data_dir != '':
        cfg.DATA_DIR = args.data_dir
    if args.max_epoch is not None:
        cfg.TRAIN.MAX_EPOCH = args.max_epoch
    if args.batch_size is not None:
        cfg.TRAIN.BATCH_SIZE = args.batch_size
    print('Using config:')
    pprint.pprint(cfg)
    if args.manualSeed is None:
        args.manualSeed = random.randint(1, 10000)
    random.seed(args.manualSeed)
    torch.manual_seed(args.manualSeed)
    torch.cuda.manual_seed_all(args.manualSeed)
    timestamp = datetime.datetime.now().strftime('%Y_%m_%d_%H_%M_%S')
    output_dir = args.output_dir or '../../../output/%s_%s_%s' % (cfg.DATASET_NAME, cfg.CONFIG_NAME, timestamp)
    stage = int(cfg.get("STAGE", 1))
    if not cfg.TRAIN.FLAG:                                             # sampling from a checkpoint (cfg.NET_G)
        algo = trainer_cls(output_dir, use_graph=False)
        if tree == "coco":                                             # S/main.py:98-100
            algo.sample('%s/test/' % cfg.DATA_DIR, num_samples=25, stage=stage, draw_bbox=True)
        elif tree == "clevr":                                          # C/main.py:91-101
            ds = datasets.ClevrTextDataset(cfg.DATA_DIR, split="test", imsize=64, transform=datasets.image_transform())
            dl = torch.utils.data.DataLoader(ds, batch_size=1, drop_last=True, shuffle=True, num_workers=int(cfg.WORKERS))
            algo.sample(dl, num_samples=25, draw_bbox=True)
        else:                                                          # M/main.py:91-93
            algo.sample(os.path.join(cfg.DATA_DIR, "test"), num_samples=25, draw_bbox=True)
        return algo
    os.makedirs(output_dir, exist_ok=True)
    if args.synthetic:
        dataset = SyntheticDataset(tree, stage, args.synthetic, seed=args.manualSeed,
                                   text_dim=cfg.TEXT.DIMENSION if "TEXT" in cfg else 0)
    elif tree == "coco":                                               # S/main.py:77-90
        resize, imsize = (76, 64) if stage == 1 else (268, 256)
        dataset = datasets.CocoTextDataset(cfg.DATA_DIR, cfg.IMG_DIR, split="train", imsize=imsize,
                                           transform=datasets.image_transform(resize), crop=True, stage=stage)
    elif tree == "clevr":                                              # C/main.py:80-85
        dataset = datasets.ClevrTextDataset(cfg.DATA_DIR, split="train", imsize=64, transform=datasets.image_transform())
    else:                                                              # M/main.py:77-83
        dataset = datasets.MnistTextDataset(cfg.DATA_DIR, split="train", imsize=64, transform=datasets.image_transform(),
                                            crop=True)
    assert len(dataset) > 0
    workers = 0 if args.synthetic else int(cfg.get("WORKERS", 0))
    dataloader = torch.utils.data.DataLoader(dataset, batch_size=cfg.TRAIN.BATCH_SIZE, drop_last=True, shuffle=True,
                                             num_workers=workers)
    algo = trainer_cls(output_dir, use_graph=args.graph)
    algo.train(dataloader, stage)
    return algo
