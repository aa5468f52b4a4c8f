"""Shared entry point of the three StackGAN-style trees (code/coco/stackgan/main.py, code/clevr/main.py,
code/multi-mnist/main.py): --cfg / --gpu / --data_dir / --manualSeed as in the reference, plus --synthetic N (train on N
synthetic items: there are no datasets on the GPU box), --max_epoch, --batch_size, --output_dir, --graph.  Each tree's
main.py binds its own cfg / trainer to `run`."""
import argparse
import datetime
import os
import pprint
import random

import torch

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
    if not cfg.TRAIN.FLAG:
        raise SystemExit("sampling is outside the train path (SURVEY.md section 8(f)); set TRAIN.FLAG")
    if not args.synthetic:
        raise SystemExit("the real-data TextDataset of this tree is outside the train hot path "
                         "(SURVEY.md section 8(f) rank 2); run with --synthetic N")
    os.makedirs(output_dir, exist_ok=True)
    dataset = SyntheticDataset(tree, stage, args.synthetic, seed=args.manualSeed,
                               text_dim=cfg.TEXT.DIMENSION if "TEXT" in cfg else 0)
    dataloader = torch.utils.data.DataLoader(dataset, batch_size=cfg.TRAIN.BATCH_SIZE, drop_last=True, shuffle=True,
                                             num_workers=0)
    algo = trainer_cls(output_dir, use_graph=args.graph)
    algo.train(dataloader, stage)
    return algo
