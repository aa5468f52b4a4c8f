"""Entry point of the coco-attngan variant (mirror of code/coco/attngan/main.py): same flags
(--cfg --gpu --resume --data_dir --manualSeed), plus --synthetic N to train on generated batches when the
COCO pickles are not present.  One process per GPU: launch with
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 main.py --cfg cfg/coco_train.yml
(`--gpu` is kept for compatibility; the device comes from LOCAL_RANK)."""
import argparse
import datetime
import os
import pprint
import random
import sys

import numpy as np
import torch
import torch.distributed as dist

if __package__ in (None, ""):
    sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(os.path.realpath(__file__)), "..", "..")))
    import mogan_loader
    mogan_loader.load()
    from mogan_amd.attngan.miscc.config import cfg, cfg_from_file
    from mogan_amd.attngan.datasets import SyntheticTextDataset, TextDataset
    from mogan_amd.attngan.trainer import condGANTrainer as trainer
    from mogan_amd.hip import lib as hiplib
else:
    from .miscc.config import cfg, cfg_from_file
    from .datasets import SyntheticTextDataset, TextDataset
    from .trainer import condGANTrainer as trainer
    from ..hip import lib as hiplib


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description='Train a AttnGAN network')
    parser.add_argument('--cfg', dest='cfg_file', help='optional config file', default='cfg/coco_train.yml', type=str)
    parser.add_argument('--gpu', dest='gpu_id', type=str, default='0')
    parser.add_argument('--resume', dest='resume', type=str, default='')
    parser.add_argument('--data_dir', dest='data_dir', type=str, default='')
    parser.add_argument('--manualSeed', type=int, help='manual seed')
    parser.add_argument('--synthetic', type=int, default=0, help='train on N generated samples instead of COCO')
    parser.add_argument('--max_epoch', type=int, default=None)
    parser.add_argument('--batch_size', type=int, default=None, help='per-GPU minibatch')
    parser.add_argument('--output_dir', type=str, default='')
    parser.add_argument('--device_feeder', action='store_true',
                        help='real data: workers only decode + resize the JPEGs; crop / flip / multi-scale / normalise run '
                             'on the GPU (attngan/feeder.py, bit-identical images, 5x less PCIe traffic)')
    parser.add_argument('--sampling', action='store_true',
                        help='evaluation (TRAIN.FLAG False): one image per caption of the whole split (sampling()) '
                             'instead of the 25 rows of sample()')
    return parser.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    if args.cfg_file is not None:
        cfg_from_file(args.cfg_file)
    cfg.GPU_ID = args.gpu_id
    if args.data_dir != '':
        cfg.DATA_DIR = args.data_dir
    if args.max_epoch is not None:
        cfg.TRAIN.MAX_EPOCH = args.max_epoch
    if args.batch_size is not None:
        cfg.TRAIN.BATCH_SIZE = args.batch_size
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if cfg.TRAIN.FLAG and not torch.cuda.is_initialized():
        # the eager multi-stream step: 4 hardware queues, and the engine's streams created / bound to them in their fixed order
        # before RCCL creates its own (hip/lib.py "hardware queues", trainer.create_engine_streams)
        hiplib.configure_hw_queues()
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        hiplib.reserve_hw_queues()
        from .trainer import create_engine_streams
        create_engine_streams()
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")       # RCCL's stream on its own priority queue (see bench.py)
        dist.init_process_group("nccl", rank=rank, world_size=world)
    if rank == 0:
        print('Using config:')
        pprint.pprint(cfg)
    if args.manualSeed is None:
        args.manualSeed = 100 if not cfg.TRAIN.FLAG else random.randint(1, 10000)
    stamp = datetime.datetime.now().strftime('%Y_%m_%d_%H_%M_%S')
    if world > 1:
        # ONE seed and ONE output directory for the job: rank 0 draws them (each rank's `random` is unseeded, so the
        # draws would differ), everybody else takes rank 0's
        shared = [args.manualSeed, stamp]
        dist.broadcast_object_list(shared, src=0)
        args.manualSeed, stamp = shared
    random.seed(args.manualSeed + rank)
    np.random.seed(args.manualSeed + rank)
    # common torch seed for the weight initialisation (the engine additionally broadcasts rank 0's weights, buffers and
    # optimizer state: TrainEngine.sync_replicas); condGANTrainer.train() re-seeds per rank afterwards for z / eps
    torch.manual_seed(args.manualSeed)
    if args.resume == '':
        output_dir = args.output_dir or '../../../output/%s_%s_%s' % (cfg.DATASET_NAME, cfg.CONFIG_NAME, stamp)
    else:
        output_dir = args.resume
    # main.py:117-121: evaluation reads the test split and the dataset hands the scaled boxes along
    split_dir, evaluate = ('train', False) if cfg.TRAIN.FLAG else ('test', True)
    with_bbox = evaluate and not args.sampling
    if args.synthetic > 0:
        dataset = SyntheticTextDataset(args.synthetic, seed=args.manualSeed, eval=with_bbox)   # one dataset, partitioned below
    else:
        dataset = TextDataset(cfg.DATA_DIR, cfg.IMG_DIR, split_dir, base_size=cfg.TREE.BASE_SIZE, eval=with_bbox,
                              raw=args.device_feeder and cfg.TRAIN.FLAG)
    assert dataset
    sampler = None
    if world > 1 and cfg.TRAIN.FLAG:
        # every rank trains on its own 1/world partition of each epoch (same permutation seed on all ranks, re-drawn per
        # epoch through sampler.set_epoch in condGANTrainer.train)
        sampler = torch.utils.data.distributed.DistributedSampler(dataset, num_replicas=world, rank=rank, shuffle=True,
                                                                  seed=int(args.manualSeed), drop_last=True)
    loader = torch.utils.data.DataLoader(dataset, batch_size=cfg.TRAIN.BATCH_SIZE, drop_last=True,
                                         shuffle=sampler is None, sampler=sampler,
                                         num_workers=min(int(cfg.WORKERS), 8 if args.synthetic else int(cfg.WORKERS)))
    algo = trainer(output_dir, loader, dataset.n_words, dataset.ixtoword, args.resume, distributed=world > 1)
    if cfg.TRAIN.FLAG:
        algo.train()
    elif args.sampling:
        algo.sampling(split_dir)                  # main.py:157 (commented alternative): the whole validation split
    elif cfg.B_VALIDATION:
        algo.sample(split_dir, num_samples=25, draw_bbox=True)                      # main.py:158
    else:
        # main.py:160 gen_example: not callable in the reference either (it passes num_samples to a two-argument
        # function and trainer.gen_example calls G_NET without boxes and labels) -- nothing to mirror
        raise SystemExit("gen_example (custom captions) is not supported: set B_VALIDATION: True")
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
