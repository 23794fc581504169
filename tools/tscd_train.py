#!/usr/bin/env python3
"""TSCD training entry (CLI of the reference's tools/tscd_train.py:27-145).  Unlike the reference it does not pin
CUDA_VISIBLE_DEVICES=0 / WORLD_SIZE=1: ``-d N`` trains data-parallel on N MI355X over RCCL."""
import argparse
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from yolox.core import Trainer, launch
from yolox.exp import get_exp
from yolox.utils import configure_nccl, configure_omp, get_num_devices


def make_parser():
    p = argparse.ArgumentParser("TSCD train parser")
    p.add_argument("-expn", "--experiment-name", type=str, default=None)
    p.add_argument("-n", "--name", type=str, default=None)
    p.add_argument("--dist-backend", default="nccl", type=str)
    p.add_argument("--dist-url", default=None, type=str)
    p.add_argument("-b", "--batch-size", type=int, default=16, help="frames per clip (lframe + gframe)")
    p.add_argument("-d", "--devices", default=1, type=int)
    p.add_argument("-f", "--exp_file", default="exps/TSCD_OVIS/ovis_tscd_large.py", type=str)
    p.add_argument("--resume", default=False, action="store_true")
    p.add_argument("-c", "--ckpt", default=None, type=str)
    p.add_argument("--num_machines", default=1, type=int)
    p.add_argument("--machine_rank", default=0, type=int)
    p.add_argument("--fp16", dest="fp16", default=True, action="store_true")
    p.add_argument("--max-iters", type=int, default=None, help="stop after this many iterations (smoke runs)")
    p.add_argument("opts", default=None, nargs=argparse.REMAINDER)
    return p


def main(exp, args):
    if exp.seed is not None:
        random.seed(exp.seed)
        torch.manual_seed(exp.seed)
    configure_nccl()
    configure_omp()
    trainer = Trainer(exp, args)
    if args.max_iters:
        trainer.before_train()
        trainer.epoch = trainer.start_epoch
        for trainer.iter in range(args.max_iters):
            out = trainer.train_one_iter()
            if trainer.rank == 0:
                print(f"iter {trainer.iter}: " + ", ".join(f"{k} {float(v.detach() if torch.is_tensor(v) else v):.3f}" for k, v in out.items()), flush=True)
        return
    trainer.train()


if __name__ == "__main__":
    args = make_parser().parse_args()
    exp = get_exp(args.exp_file, args.name)
    exp.merge(args.opts)
    if not args.experiment_name:
        args.experiment_name = exp.exp_name
    n = get_num_devices() if args.devices is None else args.devices
    assert n <= get_num_devices()
    launch(main, n, args.num_machines, args.machine_rank, backend=args.dist_backend, dist_url=args.dist_url or "auto", args=(exp, args))
