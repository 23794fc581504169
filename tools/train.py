#!/usr/bin/env python3
"""Training entry of the YOLOX base detector (CLI of the reference's tools/train.py; stage 1 of the recipe, README.md:86-89):
``python tools/train.py -f exps/default/yolox_s.py -d 8 -b 64 --fp16``.  One process per GPU over RCCL."""
import argparse
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from yolox.core import launch
from yolox.exp import get_exp
from yolox.utils import configure_nccl, configure_omp, get_num_devices


def make_parser():
    p = argparse.ArgumentParser("YOLOX train parser")
    p.add_argument("-expn", "--experiment-name", type=str, default=None)
    p.add_argument("-n", "--name", type=str, default=None)
    p.add_argument("--dist-backend", default="nccl", type=str)
    p.add_argument("--dist-url", default=None, type=str)
    p.add_argument("-b", "--batch-size", type=int, default=64, help="images per step over all devices")
    p.add_argument("-d", "--devices", default=None, type=int)
    p.add_argument("-f", "--exp_file", default="exps/default/yolox_s.py", type=str)
    p.add_argument("--resume", default=False, action="store_true")
    p.add_argument("-c", "--ckpt", default=None, type=str)
    p.add_argument("-e", "--start_epoch", default=None, type=int)
    p.add_argument("--num_machines", default=1, type=int)
    p.add_argument("--machine_rank", default=0, type=int)
    p.add_argument("--fp16", dest="fp16", default=False, action="store_true")
    p.add_argument("--cache", dest="cache", default=False, action="store_true")
    p.add_argument("opts", default=None, nargs=argparse.REMAINDER)
    return p


def main(exp, args):
    if exp.seed is not None:
        random.seed(exp.seed)
        torch.manual_seed(exp.seed)
    configure_nccl()
    configure_omp()
    exp.get_trainer(args).train()


if __name__ == "__main__":
    args = make_parser().parse_args()
    exp = get_exp(args.exp_file, args.name)
    exp.merge(args.opts)
    if not args.experiment_name:
        args.experiment_name = exp.exp_name
    n = get_num_devices() if args.devices is None else args.devices
    assert n <= get_num_devices()
    launch(main, n, args.num_machines, args.machine_rank, backend=args.dist_backend, dist_url=args.dist_url or "auto", args=(exp, args))
