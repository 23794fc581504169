#!/usr/bin/env python3
"""TSCD evaluation entry (CLI of the reference's tools/tscd_eval.py:28-183).  Runs the eval forward over the clips of
the val loader in fp16 (the reference's OVISEvaluator calls model.half(), ovis_evaluator_v2.py:167-170), carrying the
CAFM state across the clips of a video with ``resume``.  With a dataset under ``exp.data_dir`` the exp's OVISEvaluator scores
the detections (COCO bbox AP via tscd_amd's C++ COCOeval); without one the synthetic clip source is run and detections per
clip and throughput are reported."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from yolox.exp import get_exp


def make_parser():
    p = argparse.ArgumentParser("TSCD eval parser")
    p.add_argument("-f", "--exp_file", default="exps/TSCD_OVIS/ovis_tscd_large.py", type=str)
    p.add_argument("-n", "--name", type=str, default=None)
    p.add_argument("-c", "--ckpt", default=None, type=str)
    p.add_argument("--clips", type=int, default=4)
    p.add_argument("--nms", default=0.5, type=float)
    p.add_argument("opts", default=None, nargs=argparse.REMAINDER)
    return p


@torch.no_grad()
def main():
    args = make_parser().parse_args()
    exp = get_exp(args.exp_file, args.name)
    exp.merge(args.opts)
    model = exp.get_model().cuda().eval()
    if args.ckpt:
        model.load_state_dict(torch.load(args.ckpt, map_location="cuda", weights_only=True)["model"])
    loader = exp.get_eval_loader(exp.lframe_val + exp.gframe_val)
    if not hasattr(loader, "next"):                      # a real val loader: score with the exp's evaluator (N2)
        ap50_95, ap50, summary = exp.eval(model, exp.get_evaluator(loader), False, half=True)
        print(summary)
        print(f"AP50:95 = {ap50_95:.4f}  AP50 = {ap50:.4f}")
        return
    model.half()
    t0 = time.time()
    for i in range(args.clips):
        imgs, _, temb = loader.next()
        res, res_ori = model(imgs.half(), time_embedding=temb.half(), nms_thresh=args.nms, lframe=exp.lframe_val, gframe=exp.gframe_val,
                             resume=i > 0)
        torch.cuda.synchronize()
        print(f"clip {i}: detections per local frame {[0 if r is None else int(r.shape[0]) for r in res]}", flush=True)
    print(f"{args.clips / (time.time() - t0):.2f} clips/s (eval, {exp.lframe_val}+{exp.gframe_val} frames @{exp.test_size})")


if __name__ == "__main__":
    main()
