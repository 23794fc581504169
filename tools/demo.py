#!/usr/bin/env python3
"""Single-image detector demo (CLI of the reference's tools/demo.py:27-330, image mode): YOLOX on one image -- BASELINE.json
config 1.  Image decode (Pillow) and the letterbox run on the host like the reference's cv2 path; backbone, head, box decode and
the class-agnostic NMS run in tscd_amd's HIP kernels, so the model itself needs the MI355X (``--device cpu`` of the reference is
served only as far as the plumbing goes: decode, letterbox, a clear error for the forward)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

from yolox.data.data_augment import ValTransform
from yolox.data.datasets.vid import imread_bgr
from yolox.data.datasets.vid_classes import OVIS_classes, VID_classes
from yolox.exp import get_exp
from yolox.utils import fuse_model, get_model_info, postprocess, vis

IMAGE_EXT = [".jpg", ".jpeg", ".webp", ".bmp", ".png"]


def make_parser():
    p = argparse.ArgumentParser("YOLOX Demo!")
    p.add_argument("demo", default="image", nargs="?", help="demo type: image")
    p.add_argument("-expn", "--experiment-name", type=str, default=None)
    p.add_argument("-n", "--name", type=str, default=None)
    p.add_argument("--path", default="./assets/dog.jpg", help="image file or directory")
    p.add_argument("--save_result", action="store_true")
    p.add_argument("-f", "--exp_file", default="exps/default/yolox_s.py", type=str)
    p.add_argument("-c", "--ckpt", default=None, type=str)
    p.add_argument("--device", default="gpu", type=str, help="cpu or gpu")
    p.add_argument("--conf", default=0.3, type=float)
    p.add_argument("--nms", default=0.3, type=float)
    p.add_argument("--tsize", default=None, type=int)
    p.add_argument("--fp16", action="store_true")
    p.add_argument("--fuse", action="store_true")
    return p


def get_image_list(path):
    names = []
    for maindir, _, files in os.walk(path):
        names += [os.path.join(maindir, f) for f in files if os.path.splitext(f)[1] in IMAGE_EXT]
    return names


class Predictor:
    def __init__(self, model, exp, cls_names=VID_classes, device="gpu", fp16=False, legacy=False):
        self.model, self.cls_names, self.num_classes = model, cls_names, exp.num_classes
        self.confthre, self.nmsthre, self.test_size = exp.test_conf, exp.nmsthre, exp.test_size
        self.device, self.fp16, self.preproc = device, fp16, ValTransform(legacy=legacy)

    def inference(self, img):
        info = {"id": 0, "file_name": None}
        if isinstance(img, str):
            info["file_name"] = os.path.basename(img)
            img = imread_bgr(img)
        info["height"], info["width"], info["raw_img"] = img.shape[0], img.shape[1], img
        info["ratio"] = min(self.test_size[0] / img.shape[0], self.test_size[1] / img.shape[1])
        x, _ = self.preproc(img, None, self.test_size)
        x = torch.from_numpy(x).unsqueeze(0).float()
        if self.device != "gpu":
            raise RuntimeError("the detector's kernels run on an MI355X only: use --device gpu (the reference's CPU path has no "
                               "counterpart here; decode and letterbox above ran on the host)")
        x = x.cuda()
        if self.fp16:
            x = x.half()
        with torch.no_grad():
            t0 = time.time()
            outputs = postprocess(self.model(x), self.num_classes, self.confthre, self.nmsthre, class_agnostic=True)
            torch.cuda.synchronize()
            print("Infer time: {:.4f}s".format(time.time() - t0))
        return outputs, info

    def visual(self, output, info, cls_conf=0.35):
        img = info["raw_img"]
        if output is None:
            return img
        output = output.float().cpu()
        return vis(img, (output[:, 0:4] / info["ratio"]).numpy(), (output[:, 4] * output[:, 5]).numpy(), output[:, 6].numpy(), cls_conf,
                   self.cls_names)


def image_demo(predictor, vis_folder, path, save_result):
    files = sorted(get_image_list(path)) if os.path.isdir(path) else [path]
    for name in files:
        outputs, info = predictor.inference(name)
        n = 0 if outputs[0] is None else int(outputs[0].shape[0])
        print(f"{name}: {n} detections")
        if save_result:
            from PIL import Image
            os.makedirs(vis_folder, exist_ok=True)
            res = predictor.visual(outputs[0], info, predictor.confthre)
            Image.fromarray(np.ascontiguousarray(res[:, :, ::-1])).save(os.path.join(vis_folder, os.path.basename(name)))


def main(exp, args):
    if args.conf is not None:
        exp.test_conf = args.conf
    if args.nms is not None:
        exp.nmsthre = args.nms
    if args.tsize is not None:
        exp.test_size = (args.tsize, args.tsize)
    model = exp.get_model()
    if args.device == "gpu":
        model.cuda()
        print("Model Summary: {}".format(get_model_info(model, exp.test_size)))
        if args.fp16:
            model.half()
    model.eval()
    if args.ckpt:
        ckpt = torch.load(args.ckpt, map_location="cpu", weights_only=True)
        model.load_state_dict(ckpt["model"] if "model" in ckpt else ckpt)
    if args.fuse and args.device == "gpu":
        model = fuse_model(model)
    names = OVIS_classes if exp.num_classes == len(OVIS_classes) else VID_classes
    predictor = Predictor(model, exp, names, args.device, args.fp16)
    if args.demo != "image":
        raise NotImplementedError("video / webcam demos need a video decoder (OpenCV); this image has none")
    image_demo(predictor, os.path.join(getattr(exp, "output_dir", "./YOLOX_outputs"), exp.exp_name, "vis_res"), args.path, args.save_result)


if __name__ == "__main__":
    a = make_parser().parse_args()
    main(get_exp(a.exp_file, a.name), a)
