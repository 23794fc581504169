"""OVIS evaluation (interface of the reference's yolox/evaluators/ovis_evaluator_v2.py: ``OVISEvaluator`` :35-424).

``evaluate(model, distributed, half)`` runs the clips of the val loader through the model in eval mode (fp16 with
``half=True``, as the reference: model.half()), ``resume`` = "not the first clip of its video" so that the CAFM matcher
carries its state from clip to clip, converts the refined detections of the local frames to COCO records (boxes scaled back
by 1 / letterbox ratio, xywh) and scores them with the COCO bbox protocol.  Scoring runs in tscd_amd's C++ implementation
(csrc_host/cocoeval.cpp) instead of pycocotools / the reference's pybind11 extension, neither of which exists in this image.
Returns (AP@[.5:.95], AP50, summary text) like the reference.

Distributed: clips are sharded by VIDEO (yolox.data.datasets.vid.TestSampler) so that no video's state chain is cut; every
rank evaluates its videos and the detection / ground-truth records are gathered to rank 0 (``all_gather_object``)."""
import contextlib
import io
import time

import numpy as np
import torch

from yolox.data.datasets.vid_classes import OVIS_classes
from yolox.utils import is_main_process, synchronize, time_synchronized


def _xyxy2xywh(b):
    b = b.clone()
    b[:, 2] -= b[:, 0]
    b[:, 3] -= b[:, 1]
    return b


class OVISEvaluator:
    def __init__(self, dataloader, img_size, confthre, nmsthre, num_classes, testdev=False, gl_mode=False, lframe=0, gframe=32, **kwargs):
        self.dataloader, self.img_size, self.confthre, self.nmsthre, self.num_classes = dataloader, img_size, confthre, nmsthre, num_classes
        self.testdev, self.gl_mode, self.lframe, self.gframe, self.kwargs = testdev, gl_mode, lframe, gframe, kwargs
        self.class_names = list(OVIS_classes)[:num_classes] if num_classes <= len(OVIS_classes) else [str(i) for i in range(num_classes)]
        self.id = 0
        self.box_id = 0

    # ------------------------------------------------------------------------------------------------------------
    def evaluate(self, model, distributed=False, half=True, trt_file=None, decoder=None, test_size=None, img_path=None):
        model = model.eval()
        if half:
            model = model.half()
        dtype = torch.float16 if half else torch.float32
        data_list, labels_list, images = [], [], []
        inference_time, n_samples = 0.0, max(len(self.dataloader) - 1, 1)
        for cur_iter, (imgs, _, info_imgs, label, path, time_embedding) in enumerate(self.dataloader):
            with torch.no_grad():
                imgs = imgs.cuda(non_blocking=True).to(dtype)
                time_embedding = time_embedding.cuda(non_blocking=True).to(dtype)
                # first clip of a video (frame index 0 in its first file name) starts a new matcher state
                first = self.lframe == 0 or int(path[0].split("/")[-1].split(".")[0].split("_")[-1]) == 0
                timed = cur_iter < len(self.dataloader) - 1
                t0 = time.time()
                outputs, ori_res = model(imgs, time_embedding=time_embedding, lframe=self.lframe, gframe=self.gframe, resume=not first)
                if timed:
                    inference_time += time_synchronized() - t0
            if self.gl_mode:
                n_loc = int(imgs.shape[0] / 2)
                info_imgs, label = info_imgs[:n_loc], label[:n_loc]
            if self.kwargs.get("first_only", False):
                info_imgs, label = [info_imgs[0]], [label[0]]
            d, l, im = self.convert_to_coco_format(outputs, info_imgs, [x.clone() for x in label])
            data_list.extend(d)
            labels_list.extend(l)
            images.extend(im)
        stats = [inference_time, 0.0, n_samples]
        if distributed:
            import torch.distributed as dist
            world = dist.get_world_size()
            gathered = [None] * world
            dist.all_gather_object(gathered, (data_list, labels_list, images, stats))
            # image ids are rank-local: make them unique (rank-major) before merging
            data_list, labels_list, images, stats = [], [], [], [0.0, 0.0, 0]
            base = 0
            for d, l, im, st in gathered:
                data_list += [dict(x, image_id=x["image_id"] + base) for x in d]
                labels_list += [dict(x, image_id=x["image_id"] + base) for x in l]
                images += [i + base for i in im]
                base += (max(im) + 1) if im else 0
                stats = [stats[0] + st[0], stats[1] + st[1], stats[2] + st[2]]
        result = self.evaluate_prediction(data_list, labels_list, images, stats)
        self.id, self.box_id = 0, 0
        synchronize()
        return result

    def convert_to_coco_format(self, outputs, info_imgs, labels):
        """outputs: list over the local frames of [k,7] (x1,y1,x2,y2,obj,cls_score,cls_id) or None; labels: list of [n,5]
        (cls,x1,y1,x2,y2) in letterboxed coordinates -> COCO detection / annotation records in original-image pixels."""
        data_list, label_list, images = [], [], []
        for output, info_img, lab in zip(outputs, info_imgs, labels):
            scale = min(self.img_size[0] / float(info_img[0]), self.img_size[1] / float(info_img[1]))
            lab = lab.float()
            boxes_l = _xyxy2xywh(lab[:, 1:] / scale)
            for ind in range(boxes_l.shape[0]):
                label_list.append({"image_id": int(self.id), "category_id": int(lab[ind, 0]), "bbox": boxes_l[ind].tolist(), "segmentation": [],
                                   "id": self.box_id, "iscrowd": 0, "area": int(boxes_l[ind][2] * boxes_l[ind][3])})
                self.box_id += 1
            images.append(int(self.id))
            if output is not None:
                output = output.float().cpu()
                boxes = _xyxy2xywh(output[:, 0:4] / scale)
                scores = output[:, 4] * output[:, 5]
                for ind in range(boxes.shape[0]):
                    data_list.append({"image_id": int(self.id), "category_id": int(output[ind, 6]), "bbox": boxes[ind].tolist(),
                                      "score": float(scores[ind]), "segmentation": []})
            self.id += 1
        return data_list, label_list, images

    def evaluate_prediction(self, data_list, labels_list, images, statistics):
        if not is_main_process():
            return 0, 0, None
        from tscd_amd._host import cocoeval_bbox
        inference_time, nms_time, n_samples = statistics
        bs = getattr(getattr(self.dataloader, "batch_sampler", None), "batch_size", 1) or 1
        a_infer = 1000 * inference_time / max(n_samples * bs, 1)
        info = "Average forward time: {:.2f} ms, Average NMS time: {:.2f} ms, Average inference time: {:.2f} ms\n".format(a_infer, 0.0, a_infer)
        if not data_list:
            return 0, 0, info
        img_index = {i: n for n, i in enumerate(sorted(set(images)))}
        K = self.num_classes
        g = lambda k: [x[k] for x in labels_list]
        d = lambda k: [x[k] for x in data_list]
        stats, prec, rec = cocoeval_bbox(np.array(g("bbox"), np.float32).reshape(-1, 4), [img_index[i] for i in g("image_id")], g("category_id"),
                                         g("iscrowd"), g("area"), np.array(d("bbox"), np.float32).reshape(-1, 4),
                                         [img_index[i] for i in d("image_id")], d("category_id"), d("score"), len(img_index), K, full=True)
        with contextlib.redirect_stdout(io.StringIO()) as buf:
            names = ["AP   @[ IoU=0.50:0.95 | area=   all | maxDets=100 ]", "AP   @[ IoU=0.50      | area=   all | maxDets=100 ]",
                     "AP   @[ IoU=0.75      | area=   all | maxDets=100 ]", "AP   @[ IoU=0.50:0.95 | area= small | maxDets=100 ]",
                     "AP   @[ IoU=0.50:0.95 | area=medium | maxDets=100 ]", "AP   @[ IoU=0.50:0.95 | area= large | maxDets=100 ]",
                     "AR   @[ IoU=0.50:0.95 | area=   all | maxDets=  1 ]", "AR   @[ IoU=0.50:0.95 | area=   all | maxDets= 10 ]",
                     "AR   @[ IoU=0.50:0.95 | area=   all | maxDets=100 ]", "AR   @[ IoU=0.50:0.95 | area= small | maxDets=100 ]",
                     "AR   @[ IoU=0.50:0.95 | area=medium | maxDets=100 ]", "AR   @[ IoU=0.50:0.95 | area= large | maxDets=100 ]"]
            for n, v in zip(names, stats.tolist()):
                print(" Average {} = {:0.3f}".format("Precision (" + n[:2] + ")" if n.startswith("AP") else "Recall    (" + n[:2] + ")", v).replace("(AP)", " (AP)").replace("(AR)", " (AR)") + "  " + n[2:].strip())
        per_class = []
        for k in range(K):
            p = prec[:, :, k, 0, 2]
            p50 = prec[0, :, k, 0, 2]
            ap = float(p[p > -1].mean()) if (p > -1).any() else float("nan")
            ap50 = float(p50[p50 > -1].mean()) if (p50 > -1).any() else float("nan")
            per_class.append("{:<14s} AP {:6.3f}  AP50 {:6.3f}".format(self.class_names[k], ap * 100, ap50 * 100))
        info += "per class AP / AP50:\n" + "\n".join(per_class) + "\n" + buf.getvalue()
        return float(stats[0]), float(stats[1]), info
