"""COCO AP evaluation of the base detector (reference contract: yolox/evaluators/coco_evaluator.py:82-330 --
``COCOEvaluator(dataloader, img_size, confthre, nmsthre, num_classes, testdev, per_class_AP, per_class_AR)``,
``evaluate(model, distributed, half, ...) -> (AP50:95, AP50, summary)``, ``convert_to_coco_format``, ``evaluate_prediction``).

The reference hands the detections to pycocotools' COCOeval (its own compiled fast_cocoeval when present); here the ground truth comes
from the data set's own index (yolox/data/datasets/coco.py::CocoIndex) and the evaluation is csrc_host/cocoeval.cpp
(tscd_amd._host.cocoeval_bbox: COCOeval's accumulate / summarize for bounding boxes, pinned against the reference's compiled
evaluator in tests/test_cocoeval_ref_pin.py)."""
import itertools
import time

import numpy as np
import torch

from yolox.utils import is_main_process, postprocess, synchronize, time_synchronized, xyxy2xywh

_NAMES = ["AP   @[ IoU=0.50:0.95 | area=   all | maxDets=100 ]", "AP   @[ IoU=0.50      | area=   all | maxDets=100 ]",
          "AP   @[ IoU=0.75      | area=   all | maxDets=100 ]", "AP   @[ IoU=0.50:0.95 | area= small | maxDets=100 ]",
          "AP   @[ IoU=0.50:0.95 | area=medium | maxDets=100 ]", "AP   @[ IoU=0.50:0.95 | area= large | maxDets=100 ]",
          "AR   @[ IoU=0.50:0.95 | area=   all | maxDets=  1 ]", "AR   @[ IoU=0.50:0.95 | area=   all | maxDets= 10 ]",
          "AR   @[ IoU=0.50:0.95 | area=   all | maxDets=100 ]", "AR   @[ IoU=0.50:0.95 | area= small | maxDets=100 ]",
          "AR   @[ IoU=0.50:0.95 | area=medium | maxDets=100 ]", "AR   @[ IoU=0.50:0.95 | area= large | maxDets=100 ]"]


def _table(per_class, header, columns=6):
    """class / value pairs, ``columns`` cells per row, as the reference's tabulate(..., tablefmt="pipe") tables (:27-80)."""
    try:
        from tabulate import tabulate
    except ImportError:                                     # plain text when tabulate is absent
        return "\n".join("{:<16s} {:7.3f}".format(k, v) for k, v in per_class.items())
    ncol = min(columns, len(per_class) * 2)
    flat = [x for pair in per_class.items() for x in pair]
    rows = itertools.zip_longest(*[flat[i::ncol] for i in range(ncol)])
    return tabulate(rows, tablefmt="pipe", floatfmt=".3f", headers=["class", header] * (ncol // 2), numalign="left")


def per_class_AP_table(precision, class_names, header="AP", iou_index=None):
    """precision [T=10, R=101, K, A=4, M=3] -> per-class mean over recall (and IoU thresholds unless ``iou_index``), area all, 100 dets."""
    out = {}
    for k, name in enumerate(class_names):
        p = precision[:, :, k, 0, -1] if iou_index is None else precision[iou_index, :, k, 0, -1]
        p = p[p > -1]
        out[name] = float(p.mean() * 100) if p.size else float("nan")
    return _table(out, header)


def per_class_AR_table(recall, class_names):
    out = {}
    for k, name in enumerate(class_names):
        r = recall[:, k, 0, -1]
        r = r[r > -1]
        out[name] = float(r.mean() * 100) if r.size else float("nan")
    return _table(out, "AR")


class COCOEvaluator:
    def __init__(self, dataloader, img_size, confthre, nmsthre, num_classes, testdev=False, per_class_AP=False, per_class_AR=False):
        self.dataloader, self.img_size = dataloader, (img_size, img_size) if isinstance(img_size, int) else tuple(img_size)
        self.confthre, self.nmsthre, self.num_classes, self.testdev = confthre, nmsthre, num_classes, testdev
        self.per_class_AP, self.per_class_AR = per_class_AP, per_class_AR

    def evaluate(self, model, distributed=False, half=False, trt_file=None, decoder=None, test_size=None):
        """Runs the model over the evaluation loader (batches of (imgs, _, (heights, widths), ids)), NMS per image (yolox.utils.
        postprocess on the HIP kernels), and evaluates on the main process.  Switches the model to eval()."""
        if trt_file is not None:
            raise NotImplementedError("TensorRT engines are not part of this build")
        model = model.eval()
        if half:
            model = model.half()
        dev = next(model.parameters()).device
        data_list = []
        inference_time = nms_time = 0.0
        n_batches = len(self.dataloader)
        n_samples = max(n_batches - 1, 1)
        for cur_iter, (imgs, _, info_imgs, ids) in enumerate(self.dataloader):
            with torch.no_grad():
                imgs = torch.as_tensor(imgs).to(dev).to(torch.float16 if half else torch.float32)
                timed = cur_iter < n_batches - 1
                start = time.time()
                outputs = model(imgs)
                if decoder is not None:
                    outputs = decoder(outputs, dtype=outputs.type())
                infer_end = time_synchronized() if timed else start
                outputs = postprocess(outputs, self.num_classes, self.confthre, self.nmsthre)
                if timed:
                    inference_time += infer_end - start
                    nms_time += time_synchronized() - infer_end
            data_list.extend(self.convert_to_coco_format(outputs, info_imgs, ids))
        statistics = [inference_time, nms_time, n_samples]
        if distributed:
            import torch.distributed as dist
            gathered = [None] * dist.get_world_size()
            dist.all_gather_object(gathered, data_list)
            data_list = list(itertools.chain(*gathered))
            st = torch.tensor(statistics, dtype=torch.float64, device=dev)
            dist.reduce(st, dst=0)
            statistics = st.tolist()
        result = self.evaluate_prediction(data_list, statistics)
        synchronize()
        return result

    def convert_to_coco_format(self, outputs, info_imgs, ids):
        """Per kept detection {image_id, category_id (the data set's id of the class index), bbox xywh in ORIGINAL image pixels, score =
        objectness x class confidence} (:213-247)."""
        data_list = []
        class_ids = self.dataloader.dataset.class_ids
        for output, img_h, img_w, img_id in zip(outputs, info_imgs[0], info_imgs[1], ids):
            if output is None:
                continue
            output = output.float().cpu()
            scale = min(self.img_size[0] / float(img_h), self.img_size[1] / float(img_w))
            bboxes = xyxy2xywh(output[:, 0:4] / scale)
            scores = output[:, 4] * output[:, 5]
            for ind in range(bboxes.shape[0]):
                data_list.append({"image_id": int(img_id), "category_id": class_ids[int(output[ind, 6])], "bbox": bboxes[ind].numpy().tolist(),
                                  "score": float(scores[ind]), "segmentation": []})
        return data_list

    def evaluate_prediction(self, data_list, statistics):
        if not is_main_process():
            return 0, 0, None
        from tscd_amd._host import cocoeval_bbox
        inference_time, nms_time, n_samples = (float(v) for v in statistics)
        bs = getattr(self.dataloader, "batch_size", None) or getattr(getattr(self.dataloader, "batch_sampler", None), "batch_size", 1) or 1
        a_infer, a_nms = 1000 * inference_time / (n_samples * bs), 1000 * nms_time / (n_samples * bs)
        info = ", ".join("Average {} time: {:.2f} ms".format(k, v) for k, v in zip(["forward", "NMS", "inference"], [a_infer, a_nms, a_infer + a_nms])) + "\n"
        if len(data_list) == 0:
            return 0, 0, info
        ds = self.dataloader.dataset
        index = ds.coco
        img_of = {i: n for n, i in enumerate(ds.ids)}
        cat_of = {c: n for n, c in enumerate(ds.class_ids)}
        gts = [a for i in ds.ids for a in index.anns(i)]
        g = lambda k, default=None: [a.get(k, default) for a in gts]
        stats, prec, rec = cocoeval_bbox(
            np.array(g("bbox"), np.float32).reshape(-1, 4), [img_of[i] for i in g("image_id")], [cat_of[c] for c in g("category_id")],
            [int(bool(v)) for v in g("iscrowd", 0)], g("area"), np.array([d["bbox"] for d in data_list], np.float32).reshape(-1, 4),
            [img_of[d["image_id"]] for d in data_list], [cat_of[d["category_id"]] for d in data_list], [d["score"] for d in data_list],
            len(img_of), len(cat_of), full=True)
        for n, v in zip(_NAMES, stats.tolist()):
            info += " Average {} {} = {:0.3f}\n".format("Precision  (AP)" if n.startswith("AP") else "Recall     (AR)", n[2:].strip(), v)
        names = list(getattr(ds, "_classes", [str(c) for c in ds.class_ids]))
        if self.per_class_AP:
            info += "per class AP:\n" + per_class_AP_table(prec, names) + "\n"
        if self.per_class_AR:
            info += "per class AR:\n" + per_class_AR_table(rec, names) + "\n"
        return float(stats[0]), float(stats[1]), info
