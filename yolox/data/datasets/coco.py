"""COCO-format detection data set for the base detector's training and evaluation (reference contract:
yolox/data/datasets/coco.py:36-234 -- constructor arguments, ``ids`` / ``class_ids`` / ``_classes`` / ``annotations``, ``load_anno``,
``load_image``, ``load_resized_img``, ``pull_item`` -> (img, labels, (h, w), [image id]), ``__getitem__`` with the pre-processing
hook).

No pycocotools and no cv2 in this image: the annotation index is a small one of our own over the JSON (``CocoIndex``: images,
categories, per-image annotation lists, the ``iscrowd=False`` filter the reference asks the COCO API for), images are decoded with PIL
into the BGR uint8 arrays cv2.imread returns, and the resize is yolox.data.data_augment.resize_bilinear (cv2.INTER_LINEAR's
arithmetic).  ``cache=True`` keeps the resized images in RAM (the reference memory-maps them to a file next to the data: same effect
for the loader, no 136 GB file here)."""
import json
import os

import numpy as np

from ..data_augment import resize_bilinear


class CocoIndex:
    """What the data set and the evaluator need of pycocotools.COCO: the JSON's images / categories and the annotations per image."""

    def __init__(self, json_path):
        with open(json_path) as fh:
            ds = json.load(fh)
        self.images = {im["id"]: im for im in ds.get("images", [])}
        self.cats = {c["id"]: c for c in ds.get("categories", [])}
        self.anns_of = {i: [] for i in self.images}
        for a in ds.get("annotations", []):
            a.pop("segmentation", None)                     # the reference drops them too (remove_useless_info): ~30 % of the memory
            self.anns_of.setdefault(a["image_id"], []).append(a)

    def img_ids(self):
        return list(self.images)                            # file order, as COCO.getImgIds() (dict order of the index)

    def cat_ids(self):
        return sorted(self.cats)

    def anns(self, img_id, iscrowd=None):
        out = self.anns_of.get(img_id, [])
        if iscrowd is None:
            return list(out)
        return [a for a in out if bool(a.get("iscrowd", 0)) == bool(iscrowd)]


def imread_bgr(path):
    """cv2.imread's result: HxWx3 uint8, channels B, G, R (None when the file cannot be read)."""
    from PIL import Image
    try:
        with Image.open(path) as im:
            rgb = np.asarray(im.convert("RGB"))
    except (OSError, ValueError):
        return None
    return np.ascontiguousarray(rgb[:, :, ::-1])


class COCODataset:
    def __init__(self, data_dir=None, json_file="instances_train2017.json", name="train2017", img_size=(416, 416), preproc=None, cache=False):
        if data_dir is None:
            data_dir = os.path.join(os.environ.get("YOLOX_DATADIR", os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(
                os.path.dirname(os.path.abspath(__file__))))), "datasets")), "COCO")
        self.data_dir, self.json_file, self.name = data_dir, json_file, name
        self.coco = CocoIndex(os.path.join(data_dir, "annotations", json_file))
        self.ids = self.coco.img_ids()
        self.class_ids = self.coco.cat_ids()
        self._classes = tuple(self.coco.cats[c]["name"] for c in self.class_ids)
        self._cls_of = {c: i for i, c in enumerate(self.class_ids)}
        self.img_size = self.input_dim = tuple(img_size) if not isinstance(img_size, int) else (img_size, img_size)
        self.preproc = preproc
        self.imgs = None
        self.annotations = [self.load_anno_from_ids(i) for i in self.ids]
        if cache:
            self.imgs = [self.load_resized_img(k) for k in range(len(self.ids))]

    def __len__(self):
        return len(self.ids)

    def load_anno_from_ids(self, id_):
        """(labels [n,5] = clipped x1 y1 x2 y2 (scaled to the input size) + class index, (h, w), resized (h, w), file name) of one image:
        non-crowd annotations with positive area whose clipped box is not inverted (coco.py:137-172)."""
        im = self.coco.images[id_]
        width, height = im["width"], im["height"]
        rows = []
        for obj in self.coco.anns(id_, iscrowd=False):
            x1, y1 = max(0, obj["bbox"][0]), max(0, obj["bbox"][1])
            x2, y2 = min(width, x1 + max(0, obj["bbox"][2])), min(height, y1 + max(0, obj["bbox"][3]))
            if obj["area"] > 0 and x2 >= x1 and y2 >= y1:
                rows.append((x1, y1, x2, y2, self._cls_of[obj["category_id"]]))
        res = np.array(rows, dtype=np.float64).reshape(-1, 5)
        r = min(self.img_size[0] / height, self.img_size[1] / width)
        res[:, :4] *= r
        file_name = im["file_name"] if "file_name" in im else "{:012}".format(id_) + ".jpg"
        return res, (height, width), (int(height * r), int(width * r)), file_name

    def load_anno(self, index):
        return self.annotations[index][0]

    def load_image(self, index):
        img = imread_bgr(os.path.join(self.data_dir, self.name, self.annotations[index][3]))
        assert img is not None, self.annotations[index][3]
        return img

    def load_resized_img(self, index):
        img = self.load_image(index)
        r = min(self.img_size[0] / img.shape[0], self.img_size[1] / img.shape[1])
        return resize_bilinear(img, int(img.shape[0] * r), int(img.shape[1] * r)).astype(np.uint8)

    def pull_item(self, index):
        res, img_info, resized_info, _ = self.annotations[index]
        if self.imgs is not None:
            img = self.imgs[index][:resized_info[0], :resized_info[1]].copy()
        else:
            img = self.load_resized_img(index)
        return img, res.copy(), img_info, np.array([self.ids[index]])

    def __getitem__(self, index):
        img, target, img_info, img_id = self.pull_item(index)
        if self.preproc is not None:
            img, target = self.preproc(img, target, self.input_dim)
        return img, target, img_info, img_id


COCO_CLASSES = (
    "person", "bicycle", "car", "motorcycle", "airplane", "bus", "train", "truck", "boat", "traffic light", "fire hydrant", "stop sign",
    "parking meter", "bench", "bird", "cat", "dog", "horse", "sheep", "cow", "elephant", "bear", "zebra", "giraffe", "backpack", "umbrella",
    "handbag", "tie", "suitcase", "frisbee", "skis", "snowboard", "sports ball", "kite", "baseball bat", "baseball glove", "skateboard",
    "surfboard", "tennis racket", "bottle", "wine glass", "cup", "fork", "knife", "spoon", "bowl", "banana", "apple", "sandwich", "orange",
    "broccoli", "carrot", "hot dog", "pizza", "donut", "cake", "chair", "couch", "potted plant", "bed", "dining table", "toilet", "tv",
    "laptop", "mouse", "remote", "keyboard", "cell phone", "microwave", "oven", "toaster", "sink", "refrigerator", "book", "clock", "vase",
    "scissors", "teddy bear", "hair drier", "toothbrush",
)
