from .data_augment import TrainTransform, TrainTransformGPU, ValTransform, Vid_Val_Transform
from .datasets import vid
from .datasets.coco import COCO_CLASSES, COCODataset
from .datasets.vid import DataPrefetcher
from .synthetic import SyntheticClipLoader, get_timing_signal_1d
