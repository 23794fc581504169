from . import vid
from .coco import COCO_CLASSES, COCODataset, CocoIndex
from .vid_classes import OVIS_classes, VID_classes
from ..mosaic import MosaicDetection
