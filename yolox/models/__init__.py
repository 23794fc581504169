from .darknet import CSPDarknet, Darknet
from .losses import IOUloss
from .yolo_pafpn import YOLOPAFPN
from .yolo_head import YOLOXHead
from .yolox import YOLOX
from .tscd_head import TSCDHead
from .tscd import TSCD
