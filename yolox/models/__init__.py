from .darknet import CSPDarknet, Darknet
from .losses import IOUloss
from .yolo_pafpn import YOLOPAFPN
from .tscd_head import TSCDHead
from .tscd import TSCD
