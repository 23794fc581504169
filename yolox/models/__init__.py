from .darknet import CSPDarknet, Darknet
from .yolo_pafpn import YOLOPAFPN
