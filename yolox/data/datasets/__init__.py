from . import vid
from .vid_classes import OVIS_classes, VID_classes
