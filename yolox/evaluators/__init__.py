from .coco_evaluator import COCOEvaluator
from .ovis_evaluator_v2 import OVISEvaluator
