from .ovis_evaluator_v2 import OVISEvaluator
