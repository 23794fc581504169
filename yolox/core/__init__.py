from .launch import launch
from .tscd_trainer import Trainer
