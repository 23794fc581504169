from .launch import launch
from .trainer import BaseTrainer
from .tscd_trainer import Trainer
