"""BaseExp plugin ABC (contract of the reference's yolox/exp/base_exp.py:17-75): the abstract factory methods, ``merge``
with type coercion of CLI key/value overrides, and a tabular ``__repr__``."""
import ast
import pprint
from abc import ABCMeta, abstractmethod

from tabulate import tabulate


class BaseExp(metaclass=ABCMeta):
    def __init__(self):
        self.seed = None
        self.output_dir = "./YOLOX_outputs"
        self.print_interval = 100
        self.eval_interval = 10

    @abstractmethod
    def get_model(self):
        ...

    @abstractmethod
    def get_data_loader(self, batch_size, is_distributed):
        ...

    @abstractmethod
    def get_optimizer(self, batch_size):
        ...

    @abstractmethod
    def get_lr_scheduler(self, lr, iters_per_epoch, **kwargs):
        ...

    @abstractmethod
    def get_evaluator(self):
        ...

    @abstractmethod
    def eval(self, model, evaluator, weights):
        ...

    def __repr__(self):
        rows = [(str(k), pprint.pformat(v)) for k, v in vars(self).items() if not k.startswith("_")]
        return tabulate(rows, headers=["keys", "values"], tablefmt="fancy_grid")

    def merge(self, cfg_list):
        assert len(cfg_list) % 2 == 0
        for k, v in zip(cfg_list[0::2], cfg_list[1::2]):
            if not hasattr(self, k):
                continue
            cur = getattr(self, k)
            if cur is not None and type(cur) != type(v):
                try:
                    v = type(cur)(v)
                except Exception:
                    v = ast.literal_eval(v)
            setattr(self, k, v)
