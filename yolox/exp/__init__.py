from .base_exp import BaseExp
from .build import get_exp
