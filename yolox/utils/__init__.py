from .boxes import adjust_box_anns, bboxes_iou, filter_box, postprocess, xyxy2cxcywh, xyxy2xywh
from .checkpoint import load_ckpt, save_checkpoint
from .dist import (all_reduce_norm, get_local_rank, get_local_size, get_num_devices, get_rank, get_world_size,
                   is_main_process, synchronize, time_synchronized)
from .ema import ModelEMA, is_parallel
from .lr_scheduler import LRScheduler
from .model_utils import fuse_model, get_model_info, replace_module
from .setup_env import configure_nccl, configure_omp
from .visualize import vis
