"""Distributed helpers (contract of the reference's yolox/utils/dist.py:41-294) over torch.distributed: backend "nccl"
is RCCL over xGMI on MI355X.  Device discovery uses torch, not nvidia-smi shell-outs."""
import time

import torch
import torch.distributed as dist

_LOCAL_PROCESS_GROUP = None


def get_num_devices():
    return torch.cuda.device_count()


def _ready():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if _ready() else 1


def get_rank():
    return dist.get_rank() if _ready() else 0


def get_local_rank():
    if not _ready():
        return 0
    return dist.get_rank(group=_LOCAL_PROCESS_GROUP) if _LOCAL_PROCESS_GROUP is not None else get_rank()


def get_local_size():
    if not _ready():
        return 1
    return dist.get_world_size(group=_LOCAL_PROCESS_GROUP) if _LOCAL_PROCESS_GROUP is not None else get_world_size()


def is_main_process():
    return get_rank() == 0


def synchronize():
    if _ready() and dist.get_world_size() > 1:
        dist.barrier()


def time_synchronized():
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    return time.time()


def all_reduce_norm(module):
    """Average BatchNorm / InstanceNorm states over ranks before eval (yolox/utils/allreduce_norm.py:59-103):
    one flattened all-reduce instead of a pickled key broadcast."""
    ws = get_world_size()
    if ws == 1:
        return
    norms = (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d, torch.nn.BatchNorm3d, torch.nn.InstanceNorm2d)
    tensors = [v for m in module.modules() if isinstance(m, norms) for v in m.state_dict().values() if v.dtype.is_floating_point]
    if not tensors:
        return
    flat = torch.cat([t.reshape(-1).float() for t in tensors])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat /= ws
    o = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[o:o + n].view_as(t))
        o += n
