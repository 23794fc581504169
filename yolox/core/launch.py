"""Process launcher (contract of the reference's yolox/core/launch.py:39-147): ``launch(main, num_gpus_per_machine, ...)``
runs ``main`` directly for one GPU and spawns one process per GPU otherwise, each joining a torch.distributed group
(backend "nccl" = RCCL over xGMI) through a 127.0.0.1 rendezvous.  Works the same under torchrun (env already set)."""
import os
import socket
from datetime import timedelta

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

DEFAULT_TIMEOUT = timedelta(minutes=30)


def _find_free_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(local_rank, main_func, world_size, num_gpus, machine_rank, backend, dist_url, args):
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local_rank)
    dist.init_process_group(backend=backend, init_method=dist_url, world_size=world_size,
                            rank=machine_rank * num_gpus + local_rank, timeout=DEFAULT_TIMEOUT)
    dist.barrier()
    main_func(*args)
    dist.destroy_process_group()


def launch(main_func, num_gpus_per_machine, num_machines=1, machine_rank=0, backend="nccl", dist_url=None, args=(),
           timeout=DEFAULT_TIMEOUT):
    world_size = num_machines * num_gpus_per_machine
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) > 1:   # torchrun
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group(backend=backend, timeout=timeout)
        main_func(*args)
        dist.destroy_process_group()
        return
    if world_size <= 1:
        main_func(*args)
        return
    if dist_url in (None, "auto"):
        assert num_machines == 1, "dist_url=auto cannot work with distributed training."
        dist_url = f"tcp://127.0.0.1:{_find_free_port()}"
    mp.start_processes(_worker, nprocs=num_gpus_per_machine, start_method="spawn",
                       args=(main_func, world_size, num_gpus_per_machine, machine_rank, backend, dist_url, args))
