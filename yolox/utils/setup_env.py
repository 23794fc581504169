"""Environment knobs (contract of yolox/utils/setup_env.py:16-48).  The reference sets InfiniBand NCCL variables; on a
single MI355X node RCCL runs over xGMI, so only the IPC mode required by this platform and OMP threads are set."""
import os


def configure_nccl():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL / tensor sharing across processes)
    os.environ.setdefault("NCCL_LAUNCH_MODE", "PARALLEL")


def configure_omp(num_threads=1):
    os.environ.setdefault("OMP_NUM_THREADS", str(num_threads))
