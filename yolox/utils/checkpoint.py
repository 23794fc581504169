"""Checkpoint I/O with the reference's file contract (yolox/utils/checkpoint.py:11-43: ``<name>_ckpt.pth`` holding
``{"start_epoch", "model", "optimizer"}``, ``best_ckpt.pth`` a copy of the file that set the best AP; ``load_ckpt`` keeps only
entries whose name and shape match) and a writer that does not stall the training loop.

``torch.save`` of a TSCD-L state (model + momentum buffers, ~0.5 GB) every 2000 iterations blocks the reference's loop for the
whole serialisation.  Here the state is snapshotted into pinned host buffers by asynchronous copies queued on the current stream
(so the snapshot is the state at the call, and later optimizer steps simply queue behind the copies), and a worker thread waits
for the copies and writes the files.  ``wait()`` joins the writer (end of training, or before a checkpoint is read back)."""
import os
import shutil
import threading

import torch

_writer = None          # at most one checkpoint in flight
_failure = None         # exception of the last writer thread that failed, re-raised on the training thread by wait()
_pinned = {}            # (key path, shape, dtype) -> pinned buffer, reused from one checkpoint to the next


def load_ckpt(model, ckpt):
    own = model.state_dict()
    keep = {k: v for k, v in ckpt.items() if k in own and own[k].shape == v.shape}
    model.load_state_dict(keep, strict=False)
    return model


def _snapshot(obj, path=""):
    """Nested dicts / lists of tensors -> the same structure on the host; CUDA tensors go through reused pinned buffers."""
    if torch.is_tensor(obj):
        if not obj.is_cuda:
            return obj.detach().clone()
        key = (path, tuple(obj.shape), obj.dtype)
        buf = _pinned.get(key)
        if buf is None:
            buf = _pinned[key] = torch.empty(obj.shape, dtype=obj.dtype, pin_memory=True)
        buf.copy_(obj.detach(), non_blocking=True)
        return buf
    if isinstance(obj, dict):
        return {k: _snapshot(v, f"{path}/{k}") for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_snapshot(v, f"{path}/{i}") for i, v in enumerate(obj))
    return obj


def wait():
    """Joins the writer in flight and re-raises, on the calling (training) thread, whatever made it fail -- a full disk or an
    unwritable directory must stop the run, not be reported as a checkpoint that was never written."""
    global _writer, _failure
    if _writer is not None:
        _writer.join()
        _writer = None
    if _failure is not None:
        err, _failure = _failure, None
        raise RuntimeError(f"asynchronous checkpoint write failed: {err!r}") from err


def save_checkpoint(state, is_best, save_dir, model_name="", non_blocking=True):
    """Writes ``<save_dir>/<model_name>_ckpt.pth`` (and ``best_ckpt.pth`` when ``is_best``).  With ``non_blocking`` the call returns
    once the device-to-host copies are queued; the previous checkpoint's writer is joined first (its pinned buffers are reused)."""
    global _writer
    os.makedirs(save_dir, exist_ok=True)
    filename = os.path.join(save_dir, model_name + "_ckpt.pth")
    if not non_blocking or not torch.cuda.is_available():
        wait()
        torch.save(state, filename)
        if is_best:
            shutil.copyfile(filename, os.path.join(save_dir, "best_ckpt.pth"))
        return filename
    wait()
    host = _snapshot(state)
    done = torch.cuda.Event()
    done.record()
    dev = torch.cuda.current_device()

    def write():
        global _failure
        tmp = filename + ".tmp"
        try:
            torch.cuda.set_device(dev)
            done.synchronize()
            torch.save(host, tmp)
            os.replace(tmp, filename)                 # a reader never sees a half-written file
            if is_best:
                shutil.copyfile(filename, os.path.join(save_dir, "best_ckpt.pth"))
        except BaseException as ex:                   # surfaced by wait() / the next save_checkpoint
            _failure = ex
            try:
                os.remove(tmp)
            except OSError:
                pass

    _writer = threading.Thread(target=write, name="tscd-ckpt-writer", daemon=False)
    _writer.start()
    return filename
