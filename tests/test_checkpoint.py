"""Checkpoint file contract (reference yolox/utils/checkpoint.py:11-43, tscd_trainer.py:287-293, 346-347, 394-425, 454-468)."""
import os
import types

import pytest
import torch


def test_save_checkpoint_files_and_shape_tolerant_load(tmp_path):
    from yolox.utils import checkpoint
    m = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 2))
    state = {"start_epoch": 3, "model": m.state_dict(), "optimizer": {"state": {0: {"momentum_buffer": torch.ones(3, 4)}}, "param_groups": [{"lr": 0.1}]}}
    f = checkpoint.save_checkpoint(state, True, str(tmp_path), "last_epoch")
    checkpoint.wait()
    assert os.path.basename(f) == "last_epoch_ckpt.pth" and os.path.isfile(tmp_path / "best_ckpt.pth")
    back = torch.load(f, weights_only=True)
    assert back["start_epoch"] == 3 and torch.equal(back["model"]["0.weight"], m[0].weight)
    assert torch.equal(back["optimizer"]["state"][0]["momentum_buffer"], torch.ones(3, 4))
    # load_ckpt keeps entries whose name and shape match
    other = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 5))
    before = other[1].weight.clone()
    checkpoint.load_ckpt(other, back["model"])
    assert torch.equal(other[0].weight, m[0].weight) and torch.equal(other[1].weight, before)


@pytest.mark.gpu
def test_async_checkpoint_is_a_snapshot_of_the_call(tmp_path):
    """The writer thread sees the state at the call even when the tensors are overwritten right after it."""
    from yolox.utils import checkpoint
    w = torch.full((1 << 22,), 3.0, device="cuda")
    state = {"start_epoch": 1, "model": {"w": w}, "optimizer": {}}
    f = checkpoint.save_checkpoint(state, False, str(tmp_path), "latest", non_blocking=True)
    w.mul_(2.0)                                   # the "next optimizer step"
    f2 = checkpoint.save_checkpoint({"start_epoch": 2, "model": {"w": w}, "optimizer": {}}, False, str(tmp_path), "other", non_blocking=True)
    checkpoint.wait()
    assert float(torch.load(f, weights_only=True)["model"]["w"].mean()) == 3.0
    assert float(torch.load(f2, weights_only=True)["model"]["w"].mean()) == 6.0
    assert not os.path.exists(f + ".tmp")


@pytest.mark.gpu
def test_trainer_loop_checkpoint_contract_and_resume(tmp_path, monkeypatch):
    """Two tiny epochs of the trainer on the synthetic clip source: ``latest`` at iteration 0 of each epoch and at each epoch end,
    ``last_epoch`` after every evaluation, ``best_ckpt`` only when the AP improves; --resume restores model, optimizer and epoch."""
    from yolox.core.tscd_trainer import Trainer
    from yolox.exp import get_exp
    from yolox.utils import checkpoint
    exp = get_exp("exps/TSCD_OVIS/ovis_tscd_base.py")
    exp.depth, exp.width = 0.33, 0.25
    exp.input_size = exp.test_size = (128, 128)
    exp.lframe, exp.gframe = 2, 2
    exp.max_epoch, exp.eval_interval = 2, 1
    exp.output_dir = str(tmp_path)
    exp.data_dir = str(tmp_path / "no_data")      # -> synthetic clips
    exp.multiscale_range = 0
    args = types.SimpleNamespace(batch_size=4, fp16=True, experiment_name="t", ckpt=None, resume=False)
    tr = Trainer(exp, args, val_loader=object())
    saves = []
    real_save = checkpoint.save_checkpoint

    def spy(state, is_best, save_dir, model_name="", non_blocking=True):
        saves.append((model_name, bool(is_best), state["start_epoch"]))
        return real_save(state, is_best, save_dir, model_name, non_blocking)
    monkeypatch.setattr(checkpoint, "save_checkpoint", spy)
    aps = iter([0.30, 0.20])
    monkeypatch.setattr(Trainer, "evaluate", lambda self: (next(aps), 0.5, "summary"))
    monkeypatch.setattr(type(exp), "get_data_loader", lambda self, batch_size, is_distributed, **kw: _Short(self._synthetic(batch_size, 128, is_distributed), 3), raising=True)
    tr.train()
    assert saves == [("latest", False, 1), ("latest", False, 1), ("last_epoch", True, 1),
                     ("latest", False, 2), ("latest", False, 2), ("last_epoch", False, 2)]
    d = tmp_path / "t"
    for name in ("latest_ckpt.pth", "last_epoch_ckpt.pth", "best_ckpt.pth"):
        assert os.path.isfile(d / name), name
    assert torch.load(d / "best_ckpt.pth", weights_only=True)["start_epoch"] == 1
    last = torch.load(d / "latest_ckpt.pth", weights_only=True)
    assert last["start_epoch"] == 2 and set(last) == {"start_epoch", "model", "optimizer"}
    assert tr.best_ap == 0.30
    # resume
    args2 = types.SimpleNamespace(batch_size=4, fp16=True, experiment_name="t", ckpt=None, resume=True)
    tr2 = Trainer(exp, args2)
    tr2.before_train()
    assert tr2.start_epoch == 2
    ema_sd = {k: v for k, v in (tr.ema_model.ema.state_dict() if tr.use_model_ema else tr.model.state_dict()).items()}
    for k, v in tr2.model.state_dict().items():
        assert torch.equal(v, ema_sd[k]), k


class _Short:
    """The first n clips of a clip source per epoch."""

    def __init__(self, src, n):
        self.src, self.n = src, n

    def __len__(self):
        return self.n

    def next(self):
        return self.src.next()


@pytest.mark.gpu
def test_failed_asynchronous_write_is_raised_on_the_training_thread(tmp_path):
    """an exception in the writer thread (disk full, permission) used to be printed by the thread and lost -- training went
    on reporting checkpoints that were never written.  wait() and the next save_checkpoint re-raise it; no .tmp file is left."""
    from yolox.utils import checkpoint
    d = tmp_path / "ck"
    state = {"start_epoch": 1, "model": {"w": torch.ones(4, device="cuda")}, "optimizer": {}}
    checkpoint.save_checkpoint(state, False, str(d), "latest")
    checkpoint.wait()
    # make the target unwritable in a way that also holds for root: the checkpoint's name is taken by a directory
    os.remove(d / "latest_ckpt.pth")
    os.makedirs(d / "latest_ckpt.pth" / "x")
    checkpoint.save_checkpoint(state, True, str(d), "latest")
    with pytest.raises(RuntimeError, match="asynchronous checkpoint write failed"):
        checkpoint.wait()
    assert not os.path.exists(d / "latest_ckpt.pth.tmp") and not os.path.exists(d / "best_ckpt.pth")
    checkpoint.wait()                                       # raised once, then clear
    checkpoint.save_checkpoint(state, False, str(d), "other")
    checkpoint.wait()
    assert os.path.isfile(d / "other_ckpt.pth")
