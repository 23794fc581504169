"""Plugin surface: get_exp loads this repository's TSCD exp files, which build the mirror model with the
reference's attribute names, optimizer grouping and LR schedule.  CPU only."""
import math

import torch


def test_exp_builds_model_optimizer_scheduler():
    from yolox.exp import BaseExp, get_exp
    exp = get_exp("exps/TSCD_OVIS/ovis_tscd_large.py")
    assert isinstance(exp, BaseExp)
    exp.merge(["depth", "0.33", "width", "0.125", "max_epoch", "3"])
    assert exp.depth == 0.33 and exp.max_epoch == 3
    for k in ("lframe", "gframe", "lframe_val", "gframe_val", "input_size", "test_size", "ema", "basic_lr_per_img", "seed",
              "print_interval", "eval_interval", "no_aug_epochs", "warmup_epochs", "output_dir", "max_epoch"):
        assert hasattr(exp, k), k
    model = exp.get_model()
    assert len(model.state_dict()) == 630
    frozen = [k for k, p in model.named_parameters() if not p.requires_grad]
    assert any(k.startswith("backbone.") for k in frozen) and any(k.startswith("head.stems") for k in frozen)
    assert all(p.requires_grad for k, p in model.named_parameters() if k.startswith("head.cls_convs2"))
    opt = exp.get_optimizer(16)
    assert len(opt.param_groups) == 4 and opt.param_groups[3]["lr"] == opt.param_groups[0]["lr"] * exp.stem_lr_ratio
    sch = exp.get_lr_scheduler(exp.basic_lr_per_img * 16, 100)
    lr0 = exp.basic_lr_per_img * 16
    assert abs(sch.update_lr(1) - (0.1 * lr0 + 0.5 * 0.9 * lr0 * (1 + math.cos(math.pi * 1 / 100)))) < 1e-12   # warmup_epochs = 0
    assert sch.update_lr(250) == lr0 * exp.min_lr_ratio      # last no_aug epochs sit at min lr
    sz = exp.random_resize(None, 0, 0, False)
    assert sz[0] % 64 == 0 and sz[1] % 64 == 0


def test_lr_schedules_closed_forms():
    from yolox.utils import LRScheduler
    s = LRScheduler("yoloxwarmcos", 0.01, 10, 5, warmup_epochs=1, warmup_lr_start=0, no_aug_epochs=1, min_lr_ratio=0.05)
    assert abs(s.update_lr(5) - 0.01 * 0.25) < 1e-12 and s.update_lr(45) == 0.01 * 0.05
    assert abs(LRScheduler("cos", 1.0, 10, 1).update_lr(5) - 0.5) < 1e-12
    assert abs(LRScheduler("multistep", 1.0, 10, 10, milestones=[3, 6]).update_lr(70) - 0.01) < 1e-12


def test_synthetic_loader_is_rank_sharded():
    from yolox.data import SyntheticClipLoader, get_timing_signal_1d
    a = SyntheticClipLoader(4, 64, num_clips=10, device="cpu", rank=0, world_size=2)
    b = SyntheticClipLoader(4, 64, num_clips=10, device="cpu", rank=1, world_size=2)
    assert set(a.ids).isdisjoint(b.ids) and len(a) + len(b) == 10
    imgs, t, te = a.next()
    assert imgs.shape == (4, 3, 64, 64) and t.shape == (4, 100, 5) and te.shape == (4, 256)
    assert torch.allclose(te, get_timing_signal_1d(torch.arange(4), 256))
