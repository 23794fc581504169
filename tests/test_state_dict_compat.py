"""Drop-in contract: the mirror modules expose exactly the reference's state_dict keys and shapes, so
reference checkpoints load with strict=True.  CPU-only (construction needs no GPU; the library must load)."""
import torch


def build_model(cfg, **more):
    from yolox.models import TSCD, TSCDHead, YOLOPAFPN
    kwargs = dict(reconf=True, ota_mode=True, use_pre_nms=False, cat_ota_fg=False, agg_type="mca", decouple_reg=True,
                  minimal_limit=cfg.minimal_limit, maximal_limit=cfg.maximal_limit, conf_sim_thresh=cfg.conf_sim_thresh)
    kwargs.update(more)
    backbone = YOLOPAFPN(cfg.depth, cfg.width, in_channels=[256, 512, 1024])
    head = TSCDHead(cfg.num_classes, cfg.width, in_channels=[256, 512, 1024], heads=4, **kwargs)
    m = TSCD(backbone, head)
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.eps, mod.momentum = 1e-3, 0.03
    return m


def test_state_dict_keys_and_shapes_match_reference(golden):
    cfg, sd, vec, meta = golden
    m = build_model(cfg)
    ours = m.state_dict()
    assert set(ours.keys()) == set(sd.keys())
    assert len(ours) == 630
    for k, v in sd.items():
        assert tuple(ours[k].shape) == tuple(v.shape), k
    m.load_state_dict(sd, strict=True)
    # the Haar buffers are regenerated analytically and must equal the reference's pywt-derived ones
    for k in ("head.edge_enhance_reg.0.0.dwt.w_lh", "head.edge_enhance_reg.1.0.dwt.w_hl", "head.edge_enhance_reg.2.0.idwt.filters"):
        fresh = build_model(cfg).state_dict()[k]
        assert torch.allclose(fresh, sd[k], atol=1e-7), k


def test_library_exports_every_declared_symbol():
    import ctypes
    import re
    import os
    from tscd_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "tscd_hip.h")).read()
    names = set(re.findall(r"\b(tscd_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 25
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), n
    assert set(_lib.SIGNATURES) | set(_lib.PLAIN) == names


def test_host_library_exports_every_declared_symbol():
    import ctypes
    import re
    import os
    from tscd_amd import _host
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "tscd_host.h")).read()
    names = set(re.findall(r"\b(tscd_[a-z0-9_]+)\s*\(", hdr))
    assert names == {"tscd_lap_solve_f64", "tscd_lap_solve_f16", "tscd_lap_chain_f16", "tscd_lap_fixed_point_available",
                     "tscd_lap_debug_steps", "tscd_lap_debug_width", "tscd_lap_chain_set_parallel", "tscd_lap_chain_spec_hits", "tscd_cocoeval_bbox",
                     "tscd_lap_chain_f16_table", "tscd_latch_create", "tscd_latch_destroy", "tscd_latch_signal", "tscd_latch_generation",
                     "tscd_latch_timeouts", "tscd_latch_wait", "tscd_latch_ticket", "tscd_latch_wait_cb", "tscd_latch_set_spin"}
    lib = ctypes.CDLL(_host.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), n
