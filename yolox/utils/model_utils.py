"""``fuse_model`` / ``get_model_info`` / ``replace_module`` of the reference's yolox/utils/model_utils.py:14-106."""
import torch
import torch.nn as nn

__all__ = ["fuse_model", "get_model_info", "replace_module"]


def get_model_info(model, tsize):
    """"Params: x.xxM, Gflops: y.yy" for a (H, W) input.  The reference profiles with thop on a stride-64 dummy image and
    rescales; thop is not in this image, so the conv / linear MACs are counted analytically from the module shapes with
    forward hooks on a CPU meta pass over the same stride-64 dummy input."""
    params = sum(p.numel() for p in model.parameters()) / 1e6
    stride = 64
    macs = 0

    def conv_hook(m, inp, out):
        nonlocal macs
        macs += out.numel() // out.shape[1] * m.weight.numel() if out.dim() == 4 else 0

    hooks, shapes = [], {}
    # analytic count: every Conv2d of the model sees a map of size stride/its cumulative stride -- derive it from a walk over
    # the known TSCD / YOLOX topology is brittle, so count per conv from recorded output shapes when the model can run on
    # this device, else fall back to parameters only
    dev = next(model.parameters()).device
    flops = float("nan")
    if dev.type == "cuda":
        import copy
        m = copy.deepcopy(model).eval()
        recs = []

        def rec(mod, inp, out):
            o = out
            if isinstance(o, torch.Tensor) and o.dim() == 4:
                # BaseConv works on NHWC tensors: pixels = numel / channels
                px = o.numel() // mod.conv.out_channels
                recs.append(px * mod.conv.weight.numel())

        from yolox.models.network_blocks import BaseConv
        for mod in m.modules():
            if isinstance(mod, BaseConv):
                hooks.append(mod.register_forward_hook(rec))
        try:
            with torch.no_grad():
                img = torch.zeros((1, 3, stride, stride), device=dev)
                try:
                    m(img)
                except TypeError:
                    m(img, lframe=0, gframe=1)
            flops = 2 * sum(recs) * tsize[0] * tsize[1] / stride / stride / 1e9
        except Exception:
            flops = float("nan")
        for h in hooks:
            h.remove()
    return "Params: {:.2f}M, Gflops: {:.2f}".format(params, flops)


def fuse_model(model):
    """The reference folds every BatchNorm into its conv and swaps BaseConv.forward for ``fuseforward`` (model_utils.py:62-71).
    The BaseConv of this package ALWAYS runs conv + folded BatchNorm + SiLU as one kernel epilogue in eval mode, so there is
    nothing left to rewrite; the fold itself (scale / shift from the running statistics) is cached on the module here so that
    later eval forwards skip its one small launch per conv.  Returns the model (eval mode expected)."""
    from tscd_amd import ops
    from yolox.models.network_blocks import BaseConv
    for m in model.modules():
        if isinstance(m, BaseConv) and m.conv.weight.is_cuda:
            bn = m.bn
            m._folded = ops.bn_fold(bn.weight.float(), bn.bias.float(), bn.running_mean.float(), bn.running_var.float(), bn.eps)
            m._folded_ver = (bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version, bn.weight.data_ptr())
    return model


def replace_module(module, replaced_module_type, new_module_type, replace_func=None):
    def default(replaced_module_type, new_module_type):
        return new_module_type()
    replace_func = replace_func or default
    model = module
    if isinstance(module, replaced_module_type):
        model = replace_func(replaced_module_type, new_module_type)
    else:
        for name, child in module.named_children():
            new_child = replace_module(child, replaced_module_type, new_module_type)
            if new_child is not child:
                model.add_module(name, new_child)
    return model
