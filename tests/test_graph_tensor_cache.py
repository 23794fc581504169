"""tscd_amd.graphs caches tuple(module.parameters()) / tuple(module.buffers()) (walking CSPDarknet + PAFPN costs ~2 ms per call and
the graph-validity check runs every step).  The cache must notice when .half() / .to() replaced the tensors."""
import copy

import torch


def test_cached_tensor_lists_follow_dtype_conversions_and_copies():
    from tscd_amd import graphs
    m = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 1), torch.nn.BatchNorm2d(4), torch.nn.Sequential(torch.nn.BatchNorm2d(4)))
    p1, b1 = graphs._tensors_of(m)
    assert len(p1) == len(list(m.parameters())) and len(b1) == len(list(m.buffers()))
    assert graphs._tensors_of(m)[1] is b1                                   # cached: the same tuple object
    m.half()                                                                # Module._apply replaces floating-point buffers
    p2, b2 = graphs._tensors_of(m)
    assert b2 is not b1 and all(t.dtype == torch.float16 for t in b2 if t.is_floating_point())
    assert all(a is b for a, b in zip(b2, m.buffers()))
    m2 = copy.deepcopy(m).float()                                           # a copy has its own entry and does not disturb the original's
    assert all(t.dtype == torch.float32 for t in graphs._tensors_of(m2)[1] if t.is_floating_point())
    assert graphs._tensors_of(m)[1] is b2
    v0 = graphs._state_versions(m, training=False)
    with torch.no_grad():
        m[1].running_mean.add_(1.0)                                         # a buffer write invalidates evaluation graphs ...
    assert graphs._state_versions(m, training=False) != v0
    assert graphs._state_versions(m, training=True) == graphs._state_versions(m, training=True)
    vt = graphs._state_versions(m, training=True)
    with torch.no_grad():
        m[0].weight.mul_(2.0)                                               # ... a parameter write every graph
    assert graphs._state_versions(m, training=True) != vt
    graphs.clear()
