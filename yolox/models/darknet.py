"""CSPDarknet backbone on the MI355X conv kernels (interface of the reference's yolox/models/darknet.py:98-180).

Takes the NCHW image tensor of the reference contract, runs NHWC fp16 internally and returns NHWC maps; the
public NCHW view is produced by YOLOPAFPN.
"""
import torch.nn as nn

from .network_blocks import BaseConv, CSPLayer, Focus, SPPBottleneck


class CSPDarknet(nn.Module):
    def __init__(self, dep_mul, wid_mul, out_features=("dark3", "dark4", "dark5"), depthwise=False, act="silu"):
        super().__init__()
        assert out_features, "please provide output features of Darknet"
        if depthwise:
            raise NotImplementedError("depthwise convs are not on the TSCD path")
        self.out_features = out_features
        bc = int(wid_mul * 64)
        bd = max(round(dep_mul * 3), 1)
        self.stem = Focus(3, bc, ksize=3, act=act)
        self.dark2 = nn.Sequential(BaseConv(bc, bc * 2, 3, 2, act=act),
                                   CSPLayer(bc * 2, bc * 2, n=bd, depthwise=depthwise, act=act))
        self.dark3 = nn.Sequential(BaseConv(bc * 2, bc * 4, 3, 2, act=act),
                                   CSPLayer(bc * 4, bc * 4, n=bd * 3, depthwise=depthwise, act=act))
        self.dark4 = nn.Sequential(BaseConv(bc * 4, bc * 8, 3, 2, act=act),
                                   CSPLayer(bc * 8, bc * 8, n=bd * 3, depthwise=depthwise, act=act))
        self.dark5 = nn.Sequential(BaseConv(bc * 8, bc * 16, 3, 2, act=act),
                                   SPPBottleneck(bc * 16, bc * 16, activation=act),
                                   CSPLayer(bc * 16, bc * 16, n=bd, shortcut=False, depthwise=depthwise, act=act))

    def forward(self, x, outs=None):
        """outs: optional dict name -> preallocated NHWC destination (a concat slice owned by the neck)."""
        outs = outs or {}
        res = {}
        x = self.stem(x)
        res["stem"] = x
        x = self.dark2(x)
        res["dark2"] = x
        x = self.dark3[1](self.dark3[0](x), out=outs.get("dark3"))
        res["dark3"] = x
        x = self.dark4[1](self.dark4[0](x), out=outs.get("dark4"))
        res["dark4"] = x
        x = self.dark5(x)
        res["dark5"] = x
        return {k: v for k, v in res.items() if k in self.out_features}


# The reference's package also exports Darknet (YOLOv3); it is not on the TSCD path.
class Darknet(nn.Module):
    def __init__(self, *a, **k):
        raise NotImplementedError("Darknet53 is out of scope of the TSCD hot path (SURVEY.md section 2)")
