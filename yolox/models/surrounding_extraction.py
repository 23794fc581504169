"""WaveletsHFBlock on the MI355X kernels (interface of the reference's surrounding_extraction.py:79-127,215-267).

Haar DWT -> drop LL -> 1x1 conv + ReLU on the high-frequency bands -> IDWT -> multiply with a 3x3 conv + ReLU of
the input.  The DWT/IDWT taps are constants (+-1/2), so the analysis and synthesis steps are two HBM-bound HIP
kernels (haar_hf / haar_idwt_mul) and both convs are the MFMA implicit GEMM with a bias+ReLU epilogue.  The
``dwt.w_*`` / ``idwt.filters`` buffers are kept only for state_dict compatibility.
"""
import math

import torch
import torch.nn as nn

from tscd_amd import functional as TF


def _haar_buffers():
    s = 1.0 / math.sqrt(2.0)
    dec_lo, dec_hi = torch.tensor([s, s]), torch.tensor([s, -s])      # pywt dec_* reversed, as the reference does
    rec_lo, rec_hi = torch.tensor([s, s]), torch.tensor([s, -s])
    outer = lambda a, b: (a.unsqueeze(0) * b.unsqueeze(1)).unsqueeze(0).unsqueeze(0)
    dwt = dict(w_ll=outer(dec_lo, dec_lo), w_lh=outer(dec_lo, dec_hi), w_hl=outer(dec_hi, dec_lo), w_hh=outer(dec_hi, dec_hi))
    idwt = torch.cat([outer(rec_lo, rec_lo), outer(rec_lo, rec_hi), outer(rec_hi, rec_lo), outer(rec_hi, rec_hi)], dim=0)
    return dwt, idwt


class DWT_2D(nn.Module):
    def __init__(self, wave="haar"):
        super().__init__()
        assert wave == "haar"
        for k, v in _haar_buffers()[0].items():
            self.register_buffer(k, v)


class IDWT_2D(nn.Module):
    def __init__(self, wave="haar"):
        super().__init__()
        assert wave == "haar"
        self.register_buffer("filters", _haar_buffers()[1])


class WaveletsHFBlock(nn.Module):
    def __init__(self, in_channels):
        super().__init__()
        self.dwt = DWT_2D("haar")
        self.idwt = IDWT_2D("haar")
        self.filter1 = nn.Sequential(nn.Conv2d(in_channels * 3, in_channels * 3, 1, padding=0), nn.ReLU(inplace=True))
        self.filter2 = nn.Sequential(nn.Conv2d(in_channels, in_channels, 3, padding=1), nn.ReLU(inplace=True))

    def forward(self, x):
        """x: NHWC fp16 [B,H,W,C] (H, W even) -> [B,H,W,C]."""
        hf = TF.haar_hf(x)
        hf = TF.conv_bias(hf, self.filter1[0].weight, self.filter1[0].bias, relu=True)
        content = TF.conv_bias(x, self.filter2[0].weight, self.filter2[0].bias, relu=True)
        return TF.haar_idwt_mul(hf, content)
