#!/usr/bin/env python3
"""Ablation builds of the MCA backward's delta pass (timing only: the results of an ablated build are meaningless).  Writes
tools/_ablate/libtscd_hip_mca<N>.so for N = 0 (as committed), 1 no key-fragment loads, 2 no exp, 3 no barriers inside the visit, 4 no
global stores of dM / A, 5 no MFMAs, 6 no barriers and no exchange writes.  Run with
TSCD_HIP_LIB=tools/_ablate/libtscd_hip_mca<N>.so python tools/mca_bwd_ab.py   (tools/mca_delta_ablate_run.sh)."""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(ROOT, "tscd_amd", "csrc")
LIBDIR = os.path.join(ROOT, "tscd_amd", "lib")
OUT = os.path.join(HERE, "_ablate")
os.makedirs(OUT, exist_ok=True)
src = open(os.path.join(CSRC, "mca_bwd.hip")).read()
a = src.index("template <int DP, bool NEED_REG, bool CLS_G>\n__global__ __launch_bounds__(256, 2) void mca_bwd_delta_kernel")
b = src.index("__global__ void mca_bwd_rowstat_kernel")
head, body, tail = src[:a], src[a:b], src[b:]


def variant(n):
    k = body
    if n == 1:      # operands of the products from loop-invariant registers
        k = k.replace("*(const f16x8*)(p.kF_c + o)", "qc[ks]").replace("*(const f16x8*)(p.kF_r + o)", "qr[ks]")
        k = k.replace("*(const f16x8*)(p.vnF_c + o)", "vlc[ks]").replace("*(const f16x8*)(p.vrF_c + o)", "gOc[ks]")
        k = k.replace("*(const f16x8*)(p.vnF_r + o)", "vlr[ks]").replace("*(const f16x8*)(p.vrF_r + o)", "gOr[ks]")
    if n == 2:
        k = k.replace("__builtin_amdgcn_exp2f(", "(").replace("__expf(", "(")
    if n in (3, 6):
        i = k.index("for (int j = s; j < nblk; j += p.S) {")
        j = k.index("#pragma unroll\n  for (int e = 0; e < 16; ++e) {\n#pragma unroll\n    for (int o = 16;")
        loop = k[i:j].replace("__syncthreads();", "")
        if n == 6:      # ... and no exchange writes either (the phases read stale LDS)
            loop = re.sub(r"\n\s*X\[[0-2]\]\[h\]\[row\]\[col\] = [^;]*;", "", loop)
        k = k[:i] + loop + k[j:]
    if n == 4:
        k = re.sub(r"\n\s*\*\(f32x4\*\)\(p\.dMq[^\n]*", "\n      if (dm[0] == 123.456f) p.dMq[0] = dm[1];", k)
        k = re.sub(r"\n\s*\*\(f32x4\*\)\(p\.dMk \+ o\) = [^\n]*", "", k)
        k = re.sub(r"\n\s*\*\(f16x4\*\)\(p\.Ac?r?k \+ o\) = [^\n]*", "", k)
        k = k.replace("      if (CLS_G)\n      if (NEED_REG)\n", "")
    if n == 5:
        k = re.sub(r"__builtin_amdgcn_mfma_f32_32x32x16_f16\((\w+)\[ks\], (\w+), (\w+), 0, 0, 0\)", r"abl_acc(\3, \1[ks], \2)", k)
    pre = "__device__ __forceinline__ f32x16 abl_acc(f32x16 c, f16x8 a, f16x8 b) { c[0] += (float)a[0] * (float)b[0]; return c; }\n" if n == 5 else ""
    return head + pre + k + tail


for n in [int(x) for x in sys.argv[1:]] or range(7):
    path = os.path.join(OUT, f"mca_bwd_abl{n}.hip")
    open(path, "w").write(variant(n))
    obj = os.path.join(OUT, f"mca_bwd_abl{n}.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-Wno-unused-value", "-I", CSRC,
                           "-c", path, "-o", obj])
    objs = [os.path.join(LIBDIR, f) for f in sorted(os.listdir(LIBDIR)) if f.endswith(".o") and f != "mca_bwd.o"] + [obj]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(OUT, f"libtscd_hip_mca{n}.so")] + objs)
    print("built", n, flush=True)
