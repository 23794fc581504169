#!/usr/bin/env python3
"""Like tools/conv_ablate_gen.py, for conv_igemm_glds_kernel (the 128-channel-wide LDS-DMA variant): writes tools/_ablate/conv_abl_glds.hip
with run-time switches (1 no LDS-DMA after the prologue, 2 no weight (B) pieces after the prologue, 4 no fragment reads, 8 no MFMAs, 64 no main loop, 128 no epilogue) and builds
tools/_ablate/libabl_glds.so.  Timing only -- the results are garbage.  The library is linked -Bsymbolic: its kernels' host stubs have the same names as the product library's (loaded RTLD_GLOBAL for tscd_set_error) and must not be interposed by them."""
import os, subprocess
here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(here)
out = os.path.join(here, "_ablate")
os.makedirs(out, exist_ok=True)
src = open(os.path.join(root, "tscd_amd", "csrc", "conv_igemm.hip")).read()
a = src.index("template <int BM, int BN, int WAVES_N>\n__global__ __launch_bounds__(128 * WAVES_N")
b = src.index("// Main-loop variant 3 (Cin % 64 == 0, Cout > 128)")
head, k, tail = src[:a], src[a:b], src[b:]
def rep(x, y, cnt=1):
    global k
    assert k.count(x) == cnt, (x, k.count(x))
    k = k.replace(x, y)
rep("  auto issue = [&](int stage) {\n    char* sbase = smem + stage * STAGE;", "  int n_issued = 0;\n  auto issue = [&](int stage) {\n    char* sbase = smem + stage * STAGE;\n    const bool dma = !(g_abl_dev & 1) || n_issued < NS; ++n_issued;")
rep("      const f16* src = ((a_mask[i] >> ld_tap) & 1u) ? a_src[i] + ld_tapoff + ld_cin : g_zero_chunk;\n      __builtin_amdgcn_global_load_lds(",
    "      const f16* src = ((a_mask[i] >> ld_tap) & 1u) ? a_src[i] + ld_tapoff + ld_cin : g_zero_chunk;\n      if (dma) __builtin_amdgcn_global_load_lds(")
rep("    for (int j = 0; j < B_LD; ++j) {\n      __builtin_amdgcn_global_load_lds(", "    for (int j = 0; j < B_LD; ++j) {\n      if (dma && (!(g_abl_dev & 2) || n_issued <= NS)) __builtin_amdgcn_global_load_lds(")
rep("    const char* sb = smem + (t % NS) * STAGE;\n", "    const char* sb = smem + (t % NS) * STAGE;\n    if ((g_abl_dev & 4) && t > 1) return;\n")
rep("        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[S][ks][i], fb[S][ks][j], acc[i][j], 0, 0, 0);\n  };",
    "        for (int j = 0; j < TN; ++j) { if (!(g_abl_dev & 8)) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[S][ks][i], fb[S][ks][j], acc[i][j], 0, 0, 0); else asm volatile(\"\" :: \"v\"(fa[S][ks][i]), \"v\"(fb[S][ks][j])); }\n  };")
rep("  int kt = 0;\n  for (; kt + 2 < nk; kt += 2) {", "  int kt = 0;\n  if (g_abl_dev & 64) { wait_vmcnt<0>(); kt = nk; }\n  for (; kt + 2 < nk; kt += 2) {")
rep("  if (nk - kt == 2) {\n    phase(std::integral_constant<int, 0>{}, kt);\n    mfmas(std::integral_constant<int, 1>{});\n  } else {\n    mfmas(std::integral_constant<int, 0>{});\n  }",
    "  if (nk - kt == 2) {\n    phase(std::integral_constant<int, 0>{}, kt);\n    mfmas(std::integral_constant<int, 1>{});\n  } else if (!(g_abl_dev & 64)) {\n    mfmas(std::integral_constant<int, 0>{});\n  }")
rep("  conv_epilogue<BM, BN, WAVES_M, WAVES_N, TM, TN>(p, acc, (f16*)smem, m0, n0, mt, tid, lane, wm, wn);\n}",
    "  if (!(g_abl_dev & 128)) conv_epilogue<BM, BN, WAVES_M, WAVES_N, TM, TN>(p, acc, (f16*)smem, m0, n0, mt, tid, lane, wm, wn);\n  else if (acc[0][0][0] == 123.f) p.y[0] = (f16)acc[0][0][3];\n}")
k = k.replace("g_abl_dev", "p.abl")
head = head.replace("struct ConvParams {", "struct ConvParams {\n  int abl;")
tail = tail.replace("static int g_halo_auto = 1;", "static int g_abl = 0;\nextern \"C\" void tscd_conv_set_abl(int v) { g_abl = v; }\nstatic int g_halo_auto = 1;")
assert tail.count("  ConvParams p;") == 1
tail = tail.replace("  ConvParams p;", "  ConvParams p; p.abl = g_abl;")
open(os.path.join(out, "conv_abl_glds.hip"), "w").write(head + k + tail)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wl,-Bsymbolic", "-Wno-unused-result", "-I",
                       os.path.join(root, "tscd_amd", "csrc"), "-shared", os.path.join(out, "conv_abl_glds.hip"), "-o", os.path.join(out, "libabl_glds.so")])
print(os.path.join(out, "libabl_glds.so"))
