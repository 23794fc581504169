#!/usr/bin/env python3
"""Writes tools/_ablate/conv_abl.hip -- tscd_amd/csrc/conv_igemm.hip with run-time switches that skip one ingredient of the
halo-staged kernel at a time (LDS-DMA, fragment reads, MFMAs, barriers, main loop, epilogue, stores: timing only, the results are
garbage) -- and builds tools/_ablate/libabl.so from it.  tools/conv_ablate_run.py times the switches on the dominant shape
(profiles/r02_conv_halo_ablation.txt).  The patches are exact-text replacements that assert their match count: when the kernel
changes, this script fails instead of patching the wrong place."""
import os, re
here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(here)
out = os.path.join(here, "_ablate")
os.makedirs(out, exist_ok=True)
src = open(os.path.join(root, "tscd_amd", "csrc", "conv_igemm.hip")).read()
assert src.count("if (m < p.M && n < p.Cout) *(f16x8*)(p.y + (long)m * p.ldy + n) = *(const f16x8*)(ctile + row * CP + ch * 8);") == 1
src = src.replace("if (m < p.M && n < p.Cout) *(f16x8*)(p.y + (long)m * p.ldy + n) = *(const f16x8*)(ctile + row * CP + ch * 8);",
  "if (m < p.M && n < p.Cout) { f16x8 vv = *(const f16x8*)(ctile + row * CP + ch * 8); if (!(p.abl & 256)) *(f16x8*)(p.y + (long)m * p.ldy + n) = vv; else asm volatile(\"\" :: \"v\"(vv)); }")
src = src.replace("  if (stats && tid < BN) {\n    if constexpr (SPLIT_STATS) {", "  if (stats && tid < BN && !(p.abl & 512)) {\n    if constexpr (SPLIT_STATS) {")
src = src.replace("struct ConvParams {", "struct ConvParams {\n  int abl;")
cut = src.index("template <int TM>\n__global__ __launch_bounds__(512, 1) void conv_igemm_halo_kernel")
head, src = src[:cut], src[cut:]
def rep(a, b, cnt=1):
    global src
    assert src.count(a) == cnt, (a, src.count(a))
    src = src.replace(a, b)
rep("void conv_igemm_halo_kernel(ConvParams p, int HRP) {", "void conv_igemm_halo_kernel(ConvParams p, int HRP, int abl) {")
# B DMA
rep("""    const int koff = ld_tap * p.Cin + ld_cin;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[j] + koff),
                                       (__attribute__((address_space(3))) void*)(sbase + (wave * 2 + j) * 1024), 16, 0, 0);""",
"""    const int koff = ld_tap * p.Cin + ld_cin;
    if (!(abl & 1) || ld_tile < 6) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[j] + koff),
                                       (__attribute__((address_space(3))) void*)(sbase + (wave * 2 + j) * 1024), 16, 0, 0);
    }""")
rep("      if (grp == 1 - TAP) issue_a(chunk_idx + 1);", "      if (grp == 1 - TAP && !(abl & 2)) issue_a(chunk_idx + 1);")
# LDS reads
rep("""#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const char* ptr = sa + a_pre[TAP][i];
        fa[0][i] = *(const f16x8*)(ptr);
        fa[1][i] = *(const f16x8*)(ptr + 512);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        fb[0][j] = *(const f16x8*)(sb + b_off[j]);
        fb[1][j] = *(const f16x8*)(sb + b_off[j] + 512);
      }""", """      if (!(abl & 4) || u < 2) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const char* ptr = sa + a_pre[TAP][i];
        fa[0][i] = *(const f16x8*)(ptr);
        fa[1][i] = *(const f16x8*)(ptr + 512);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        fb[0][j] = *(const f16x8*)(sb + b_off[j]);
        fb[1][j] = *(const f16x8*)(sb + b_off[j] + 512);
      }
      }""")
# barriers inside step
rep("""    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---------------- M(u) ----------------""", """    __builtin_amdgcn_sched_barrier(0);
    if (!(abl & 16)) __builtin_amdgcn_s_barrier();
    // ---------------- M(u) ----------------""")
rep("""      wait_tail(u + 1 + grp);
    }
    __builtin_amdgcn_s_barrier();
  };""", """      wait_tail(u + 1 + grp);
    }
    if (!(abl & 16)) __builtin_amdgcn_s_barrier();
  };""")
# MFMAs
rep("""    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[ks][i], fb[ks][j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);""", """    if (!(abl & 32)) __builtin_amdgcn_s_setprio(1);
    if (!(abl & 8)) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[ks][i], fb[ks][j], acc[i][j], 0, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < TM; ++i) asm volatile("" :: "v"(fa[0][i]), "v"(fa[1][i]));
#pragma unroll
      for (int j = 0; j < TN; ++j) asm volatile("" :: "v"(fb[0][j]), "v"(fb[1][j]));
    }
    if (!(abl & 32)) __builtin_amdgcn_s_setprio(0);""")
rep("""  int ci = 0;
  for (; ci + 1 < nchunks; ++ci) { HALO_CHUNK(std::true_type, ci) }
  { HALO_CHUNK(std::false_type, ci) }""", """  int ci = 0;
  if (!(abl & 64)) {
  for (; ci + 1 < nchunks; ++ci) { HALO_CHUNK(std::true_type, ci) }
  { HALO_CHUNK(std::false_type, ci) }
  } else { wait_vmcnt_v5<0>(); }""")
rep("  conv_epilogue<BM, BN, WAVES_M, WAVES_N, TM, TN>(p, acc, (f16*)smem, m0, n0, mt, tid, lane, wm, wn);\n}",
    "  if (!(abl & 128)) conv_epilogue<BM, BN, WAVES_M, WAVES_N, TM, TN>(p, acc, (f16*)smem, m0, n0, mt, tid, lane, wm, wn);\n  else if (acc[0][0][0] == 123.f) p.y[0] = (f16)acc[1][1][3];\n}")
rep("hipLaunchKernelGGL((conv_igemm_halo_kernel<3>), dim3(p.mtiles * p.ntiles), dim3(512), 0, (hipStream_t)stream, p, hrp);",
    "hipLaunchKernelGGL((conv_igemm_halo_kernel<3>), dim3(p.mtiles * p.ntiles), dim3(512), 0, (hipStream_t)stream, p, hrp, g_abl);")
rep("hipLaunchKernelGGL((conv_igemm_halo_kernel<4>), dim3(p.mtiles * p.ntiles), dim3(512), 0, (hipStream_t)stream, p, hrp);",
    "hipLaunchKernelGGL((conv_igemm_halo_kernel<4>), dim3(p.mtiles * p.ntiles), dim3(512), 0, (hipStream_t)stream, p, hrp, g_abl);")
rep("static int g_halo_auto = 1;", "static int g_abl = 0;\nextern \"C\" void tscd_conv_set_abl(int v) { g_abl = v; }\nstatic int g_halo_auto = 1;")
src = src.replace("  ConvParams p;", "  ConvParams p; p.abl = g_abl;")
open(os.path.join(out, "conv_abl.hip"), "w").write(head + src)
import subprocess
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wl,-Bsymbolic", "-Wno-unused-result", "-I",
                       os.path.join(root, "tscd_amd", "csrc"), "-shared", os.path.join(out, "conv_abl.hip"), "-o", os.path.join(out, "libabl.so")])
print(os.path.join(out, "libabl.so"))
