#!/usr/bin/env python3
"""Writes tools/_ablate/conv_stamp.hip: conv_igemm_glds_kernel with s_memrealtime stamps (100 MHz constant clock) at four points of every
k-step -- after the fragment-read wait that opens the step, after the counted vmcnt wait, after the barrier, after the LDS-DMA issue --
summed per wave and written to a debug buffer at the end; builds tools/_ablate/libstamp.so (-Bsymbolic).  The stamps sit where no LDS
read is outstanding, so their own lgkmcnt(0) waits do not serialise anything that was not already waited for.
tools/conv_stamp_run.py prints where a step's time goes."""
import os, subprocess
here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(here)
out = os.path.join(here, "_ablate")
os.makedirs(out, exist_ok=True)
src = open(os.path.join(root, "tscd_amd", "csrc", "conv_igemm.hip")).read()
a = src.index("template <int BM, int BN, int WAVES_N>\n__global__ __launch_bounds__(128 * WAVES_N")
b = src.index("// Main-loop variant 3 (Cin % 64 == 0, Cout > 128)")
head, k, tail = src[:a], src[a:b], src[b:]
def rep(x, y):
    global k
    assert k.count(x) == 1, (x[:60], k.count(x))
    k = k.replace(x, y)
rep("""  auto phase = [&](auto setc, int kt) {
    constexpr int S = decltype(setc)::value;
    __builtin_amdgcn_sched_barrier(0);                       // keep the previous step's MFMAs ahead of the wait
    __builtin_amdgcn_s_waitcnt(0xC07F);                      // lgkmcnt(0): this wave's reads of tile kt are in registers ...
    wait_tile(kt + 1);
    __builtin_amdgcn_s_barrier();                            // ... everyone's are: tile kt's buffer may be refilled
    __builtin_amdgcn_sched_barrier(0);
    if (kt + NS < nk) issue((kt + NS) % NS);
    read_frags(std::integral_constant<int, 1 - S>{}, kt + 1);
    mfmas(setc);
  };""", """  unsigned long long tsum[4] = {0, 0, 0, 0}, tprev = 0;
  auto stamp = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
  };
  auto phase = [&](auto setc, int kt) {
    constexpr int S = decltype(setc)::value;
    __builtin_amdgcn_sched_barrier(0);                       // keep the previous step's MFMAs ahead of the wait
    __builtin_amdgcn_s_waitcnt(0xC07F);                      // lgkmcnt(0): this wave's reads of tile kt are in registers ...
    const unsigned long long t0 = stamp();
    if (tprev) tsum[3] += t0 - tprev;                        // DMA issue end -> next step start: reads, MFMAs, loop
    wait_tile(kt + 1);
    const unsigned long long t1 = stamp();
    __builtin_amdgcn_s_barrier();                            // ... everyone's are: tile kt's buffer may be refilled
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t2 = stamp();
    if (kt + NS < nk) issue((kt + NS) % NS);
    const unsigned long long t3 = stamp();
    tsum[0] += t1 - t0; tsum[1] += t2 - t1; tsum[2] += t3 - t2; tprev = t3;
    read_frags(std::integral_constant<int, 1 - S>{}, kt + 1);
    mfmas(setc);
  };""")
rep("  conv_epilogue<BM, BN, WAVES_M, WAVES_N, TM, TN>(p, acc, (f16*)smem, m0, n0, mt, tid, lane, wm, wn);\n}",
    """  const unsigned long long tend = stamp();
  conv_epilogue<BM, BN, WAVES_M, WAVES_N, TM, TN>(p, acc, (f16*)smem, m0, n0, mt, tid, lane, wm, wn);
  if (p.dbg && lane == 0) {
    unsigned long long* d = p.dbg + ((long)blockIdx.x * NW + wave) * 8;
    d[0] = tsum[0]; d[1] = tsum[1]; d[2] = tsum[2]; d[3] = tsum[3]; d[4] = tend - tstart; d[5] = stamp() - tend; d[6] = nk;
  }
}""")
rep("  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);\n  const int wm = wave / WAVES_N, wn = wave % WAVES_N;",
    "  const unsigned long long tstart = __builtin_amdgcn_s_memrealtime();\n  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);\n  const int wm = wave / WAVES_N, wn = wave % WAVES_N;")
head = head.replace("struct ConvParams {", "struct ConvParams {\n  unsigned long long* dbg;")
tail = tail.replace("static int g_halo_auto = 1;", "static unsigned long long* g_dbg = nullptr;\nextern \"C\" void tscd_conv_set_dbg(void* p) { g_dbg = (unsigned long long*)p; }\nstatic int g_halo_auto = 1;")
assert tail.count("  ConvParams p;") == 1
tail = tail.replace("  ConvParams p;", "  ConvParams p; p.dbg = g_dbg;")
open(os.path.join(out, "conv_stamp.hip"), "w").write(head + k + tail)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wl,-Bsymbolic", "-Wno-unused-result", "-I",
                       os.path.join(root, "tscd_amd", "csrc"), "-shared", os.path.join(out, "conv_stamp.hip"), "-o", os.path.join(out, "libstamp.so")])
print(os.path.join(out, "libstamp.so"))
