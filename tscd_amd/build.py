"""Builds tscd_amd/lib/libtscd_hip.so (hipcc, gfx950) from tscd_amd/csrc/*.hip -- in-tree, no JIT cache."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libtscd_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", 
         "-Wno-unused-result", "-I", CSRC]


HOST_SRCS = sorted(glob.glob(os.path.join(HERE, "csrc_host", "*.cpp")))      # lap.cpp (assignment solver), cocoeval.cpp (COCO bbox AP)
HOST_LIB = os.path.join(LIBDIR, "libtscd_host.so")


def build_host(force=False, verbose=True):
    """g++ build of the host-side code (no -march flags: SIMD variants are selected at run time)."""
    if not force and os.path.exists(HOST_LIB) and os.path.getmtime(HOST_LIB) > max(os.path.getmtime(s) for s in HOST_SRCS):
        return HOST_LIB
    os.makedirs(LIBDIR, exist_ok=True)
    cmd = [os.environ.get("CXX", "g++"), "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-fast-math", "-pthread"] + HOST_SRCS + ["-o", HOST_LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return HOST_LIB


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    srcs = glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h"))
    return any(os.path.getmtime(s) > t for s in srcs)


def build(force=False, verbose=True):
    build_host(force, verbose)
    if not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    objs = []
    procs = []
    for src in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
        obj = os.path.join(LIBDIR, os.path.basename(src)[:-4] + ".o")
        hdr_t = max(os.path.getmtime(h) for h in glob.glob(os.path.join(CSRC, "*.h")))
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), hdr_t):
            continue
        cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
