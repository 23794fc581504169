"""TEST INFRASTRUCTURE -- builds oracle/_ref/ from the reference's OWN native source, where it lies under /root/reference.

The reference's only native component on the evaluation path is its fast COCO evaluator (yolox/layers/cocoeval/cocoeval.cpp + .h,
SURVEY.md F2: plain pybind11 C++, no torch, no generated code): COCOevalEvaluateImages (per-image detection <-> ground-truth matching)
and COCOevalAccumulate (precision / recall curves).  It compiles from those two files with g++ and the pybind11 headers of this image:

    python -m oracle.build_ref        ->  oracle/_ref/fast_cocoeval_ref<ext suffix>.so

Nothing is copied: the source is compiled in place and only the shared object lands in oracle/_ref/ (git-ignored, NOT gpurun-ignored:
it travels to the GPU box like the product's own built libraries).  tests/test_cocoeval_ref_pin.py pins the product's C++ evaluator
(tscd_amd/csrc_host/cocoeval.cpp) and the plain-Python restatement (oracle/cocoeval_ref.py) against it.  The prebuilt
yolox/layers/fast_cocoeval.cpython-37m-*.so that ships inside the reference is never loaded."""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = "/root/reference/yolox/layers/cocoeval/cocoeval.cpp"
OUT_DIR = os.path.join(HERE, "_ref")
MOD = "fast_cocoeval_ref"


def out_path():
    return os.path.join(OUT_DIR, MOD + sysconfig.get_config_var("EXT_SUFFIX"))


def build(force=False, verbose=True):
    """Returns the path of the built module, or None when the reference tree is not present (the GPU box: prebuilt file travels)."""
    out = out_path()
    if not os.path.isfile(REF_SRC):
        return out if os.path.isfile(out) else None
    if not force and os.path.isfile(out) and os.path.getmtime(out) > os.path.getmtime(REF_SRC):
        return out
    import pybind11
    os.makedirs(OUT_DIR, exist_ok=True)
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-shared", "-fPIC", "-fvisibility=hidden", f"-DTORCH_EXTENSION_NAME={MOD}",
           "-I", pybind11.get_include(), "-I", sysconfig.get_paths()["include"], "-I", os.path.dirname(REF_SRC), REF_SRC, "-o", out]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


def load():
    """The compiled reference module, or None when it has not been built."""
    import importlib.util
    out = out_path()
    if not os.path.isfile(out):
        return None
    spec = importlib.util.spec_from_file_location(MOD, out)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
