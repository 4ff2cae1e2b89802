"""Builds libstabletts_hip.so (gfx950) in-tree with hipcc.  No GPU needed (cross-compile).

    python -m stabletts_amd.build [--force]

Developer A/B builds: ST_BUILD_DEFS="-DST_STORE_WT=0" ST_BUILD_OUT=/path/variant.so python -m stabletts_amd.build
writes a second library (own object directory) that STABLETTS_HIP_LIB=/path/variant.so makes _lib.py load.
"""
import concurrent.futures as cf
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.environ.get("ST_BUILD_OUT") or os.path.join(HERE, "libstabletts_hip.so")
OBJ = os.path.join(HERE, "csrc", "build") if not os.environ.get("ST_BUILD_OUT") else LIB + ".obj"
SOURCES = ["engine.cpp", "conv_gemm2_bf16.hip", "conv_gemm2_f16.hip",
           "attention.hip", "misc_kernels.hip", "adaptive_ode.hip"]
HEADERS = ["common.h", "launch.h", "conv_gemm2_impl.h", "conv_gemm2_inst.h", os.path.join("..", "..", "include", "stabletts_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result"]
FLAGS += os.environ.get("ST_BUILD_DEFS", "").split()


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (needed to build libstabletts_hip.so for gfx950)")


def _digest():
    h = hashlib.sha256()
    for f in SOURCES + HEADERS:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    """Compile every translation unit and link the shared library. Returns its path."""
    stamp = os.path.join(OBJ, "stamp")
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dig:
        return LIB
    hipcc = _hipcc()
    os.makedirs(OBJ, exist_ok=True)

    def compile_one(src):
        obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
        cmd = [hipcc, *FLAGS, "-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with cf.ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 2)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    with open(stamp, "w") as fh:
        fh.write(dig)
    if verbose:
        print(f"built {LIB} ({os.path.getsize(LIB) / 1e6:.2f} MB)")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
