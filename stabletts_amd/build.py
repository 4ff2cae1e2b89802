"""Builds libstabletts_hip.so (gfx950) in-tree with hipcc.  No GPU needed (cross-compile).

    python -m stabletts_amd.build [--force]

Developer A/B builds: ST_BUILD_DEFS="-DST_STORE_WT=0" ST_BUILD_OUT=/path/variant.so python -m stabletts_amd.build
writes a second library (own object directory) that STABLETTS_HIP_LIB=/path/variant.so makes _lib.py load.
"""
import concurrent.futures as cf
import hashlib
import json
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.environ.get("ST_BUILD_OUT") or os.path.join(HERE, "libstabletts_hip.so")
OBJ = os.path.join(HERE, "csrc", "build") if not os.environ.get("ST_BUILD_OUT") else LIB + ".obj"
SOURCES = ["engine.cpp", "engine_train.cpp", "engine_vocos.cpp", "vocos_kernels.hip", "conv_gemm2_bf16.hip", "conv_gemm2_f16.hip", "ffn_fused_bf16.hip", "ffn_fused_f16.hip", "ffn_wino_f16.hip", "qkv_ws.hip", "oproj_ws.hip",
           "attention.hip", "attention_bwd.hip", "train_kernels.hip", "wgrad_tn.hip", "misc_kernels.hip", "align_kernels.hip", "adaptive_ode.hip"]
HEADERS = ["common.h", "launch.h", "train_launch.h", "vocos_launch.h", "engine_internal.h", "conv_gemm2_impl.h", "conv_gemm_phased.h", "conv_gemm2_inst.h", "ffn_fused.h", "ffn_wino.h", os.path.join("..", "..", "include", "stabletts_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result",
         "-Rpass-analysis=kernel-resource-usage"]
# per-source flags: the attention kernels keep their fp32 row-sum adds scalar (common.h: add_f32_scalar)
# (the attention BACKWARD with the same flag: ~100 packed fp32 ops per key tile gone, 229 -> 164 VGPRs for dQ pass 2 -- and no change of the step,
#  18.87 vs 18.91 ms paired: profiles/r06_ab_attn_bwd_noslp_null.txt; not set)
EXTRA_FLAGS = {"attention.hip": ["-fno-slp-vectorize"]}
RESOURCES = os.path.join(OBJ, "kernel_resources.json")    # per-kernel VGPR / SGPR / scratch / occupancy of the last build
FLAGS += os.environ.get("ST_BUILD_DEFS", "").split()


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (needed to build libstabletts_hip.so for gfx950)")


def _parse_resources(stderr, table):
    """Collects hipcc's -Rpass-analysis=kernel-resource-usage remarks into `table` (demangled-ish kernel name ->
    dict) and returns the remaining compiler output (real warnings)."""
    import re
    rest, cur = [], None
    keys = {"TotalSGPRs": "sgpr", "VGPRs": "vgpr", "AGPRs": "agpr", "ScratchSize [bytes/lane]": "scratch",
            "Occupancy [waves/SIMD]": "occupancy", "VGPRs Spill": "vgpr_spill", "SGPRs Spill": "sgpr_spill",
            "LDS Size [bytes/block]": "lds_static"}
    lines = stderr.split("\n")
    i = 0
    while i < len(lines):
        ln = lines[i]
        if "[-Rpass-analysis=kernel-resource-usage]" in ln:
            m = re.search(r"remark:\s+Function Name: (\S+)", ln)
            if m:
                cur = table.setdefault(m.group(1), {})
                i += 3 if i + 2 < len(lines) and lines[i + 2].strip().startswith("|") else 1    # source excerpt + caret
                continue
            m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\S+)", ln)
            if m and cur is not None and m.group(1).strip() in keys:
                try:
                    cur[keys[m.group(1).strip()]] = int(m.group(2))
                except ValueError:
                    pass
            i += 1
            continue
        if ln.startswith("In file included from") or re.match(r"^\d+ (warning|remark)s? generated", ln.strip()):
            i += 1
            continue
        rest.append(ln)
        i += 1
    return "\n".join(l for l in rest if l.strip())


def _digest():
    h = hashlib.sha256()
    for f in SOURCES + HEADERS:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    h.update((" ".join(FLAGS) + repr(sorted(EXTRA_FLAGS.items()))).encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    """Compile every translation unit and link the shared library. Returns its path."""
    stamp = os.path.join(OBJ, "stamp")
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dig:
        return LIB
    hipcc = _hipcc()
    os.makedirs(OBJ, exist_ok=True)

    resources = {}

    def compile_one(src):
        obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
        cmd = [hipcc, *FLAGS, *EXTRA_FLAGS.get(src, []), "-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        rest = _parse_resources(r.stderr, resources)
        if verbose and rest.strip():
            print(rest, file=sys.stderr)
        return obj

    with cf.ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 2)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    with open(RESOURCES, "w") as fh:
        json.dump(resources, fh, indent=1, sort_keys=True)
    spilled = {k: v for k, v in resources.items() if v.get("scratch", 0) or v.get("vgpr_spill", 0)}
    if spilled:      # a kernel that touches scratch lost >= 20 % in every measurement of this project: refuse to ship it
        raise RuntimeError("kernels use scratch memory / spill registers (see " + RESOURCES + "): " + ", ".join(sorted(spilled)))
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
