"""Build libdinodet.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

    python -m dinov2_od_amd._build [--force] [--tuning]

hipcc cross-compiles without a GPU.  Objects go to build/obj, the library to
dinov2_od_amd/lib/libdinodet.so (git-ignored, but it travels with the gpurun snapshot).
--tuning: the -DDINODET_TUNING build (in-kernel time stamps, register-only MFMA probes, tile / schedule overrides through
DINODET_* variables: include/dinodet_tuning.h) -> lib/libdinodet_tuning.so, objects in build/obj_tuning; tools/ load it with
DINODET_LIB=dinov2_od_amd/lib/libdinodet_tuning.so.  The release library has none of that.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libdinodet.so")
OBJ = os.path.join(ROOT, "build", "obj")
SOURCES = ["dod_api.hip", "gemm_bf16.hip", "gemm_f32.hip", "attn_bf16.hip", "attn_f32.hip", "rowops.hip", "deform.hip",
           "postproc.hip", "matchcost.hip", "gemm_fp8.hip", "attn_x3.hip", "gemm_x3.hip", "preproc.hip", "attn_f32m.hip", "gemm_pp.hip", "dec_train.hip", "patch_embed.hip"]
HEADERS = [os.path.join(CSRC, "dod_common.h"), os.path.join(CSRC, "gemm_epi.h"), os.path.join(ROOT, "include", "dinodet.h"),
           os.path.join(ROOT, "include", "dinodet_tuning.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-I" + os.path.join(ROOT, "include")]


# per-source extra flags.  The flash attention kernels are built WITHOUT SLP vectorisation: hipcc packs adjacent fp32 adds / multiplies /
# fmas of the softmax into v_pk_*_f32, and on gfx950 packed fp32 does not overlap an executing MFMA (it takes the sum of the two times,
# tools/probes/mfma_valu_overlap.hip, profiles/r04_mfma_valu_overlap_probe.txt) while plain VALU does.
EXTRA = {"attn_bf16.hip": ["-fno-slp-vectorize"], "attn_x3.hip": ["-fno-slp-vectorize"]}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _digest(paths):
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _compile(src, tuning=False):
    s = os.path.join(CSRC, src)
    o = os.path.join(OBJ + ("_tuning" if tuning else ""), src.replace(".hip", ".o"))
    stamp = o + ".sha"
    extra = EXTRA.get(src, []) + (["-DDINODET_TUNING"] if tuning else [])
    d = _digest([s] + HEADERS) + "|" + " ".join(extra)
    if os.path.exists(o) and os.path.exists(stamp) and open(stamp).read() == d:
        return o, False
    cmd = [_hipcc()] + FLAGS + extra + ["-c", s, "-o", o]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    with open(stamp, "w") as f:
        f.write(d)
    return o, True


def build(force=False, verbose=True, tuning=False):
    obj = OBJ + ("_tuning" if tuning else "")
    LIB = os.path.join(LIBDIR, "libdinodet_tuning.so" if tuning else "libdinodet.so")
    os.makedirs(obj, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    if force:
        for f in os.listdir(obj):
            os.remove(os.path.join(obj, f))
    with ThreadPoolExecutor(max_workers=min(6, len(SOURCES))) as ex:
        res = list(ex.map(lambda src: _compile(src, tuning), SOURCES))
    objs = [o for o, _ in res]
    if any(ch for _, ch in res) or not os.path.exists(LIB):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"built {LIB}")
    elif verbose:
        print(f"{LIB} up to date")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, tuning="--tuning" in sys.argv)
