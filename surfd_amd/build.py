"""Builds libsurfd_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

In-tree build: objects under surfd_amd/lib/obj/, library at surfd_amd/lib/libsurfd_hip.so
(git-ignored, but shipped to the GPU box with the working tree).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libsurfd_hip.so")
SOURCES = ["core.hip", "decoder.hip", "grid.hip", "unet.hip", "conv_f16x2.hip", "sampler.hip", "xattn.hip", "mcubes.cpp"]   # .cpp = host-only code
# -ffp-contract=off: HIP's __fmul_rn/__fadd_rn are plain operators, so with the default
# "fast" contraction the compiler would fuse the separately rounded steps that mirror torch's
# fp32 op sequence (grid coordinates, posterior updates) into FMAs.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-pass-failed", "-ffp-contract=off"]
FLAGS += os.environ.get("SURFD_EXTRA_HIPCC_FLAGS", "").split()      # debugging builds only (e.g. -DSURFD_DEC_STAMPS)


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (needed to build libsurfd_hip.so for gfx950)")


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = True) -> str:
    hipcc = _hipcc()
    os.makedirs(os.path.join(LIBDIR, "obj"), exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "surfd_hip.h"))
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    jobs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(LIBDIR, "obj", os.path.splitext(s)[0] + ".o")
        if force or _newer(obj, [src] + headers):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        if src.endswith(".cpp"):      # host-only translation unit: no offload
            cmd = [hipcc, "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-x", "c++", "-c", src, "-o", obj]
        else:
            cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr[-4000:]}")
        if verbose:
            print(f"[surfd_amd.build] compiled {os.path.basename(src)}", file=sys.stderr)

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        list(ex.map(compile_one, jobs))
    objs = [os.path.join(LIBDIR, "obj", os.path.splitext(s)[0] + ".o") for s in srcs]
    if force or jobs or _newer(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
        if verbose:
            print(f"[surfd_amd.build] linked {LIB}", file=sys.stderr)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv))
