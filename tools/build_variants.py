#!/usr/bin/env python3
"""Builds differently compiled copies of libsurfd_hip.so for A/B timing on the GPU box (selected with SURFD_LIB=...):
    python tools/build_variants.py name=decoder.hip:-DFLAG[,-DFLAG2] [name2=file@gitrev] ...
`file:-Dflags` recompiles that one source with extra flags; `file@rev` takes the source text from a git revision;
`file+path[:-Dflags]` takes it from another file (an experiment that is not to touch the tree's source).
Every other object is shared with the regular build.  Output: surfd_amd/lib/variants/libsurfd_hip_<name>.so"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from surfd_amd import build as B

B.build_library(verbose=False)
out_dir = os.path.join(B.LIBDIR, "variants")
os.makedirs(out_dir, exist_ok=True)
for spec in sys.argv[1:]:
    name, rest = spec.split("=", 1)
    flags, rev, alt = [], None, None
    if "+" in rest:
        src_name, tail = rest.split("+", 1)
        alt, _, fl = tail.partition(":")
        flags = fl.split(",") if fl else []
    elif "@" in rest:
        src_name, rev = rest.split("@", 1)
    elif ":" in rest:
        src_name, fl = rest.split(":", 1)
        flags = fl.split(",")
    else:
        src_name = rest
    src = os.path.join(B.CSRC, src_name)
    tmp = None
    if rev or alt:
        text = open(alt).read() if alt else subprocess.check_output(["git", "show", f"{rev}:surfd_amd/csrc/{src_name}"], cwd=ROOT, text=True)
        tmp = os.path.join(B.CSRC, f"_variant_{name}_{src_name}")       # next to its headers
        open(tmp, "w").write(text)
        src = tmp
    obj = os.path.join(out_dir, f"{name}_{os.path.splitext(src_name)[0]}.o")
    try:
        subprocess.check_call([B._hipcc()] + B.FLAGS + ["-DSURFD_ALLOW_UNSAFE_VARIANTS"] + flags + ["-c", src, "-o", obj])      # experiments: the fence of the unsafe variants is lifted here and only here
    finally:
        if tmp:
            os.remove(tmp)
    objs = [obj if s == src_name else os.path.join(B.LIBDIR, "obj", os.path.splitext(s)[0] + ".o") for s in B.SOURCES]
    lib = os.path.join(out_dir, f"libsurfd_hip_{name}.so")
    subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    print("built", lib)
