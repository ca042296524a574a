#!/usr/bin/env python3
"""Register / scratch metadata of every gfx950 kernel in libsurfd_hip.so, read from the code objects' ELF notes
(llvm-objdump --offloading + llvm-readelf --notes; no GPU needed).   python tools/kernel_regs.py [filter]"""
import os, re, shutil, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
FIELDS = (".vgpr_count", ".agpr_count", ".sgpr_count", ".vgpr_spill_count", ".sgpr_spill_count", ".private_segment_fixed_size",
          ".group_segment_fixed_size")


def kernel_metadata(lib=None):
    """-> {demangled kernel name: {field: int}} for every kernel of the library's gfx950 code objects."""
    lib = lib or os.path.join(ROOT, "surfd_amd", "lib", "libsurfd_hip.so")
    out = {}
    with tempfile.TemporaryDirectory() as td:
        loc = os.path.join(td, "lib.so")
        shutil.copy(lib, loc)
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", loc], check=True, capture_output=True, cwd=td)
        for f in sorted(os.listdir(td)):
            if "gfx950" not in f:
                continue
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", os.path.join(td, f)], check=True, capture_output=True, text=True).stdout
            for blk in notes.split("  - .agpr_count:")[1:]:
                blk = ".agpr_count:" + blk
                m = re.search(r"\.name:\s+(\S+)", blk)
                if not m:
                    continue
                name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
                out[name] = {k: int(re.search(re.escape(k) + r":\s+(\d+)", blk).group(1)) for k in FIELDS if re.search(re.escape(k) + r":\s+(\d+)", blk)}
    return out


if __name__ == "__main__":
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    md = kernel_metadata()
    print(f"{'kernel':70s} vgpr agpr sgpr vspill sspill scratchB ldsB")
    for k in sorted(md):
        if flt in k:
            v = md[k]
            short = re.sub(r"\(.*", "", k)[:70]
            print(f"{short:70s} {v.get('.vgpr_count', 0):4d} {v.get('.agpr_count', 0):4d} {v.get('.sgpr_count', 0):4d} {v.get('.vgpr_spill_count', 0):6d} "
                  f"{v.get('.sgpr_spill_count', 0):6d} {v.get('.private_segment_fixed_size', 0):8d} {v.get('.group_segment_fixed_size', 0):5d}")
