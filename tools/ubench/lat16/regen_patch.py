#!/usr/bin/env python3
"""Regenerates hook.patch against the current surfd_amd/csrc/conv_f16x2.hip (run after that file changes; tests/test_experiments_cpu.py
checks that the patch applies): the device helpers that conv2_dev.h holds are cut out of conv_f16x2.hip, the header is included, the
sixteen-wave form gets its early dispatch in launch_conv2, its declaration and its place in the build list."""
import os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
paths = ["surfd_amd/csrc/conv_f16x2.hip", "surfd_amd/csrc/unet_plan.h", "surfd_amd/build.py"]
assert not subprocess.run(["git", "status", "--porcelain", "--"] + paths, cwd=ROOT, capture_output=True, text=True).stdout, "commit the sources first"
src = open(os.path.join(ROOT, paths[0])).read()

def cut(text, start, end):
    i = text.index(start); j = text.index(end, i) + len(end)
    return text[:i] + text[j:]

new = src
new = cut(new, "typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));", "typedef float f32x2 __attribute__((ext_vector_type(2)));\n")
new = cut(new, "// SiLU of the operand staging: x * 1 / (1 + e^-x) with the hardware reciprocal", "    return v * __frcp_rn(1.f + __expf(-v));\n#endif\n}\n")
new = cut(new, "// Kernel-argument prefetch.  Conv2Args is 6-7 cache lines", ': "=&s"(d0), "=&s"(d1), "=&s"(d2), "=&s"(d3), "=&s"(d4), "=&s"(d5), "=&s"(d6), "=&s"(d7) : "s"(ka));\n#endif\n}\n')
new = cut(new, "// GroupNorm statistics in the wave (SURFD_C2_GNW = 1, the default since round 6).",
          '__device__ __forceinline__ void lds_bar() {\n    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");\n    __builtin_amdgcn_s_barrier();\n    asm volatile("" ::: "memory");\n}\n')
new = new.replace('#include "unet_plan.h"\n', '#include "unet_plan.h"\n#include "conv2_dev.h"\n', 1)
old = "    const bool wide = u->wide_batch > 0;\n"
assert new.count(old) == 1
new = new.replace(old, old + '''    // narrow loops (latency form): the sixteen-wave kernel (conv_lat16.hip) wherever it covers the layer; SURFD_CONV2_LAT16=0 keeps
    // the four-wave form everywhere (A/B timing, and the reference point of the tests that compare the two)
    static const int lat16_env = env_int("SURFD_CONV2_LAT16", 1);
    if (!wide && lat16_env) {
        const int rx = launch_conv2x(u, c, B, L, io, st);
        if (rx <= 0) return rx;
    }
''')
old = "    return SURFD_OK;\n}\n\n}  // namespace surfd\n"
assert new.count(old) == 1
new = new.replace(old, "    return conv2x_set_attributes();\n}\n\n}  // namespace surfd\n")
plan = open(os.path.join(ROOT, paths[1])).read()
decl = "int launch_conv2(surfd_unet *u, const ConvPlan &c, int B, int L, const ConvLaunchIO &io, hipStream_t st);\n"
assert plan.count(decl) == 1
plan2 = plan.replace(decl, decl + "// conv_lat16.hip: the sixteen-wave latency form of the same convolution (narrow loops); same return convention\n"
                     "int launch_conv2x(surfd_unet *u, const ConvPlan &c, int B, int L, const ConvLaunchIO &io, hipStream_t st);\nint conv2x_set_attributes();\n")
build = open(os.path.join(ROOT, paths[2])).read()
build2 = build.replace('"conv_f16x2.hip", "sampler.hip"', '"conv_f16x2.hip", "conv_lat16.hip", "sampler.hip"')
assert build2 != build
try:
    for p, t in zip(paths, (new, plan2, build2)):
        open(os.path.join(ROOT, p), "w").write(t)
    diff = subprocess.run(["git", "diff", "--"] + paths, cwd=ROOT, capture_output=True, text=True, check=True).stdout
finally:
    subprocess.run(["git", "checkout", "--"] + paths, cwd=ROOT, check=True)
open(os.path.join(ROOT, "tools", "ubench", "lat16", "hook.patch"), "w").write(diff)
subprocess.run(["git", "apply", "--check", "tools/ubench/lat16/hook.patch"], cwd=ROOT, check=True)
print("hook.patch regenerated:", diff.count("\n"), "lines")
