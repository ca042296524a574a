"""Register / scratch budget of the hot kernels, read from the gfx950 code objects inside libsurfd_hip.so (ELF notes; no
GPU needed).  VERDICT r2: the f16x2 decoder kernels spilled 132 / 371 registers; a spilled weight stage inside a GEMM is a
drain of the software-pipelined weight stream.  The forward kernel must not spill at all, the gradient kernel (both
accumulator sets + 11 layers of ReLU gates) stays within a bound, the reverse loop's conv kernel keeps two workgroups per CU."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def meta():
    spec = importlib.util.spec_from_file_location("kernel_regs", os.path.join(ROOT, "tools", "kernel_regs.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.kernel_metadata()


def _one(meta, prefix):
    hits = [v for k, v in meta.items() if k.startswith(prefix)]
    assert len(hits) == 1, (prefix, [k for k in meta if prefix.split("<")[0] in k])
    return hits[0]


def test_forward_decoder_kernel_does_not_spill(meta):
    k = _one(meta, "void surfd::decoder_kernel<false, true>")
    assert k[".vgpr_spill_count"] == 0 and k[".private_segment_fixed_size"] == 0 and k[".sgpr_spill_count"] == 0
    assert k[".vgpr_count"] <= 512


def test_eight_wave_forward_kernel_fits_two_waves_per_simd(meta):
    """The default forward kernel: 512 threads, two waves per SIMD -> at most 256 registers per lane.  Its few spilled
    registers are scalars of the point-source descriptor parked over the whole layer loop (SGPR pressure), stored once per
    kernel and reloaded in the fetch / output phases of a tile — there is no scratch access inside the block loop (checked on
    the ISA when the kernel was written; here: the bound)."""
    k = _one(meta, "surfd::decoder_fwd8_kernel")
    assert k[".vgpr_count"] + k.get(".agpr_count", 0) <= 256
    assert k[".vgpr_spill_count"] <= 16 and k[".private_segment_fixed_size"] <= 64          # 13 / 56 B since the fetch / encode phases form their addresses in place


def test_gradient_decoder_kernel_spill_bound(meta):
    k = _one(meta, "void surfd::decoder_kernel<true, true>")
    # round 2: 371, mid round 3: 101.  What is left: 16 accumulator registers parked across the first epilogue of a tile and
    # a dozen per-tile scalars; the 44 words of ReLU gates live in scratch by design (indexed by the layer loop) and are not spills
    assert k[".vgpr_spill_count"] <= 40
    assert k[".private_segment_fixed_size"] <= 384


def test_conv_kernels_keep_two_workgroups_per_cu(meta):
    for name in ("void surfd::conv2_kernel<8, false, false, false, false, false>", "void surfd::conv2_kernel<8, false, true, false, false, false>"):
        k = _one(meta, name)
        assert k[".vgpr_spill_count"] == 0 and k[".private_segment_fixed_size"] == 0
        assert k[".vgpr_count"] + k.get(".agpr_count", 0) <= 256          # 2 waves per SIMD


def test_lean_conv_kernel_fits_three_workgroups_per_cu(meta):
    """The wide loops' default kernel (round 4): three 256-thread workgroups per CU need <= 168 registers per lane, no
    scratch (the build for four — 128 registers — spills 63 and measured 30 % slower), <= 112 SGPRs (the hardware admits
    floor(800 / (ceil(sgpr / 16) * 16 + 16)) blocks of 256 threads per CU)."""
    k = _one(meta, "void surfd::conv2_kernel<8, false, true, true, false, false>")
    # round 5 allowed ONE spilled register here (the word of the weight prefetch parked to the kernel's end — a private scratch
    # segment for every dispatch of the hot kernel); with the in-wave GroupNorm statistics of round 6 the registers the
    # LDS exchange needed are free and nothing spills
    assert k[".vgpr_spill_count"] == 0 and k[".private_segment_fixed_size"] == 0
    assert k[".vgpr_count"] + k.get(".agpr_count", 0) <= 168
    assert k[".sgpr_count"] <= 112


def test_two_column_tile_lean_kernel_fits_three_workgroups_per_cu(meta):
    """NT2 (two column tiles per wave on <= 128-channel K blocks): same budget as the one-tile lean form; the two registers it
    spills are outside the K loop."""
    k = _one(meta, "void surfd::conv2_kernel<8, false, true, true, false, true>")
    assert k[".vgpr_count"] + k.get(".agpr_count", 0) <= 168 and k[".vgpr_spill_count"] <= 4 and k[".private_segment_fixed_size"] <= 16
