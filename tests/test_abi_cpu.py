"""CPU-only checks of the C-ABI boundary: the library builds/loads without a GPU, exports every
symbol include/surfd_hip.h declares, enumerates the reference checkpoint layouts, and reports
errors through return codes (never by falling back to a CPU computation)."""
import ctypes as C
import os
import re

import pytest
import torch

from surfd_amd import _native as N
from surfd_amd.spec import DecoderConfig, UNetConfig, decoder_param_spec, unet_param_spec

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(N.LIB_PATH):
        from surfd_amd.build import build_library
        build_library()
    return N.lib()


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "surfd_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(surfd_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(lib):
    declared = _declared_symbols()
    assert len(declared) >= 35
    raw = C.CDLL(N.LIB_PATH)
    for sym in declared:
        assert hasattr(raw, sym), f"{sym} declared in surfd_hip.h but not exported"
    assert set(N.EXPORTED_SYMBOLS) == set(declared), set(N.EXPORTED_SYMBOLS) ^ set(declared)


def test_integration_appendix_lists_every_export():
    """INTEGRATION.md's appendix (tools/abi_table.py) names, for every export, the reference interface it stands for and the
    module that binds it; an export added to the header without regenerating the appendix fails here."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    appendix = doc[doc.index("## Appendix: every export"):]
    missing = [s for s in _declared_symbols() if f"| `{s}` |" not in appendix]
    assert not missing, missing


def test_version_and_no_device(lib):
    assert lib.surfd_abi_version() == 1
    assert lib.surfd_device_count() >= 0


# the experiment macros of the kernel sources and the values the PRODUCT build carries: a library built with
# anything else — SURFD_EXTRA_HIPCC_FLAGS, a variant copied over the product path — fails here, not in a 1e-3 error weeks later
SHIPPED_BUILD = {"C2_GNW": "1", "C2_GNPAD": "4", "C2_PFN": "1", "C2_PFN_FORMS": "1", "C2_PFN_N": "1", "C2_KAPF": "1",
                 "C2_FAST_RCP": "1", "C2_EPI_LATE": "1", "C2_LAT_D": "2", "C2_DEEP_D": "3", "C2_ZB": "4", "C2_DIST": "1", "C2_LEAN_WAVES": "3", "C2_LEAN_U": "2",
                 "C2_PLANE_LEAN": "11264", "C2_ABLATE": "0", "C2_DBG_POISON": "0", "C2_PROBE": "0", "C2_STAMPS": "0",
                 "DEC_OVL": "0", "DEC_MIX": "1", "DEC_GRAD_W": "2", "DEC_GRAD_MIX": "0", "DEC_WS_AHEAD": "0", "DEC_REQ_EARLY": "2",
                 "DEC_FWD_STAGED": "0", "DEC_XCD_STAGGER": "0", "DEC_CLOCK": "1", "DEC_W_NT": "0", "DEC_STAMPS": "0"}


def test_shipped_library_is_built_with_the_default_configuration(lib):
    cfg = dict(kv.split("=", 1) for kv in lib.surfd_build_config().decode().split())
    assert cfg.pop("abi") == "1"
    assert cfg.pop("unsafe_variants") == "0", cfg
    bpipe = cfg.pop("C2_BPIPE")
    assert bpipe in ("0", "1")                       # the operand-pipelined K loop: either is a validated configuration
    assert cfg == SHIPPED_BUILD, {k: (cfg.get(k), SHIPPED_BUILD.get(k)) for k in set(cfg) | set(SHIPPED_BUILD) if cfg.get(k) != SHIPPED_BUILD.get(k)}


@pytest.mark.parametrize("flag", ["-DSURFD_C2_GNW=0", "-DSURFD_C2_GNPAD=0", "-DSURFD_C2_LAT_D=3", "-DSURFD_C2_ABLATE=1", "-DSURFD_C2_BPIPE=2", "-DSURFD_C2_PROBE"])
def test_unsafe_variants_do_not_compile_without_the_override(flag):
    """The fence itself: the preprocessor stops a build that selects a variant recorded as wrong / not bit-stable (checked with
    the preprocessor alone, -E: seconds, no code generation), and -DSURFD_ALLOW_UNSAFE_VARIANTS lifts it."""
    import subprocess
    from surfd_amd import build as B
    src = os.path.join(B.CSRC, "conv_f16x2.hip")
    cmd = [B._hipcc(), "--offload-arch=gfx950", "-std=c++17", "--cuda-device-only", "-E", "-o", os.devnull, src, flag]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode != 0 and "SURFD_ALLOW_UNSAFE_VARIANTS" in r.stderr, r.stderr[-600:]
    r = subprocess.run(cmd + ["-DSURFD_ALLOW_UNSAFE_VARIANTS"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-600:]


def _unet_cfg(ncls=0, mult=(1, 2, 4, 4)):
    return N.UNetCfg(1, 224, 1, 2, len(mult), (C.c_int * 8)(*mult), 3, (C.c_int * 8)(4, 2, 1), 8, 512, ncls)


@pytest.mark.parametrize("ncls", [0, 9])
def test_unet_plan_matches_checkpoint_layout(lib, ncls):
    h = C.c_void_p()
    N.check(lib.surfd_unet_create(C.byref(_unet_cfg(ncls)), C.byref(h)))
    got = []
    for i in range(lib.surfd_unet_num_params(h)):
        key, shp, nd = C.c_char_p(), (C.c_int64 * 4)(), C.c_int()
        N.check(lib.surfd_unet_param_info(h, i, C.byref(key), shp, C.byref(nd)))
        got.append(("Unet." + key.value.decode(), tuple(shp[:nd.value])))
    assert got == unet_param_spec(UNetConfig(num_classes=ncls or None))
    lib.surfd_unet_destroy(h)


@pytest.mark.parametrize("D", [32, 64])
def test_decoder_layout(lib, D):
    h = C.c_void_p()
    N.check(lib.surfd_decoder_create(63, D, 512, 5, C.byref(h)))
    got = []
    for i in range(lib.surfd_decoder_num_params(h)):
        key, shp, nd = C.c_char_p(), (C.c_int64 * 4)(), C.c_int()
        N.check(lib.surfd_decoder_param_info(h, i, C.byref(key), shp, C.byref(nd)))
        got.append((key.value.decode(), tuple(shp[:nd.value])))
    assert got == decoder_param_spec(DecoderConfig(latent_dim=D))
    lib.surfd_decoder_destroy(h)


def test_errors_are_return_codes(lib):
    h = C.c_void_p()
    rc = lib.surfd_decoder_create(63, 32, 256, 5, C.byref(h))          # hidden_dim the kernels are not built for
    assert rc == -4 and b"512" in lib.surfd_last_error()
    rc = lib.surfd_grid_create(100, C.byref(h))                          # not a power of two
    assert rc == -4
    bad = _unet_cfg()
    bad.model_channels = 100
    assert lib.surfd_unet_create(C.byref(bad), C.byref(h)) == -4
    with pytest.raises(RuntimeError, match="libsurfd_hip error"):
        N.check(lib.surfd_unet_create(None, C.byref(h)))
    # a grid handle without thresholds refuses to run (state error), nothing is computed on the host
    g = C.c_void_p()
    N.check(lib.surfd_grid_create(64, C.byref(g)))
    assert lib.surfd_grid_begin(g, None, None, None) == -2
    lib.surfd_grid_destroy(g)


def test_product_modules_refuse_cpu():
    """No CPU fallback: the drop-in modules raise on CPU tensors instead of computing."""
    from surfd_amd.cbndec import CbnDecoder, CoordsEncoder
    from surfd_amd.mdm import MDM
    dec = CbnDecoder(63, 32, 512, 5)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        dec(CoordsEncoder().encode(torch.zeros(1, 4, 3)), torch.zeros(1, 32))
    m = MDM(cond_mode="no_cond", unet_cfg=UNetConfig(channel_mult=(1,), num_res_blocks=1, attention_resolutions=()))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 1, 32), torch.zeros(1, dtype=torch.long), y={})


def test_no_product_import_of_oracle():
    pkg = os.path.join(ROOT, "surfd_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
