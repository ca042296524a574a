"""Native UDF marching cubes (surfd_amd/csrc/mcubes.cpp, host code) — SURVEY.md §8 f1.

Parity is BIT-EXACT (integer faces, identical float32 vertices):
  * against committed fixtures made by the reference's own compiled extension (tools/make_golden.py g13): counts and
    SHA-256 of the arrays for nine grids up to 256^3 (+ 512^3 counts), full arrays for the 48^3 ones;
  * against the reference extension itself (oracle/_ref, built by oracle/build_ref.py from the reference's .pyx where
    it lies) on more grids, when it is present — here always, on the GPU box if the snapshot carried it.
Plus mesh properties that need no oracle, the argument contract of udf_mc_lewiner, and the case tables."""
import hashlib
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import mc_fields  # noqa: E402

from surfd_amd import mcubes  # noqa: E402


def _sha(a, dt):
    return hashlib.sha256(np.ascontiguousarray(a, dt).tobytes()).hexdigest()


CASES = [("two_spheres", 48), ("two_spheres", 96), ("open_sheet", 48), ("open_sheet", 96), ("noisy_blob", 48),
         ("noisy_blob", 96), ("thin_shell", 64), ("thin_shell", 128)]


@pytest.mark.parametrize("name,N", CASES)
def test_bit_exact_vs_reference_fixtures(golden, name, N):
    g = golden("g13_marching_cubes")
    udf, grads = mc_fields.FIELDS[name](N)
    tag = f"{name}_{N}"
    if hashlib.sha256(udf.tobytes() + grads.tobytes()).hexdigest() != str(g[tag + "_input_sha256"]):
        pytest.skip("this host's libm rounds the synthetic grid differently from the fixture's (the extension-based test covers it)")
    v, f, n, val = mcubes.udf_mc_lewiner(udf, grads)
    assert (len(v), len(f)) == (int(g[tag + "_nv"]), int(g[tag + "_nf"]))
    assert _sha(f, np.int32) == str(g[tag + "_faces_sha256"])
    assert _sha(v, np.float32) == str(g[tag + "_verts_sha256"])
    if tag + "_verts" in g.files:
        np.testing.assert_array_equal(v, g[tag + "_verts"])
        np.testing.assert_array_equal(f, g[tag + "_faces"])
        np.testing.assert_array_equal(n, g[tag + "_normals"])


def test_thin_shell_256_counts_of_the_survey(golden):
    """55 756 vertices / 110 899 faces at 256^3 (and the arrays' hashes)."""
    g = golden("g13_marching_cubes")
    udf, grads = mc_fields.thin_shell(256)
    v, f, _, _ = mcubes.udf_mc_lewiner(udf, grads)
    assert (len(v), len(f)) == (55756, 110899) == (int(g["thin_shell_256_nv"]), int(g["thin_shell_256_nf"]))
    if hashlib.sha256(udf.tobytes() + grads.tobytes()).hexdigest() != str(g["thin_shell_256_input_sha256"]):
        return                     # another CPU's sqrt: counts are the portable part
    assert _sha(f, np.int32) == str(g["thin_shell_256_faces_sha256"]) and _sha(v, np.float32) == str(g["thin_shell_256_verts_sha256"])


@pytest.fixture(scope="module")
def reference():
    from oracle import build_ref
    cy = build_ref.load()
    if cy is None:
        pytest.skip("oracle/_ref not built and no reference tree to build it from")
    return build_ref, cy, mcubes.lut_tables()


@pytest.mark.parametrize("name,N,seed", [("noisy_blob", 40, 1), ("noisy_blob", 56, 2), ("noisy_blob", 72, 3), ("two_spheres", 33, 0),
                                         ("open_sheet", 57, 0)])
def test_bit_exact_vs_reference_extension(reference, name, N, seed):
    build_ref, cy, tables = reference
    udf, grads = mc_fields.noisy_blob(N, seed) if name == "noisy_blob" else mc_fields.FIELDS[name](N)
    rv, rf, rn, rval = build_ref.reference_udf_mc(cy, tables, udf, grads)
    v, f, n, val = mcubes.udf_mc_lewiner(udf, grads)
    np.testing.assert_array_equal(f, rf)
    np.testing.assert_array_equal(v, rv)
    np.testing.assert_array_equal(n, rn)
    np.testing.assert_array_equal(val, rval)


def test_mesh_properties_closed_surface():
    udf, grads = mc_fields.two_spheres(64)
    v, f, n, val = mcubes.udf_mc_lewiner(udf, grads, spacing=[2.0 / 63] * 3)
    assert v.dtype == np.float64 and f.dtype == np.int32          # spacing multiply is float64, as the reference's
    assert f.min() == 0 and f.max() == len(v) - 1
    p = v - 1.0                                                   # coords_range[0] shift of get_mesh_from_udf
    # (vertex columns follow the volume's axes: axis 0, 1, 2 = the field's x, y, z)
    d = np.minimum(np.abs(np.linalg.norm(p - np.array([-0.2, 0.0, 0.05]), axis=1) - 0.45),
                   np.abs(np.linalg.norm(p - np.array([0.25, 0.1, -0.05]), axis=1) - 0.4))
    assert d.max() < 2.0 / 63                                     # every vertex within one voxel of the true surface
    # closed surface: every undirected edge belongs to exactly two triangles
    e = np.sort(np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]), axis=1)
    _, counts = np.unique(e, axis=0, return_counts=True)
    assert (counts == 2).mean() > 0.999
    np.testing.assert_allclose(np.linalg.norm(n, axis=1)[np.linalg.norm(n, axis=1) > 0], 1.0, atol=1e-5)


def test_argument_contract():
    udf, grads = mc_fields.two_spheres(16)
    with pytest.raises(ValueError):
        mcubes.udf_mc_lewiner(udf[0], grads)
    with pytest.raises(ValueError):
        mcubes.udf_mc_lewiner(udf, grads, spacing=(1, 1))
    with pytest.raises(ValueError):
        mcubes.udf_mc_lewiner(udf, grads, step_size=0)
    with pytest.raises(RuntimeError, match="No surface"):
        mcubes.udf_mc_lewiner(np.full((8, 8, 8), 0.1, np.float32), np.zeros((8, 8, 8, 3), np.float32))
    v1, f1, _, _ = mcubes.udf_mc_lewiner(udf, grads, gradient_direction="ascent")
    v2, f2, _, _ = mcubes.udf_mc_lewiner(udf, grads)
    np.testing.assert_array_equal(f1, np.fliplr(f2))
    np.testing.assert_array_equal(v1, v2)


def test_case_tables_are_lewiners(golden):
    """The case tables compiled into the library ARE the reference's: SHA-256 and shape of every table against the
    hashes tools/make_golden.py took from /root/reference/meshudf/_marching_cubes_lewiner_luts.py (and of the three
    edge-corner tables of _marching_cubes_lewiner.py:171-224) — holds on any host, also where oracle/_ref is driven
    with the library's own tables.  Plus a few structural invariants."""
    import hashlib
    t = mcubes.lut_tables()
    g = golden("g13_lut_sha256")
    names = sorted(k[:-7] for k in g.files if k.endswith("_sha256"))
    assert sorted(t) == names and len(t) == 51
    for name in names:
        assert tuple(int(v) for v in g[name + "_shape"]) == t[name].shape, name
        assert hashlib.sha256(np.ascontiguousarray(t[name], np.int8).tobytes()).hexdigest() == str(g[name + "_sha256"]), name
    assert t["CASES"].shape == (256, 2) and t["TILING13_3"].shape == (2, 12, 30) and t["SUBCONFIG13"].shape == (64,)
    assert t["CASES"][0, 0] == 0 and t["CASES"][255, 0] == 0
    for name, ar in t.items():
        if name.startswith("TILING"):
            assert ar.min() >= 0 and ar.max() <= 12, name      # every fan entry is an edge 0..11 or the interior vertex 12
    # a sign pattern and its complement are the same Lewiner case
    assert all(t["CASES"][i, 0] == t["CASES"][255 - i, 0] for i in range(256))


@pytest.mark.parametrize("name,n", [("two_spheres", 48), ("open_sheet", 48), ("noisy_blob", 48), ("thin_shell", 64), ("noisy_blob", 96)])
def test_band_mesher_equals_dense_mesher(name, n):
    """f1's sparse hand-off, host half: the mesher fed ONLY the voxels with udf <= 1.74 voxel (index, value, gradient —
    what the device compacts) gives the mesh of the dense volumes bit for bit, shape after shape on one scratch volume
    (which must come back clean), also where the band's order is not the voxel order."""
    udf, grads = mc_fields.FIELDS[name](n)
    v0, f0, n0, val0 = mcubes.udf_mc_lewiner(udf, grads)
    idx, packed = mcubes.band_of(udf, grads)
    assert 0 < len(idx) <= udf.size
    sc = mcubes.McScratch(n)
    for rep in range(2):
        order = np.arange(len(idx)) if rep == 0 else np.random.default_rng(0).permutation(len(idx))
        v, f, nr, val = sc.mesh(idx[order], packed[order], len(idx))
        np.testing.assert_array_equal(f, f0)
        np.testing.assert_array_equal(v, v0.astype(np.float32))
        np.testing.assert_array_equal(nr, n0)
        np.testing.assert_array_equal(val, val0)
    # an empty band: no surface, and a band entry above the threshold is refused
    ev, ef, _, _ = sc.mesh(np.zeros(0, np.int32), np.zeros((0, 4), np.float32), 0)
    assert len(ev) == 0 and len(ef) == 0
    bad = packed.copy(); bad[0, 0] = 0.5
    with pytest.raises(RuntimeError, match="above the band threshold"):
        sc.mesh(idx, bad, len(idx))
    v, f, _, _ = sc.mesh(idx, packed, len(idx))                  # the refused call left the scratch clean
    np.testing.assert_array_equal(f, f0)


def test_mesher_rejects_steps_that_do_not_fit():
    """step > min(dim) - 1 would read past the volume."""
    udf, grads = mc_fields.FIELDS["two_spheres"](48)
    for vol, g, step in [(udf[:2, :2, :2], grads[:2, :2, :2], 2), (udf[:3, :3, :3], grads[:3, :3, :3], 3)]:
        with pytest.raises((RuntimeError, ValueError)):
            mcubes.udf_mc_lewiner(np.ascontiguousarray(vol), np.ascontiguousarray(g), step_size=step)
    with pytest.raises((RuntimeError, ValueError)):
        mcubes._run(np.ascontiguousarray(udf[:2, :2, :2]), None, 2, 0.01, True)
