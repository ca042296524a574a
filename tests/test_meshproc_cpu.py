"""surfd_amd.meshproc — numpy restatements of the trimesh calls behind get_mesh_from_udf.

No trimesh exists in this environment: these are pinned by hand-checkable fixtures ("parity unpinned vs trimesh");
each test states the trimesh behaviour it encodes."""
import numpy as np
import pytest

from surfd_amd import meshproc as mp

TET_V = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], dtype=np.float64)
TET_F = np.array([[0, 2, 1], [0, 1, 3], [1, 2, 3], [0, 3, 2]])         # outward winding


def _edge_use(faces):
    e, _ = mp.edges_of_faces(faces)
    _, c = np.unique(np.sort(e, axis=1), axis=0, return_counts=True)
    return c


def test_edges_are_face_major_directed_triples():
    e, ef = mp.edges_of_faces(np.array([[5, 6, 7], [7, 6, 8]]))
    np.testing.assert_array_equal(e, [[5, 6], [6, 7], [7, 5], [7, 6], [6, 8], [8, 7]])
    np.testing.assert_array_equal(ef, [0, 0, 0, 1, 1, 1])


def test_cull_and_merge_keeps_first_occurrence_order():
    """process(): unreferenced vertices vanish, coincident ones (to 1e-8) merge, survivors stay in first-seen order."""
    v = np.array([[9, 9, 9],            # 0 unreferenced
                  [1, 0, 0],            # 1
                  [0, 0, 0],            # 2
                  [0, 1, 0],            # 3
                  [1, 0, 0 + 4e-9],     # 4 == 1 within 1e-8
                  [0, 0, 1]], float)    # 5
    f = np.array([[2, 1, 3], [2, 4, 5]])
    nv, nf = mp.cull_and_merge(v, f)
    np.testing.assert_allclose(nv, [[1, 0, 0], [0, 0, 0], [0, 1, 0], [0, 0, 1]], atol=1e-8)
    np.testing.assert_array_equal(nf, [[1, 0, 2], [1, 0, 3]])
    # idempotent
    v2, f2 = mp.cull_and_merge(nv, nf)
    np.testing.assert_array_equal(v2, nv); np.testing.assert_array_equal(f2, nf)
    # non-finite vertices take their faces with them
    v[3, 0] = np.nan
    nv, nf = mp.cull_and_merge(v, f)
    assert len(nf) == 1 and np.isfinite(nv).all() and len(nv) == 3


def test_duplicate_faces_ignore_winding_and_follow_sorted_key_order():
    f = np.array([[3, 4, 5], [0, 1, 2], [2, 1, 0], [1, 2, 0], [5, 3, 4], [0, 1, 3]])
    out = mp.drop_duplicate_faces(f)
    # survivors = first occurrences, ordered by (max, mid, min) vertex: {0,1,2} < {0,1,3} < {3,4,5}
    np.testing.assert_array_equal(out, [[0, 1, 2], [0, 1, 3], [3, 4, 5]])
    assert mp.drop_duplicate_faces(np.zeros((0, 3), int)).shape == (0, 3)


def test_degenerate_faces_by_height():
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [2, 0, 0], [0.5, 1e-9, 0], [0, 0, 0]], float)
    f = np.array([[0, 1, 2],      # fine
                  [0, 1, 3],      # collinear: zero area
                  [0, 1, 4],      # height 1e-9 < 1e-8
                  [0, 5, 2]])     # two coincident corners: an edge of zero length
    np.testing.assert_array_equal(mp.drop_degenerate_faces(v, f), [[0, 1, 2]])


def test_border_rows_and_watertightness():
    assert len(mp.border_edge_rows(TET_F)) == 0                      # closed: every edge twice
    open_f = TET_F[:3]
    rows = mp.border_edge_rows(open_f)
    e, _ = mp.edges_of_faces(open_f)
    assert sorted(map(tuple, np.sort(e[rows], axis=1))) == [(0, 2), (0, 3), (2, 3)]


def test_single_triangle_hole_is_closed_with_consistent_winding():
    filled = mp.fill_small_holes(TET_V, TET_F[:3])
    assert len(filled) == 4 and (_edge_use(filled) == 2).all()
    # consistent orientation: every directed edge appears once in each direction
    e, _ = mp.edges_of_faces(filled)
    assert set(map(tuple, e)) == set(map(tuple, e[:, ::-1]))
    # outward: the signed volume stays positive
    t = TET_V[filled]
    assert np.einsum("ij,ij->i", t[:, 0], np.cross(t[:, 1], t[:, 2])).sum() > 0


def test_quad_hole_gets_two_triangles_and_large_holes_stay_open():
    # a cube as 12 triangles, remove both triangles of the top face -> a 4-edge hole
    V = np.array([[x, y, z] for z in (0, 1) for y in (0, 1) for x in (0, 1)], float)
    quads = [(0, 2, 3, 1), (4, 5, 7, 6), (0, 1, 5, 4), (2, 6, 7, 3), (0, 4, 6, 2), (1, 3, 7, 5)]
    F = np.array([t for a, b, c, d in quads for t in ((a, b, c), (a, c, d))])
    assert (_edge_use(F) == 2).all()
    holed = np.array([f for f in F if not set(f) <= {4, 5, 6, 7}])
    assert len(holed) == 10
    filled = mp.fill_small_holes(V, holed)
    assert len(filled) == 12 and (_edge_use(filled) == 2).all()
    e, _ = mp.edges_of_faces(filled)
    assert set(map(tuple, e)) == set(map(tuple, e[:, ::-1]))
    # a 6-edge hole (two adjacent cube faces removed) is left alone
    holed6 = np.array([f for f in F if not (set(f) <= {4, 5, 6, 7} or set(f) <= {1, 3, 5, 7})])
    assert len(mp.fill_small_holes(V, holed6)) == len(holed6)


def test_border_smoothing_moves_only_the_border_and_converges_to_neighbour_means():
    # a 5x5 flat grid of quads (two triangles each); lift one border vertex and one interior vertex
    n = 5
    V = np.array([[i, j, 0.0] for i in range(n) for j in range(n)])
    idx = lambda i, j: i * n + j
    F = np.array([t for i in range(n - 1) for j in range(n - 1)
                  for t in ((idx(i, j), idx(i + 1, j), idx(i + 1, j + 1)), (idx(i, j), idx(i + 1, j + 1), idx(i, j + 1)))])
    V[idx(0, 2), 2] = 1.0
    V[idx(2, 2), 2] = 1.0
    one = mp.smooth_borders(V, F, lam=0.3, iterations=1)
    assert one[idx(2, 2), 2] == 1.0                                   # interior untouched
    assert one[idx(0, 2), 2] == pytest.approx(0.7)                    # 1 + 0.3 * (0 - 1): both border neighbours at 0
    assert one[idx(0, 1), 2] == pytest.approx(0.15)                   # 0 + 0.3 * ((1 + 0) / 2 - 0), from the OLD state
    many = mp.smooth_borders(V, F)
    assert abs(many[idx(0, 2), 2]) < 0.2 and many[idx(2, 2), 2] == 1.0
    np.testing.assert_array_equal(many[:, :2][[idx(1, 1), idx(2, 2)]], V[:, :2][[idx(1, 1), idx(2, 2)]])
    # closed mesh: nothing moves
    np.testing.assert_array_equal(mp.smooth_borders(TET_V, TET_F), TET_V)


def test_angle_weighted_normals():
    n = mp.vertex_normals_by_angle(TET_V, TET_F)
    np.testing.assert_allclose(np.linalg.norm(n, axis=1), 1.0, atol=1e-12)
    np.testing.assert_allclose(n[0], -np.ones(3) / np.sqrt(3), atol=1e-12)     # three right angles, three axis faces
    flat = mp.vertex_normals_by_angle(np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0]], float), np.array([[0, 1, 2], [1, 3, 2]]))
    np.testing.assert_allclose(flat, np.tile([0, 0, 1.0], (4, 1)), atol=1e-12)


def test_clean_until_stable_pipeline():
    # tetrahedron with a duplicated vertex, a duplicated face, a degenerate face and one face missing
    V = np.vstack([TET_V, TET_V[1] + 1e-10, [[5, 5, 5]]])
    F = np.array([[0, 2, 1], [0, 4, 3], [1, 2, 3], [2, 1, 0], [1, 4, 2]])       # [0,3,2] missing; [1,4,2] collapses
    v, f = mp.clean_until_stable(V, F)
    assert len(v) == 4 and len(f) == 4 and (_edge_use(f) == 2).all()


def test_native_obj_writer_text(tmp_path):
    """meshproc.write_obj goes through the library's writer (surfd_write_obj); its text is the per-line formatting the
    module documents ('v' lines with 6 decimals, 1-based 'f' lines), incl. rounding ties and an empty mesh."""
    rng = np.random.default_rng(1)
    v = rng.normal(size=(500, 3)) * np.array([1.0, 1e-3, 1e3])
    v[0] = [0.0000005, -0.0000005, 1.0000005]
    f = rng.integers(0, 500, size=(900, 3))
    p = tmp_path / "a" / "m.obj"                         # the directory is created
    mp.write_obj(str(p), v, f)
    want = "# surfd_amd mesh\n" + "".join(f"v {x:.6f} {y:.6f} {z:.6f}\n" for x, y, z in v) + "".join(f"f {a} {b} {c}\n" for a, b, c in f + 1)
    assert p.read_text() == want
    mp.write_obj(str(p), np.zeros((0, 3)), np.zeros((0, 3), np.int64))
    assert p.read_text() == "# surfd_amd mesh\n"


def _clean_spec(vertices, faces, max_iter=10):
    """The cleaning sequence written out call by call as the reference performs it (meshudf.py:380-404), without the
    shortcuts of meshproc.clean_until_stable."""
    v, f = mp.cull_and_merge(vertices, faces)
    f = mp.drop_duplicate_faces(f)
    f = mp.drop_degenerate_faces(v, f)
    f = mp.fill_small_holes(v, f)
    v, f = mp.cull_and_merge(v, f)
    counts, rounds = (0, 0), 0
    while counts != (len(v), len(f)) and rounds < max_iter:
        v, f = mp.cull_and_merge(v, f)
        f = mp.drop_duplicate_faces(f)
        f = mp.drop_degenerate_faces(v, f)
        counts = (len(v), len(f))
        rounds += 1
        v, f = mp.cull_and_merge(v, f)
    return v, f


@pytest.mark.parametrize("seed", range(6))
def test_clean_until_stable_equals_the_unabridged_sequence(seed):
    """Skipping the provably idempotent re-processing steps must not change a single vertex or face — on messy meshes:
    duplicated and unreferenced vertices, duplicate / degenerate / sliver faces, holes, non-finite coordinates."""
    rng = np.random.default_rng(seed)
    n = 60
    v = rng.normal(size=(n, 3))
    v[rng.integers(0, n, 8)] = v[rng.integers(0, n, 8)]                  # coincident vertices
    v[5] = v[6] + 1e-10                                                  # merge within tolerance
    if seed % 2:
        v[7] = [np.nan, 0.0, 0.0]
    f = rng.integers(0, n, size=(150, 3))
    f = np.vstack([f, f[:10][:, ::-1], f[10:15], np.stack([f[:5, 0], f[:5, 0], f[:5, 1]], 1)])   # flipped / repeated / degenerate
    a, b = mp.clean_until_stable(v, f)
    c, d = _clean_spec(v, f)
    assert np.array_equal(a, c) and np.array_equal(b, d)
    assert np.array_equal(*[x[0] for x in (mp.clean_until_stable(a, b), (a, b))])            # stable: cleaning again changes nothing


def test_clean_until_stable_fast_path_on_clean_meshes():
    """A mesh with nothing to remove and no small hole (what marching cubes usually delivers) leaves the cleaning after
    the first pass; the result must still be the unabridged sequence's, face order included."""
    from surfd_amd import mcubes
    ax = np.linspace(-1, 1, 40, dtype=np.float32)
    z, y, x = np.meshgrid(ax, ax, ax, indexing="ij")
    vol = np.sqrt(x * x + y * y + (1.3 * z) ** 2) - 0.6
    for classic in (True, False):
        v, f = mcubes.marching_cubes(np.ascontiguousarray(vol), 0.0, classic=classic)
        a, b = mp.clean_until_stable(v, f)
        c, d = _clean_spec(v, f)
        assert np.array_equal(a, c) and np.array_equal(b, d)
        assert len(b) == len(f)                                         # really the fast path: nothing removed
        # an open mesh (big hole: not filled) and a shuffled one take it too
        keep = v[f].mean(axis=1)[:, 2] < 25.0
        perm = np.random.default_rng(3).permutation(int(keep.sum()))
        a, b = mp.clean_until_stable(v, f[keep][perm])
        c, d = _clean_spec(v, f[keep][perm])
        assert np.array_equal(a, c) and np.array_equal(b, d)


# ---- randomised invariants (hypothesis): what must hold for ANY mesh, whatever trimesh's exact face order would be -----
from hypothesis import given, settings, strategies as st    # noqa: E402


@st.composite
def _meshes(draw):
    n = draw(st.integers(4, 40))
    m = draw(st.integers(1, 80))
    seed = draw(st.integers(0, 2 ** 31 - 1))
    rng = np.random.default_rng(seed)
    v = np.round(rng.normal(size=(n, 3)), draw(st.integers(0, 3)))          # coarse rounding makes coincident vertices likely
    f = rng.integers(0, n, size=(m, 3))
    return v, f


@settings(max_examples=60, deadline=None)
@given(_meshes())
def test_cull_and_merge_invariants(mesh):
    v, f = mesh
    nv, nf = mp.cull_and_merge(v, f)
    assert len(nf) == len(f)                                                 # faces are re-indexed, never dropped
    np.testing.assert_allclose(nv[nf], v[f], atol=mp.MERGE_TOL)               # every corner keeps its position
    assert len(np.unique(np.round(nv / mp.MERGE_TOL).astype(np.int64), axis=0)) == len(nv)    # survivors are distinct
    assert set(np.unique(nf)) == set(range(len(nv)))                         # and all referenced
    v2, f2 = mp.cull_and_merge(nv, nf)
    assert np.array_equal(v2, nv) and np.array_equal(f2, nf)                 # idempotent


@settings(max_examples=60, deadline=None)
@given(_meshes())
def test_duplicate_and_component_invariants(mesh):
    v, f = mesh
    g = mp.drop_duplicate_faces(f)
    keys = {tuple(sorted(t)) for t in f.tolist()}
    assert len(g) == len(keys) and {tuple(sorted(t)) for t in g.tolist()} == keys          # one face per vertex triple
    assert np.array_equal(mp.drop_duplicate_faces(g), g)                                  # idempotent, order included
    # connected components against a plain union-find over shared vertices
    parent = list(range(len(v)))

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a
    for a, b, c in f.tolist():
        ra, rb, rc = find(a), find(b), find(c)
        parent[rb] = ra
        parent[find(rc)] = ra
    want = [find(t[0]) for t in f.tolist()]
    lab = mp.face_components(f, len(v))
    assert len(set(zip(want, lab.tolist()))) == len(set(want)) == len(set(lab.tolist()))   # same partition of the faces
    # the size filter keeps exactly the faces of large enough components
    sizes = {r: want.count(r) for r in set(want)}
    k = max(sizes.values())
    kv, kf = mp.keep_components_with_at_least(v, f, k)
    assert len(kf) == sum(s for s in sizes.values() if s >= k)


def _open_sheet(n=9, h=0.1):
    """flat n x n vertex grid in the z = 0 plane, two triangles per cell"""
    ii, jj = np.mgrid[0:n, 0:n]
    v = np.stack([ii.ravel() * h, jj.ravel() * h, np.zeros(n * n)], 1).astype(np.float64)
    idx = lambda i, j: i * n + j
    f = []
    for i in range(n - 1):
        for j in range(n - 1):
            f.append([idx(i, j), idx(i + 1, j), idx(i + 1, j + 1)])
            f.append([idx(i, j), idx(i + 1, j + 1), idx(i, j + 1)])
    return v, np.array(f, dtype=np.int64)


def test_laplacian_smoothing_keeps_the_border_of_an_open_sheet():
    """MeshLab's apply_coord_laplacian_smoothing defaults (boundary=True, cotangent weights) smooth border
    vertices only along the border polyline.  On a flat open sheet: nothing leaves the plane, every non-corner border vertex
    stays ON its border line and at its place (evenly spaced neighbours), only the four corners round off; with
    boundary=False the border is pulled inwards (what the previous uniform umbrella did)."""
    n, h = 9, 0.1
    v, f = _open_sheet(n, h)
    out = mp.laplacian_smooth(v, f, steps=3)
    assert np.abs(out[:, 2]).max() == 0.0
    side = (n - 1) * h
    on_left = np.isclose(v[:, 0], 0.0) & (v[:, 1] > 1e-9) & (v[:, 1] < side - 1e-9)
    far_from_corner = on_left & (v[:, 1] > 3.5 * h) & (v[:, 1] < side - 3.5 * h)
    np.testing.assert_allclose(out[far_from_corner], v[far_from_corner], atol=1e-12)          # on the line x = 0, not moved along it
    assert np.abs(out[on_left, 0]).max() < 0.2 * h                               # next to a rounded corner: dragged along by it, a little
    corner = np.isclose(v[:, 0], 0.0) & np.isclose(v[:, 1], 0.0)
    assert 0 < out[corner, 0][0] < h and 0 < out[corner, 1][0] < h                # corners round off, by less than a cell
    interior = (v[:, 0] > 3.5 * h) & (v[:, 0] < side - 3.5 * h) & (v[:, 1] > 3.5 * h) & (v[:, 1] < side - 3.5 * h)
    np.testing.assert_allclose(out[interior], v[interior], atol=1e-12)            # a regular interior is a fixed point
    shrunk = mp.laplacian_smooth(v, f, steps=3, boundary=False)
    assert shrunk[far_from_corner, 0].min() > 0.2 * h                             # without the border rule the whole hem creeps inwards
    # uniform weights, border rule on: same fixed points
    uni = mp.laplacian_smooth(v, f, steps=3, cotangent=False)
    assert np.abs(uni[far_from_corner, 0]).max() < 1e-12
