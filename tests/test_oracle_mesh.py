"""CPU tests of the contouring / component-filter oracle ("next" rows D.2, D.3) and of the
libstdc++ unordered_set replay used by the HIP kernel (csrc/asr_uset.h).  No GPU."""
import ctypes
import hashlib

import numpy as np

import parity
from oracle import oracle as O


def _uset_emulated(xs):
    from asr_hip import _lib
    xs = np.ascontiguousarray(xs, np.uint32)
    out = np.zeros(len(xs), np.uint32)
    rc = _lib.load().asr_hip_unordered_set_order(xs.ctypes.data_as(ctypes.c_void_p), int(len(xs)),
                                                 out.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    return out


def test_unordered_set_replay_matches_libstdcxx():
    """the device-side replay == iteration order of a real std::unordered_set<size_t>"""
    rng = np.random.default_rng(0)
    for trial in range(3000):
        n = int(rng.integers(1, 30))
        hi = int(rng.choice([40, 1000, 2**20, 2**31 - 1]))
        xs = np.sort(rng.choice(hi, size=n, replace=False)).astype(np.uint64)
        if trial % 3 == 0:
            xs = rng.permutation(xs)  # the kernel inserts ascending, the rule holds for any order
        want = O.unordered_set_order(xs)
        got = _uset_emulated(xs.astype(np.uint32))
        assert np.array_equal(want.astype(np.uint32), got), (xs, want, got)


def test_unordered_set_replay_rejects_more_than_cap():
    from asr_hip import _lib
    xs = np.arange(30, dtype=np.uint32)
    out = np.zeros(30, np.uint32)
    assert _lib.load().asr_hip_unordered_set_order(xs.ctypes.data_as(ctypes.c_void_p), 30,
                                                   out.ctypes.data_as(ctypes.c_void_p)) != 0


def test_contour_sphere_properties():
    """analytic sphere SDF on the adaptive grid: every vertex on the sphere, consistent orientation,
    one component"""
    g, du, values = parity.sphere_field(20000, seed=0)
    v, t = O.create_triangle_mesh(values, du, g["voxel_centers"], 1.0)
    assert v.shape[0] > 1000 and t.shape[0] > 2000
    r = np.linalg.norm(v, axis=1)
    assert abs(r - 1).max() < 0.02
    assert t.min() >= 0 and t.max() < v.shape[0]
    # the edge is oriented from the lower to the higher value (contouring.cpp:357): one winding
    a, b, c = v[t[:, 0]], v[t[:, 1]], v[t[:, 2]]
    nrm = np.cross(b - a, c - a)
    cen = (a + b + c) / 3
    s = np.sign((nrm * cen).sum(1))
    assert (s == s[0]).mean() > 0.99
    v2, t2 = O.remove_connected_components(v, t, 1, 3)
    assert v2.shape == v.shape and np.array_equal(t, t2)


def test_contour_threshold_gates_unsigned_value():
    g, du, values = parity.sphere_field(5000, seed=1)
    v, t = O.create_triangle_mesh(values, du, g["voxel_centers"], 1.0)
    far = values.copy()
    far[:, 1] = 5.0  # unsigned distance above the threshold everywhere -> no crossing is accepted
    v0, t0 = O.create_triangle_mesh(far, du, g["voxel_centers"], 1.0)
    assert v.shape[0] > 0 and v0.shape[0] == 0 and t0.shape[0] == 0


def test_contour_empty_inputs():
    v, t = O.create_triangle_mesh(np.zeros((0, 2), np.float32), np.zeros((0, 8), np.int64),
                                  np.zeros((0, 3), np.float32))
    assert v.shape == (0, 3) and t.shape == (0, 3)


def test_remove_components_order_and_ties():
    """postprocess.cpp:148-166: largest first, ties -> the later component first; compaction keeps order"""
    # components: A = verts 0..2 (1 tri), B = verts 3..6 (2 tris), C = verts 7..9 (1 tri), isolated 10
    v = np.arange(33, dtype=np.float32).reshape(11, 3)
    t = np.array([[0, 1, 2], [3, 4, 5], [4, 5, 6], [7, 8, 9]], np.int32)
    v1, t1 = O.remove_connected_components(v, t, 1, 3)
    assert np.array_equal(v1, v[3:7]) and np.array_equal(t1, [[0, 1, 2], [1, 2, 3]])
    v2, t2 = O.remove_connected_components(v, t, 2, 3)  # B, then C (tie A/C -> larger label)
    assert np.array_equal(v2, v[3:10]) and np.array_equal(t2, [[0, 1, 2], [1, 2, 3], [4, 5, 6]])
    v3, t3 = O.remove_connected_components(v, t, 2**63 - 1, 3)
    assert np.array_equal(v3, v[:10]) and np.array_equal(t3, t)
    v4, t4 = O.remove_connected_components(v, t, 2**63 - 1, 4)
    assert np.array_equal(v4, v[3:7])
    v5, t5 = O.remove_connected_components(v, t, 2**63 - 1, 1)  # the isolated vertex survives
    assert v5.shape[0] == 11 and np.array_equal(t5, t)


def test_contour_digest_is_stable():
    """regression pin of the oracle's own output (NOT a reference vector: the reference cannot be
    built here)"""
    g, du, values = parity.sphere_field(5000, seed=1, noise=0.3)
    v, t = O.create_triangle_mesh(values, du, g["voxel_centers"], 1.0)
    h = hashlib.sha256(v.tobytes() + t.tobytes()).hexdigest()
    assert (v.shape[0], t.shape[0], h[:16]) == parity.SPHERE_MESH_PIN, (v.shape[0], t.shape[0], h[:16])


def test_density_inlier_matches_reference_quirk():
    """preprocess.cpp:53-60 compares the counts after partial_sort permuted them"""
    from asr_hip import ops
    rng = np.random.default_rng(0)
    counts = rng.integers(1, 50, 1000)
    inl = ops.density_inlier(counts, 10.0)
    assert inl.shape == (1000,) and not inl[:100].any()  # the 100 smallest now sit in front
    thr = np.sort(counts)[99]
    assert inl.sum() == (counts > thr).sum()
