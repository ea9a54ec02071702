"""Pins for the integer rows a1-a7: the hand-worked trees of tests/golden/micro_trees.py (every entry derived
on paper from /root/reference/cpp/lib/octree.cpp:110-228 and grid.cpp:99-243, lines cited there) against the
oracle (CPU) and against the HIP path through the C ABI (GPU)."""
import importlib.util
import os

import numpy as np
import pytest

from oracle import oracle as O

_spec = importlib.util.spec_from_file_location(
    "micro_trees", os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "micro_trees.py"))
micro = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(micro)


def _check(t, nodes, leaves, grids):
    assert nodes.tolist() == t["nodes"] and leaves.tolist() == t["leaves"]
    for lvl, name in ((0, "grid0"), (1, "grid1")):
        g, want = grids[lvl], t[name]
        assert g["neighbors_index"].tolist() == want["index"], name
        assert g["neighbors_kernel_index"].tolist() == want["kernel_index"], name
        assert g["neighbors_row_splits"].tolist() == want["row_splits"], name
        assert g["up_neighbors_index"].tolist() == want["up_index"], name
        assert g["up_neighbors_kernel_index"].tolist() == want["up_kernel_index"], name
        assert g["up_neighbors_row_splits"].tolist() == list(range(len(want["up_index"]) + 1))
    assert np.array_equal(grids[0]["voxel_centers"], np.array(t["grid0"]["centers"], np.float32))
    assert np.array_equal(grids[0]["voxel_sizes"], np.array(t["grid0"]["sizes"], np.float32))
    for lvl in range(1, 5):
        assert grids[lvl]["voxel_keys"].tolist() == t["coarse_keys"][lvl - 1], lvl


@pytest.mark.parametrize("name", ["A", "B"])
def test_oracle_reproduces_the_hand_worked_tree(name):
    t = micro.TREES[name]
    o = O.Oracle()
    o.build_octree(np.array(t["points"], np.float32), np.array(t["radii"], np.float32),
                   np.array(micro.BBOX[0], np.float32), np.array(micro.BBOX[1], np.float32))
    _check(t, o.nodes, o.leaves, o.create_grids(5))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["A", "B"])
def test_hip_reproduces_the_hand_worked_tree(gpu, name):
    import adaptivesurfacereconstruction as asr
    t = micro.TREES[name]
    tree = asr.create_octree(np.array(t["points"], np.float32), np.array(t["radii"], np.float32),
                             micro.BBOX[0], micro.BBOX[1])
    grids = asr.create_grids_from_octree(tree, 5, voxel_info_all_levels=True)
    for g in grids:  # the module omits empty arrays (cpp/pybind/module.cpp:163-228); level 4 has no up lists
        for k in ("up_neighbors_index", "up_neighbors_kernel_index"):
            g.setdefault(k, np.zeros(0, np.int32))
        g.setdefault("up_neighbors_row_splits", np.zeros(1, np.int64))
    nodes = tree.nodes.cpu().numpy().view(np.uint64)
    leaves = tree.leaves.cpu().numpy().view(np.uint64)
    _check(t, nodes, leaves, grids)


@pytest.mark.parametrize("mode", [0, 1], ids=["round-synchronous", "sequential-walk"])
@pytest.mark.parametrize("name", ["C", "D"])
def test_oracle_balance_inserts_the_hand_worked_nodes(name, mode):
    """BalanceFaces (/root/reference/cpp/lib/octree.cpp:152-206) has to ADD sibling groups in these trees; both
    statements of the oracle (mode 0 = round-synchronous leaf test, mode 1 = the reference's sequential walk) give
    the hand-derived node and leaf sets"""
    t = micro.BALANCE_TREES[name]
    o = O.Oracle()
    o.build_octree(np.array(t["points"], np.float32), np.array(t["radii"], np.float32),
                   np.array(micro.BBOX[0], np.float32), np.array(micro.BBOX[1], np.float32), mode=mode)
    assert o.nodes.tolist() == t["nodes"] and o.leaves.tolist() == t["leaves"]
    assert len(t["nodes"]) > len(t["nodes_unbalanced"]) and set(t["nodes_unbalanced"]) < set(t["nodes"])
    assert o.balance_rounds == t["balance_rounds"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["C", "D"])
def test_hip_balance_inserts_the_hand_worked_nodes(gpu, name):
    import adaptivesurfacereconstruction as asr
    t = micro.BALANCE_TREES[name]
    tree = asr.create_octree(np.array(t["points"], np.float32), np.array(t["radii"], np.float32),
                             micro.BBOX[0], micro.BBOX[1])
    assert tree.nodes.cpu().numpy().view(np.uint64).tolist() == t["nodes"]
    assert tree.leaves.cpu().numpy().view(np.uint64).tolist() == t["leaves"]
    # the grids built on the balanced leaves: the 55-slot lists of the oracle on the same (hand-checked) leaf set
    o = O.Oracle()
    o.build_octree(np.array(t["points"], np.float32), np.array(t["radii"], np.float32),
                   np.array(micro.BBOX[0], np.float32), np.array(micro.BBOX[1], np.float32))
    want = o.create_grids(5)
    got = asr.create_grids_from_octree(tree, 5, voxel_info_all_levels=True)
    for g, w in zip(got, want):
        for k in ("voxel_keys", "neighbors_index", "neighbors_kernel_index", "neighbors_row_splits"):
            assert np.array_equal(g[k], w[k]), k
