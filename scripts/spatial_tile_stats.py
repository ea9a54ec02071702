"""CPU-side study for the spatially tiled sparse conv (round 4): for Morton-contiguous tiles of R output rows of
a grid level, how many unique input rows a tile touches, and how many 16-row MFMA sets the tile costs
  (A) output-stationary: rows sorted by slot mask inside the tile, 16-row groups, sets = sum popcount(group union)
  (B) slot-major: per slot the tile's pairs in groups of 16, sets = sum ceil(c_k / 16)
against the ideal pairs / 16.  Geometry from the oracle (test infrastructure).
usage: python scripts/spatial_tile_stats.py [points]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "adaptive-surface-reconstruction_amd"), os.path.join(REPO, "tests")]
import numpy as np
from asr_hip import synth
from oracle import oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
pts, nrm = synth.scan_cloud(n, seed=1000, device="cpu")
pts = pts.numpy()
radii = synth.knn_radii(pts, 24)
bb = synth.bounding_box(pts, 0.1)
o = O.Oracle()
o.build_octree(pts, radii, bb[0], bb[1], 1.0, 21)
grids = o.create_grids(5)


def level_of(keys):
    return (63 - np.array([int(k).bit_length() - 1 for k in keys]) * 0 - 0)  # placeholder


def spatial_order(keys):
    keys = keys.astype(np.uint64)
    lev = np.array([(int(k).bit_length() - 1) // 3 for k in keys], dtype=np.int64)
    mort = keys ^ (np.uint64(1) << (3 * lev).astype(np.uint64))
    norm = mort << (3 * (21 - lev)).astype(np.uint64)
    return np.lexsort((lev, norm))  # by normalised morton, ties (ancestors cannot coexist as leaves) by level


for lvl in range(3):
    g = grids[lvl]
    keys = g["voxel_keys"]
    idx = g["neighbors_index"].astype(np.int64)
    kidx = g["neighbors_kernel_index"].astype(np.int64)
    rs = g["neighbors_row_splits"].astype(np.int64)
    v = len(keys)
    rows = np.repeat(np.arange(v), np.diff(rs))
    order = spatial_order(keys)
    pos = np.empty(v, np.int64)
    pos[order] = np.arange(v)
    prow = pos[rows]            # tile-order position of each pair's output row
    for R in (128, 256, 512, 1024):
        ntile = (v + R - 1) // R
        tile_of_pair = prow // R
        # unique inputs per tile
        tu = np.unique(tile_of_pair * v + idx)
        uniq = len(tu)
        # how many of the unique inputs are inside the tile itself
        inside = np.count_nonzero(pos[tu % v] // R == tu // v)
        # (B) slot-major sets
        ts = tile_of_pair * 64 + kidx
        cnt = np.bincount(ts, minlength=ntile * 64)
        setsB = np.sum((cnt + 15) // 16)
        slots_per_tile = np.count_nonzero(cnt) / ntile
        # (A) rows sorted by mask inside the tile
        mask = np.zeros(v, np.uint64)
        np.bitwise_or.at(mask, rows, (np.uint64(1) << kidx.astype(np.uint64)))
        m_t = mask[order]
        tile_id = np.arange(v) // R
        o2 = np.lexsort((m_t, tile_id))
        m_s = m_t[o2]
        # groups of 16 inside tiles (tiles are multiples of 16 rows)
        pad = (-v) % 16
        mp = np.concatenate([m_s, np.zeros(pad, np.uint64)]).reshape(-1, 16)
        un = np.bitwise_or.reduce(mp, axis=1)
        pc = np.array([bin(int(x)).count("1") for x in un])
        setsA = pc.sum()
        ideal = len(idx) / 16
        print("level %d V %d pairs/row %.2f | R %4d: unique inputs/R %.2f (inside %.2f) slots/tile %.1f | "
              "eff A %.2f  eff B %.2f" % (lvl, v, len(idx) / v, R, uniq / v, inside / v, slots_per_tile,
                                          ideal / setsA, ideal / setsB), flush=True)
