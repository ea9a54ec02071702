"""Hand-worked known answers for rows a1-a7 (keys, octree closure, 55-slot CSR, CombineSiblings, voxel info):
two trees small enough to derive every array entry on paper from the reference's rules.  Each block cites
the reference lines the entries follow from (paths relative to /root/reference).  Data only -- checked
against the oracle (CPU) and against the HIP path (GPU) by tests/test_micro_trees.py.

Conventions used below
  * location code = Morton(x, y, z) | 1 << 3*level, x in bit 0 (cpp/lib/octreebase.h:59-65, zindex.h:34-54);
  * kernel slots (cpp/lib/grid.cpp:99-170): 0 = self; 1..6 = same level -x,+x,-y,+y,-z,+z (:54-56,109-124);
    7..30 = the four child-level cells across each face, faces in the order -x,+x,-y,+y,-z,+z
    (:66-83,127-146); 31..54 = parent-level face neighbour, slot 31 + 4*face +
    parent_kernel_index_offset[face][neighbour_key & 7] (:58-64,149-170);
  * bounding box [0,1]^3: root edge 1, voxel size 2^-level, integer offset 0 (cpp/lib/octree.cpp:20-42).
"""

BBOX = ([0.0, 0.0, 0.0], [1.0, 1.0, 1.0])

# ---- tree A: one point ---------------------------------------------------------------------------------
# radius 0.3: the last level whose voxel size (1, 0.5, 0.25, ...) is >= 0.3 is level 1 (octree.h:42-47);
# (0.2,0.2,0.2) lies in cell (0,0,0) of level 1 -> key 0 | 1<<3 = 8.  CreateAncestorsAndSiblings
# (octree.cpp:110-150) adds the seven siblings 9..15 and the root 1; BalanceFaces (:152-206) finds no parent
# level below the root; leaves = nodes without first child (:208-228) = 8..15.
TREE_A = {
    "points": [[0.2, 0.2, 0.2]], "radii": [0.3],
    "nodes": [1, 8, 9, 10, 11, 12, 13, 14, 15],
    "leaves": [8, 9, 10, 11, 12, 13, 14, 15],
    # grid 0: voxel i has cell (i&1, i>>1&1, i>>2&1); per axis the neighbour is i ^ (1<<axis), on the + side
    # (slot 2/4/6) when the bit is 0 and on the - side (slot 1/3/5) when it is 1; no other level exists
    "grid0": {
        "index": [0, 1, 2, 4,  1, 0, 3, 5,  2, 3, 0, 6,  3, 2, 1, 7,  4, 5, 6, 0,  5, 4, 7, 1,  6, 7, 4, 2,  7, 6, 5, 3],
        "kernel_index": [0, 2, 4, 6,  0, 1, 4, 6,  0, 2, 3, 6,  0, 1, 3, 6,  0, 2, 4, 5,  0, 1, 4, 5,  0, 2, 3, 5,  0, 1, 3, 5],
        "row_splits": [0, 4, 8, 12, 16, 20, 24, 28, 32],
        "centers": [[0.25 + 0.5 * (i & 1), 0.25 + 0.5 * (i >> 1 & 1), 0.25 + 0.5 * (i >> 2 & 1)] for i in range(8)],
        "sizes": [0.5] * 8,
        # CombineSiblings (grid.cpp:177-243): 8 consecutive siblings -> parent 1, slot = child id
        "up_index": [0] * 8, "up_kernel_index": [0, 1, 2, 3, 4, 5, 6, 7],
    },
    # grids 1..4: the root alone; it cannot merge (slot 8 = carried, grid.cpp:206-242)
    "coarse_keys": [[1], [1], [1], [1]],
    "grid1": {"index": [0], "kernel_index": [0], "row_splits": [0, 1], "up_index": [0], "up_kernel_index": [8]},
}

# ---- tree B: two points on different levels --------------------------------------------------------------
# A = (0.2,0.2,0.2), r 0.3 -> level 1, key 8 as above.  B = (0.9,0.9,0.9), r 0.2 -> level 2 (0.25 >= 0.2 > 0.125),
# cell (3,3,3): Morton = 0b111111 = 63, key 63 | 1<<6 = 127.  Closure: siblings 120..126, parent 15 and its
# siblings 8..14 (already there), root.  BalanceFaces: the parent 15 = cell (1,1,1) of level 1 has the face
# neighbours 14, 13, 11 (present) and three outside the cube -> nothing to add.  Leaves: 15 has the first child
# 120, so it is inner; leaves = 8..14, 120..127 (sorted as integers).
# Voxel index: i = 0..6 -> key 8+i, level-1 cell (i&1, i>>1&1, i>>2&1);
#              7+j (j = a + 2b + 4c) -> key 120+j, level-2 cell (2+a, 2+b, 2+c).
_ROWS_B = [
    # level 1.  The + side neighbour of rows 3 / 5 / 6 along z / y / x would be key 15, an inner node: absent; the
    # four level-2 cells across that face are found instead (child slots 7 + 4*face + n)
    [(0, 0), (2, 1), (4, 2), (6, 4)],
    [(0, 1), (1, 0), (4, 3), (6, 5)],
    [(0, 2), (2, 3), (3, 0), (6, 6)],
    [(0, 3), (1, 2), (3, 1), (27, 7), (28, 8), (29, 9), (30, 10)],       # face +z: cells (2..3, 2..3, 2)
    [(0, 4), (2, 5), (4, 6), (5, 0)],
    [(0, 5), (1, 4), (5, 1), (19, 7), (20, 8), (21, 11), (22, 12)],      # face +y: cells (2..3, 2, 2..3)
    [(0, 6), (3, 4), (5, 2), (11, 7), (12, 9), (13, 11), (14, 13)],      # face +x: cells (2, 2..3, 2..3)
    # level 2.  Inside the octant the neighbour along an axis is j ^ (1<<axis).  Across the three faces that
    # look at the coarser octants the same-level cell (1, ., .) does not exist; its parent does:
    #   face -x: neighbour_key & 7 = 1 + 2b + 4c, offset row {-1,0,-1,1,-1,2,-1,3} -> b + 2c, parent 14 = voxel 6
    #   face -y: neighbour_key & 7 = a + 2 + 4c,  offset row {-1,-1,0,1,-1,-1,2,3} -> a + 2c, parent 13 = voxel 5
    #   face -z: neighbour_key & 7 = a + 2b + 4,  offset row {-1,-1,-1,-1,0,1,2,3} -> a + 2b, parent 11 = voxel 3
    [(0, 7), (2, 8), (4, 9), (6, 11), (31, 6), (39, 5), (47, 3)],
    [(0, 8), (1, 7), (4, 10), (6, 12), (40, 5), (48, 3)],
    [(0, 9), (2, 10), (3, 7), (6, 13), (32, 6), (49, 3)],
    [(0, 10), (1, 9), (3, 8), (6, 14), (50, 3)],
    [(0, 11), (2, 12), (4, 13), (5, 7), (33, 6), (41, 5)],
    [(0, 12), (1, 11), (4, 14), (5, 8), (42, 5)],
    [(0, 13), (2, 14), (3, 11), (5, 9), (34, 6)],
    [(0, 14), (1, 13), (3, 12), (5, 10)],
]
TREE_B = {
    "points": [[0.2, 0.2, 0.2], [0.9, 0.9, 0.9]], "radii": [0.3, 0.2],
    "nodes": [1] + list(range(8, 16)) + list(range(120, 128)),
    "leaves": list(range(8, 15)) + list(range(120, 128)),
    "grid0": {
        "index": [idx for row in _ROWS_B for _, idx in row],
        "kernel_index": [slot for row in _ROWS_B for slot, _ in row],
        "row_splits": [sum(len(r) for r in _ROWS_B[:i]) for i in range(len(_ROWS_B) + 1)],   # 81 pairs
        "centers": [[0.25 + 0.5 * (i & 1), 0.25 + 0.5 * (i >> 1 & 1), 0.25 + 0.5 * (i >> 2 & 1)] for i in range(7)] +
                   [[0.625 + 0.25 * (j & 1), 0.625 + 0.25 * (j >> 1 & 1), 0.625 + 0.25 * (j >> 2 & 1)] for j in range(8)],
        "sizes": [0.5] * 7 + [0.25] * 8,
        # CombineSiblings: 120..127 are eight consecutive siblings -> parent 15, slot = child id; 8..14 are only
        # seven of their family -> carried (slot 8).  Coarse keys sorted: 8..15, so the parent is voxel 7
        "up_index": [0, 1, 2, 3, 4, 5, 6] + [7] * 8,
        "up_kernel_index": [8] * 7 + [0, 1, 2, 3, 4, 5, 6, 7],
    },
    "coarse_keys": [list(range(8, 16)), [1], [1], [1]],
    # grid 1 = the complete level-1 family: the CSR of tree A's grid 0
    "grid1": {"index": TREE_A["grid0"]["index"], "kernel_index": TREE_A["grid0"]["kernel_index"],
              "row_splits": TREE_A["grid0"]["row_splits"], "up_index": [0] * 8,
              "up_kernel_index": [0, 1, 2, 3, 4, 5, 6, 7]},
}

# ---- tree C: BalanceFaces inserts a sibling group ---------------------------------------------------------
# One point P = (0.45, 0.05, 0.05), r 0.1 -> level 3 (0.125 >= 0.1 > 0.0625, octree.h:42-47); cell
# (floor(0.45 * 8), 0, 0) = (3, 0, 0): Morton = x bits 0 and 1 at positions 0 and 3 = 9, key 9 | 1<<9 = 521.
# CreateAncestorsAndSiblings (octree.cpp:110-150): siblings 520..527 (children of 521>>3 = 65 = level-2 cell
# (1,0,0)), parent group 64..71 (children of 8), 8..15, root 1: 25 nodes.
# BalanceFaces (octree.cpp:152-206), first-sibling queue {8, 64, 520} (:160-166):
#   8   : HasFirstChild (64 exists) -> not a leaf, skipped (:175-176);
#   64  : its first child 512 does not exist, so the GROUP counts as a leaf although its sibling 65 is refined
#         ("process only the first sibling", :163-164); parent (0,0,0) of level 1, face neighbours 9, 10, 12 exist,
#         the other three are outside the cube (INVALID_KEY, :188) -> nothing;
#   520 : leaf; parent 65 = level-2 cell (1,0,0).  -x: 64 exists; -y, -z: outside; +y: (1,1,0) = 3|64 = 67 exists;
#         +z: (1,0,1) = 5|64 = 69 exists; +x: (2,0,0) = Morton 8 | 64 = 72 is NOT a node: the while loop (:191-201)
#         inserts the sibling group 72..79, queues 72, then key >>= 3 = 9 is found and the loop ends.
# Second round {72}: leaf; parent 9 = level-1 cell (1,0,0): 8, 11, 13 exist, (2,0,0) is outside -> nothing.
# Leaves (nodes without first child, :208-228): 9 now has the first child 72 and becomes inner.
TREE_C = {
    "points": [[0.45, 0.05, 0.05]], "radii": [0.1],
    "nodes_unbalanced": [1] + list(range(8, 16)) + list(range(64, 72)) + list(range(520, 528)),
    "nodes": [1] + list(range(8, 16)) + list(range(64, 80)) + list(range(520, 528)),
    "leaves": list(range(10, 16)) + [64] + list(range(66, 72)) + list(range(72, 80)) + list(range(520, 528)),
    "balance_rounds": 2,
}

# ---- tree D: one walk of the while loop inserts two levels --------------------------------------------------
# The same point with r 0.05 -> level 4 (0.0625 >= 0.05); cell (floor(0.45 * 16), 0, 0) = (7, 0, 0): x = 0b111 at
# Morton positions 0, 3, 6 = 73, key 73 | 1<<12 = 4169.  Closure: 4168..4175 (children of 521), 520..527, 64..71,
# 8..15, 1.  BalanceFaces queue {8, 64, 520, 4168}:
#   4168: leaf; parent 521 = level-3 cell (3,0,0); +x neighbour (4,0,0): x = 0b100 -> Morton 64, key 64 | 512 = 576,
#         not a node: the loop inserts 576..583 and queues 576; key >>= 3 = 72 (level-2 cell (2,0,0)) is not a
#         node either: inserts 72..79, queues 72; key >>= 3 = 9 exists.  Its other face neighbours 520, 523, 525
#         are siblings;
#   520 : its first child (520<<3 = 4160) does not exist -> leaf group; parent 65: +x neighbour 72 -- present or
#         not depending on whether 4168 was visited first; either way 72..79 ends up in the set;
#   8, 64 as in tree C.
# Second round {576, 72} (either order): 576 is a leaf, parent 72 = level-2 cell (2,0,0): 65, 73, 74, 76 exist, the
# rest is outside; 72 now has the first child 576 -> skipped.  The reference's sequential walk and the
# round-synchronous statement used here give the same set.
TREE_D = {
    "points": [[0.45, 0.05, 0.05]], "radii": [0.05],
    "nodes_unbalanced": [1] + list(range(8, 16)) + list(range(64, 72)) + list(range(520, 528)) + list(range(4168, 4176)),
    "nodes": [1] + list(range(8, 16)) + list(range(64, 80)) + list(range(520, 528)) + list(range(576, 584)) +
             list(range(4168, 4176)),
    # inner: 1, 8, 9 (child 72), 65 (child 520), 72 (child 576), 521 (child 4168)
    "leaves": list(range(10, 16)) + [64] + list(range(66, 72)) + list(range(73, 80)) + [520] + list(range(522, 528)) +
              list(range(576, 584)) + list(range(4168, 4176)),
    "balance_rounds": 2,
}

TREES = {"A": TREE_A, "B": TREE_B}
BALANCE_TREES = {"C": TREE_C, "D": TREE_D}
