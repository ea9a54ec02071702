"""Shared reader of tests/golden/unet_layers_d4_1k.npz: every SpecialSparseConv call of the reference's
UNet5 graph (53 calls, produced by the reference's own unchanged model code, tests/golden/make_unet_fixture.py)
with its inputs and outputs.  The CSR each call used is rebuilt by the oracle from the stored points."""
import os

import numpy as np

import parity
from asr_hip import synth
from oracle import oracle as O

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "unet_layers_d4_1k.npz")


def load():
    fx = np.load(PATH)
    item = parity.oracle_geometry(fx["points"], fx["radii"], fx["bb_min"], fx["bb_max"])
    assert [len(item["voxel_sizes%d" % i]) for i in range(5)] == fx["voxels"].tolist()
    weights = synth.make_weights(int(fx["channel_div"]), seed=int(fx["seed"]))
    layers = []
    for i in range(int(fx["num_layers"])):
        name = str(fx["layer%d_name" % i])
        K, lo, li = (int(x) for x in fx["layer%d_meta" % i])
        if K == 55:
            csr = (item["neighbors_index%d" % lo], item["neighbors_kernel_index%d" % lo], item["neighbors_row_splits%d" % lo])
        elif lo < li:  # up: rows on the finer grid, inputs from the coarser one
            csr = (item["up_neighbors_index%d" % lo], item["up_neighbors_kernel_index%d" % lo],
                   item["up_neighbors_row_splits%d" % lo])
        else:          # down: rows on the coarser grid = inverted up lists of the finer one
            idx, rs, attr = O.invert_neighbors_list(len(item["voxel_sizes%d" % lo]), item["up_neighbors_index%d" % li],
                                                    item["up_neighbors_row_splits%d" % li],
                                                    item["up_neighbors_kernel_index%d" % li])
            csr = (idx, attr, rs)
        imp = fx["layer%d_imp" % i]
        layers.append(dict(name=name, K=K, csr=csr, inp=fx["layer%d_inp" % i], imp=imp if imp.size else None,
                           out=fx["layer%d_out" % i], oimp=fx["layer%d_oimp" % i],
                           kernel=weights[name + ".kernel"], bias=weights[name + ".bias"],
                           normalize=name.endswith("conv1b")))
    return layers
