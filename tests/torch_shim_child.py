"""Child process of tests/test_torch_shim.py: loads ONLY the C++ op library (no Python registration of the open3d
namespace in this process) and checks the four ops against the oracle, eagerly and through a TorchScript module
that was saved and loaded again -- the way a libtorch caller reaches them (cpp/lib/asr.cpp:315-326)."""
import os
import sys
import tempfile

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "tests"), os.path.join(REPO, "adaptive-surface-reconstruction_amd"), REPO]
LIB = os.path.join(REPO, "adaptive-surface-reconstruction_amd", "csrc", "libasr_open3d_ops.so")


def main():
    torch.ops.load_library(LIB)
    assert "open3d.ml.torch.ops" not in sys.modules
    for name in ("sparse_conv", "continuous_conv", "invert_neighbors_list", "reduce_subarrays_sum"):
        assert torch._C._dispatch_has_kernel_for_dispatch_key("open3d::" + name, "CUDA"), name
    if len(sys.argv) > 1 and sys.argv[1] == "load-only":
        print("SHIM LOAD OK")
        return
    import parity
    from asr_hip import synth
    from oracle import oracle as O
    dev = torch.device("cuda:0")
    p, q = synth.scan_cloud(6000, seed=3, device="cpu")
    pts, nrm = p.numpy(), q.numpy()
    rad = synth.knn_radii(pts, 24)
    bb = synth.bounding_box(pts, 0.1)
    geo = parity.oracle_geometry(pts, rad, *bb)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    empty = torch.empty(0, device=dev)
    rng = np.random.default_rng(0)
    # sparse conv (importance weighted, normalised) + reduce_subarrays_sum
    idx, kidx, rs = geo["neighbors_index1"], geo["neighbors_kernel_index1"], geo["neighbors_row_splits1"]
    v = len(rs) - 1
    f = rng.standard_normal((v, 32)).astype(np.float32)
    W = (rng.standard_normal((55, 32, 24)) * 0.1).astype(np.float32)
    nimp = rng.uniform(0.1, 1.0, size=len(idx)).astype(np.float32)
    got = torch.ops.open3d.sparse_conv(t(W), t(f), empty, t(idx), t(kidx), t(nimp), t(rs), True, 64)
    parity.assert_close(got.cpu().numpy(), O.sparse_conv(W, f, idx, kidx, nimp, rs, True))
    got = torch.ops.open3d.reduce_subarrays_sum(t(nimp), t(rs))
    parity.assert_close(got.cpu().numpy(), O.reduce_subarrays_sum(nimp, rs))
    # inversion of the up list (with attributes)
    ui, uk, ur = geo["up_neighbors_index0"], geo["up_neighbors_kernel_index0"], geo["up_neighbors_row_splits0"]
    n_coarse = len(geo["voxel_sizes1"])
    gi, gr, ga = torch.ops.open3d.invert_neighbors_list(n_coarse, t(ui), t(ur), t(uk))
    ri, rr, ra = O.invert_neighbors_list(n_coarse, ui, ur, uk)
    assert np.array_equal(gi.cpu().numpy(), ri) and np.array_equal(gr.cpu().numpy(), rr)
    assert np.array_equal(ga.cpu().numpy(), ra)
    # continuous conv of the aggregation block
    centers, sizes = geo["voxel_centers0"], geo["voxel_sizes0"]
    ai, ar = geo["aggregation_neighbors_index"], geo["aggregation_row_splits"]
    aimp = (geo["aggregation_scale_compat"] * O.window_poly6(geo["aggregation_neighbors_dist"])).astype(np.float32)
    feats = np.concatenate([nrm, np.ones((len(nrm), 1), np.float32)], 1)
    Wc = (rng.standard_normal((4, 4, 4, 4, 32)) * 0.1).astype(np.float32)
    got = torch.ops.open3d.continuous_conv(t(Wc), t(centers), t(sizes), torch.zeros(3, device=dev), t(pts), t(feats),
                                           empty, t(ai), t(aimp), t(ar), True, "ball_to_cube_radial", True, "linear", 64)
    parity.assert_close(got.cpu().numpy(), O.continuous_conv(Wc, centers, sizes, pts, feats, ai, aimp, ar, True))

    # TorchScript round trip: a scripted module that calls the ops, saved, loaded, run
    class Block(torch.nn.Module):
        def __init__(self, w):
            super().__init__()
            self.kernel = torch.nn.Parameter(w, requires_grad=False)

        def forward(self, feats, idx, kidx, imp, rs):
            none = torch.empty(0, device=feats.device)
            out = torch.ops.open3d.sparse_conv(self.kernel, feats, none, idx, kidx, imp, rs, True, 64)
            return out, torch.ops.open3d.reduce_subarrays_sum(imp, rs)

    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "block.pt")
        torch.jit.script(Block(t(W))).save(path)
        mod = torch.jit.load(path, map_location=dev)
    o, s = mod(t(f), t(idx), t(kidx), t(nimp), t(rs))
    parity.assert_close(o.cpu().numpy(), O.sparse_conv(W, f, idx, kidx, nimp, rs, True))
    parity.assert_close(s.cpu().numpy(), O.reduce_subarrays_sum(nimp, rs))

    # and with no Python in the process at all: csrc/shim_demo (libtorch) loads the op library and an archive whose
    # module carries its inputs as buffers, runs forward() and writes the result
    class Closed(torch.nn.Module):
        def __init__(self):
            super().__init__()
            for name, val in (("kernel", t(W)), ("feats", t(f)), ("idx", t(idx)), ("kidx", t(kidx)), ("imp", t(nimp)),
                              ("rs", t(rs))):
                self.register_buffer(name, val)

        def forward(self):
            none = torch.empty(0, device=self.feats.device)
            return torch.ops.open3d.sparse_conv(self.kernel, self.feats, none, self.idx, self.kidx, self.imp, self.rs,
                                                True, 64)

    import subprocess
    demo = os.path.join(os.path.dirname(LIB), "shim_demo")
    with tempfile.TemporaryDirectory() as d:
        path, outp = os.path.join(d, "closed.pt"), os.path.join(d, "out.bin")
        torch.jit.script(Closed()).save(path)
        r = subprocess.run([demo, LIB, path, outp], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        got = np.fromfile(outp, np.float32).reshape(v, -1)
    parity.assert_close(got, O.sparse_conv(W, f, idx, kidx, nimp, rs, True))
    print("SHIM OK")


if __name__ == "__main__":
    main()
