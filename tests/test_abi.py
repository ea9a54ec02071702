"""The C-ABI library loads without a GPU and exports every symbol include/asr_hip.h declares;
host-only entry points work (no compute calls)."""
import pytest
import ctypes
import os
import re

import numpy as np

from asr_hip import _lib, synth
from oracle import oracle as O

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(REPO, "include", "asr_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(asr_(?:hip_)?[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 25
    for s in declared:
        assert hasattr(lib, s), "missing export %s" % s
    assert sorted(_lib.EXPORTS) == declared
    assert lib.asr_hip_version().decode().startswith("0.2.0")


def test_struct_layouts_match_header_sizes():
    # sizes implied by the header (LP64): a mismatch would corrupt every call
    assert ctypes.sizeof(_lib.OctreeFrame) == 4 * (22 + 22 + 3 + 3 + 3)
    assert ctypes.sizeof(_lib.Weight) == 8 + 8 + 8 + 5 * 8
    assert ctypes.sizeof(_lib.ImplicitSizes) == 8 * (2 + 5 + 5 + 1)
    assert ctypes.sizeof(_lib.ImplicitParams) == 4 * (1 + 1 + 3 + 3 + 1 + 1)
    assert ctypes.sizeof(_lib.SparseConvArgs) == 224
    lib = _lib.load()
    assert lib.asr_hip_struct_size(b"asr_sparse_conv_args") == 224
    assert lib.asr_hip_struct_size(b"nope") == 0


def test_host_frame_init_matches_oracle():
    """asr_octree_frame_init is host code (cpp/lib/octree.cpp:20-42); compare with the oracle"""
    rng = np.random.default_rng(0)
    for _ in range(20):
        lo = rng.uniform(-3, 0, size=3).astype(np.float32)
        hi = (lo + rng.uniform(0.1, 5, size=3)).astype(np.float32)
        f = _lib.frame_init(lo, hi)
        o = O.Oracle()
        o.build_octree(np.zeros((0, 3), np.float32), np.zeros(0, np.float32), lo, hi)
        vs, ivs, off = o.frame()
        assert np.array_equal(np.array(f.voxel_size[:], np.float32), vs)
        assert np.array_equal(np.array(f.inv_voxel_size[:], np.float32), ivs)
        assert list(f.offset) == list(off)


def test_degenerate_bbox_is_rejected():
    import pytest
    with pytest.raises(_lib.AsrHipError):
        _lib.frame_init([0, 0, 0], [0, 0, 0])


def test_param_table_matches_survey():
    """92 200 000 parameters, names of SURVEY A.6"""
    shapes = synth.unet5_param_shapes(1)
    assert sum(int(np.prod(s)) for s in shapes.values()) == 92200000
    assert shapes["sparseconv_encblock0.conv1a.kernel"] == (55, 32, 56)
    assert shapes["sparseconv_decblock1.conv1.kernel"] == (55, 384, 128)
    assert shapes["dense_decoder3.weight"] == (2, 32)
    assert "sparseconv_down4.conv1a.kernel" not in shapes
    w = synth.make_weights(4, seed=5)
    w2 = synth.make_weights(4, seed=5)
    assert all(np.array_equal(w[k], w2[k]) for k in w)


def test_no_cpu_fallback_in_product():
    """the product ops refuse CPU tensors instead of computing something else"""
    import pytest
    import torch
    from asr_hip import ops
    with pytest.raises(_lib.AsrHipError):
        ops._dev(torch.zeros(3), torch.float32)
    if not torch.cuda.is_available():
        with pytest.raises(_lib.AsrHipError):
            _lib.Context()


def test_bench_instance_list_matches_committed_profile():
    """tests/sconv_instances.py (the instances the GPU parity tests force) == the k_sconv_mfma
    instances in the newest committed 10 M-point kernel trace of bench.py"""
    import sconv_instances as si
    path = si.latest_trace("f32")
    assert path is not None
    assert si.instances_in_trace(path) == si.BENCH_INSTANCES, path
    path = si.latest_trace("16")   # the default bench runs the bf16x3 arithmetic on k_sconv_mfma16
    assert path is not None
    assert si.instances16_in_trace(path) == si.BENCH_INSTANCES16, path


def test_options_without_a_gpu():
    """set/get option validate their arguments without touching the device"""
    lib = _lib.load()
    assert lib.asr_hip_context_set_option(None, b"overlap", ctypes.c_int64(0)) == 1
    assert lib.asr_hip_context_device(None) == -1
    # the option table (asr_hip_option_info): names + built-in defaults, no work-skipping mode among them
    names, i = {}, 0
    name, dflt = ctypes.c_char_p(), ctypes.c_int64()
    while lib.asr_hip_option_info(i, ctypes.byref(name), ctypes.byref(dflt)) == 0:
        names[name.value.decode()] = dflt.value
        i += 1
    assert names["overlap"] == 1 and names["sconv_split_rows"] == 32768 and len(names) >= 20
    assert not any("dry" in k or "ring" in k for k in names)
    assert lib.asr_hip_option_info(i, ctypes.byref(name), ctypes.byref(dflt)) == 1


def test_model_pt_loader_reads_a_torchscript_archive(tmp_path):
    """the reference ships its weights as a TorchScript file (cpp/lib/asr.cpp:138-139); the module's loader
    takes such an archive, a pickled state dict or an .npz and returns the same name -> tensor table"""
    import torch
    import parity
    import adaptivesurfacereconstruction as asr
    w = synth.make_weights(4, seed=5)
    path = str(tmp_path / "model.pt")
    parity.save_torchscript_weights(w, path)
    got = asr._load_weights(path)
    assert set(got) == set(w) and len(got) == 109
    assert all(np.array_equal(got[k].numpy(), w[k]) for k in w)
    torch.save({k: torch.from_numpy(v) for k, v in w.items()}, str(tmp_path / "sd.pt"))
    got = asr._load_weights(str(tmp_path / "sd.pt"))
    assert all(np.array_equal(got[k].numpy(), w[k]) for k in w)
    np.savez(str(tmp_path / "w.npz"), **w)
    got = asr._load_weights(str(tmp_path / "w.npz"))
    assert all(np.array_equal(got[k], w[k]) for k in w)
    assert "rocPRIM" in asr.get_third_party_notices() and asr.get_version_str().startswith("0.2.0")


def test_print_callback_registry():
    """asr::SetPrintCallbackFunction / asr::Print (cpp/lib/asr.cpp:34-47, utils.h:26): per-level callbacks, process
    wide; an invalid level is refused and changes nothing.  Host only."""
    from asr_hip import _lib
    got = []
    _lib.set_print_callback_function(got.append, ["INFO", "WARN"])
    _lib.library_print("grid building\n", 1)
    _lib.library_print("dbg", 0)      # no callback on DEBUG
    _lib.library_print("careful", 2)
    assert got == ["grid building\n", "careful"]
    with pytest.raises(RuntimeError, match="invalid verbosity level"):
        _lib.set_print_callback_function(got.append, [1, 4])
    _lib.library_print("still there", 1)
    assert got[-1] == "still there"
    _lib.set_print_callback_function(None, [0, 1, 2, 3])
    _lib.library_print("silence", 1)
    assert got[-1] == "still there"
