"""C++ TORCH_LIBRARY(open3d) shim (csrc/torch_shim/asr_open3d_ops.cpp): the ops reach the HIP kernels from a process
that never imports the Python registration -- eagerly and from a saved-and-loaded TorchScript module, which is how the
reference's C++ runs model.pt (cpp/lib/asr.cpp:315-326).  Runs in a child process: a schema can be defined once per
process and other tests import open3d.ml.torch."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CHILD = os.path.join(HERE, "torch_shim_child.py")


def _run(*args):
    env = dict(os.environ)
    env.pop("PYTHONPATH", None)
    return subprocess.run([sys.executable, CHILD, *args], capture_output=True, text=True, timeout=600, env=env)


def test_shim_library_loads_and_registers_the_four_ops():
    r = _run("load-only")
    assert r.returncode == 0 and "SHIM LOAD OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_shim_ops_match_the_oracle_eager_and_torchscript():
    r = _run()
    assert r.returncode == 0 and "SHIM OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
