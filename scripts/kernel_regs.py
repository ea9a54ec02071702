#!/usr/bin/env python3
"""VGPR / AGPR / LDS / scratch of the gfx950 kernels in a hipcc object or shared library (reads the code-object notes).
usage: scripts/kernel_regs.py <file.o|.so> [substring ...]   -- prints the kernels whose demangled name holds every substring"""
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin/"


def kernels(path):
    with tempfile.TemporaryDirectory() as d:
        fat = d + "/fat.bin"
        subprocess.check_call([LLVM + "llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", path, fat])
        dev = d + "/dev.o"
        subprocess.check_call([LLVM + "clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                               "--input=" + fat, "--output=" + dev, "--unbundle"])
        notes = subprocess.check_output([LLVM + "llvm-readelf", "--notes", dev]).decode()
    # amdhsa.kernels is a YAML list: an entry starts with "  - .agpr_count:" (keys in alphabetical order), argument entries
    # with a deeper "      - ." -- the kernel entries are the "- ." items at the shallowest indentation after "amdhsa.kernels:"
    out, cur, depth, inside = [], None, None, False
    for line in notes.splitlines():
        if line.strip().startswith("amdhsa.kernels:"):
            inside = True
            continue
        if not inside:
            continue
        m = re.match(r"(\s*)(-\s+)?\.(\w+):\s*(.*)", line)
        if not m:
            if line.strip().startswith("amdhsa."):
                inside = False
            continue
        ind, item, k, v = len(m.group(1)), m.group(2), m.group(3), m.group(4).strip()
        if item and (depth is None or ind <= depth):
            depth = ind
            cur = {}
            out.append(cur)
        if cur is None or ind > depth + 2:
            continue
        if k == "name":
            cur["name"] = v
        elif k in ("vgpr_count", "agpr_count", "sgpr_count", "group_segment_fixed_size", "private_segment_fixed_size",
                   "vgpr_spill_count"):
            cur[k] = int(v)
    out = [k for k in out if "name" in k]
    names = subprocess.run(["c++filt"], input="\n".join(k["name"] for k in out).encode(),
                           stdout=subprocess.PIPE).stdout.decode().splitlines()
    for k, n in zip(out, names):
        k["demangled"] = re.sub(r"\(anonymous namespace\)::", "", n).split("(")[0].replace("void ", "")
    return out


if __name__ == "__main__":
    for k in kernels(sys.argv[1]):
        if all(s in k["demangled"] for s in sys.argv[2:]):
            print("%-70s vgpr %3d agpr %3d lds %6d scratch %4d spill %d" % (
                k["demangled"][:70], k.get("vgpr_count", -1), k.get("agpr_count", 0), k.get("group_segment_fixed_size", 0),
                k.get("private_segment_fixed_size", 0), k.get("vgpr_spill_count", 0)))
