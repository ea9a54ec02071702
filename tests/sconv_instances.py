"""The k_sconv_mfma<NT,KC,IMP,WAVES,DUAL> template instances that `python bench.py` launches at 10 M
points (full-width UNet5), as listed by the committed rocprofv3 kernel trace of that command.  The GPU
parity tests force every one of them against the oracle (tests/test_gpu_sconv_variants.py)."""
import glob
import os
import re

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (NT, KC, IMP, WAVES, DUAL)
BENCH_INSTANCES = {
    (4, 32, 1, 8, 1), (4, 32, 0, 8, 0), (8, 16, 1, 8, 1), (8, 16, 0, 8, 0), (4, 32, 1, 4, 1),
    (4, 32, 0, 4, 0), (2, 64, 1, 4, 1), (2, 64, 0, 4, 0), (16, 16, 0, 8, 0), (2, 64, 0, 8, 0),
    (2, 32, 0, 8, 0),
}


# 16-bit kernel instances of the default bench (bf16x3 arithmetic since round 5; f16x2 in rounds 3-4 and as a sub-record
# now -- the launcher picks the same tiles for both): (NT, KC, IMP, WAVES, DUAL, MODE, PLAN) as the library's launch counters report them (template
# order is <NT, KC, WAVES, MODE, IMP, DUAL>; MODE 2 = bf16x3, 3 = f16x2; PLAN 1 = k_sconv_plan16, 0 = k_sconv_mfma16)
BENCH_SHAPES16 = {
    (4, 32, 0, 8, 1), (4, 32, 0, 8, 0), (8, 32, 0, 8, 1), (8, 32, 0, 8, 0),
    (8, 32, 0, 4, 1), (8, 32, 0, 4, 0), (2, 32, 0, 4, 1),
    (2, 32, 0, 8, 0),
}
# ... and the slot-range split (round 4: plain 55-slot layers of a grid of 2 048 .. 32 768 rows, level 4 at 10 M points;
# an eighth field, 1, in the launch counters; template argument SPLIT = true in the trace)
BENCH_SPLIT16 = {(8, 32, 0, 8, 0)}
MODE_ID = {"bf16x3": 2, "f16x2": 3}


def bench_instances16(mode):
    return {s + (MODE_ID[mode], 1) for s in BENCH_SHAPES16} | {s + (MODE_ID[mode], 1, 1) for s in BENCH_SPLIT16}


BENCH_INSTANCES16 = bench_instances16("bf16x3")  # what `python bench.py` launches


def instances16_in_trace(path):
    out = set()
    with open(path) as f:
        for line in f:
            m = re.match(r"k_sconv_(mfma|plan)16<(\d+), (\d+), (\d+), (\d+), (true|false), (true|false)(?:, (true|false))?>", line)
            if m:
                key = (int(m.group(2)), int(m.group(3)), int(m.group(6) == "true"), int(m.group(4)),
                       int(m.group(7) == "true"), int(m.group(5)), int(m.group(1) == "plan"))
                out.add(key + (1,) if m.group(8) == "true" else key)
    return out


def instances_in_trace(path):
    """set of instances named in a profiles/*_sconv_trace.csv (scripts/layer_table.py input)"""
    out = set()
    with open(path) as f:
        for line in f:  # the template arguments contain commas and the file is not quoted
            m = re.match(r"k_sconv_mfma<(\d+), (\d+), (true|false), (\d+), (true|false)>", line)
            if m:
                out.add((int(m.group(1)), int(m.group(2)), int(m.group(3) == "true"), int(m.group(4)),
                         int(m.group(5) == "true")))
    return out


def latest_trace(kind="f32"):
    """newest committed 10 M-point trace that contains launches of the f32 kernel / of the 16-bit kernel"""
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_10m_sconv_trace.csv")))
    probe = instances_in_trace if kind == "f32" else instances16_in_trace
    files = [f for f in files if probe(f)]
    return files[-1] if files else None
