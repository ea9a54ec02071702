"""Per-layer time / algorithmic TFLOP/s of the sparse convolutions from one or two
*_sconv_trace.csv files (scripts/gpu_profile.sh), 10 M-point C3 sizes."""
import sys
sys.path[:0] = ['adaptive-surface-reconstruction_amd']
from asr_hip import synth


def load(path):
    out = []
    for line in open(path).read().splitlines()[1:]:
        f = line.rsplit(",", 5)
        if "split_reduce" in f[0] and out:  # second pass of a slot-range split: part of the layer before it
            out[-1] = (out[-1][0], out[-1][1] + float(f[3]))
        elif "sconv" in f[0]:
            out.append((f[0], float(f[3])))
    return out


files = [load(p) for p in sys.argv[1:]]
shapes = {n: s for n, s in synth.unet5_param_shapes(1).items() if len(s) == 3}
seq = ['encblock0.' + c for c in ('conv1a', 'conv1b', 'conv2', 'conv3', 'conv4')]
lev = [0] * 5
for l, dn in ((1, 'down1'), (2, 'down2'), (3, 'down3'), (4, 'down3')):
    seq += [dn + '.conv1a', dn + '.conv1b'] + ['encblock%d.%s' % (l, c) for c in ('conv1a', 'conv1b', 'conv2', 'conv3', 'conv4')]
    lev += [l] * 7
for l in (3, 2, 1, 0):
    seq += ['up%d.conv1' % l] + ['decblock%d.%s' % (l, c) for c in ('conv1', 'conv2', 'conv3', 'conv4')]
    lev += [l] * 5
pairs = [19838807, 4537981, 1179462, 329893, 89914]
vox = [2572109, 600041, 160994, 45823, 12566]
tot = [0.0] * len(files)
fused = len(files[0]) == 44  # conv1a + conv1b share a launch since the second filter bank
if fused:
    keep = [i for i, n in enumerate(seq) if not n.endswith('conv1b')]
    seq, lev = [seq[i] for i in keep], [lev[i] for i in keep]
for i, n in enumerate(seq):
    k, ci, co = shapes['sparseconv_' + n + '.kernel']
    if fused and n.endswith('conv1a'):
        co += 8
    L = lev[i]
    P = pairs[L] if k == 55 else (vox[L - 1] if 'down' in n else vox[L])
    fl = 2.0 * P * ci * co
    s = "%-18s L%d %3dx%3d " % (n, L, ci, co)
    for j, f in enumerate(files):
        t = f[i][1]
        tot[j] += t
        s += "| %-14s %7.1f us %5.1f TF " % (f[i][0][12:], t, fl / t / 1e6)
    print(s)
print("total ms", [round(t / 1e3, 2) for t in tot])
