#!/bin/bash
# usage: scripts/ab_levels.sh <lib.so> [<lib.so> ...]   (on the GPU box)
# per-level U-Net kernel time of each build of the library, same box, from rocprofv3 kernel traces of scripts/unet_time.py
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  tag=$(basename $lib .so)
  out=/tmp/abl_$tag; rm -rf $out
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out -- python $root/scripts/unet_time.py --lib $root/$lib --steps 3 $ASR_AB_ARGS > /tmp/abl_$tag.log 2>&1
  t=$(find $out -name "*kernel_trace.csv" | head -1)
  python3 - "$t" "$tag" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_sconv" in r["Kernel_Name"]]
n = len(rows) // 5   # 2 warm-up + 3 timed forwards
last = rows[-n:]
by = collections.OrderedDict()
for r in last:
    name = r["Kernel_Name"].split("k_sconv_")[1].split("(")[0][:44]
    key = (name, r["Grid_Size_X"])
    by.setdefault(key, [0, 0.0])
    by[key][0] += 1
    by[key][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in by.values())
print("%s: %d launches, %.2f ms" % (sys.argv[2], len(last), tot / 1e3))
for (name, gx), (c, us) in by.items():
    print("   %-46s grid %9s x%2d %8.1f us" % (name, gx, c, us))
PY
done
