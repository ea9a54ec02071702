"""timeline of the LAST repetition in a rocprofv3 kernel trace of scripts/prof_geom.py: every kernel in start
order with its offset, duration and the idle gap in front of it (host round trips show up as gaps).
usage: python scripts/geom_timeline.py <kernel_trace.csv> [marker substring = k_octree_insert_points]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marker = sys.argv[2] if len(sys.argv) > 2 else "k_octree_insert_points"
starts = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
last = rows[starts[-1]:]
t0 = int(last[0]["Start_Timestamp"])
end_prev = t0
tot_gap = 0.0
for r in last:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    if "rocprim" in n:
        m = re.search(r"wrapped_(\w+?)_config|detail::(\w+)_kernel", n)
        n = "rocprim " + ((m.group(1) or m.group(2)) if m else "?")
    else:
        n = n.split("(")[0][:60]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - end_prev) / 1e3
    if gap > 0:
        tot_gap += gap
    print("%9.1f us  +%7.1f us  gap %7.1f  q%s  %s  [%s]" % ((s - t0) / 1e3, (e - s) / 1e3, gap, r.get("Queue_Id", "?"), n,
                                                         r.get("Grid_Size_X", "")))
    end_prev = max(end_prev, e)
print("kernels %d, wall %.1f us, idle gaps %.1f us" % (len(last), (end_prev - t0) / 1e3, tot_gap))
