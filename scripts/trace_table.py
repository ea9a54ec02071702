"""aggregates the LAST repetition in a rocprofv3 kernel trace csv: splits at occurrences of a marker kernel.
usage: python scripts/trace_table.py <kernel_trace.csv> <marker substring> [top]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marker = sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
starts = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
last = rows[starts[-1]:]
t0, t1 = int(last[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in last)
agg = collections.OrderedDict()
busy = 0
for r in last:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    if "rocprim" in n:  # keep the algorithm name and the key / value types
        import re
        m = re.search(r"wrapped_(\w+?)_config|detail::(\w+)_kernel", n)
        kt = re.search(r"(unsigned long|int|long|unsigned char|unsigned int)\*?,\s*(unsigned long|int|long|rocprim::[\w:]+)", n)
        n = "rocprim " + (m.group(1) or m.group(2) if m else "?") + (" <%s>" % kt.group(1) if kt else "")
    else:
        n = n.split("(")[0][:90]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg.setdefault(n, [0, 0.0])
    a[0] += 1
    a[1] += d
    busy += d
print("last repetition: %d kernels, wall %.2f ms, sum of kernel durations %.2f ms" % (len(last), (t1 - t0) / 1e6, busy / 1e3))
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print("%9.1f us %5d  %s" % (d, c, n))

# spans: wall time from each occurrence of kernel A to the end of the next occurrence of kernel B
SPANS = [("row_groups", "k_row_masks", "k_chunk_apply"), ("coarsen", "k_coarsen_count", "k_coarsen_up"),
         ("invert", "k_invert_hist", "k_invert_fill"), ("neighbors", "k_map_build", "k_neighbors_fill"),
         ("octree", "k_octree_insert_points", "k_collect_leaves"), ("search", "k_query_levels", "k_radius_place")]
names = [r["Kernel_Name"] for r in last]
for label, a, b in SPANS:
    total, cnt, i = 0.0, 0, 0
    while i < len(last):
        if a in names[i]:
            j = i
            while j < len(last) and b not in names[j]:
                j += 1
            if j < len(last):
                total += (int(last[j]["End_Timestamp"]) - int(last[i]["Start_Timestamp"])) / 1e3
                cnt += 1
                i = j
        i += 1
    print("span %-12s %3d x  total %9.1f us" % (label, cnt, total))
