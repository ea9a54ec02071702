"""profiles/<tag>_fetch.csv + <tag>_write.csv (scripts/gpu_pmc.sh aggregates) -> <tag>_pmc_traffic.json:
HBM bytes per sparse-conv launch, corrected as MI355X_MICROARCH.md prescribes for gfx950
(FETCH_SIZE counts half of a 16 B/lane read; both counters are in KiB)."""
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 44
precision = sys.argv[3] if len(sys.argv) > 3 else "bf16x3"  # arithmetic of the profiled bench command


def total(path, col):
    lines = open(path).read().splitlines()
    names = lines[0].split("|")
    i = names.index(col)
    return sum(float(l.split("|")[i]) for l in lines[1:] if l.startswith(("k_sconv_mfma", "k_sconv_plan"))), \
        sum(int(l.split("|")[1]) for l in lines[1:] if l.startswith(("k_sconv_mfma", "k_sconv_plan")))


fetch, n1 = total("profiles/%s_fetch.csv" % tag, "FETCH_SIZE")
write, n2 = total("profiles/%s_write.csv" % tag, "WRITE_SIZE")
assert n1 == n2 == launches, (n1, n2, launches)
hbm = (2 * fetch + write) * 1024
json.dump({"kernel": "k_sconv_plan16" if precision != "f32" else "k_sconv_mfma", "precision": precision, "launches": launches, "fetch_size_kb": fetch, "write_size_kb": write,
           "hbm_bytes_per_forward": hbm, "hbm_bytes_per_launch": hbm / launches, "points": 10_000_000,
           "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of `bench.py --steps 1 --warmup 0` "
                   "(10 M points); bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: gfx950 FETCH_SIZE reports half of a "
                   "16 B/lane read (MI355X_MICROARCH.md, HBM section); WRITE_SIZE uncalibrated"},
          open("profiles/%s_pmc_traffic.json" % tag, "w"), indent=1)
print(open("profiles/%s_pmc_traffic.json" % tag).read())
