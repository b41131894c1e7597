#!/usr/bin/env python3
"""Summarise a `tools/pmc_traffic.sh` run into profiles/r03_bench_traffic.json: mean FETCH_SIZE per launch of the
bench's dominant kernel, doubled (gfx950 counts the 128-B requests of wide coalesced reads at 64 B; MI355X_MICROARCH.md,
HBM section).  usage: tools/pmc_traffic_summary.py <dir with *_counter_collection.csv> <kernel name substring> [out.json]"""
import csv, glob, json, os, sys

src, needle = sys.argv[1], sys.argv[2]
out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r03_bench_traffic.json")
files = glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)
assert files, "no counter_collection.csv under " + src
n, tot, name = 0, 0.0, None
for f in files:
    for row in csv.DictReader(open(f, newline="")):
        if needle in row["Kernel_Name"] and row["Counter_Name"] == "FETCH_SIZE":
            n += 1
            tot += float(row["Counter_Value"])
            name = row["Kernel_Name"]
assert n, "no launches of a kernel matching " + needle
kib = tot / n
res = {"kernel": name.replace("void (anonymous namespace)::", "").split("(")[0], "counter": "FETCH_SIZE (KiB)", "launches": n,
       "fetch_size_kib_per_launch": round(kib, 1), "gfx950_correction": 2.0, "traffic_bytes_per_launch": int(round(kib * 1024 * 2.0)),
       "source": "rocprofv3 --pmc FETCH_SIZE over `bench.py --no-graph` (tools/pmc_traffic.sh, tools/pmc_traffic_summary.py); "
                 "MI355X_MICROARCH.md HBM section: FETCH_SIZE counts the 128-B requests of 16-B/lane coalesced reads at 64 B on gfx950, doubled here"}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
