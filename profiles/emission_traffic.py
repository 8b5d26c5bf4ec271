#!/usr/bin/env python3
"""profiles/emission_traffic.json from the PMC passes of profiles/pmc_passes.sh: HBM bytes per launch
of the emission kernel = 2 x FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md, HBM: on gfx950
FETCH_SIZE tallies a 128-byte request of a wide coalesced read as 64 bytes; WRITE_SIZE is exact for
16-byte-per-lane stores; both counters are in KB).  bench.py reads the file for roofline.traffic.
usage: emission_traffic.py <pmc_dir> <tag>"""
import collections
import csv
import glob
import json
import os
import sys

d, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
name = None
for f in sorted(glob.glob(f"{d}/pass*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "k_emission_sched" in r["Kernel_Name"] and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ghmm::", "")
fetch = sum(acc["FETCH_SIZE"]) / len(acc["FETCH_SIZE"])
write = sum(acc["WRITE_SIZE"]) / len(acc["WRITE_SIZE"])
traffic = int(round((2 * fetch + write) * 1024))
alg = 1032 * 300000
out = {
    "source": f"profiles/{tag}_pmc_sq_tcc.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, "
              "profiles/pmc_passes.sh = bench.py --steps 3 --warmup 1 --no-extras), profiles/emission_traffic.py",
    "kernel": name, "frames_per_launch": 300000,
    "FETCH_SIZE_KB": round(fetch, 1), "WRITE_SIZE_KB": round(write, 1),
    "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request of a wide coalesced read "
                  "(MI355X_MICROARCH.md, HBM) -> x2; the frame tiles are read with 16-byte-per-lane loads, the "
                  "access width that correction is documented for; WRITE_SIZE exact (16-byte-per-lane stores)",
    "traffic_bytes_per_launch": traffic,
    "note": f"algorithmic bytes per launch: {alg} (1 032 B x 300 000 frames): ratio {traffic / alg:.3f}; the excess is "
            "the frame tiles read twice where a wave's share of work units starts or ends inside a tile (12 waves "
            "x 256 blocks) and the B fragments per block (13 MB)",
}
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emission_traffic.json")
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out))
