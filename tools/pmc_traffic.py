"""Per-launch HBM traffic of one kernel from a rocprofv3 `--pmc FETCH_SIZE WRITE_SIZE` counter_collection CSV.

usage: python tools/pmc_traffic.py <counter_collection.csv> <kernel-name substring> [out.json]
FETCH_SIZE / WRITE_SIZE are in KiB-units of 1 KB as rocprofv3 reports them; on gfx950 FETCH_SIZE counts 128-byte
requests at 64 bytes, so it is doubled (MI355X_MICROARCH.md, section HBM)."""
import csv
import json
import sys
from collections import defaultdict

path, needle = sys.argv[1], sys.argv[2]
per_dispatch = defaultdict(dict)
with open(path, newline="") as fh:
    for row in csv.DictReader(fh):
        if needle in row["Kernel_Name"]:
            per_dispatch[row["Dispatch_Id"]][row["Counter_Name"]] = float(row["Counter_Value"])
n = len(per_dispatch)
fetch = sum(d.get("FETCH_SIZE", 0.0) for d in per_dispatch.values()) / max(n, 1)
write = sum(d.get("WRITE_SIZE", 0.0) for d in per_dispatch.values()) / max(n, 1)
out = {"kernel": needle, "launches": n, "FETCH_SIZE_KB_raw_avg": fetch, "WRITE_SIZE_KB_avg": write,
       "read_bytes_per_launch": 2.0 * fetch * 1024.0, "write_bytes_per_launch": write * 1024.0,
       "traffic_bytes_per_launch": 2.0 * fetch * 1024.0 + write * 1024.0,
       "correction": "FETCH_SIZE x 2 on gfx950; WRITE_SIZE as reported; both x 1024 B"}
print(json.dumps(out, indent=1))
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
