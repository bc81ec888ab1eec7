"""MFMA-busy fraction per kernel from two rocprofv3 counter passes of the same program:
   pass A: --pmc SQ_VALU_MFMA_BUSY_CYCLES     pass B: --pmc GRBM_GUI_ACTIVE
usage: python tools/pmc_mfma_busy.py <A counter_collection.csv> <B counter_collection.csv> [min_us]
MFMA-busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); dispatches are matched by order."""
import csv
import sys
from collections import OrderedDict, defaultdict


def load(path, counter):
    rows = OrderedDict()
    with open(path, newline="") as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] == counter:
                key = int(r["Dispatch_Id"])
                rows[key] = (r["Kernel_Name"], float(r["Counter_Value"]),
                             (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
    return [rows[k] for k in sorted(rows)]


a = load(sys.argv[1], "SQ_VALU_MFMA_BUSY_CYCLES")
b = load(sys.argv[2], "GRBM_GUI_ACTIVE")
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 200.0
assert len(a) == len(b), (len(a), len(b))
agg = defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for (ka, busy, _), (kb, gui, us) in zip(a, b):
    assert ka == kb
    if us < min_us:
        continue
    s = agg[ka]
    s[0] += 1
    s[1] += busy
    s[2] += gui
    s[3] += us
print("kernel,launches,total_ms,mfma_busy,clock_GHz")
for k, (n, busy, gui, us) in sorted(agg.items(), key=lambda kv: -kv[1][3]):
    short = k.replace("void wv::", "").split("(")[0]
    print(f'"{short}",{n},{us * 1e-3:.3f},{busy / (1024 * gui / 8):.4f},{gui / 8 / (us * 1e3):.3f}')
