"""Clock and MFMA-busy per kernel launch from ONE rocprofv3 pass with --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES:
clock = GRBM_GUI_ACTIVE / 8 XCDs / duration; busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles).
usage: python tools/pmc_clock.py <counter_collection.csv> [min_us]"""
import csv, sys
from collections import defaultdict
d = defaultdict(dict)
with open(sys.argv[1], newline="") as fh:
    for r in csv.DictReader(fh):
        k = int(r["Dispatch_Id"])
        d[k]["name"] = r["Kernel_Name"].replace("void wv::", "").split("(")[0]
        d[k]["us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
        d[k]["grid"] = int(r["Grid_Size"])
        d[k][r["Counter_Name"]] = float(r["Counter_Value"])
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 500.0
print("dispatch,kernel,grid,dur_us,clock_GHz,mfma_busy")
for k in sorted(d):
    r = d[k]
    if r["us"] < min_us or "GRBM_GUI_ACTIVE" not in r:
        continue
    cyc = r["GRBM_GUI_ACTIVE"] / 8.0
    print(f'{k},"{r["name"]}",{r["grid"]},{r["us"]:.1f},{cyc / (r["us"] * 1e3):.3f},{r.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * cyc):.3f}')
