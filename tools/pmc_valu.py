"""VALU / issue counters per kernel from two rocprofv3 counter passes of the same program:
   pass A: --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY
           SQ_ACTIVE_INST_ANY SQ_INSTS_SALU
   pass B: --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
usage: python tools/pmc_valu.py <A counter_collection.csv> <B counter_collection.csv> [name filter ...]
Per kernel (largest launch of each name): duration, clock (GRBM_GUI_ACTIVE / 8 / duration), resident waves per SIMD
(4 x SQ_WAVE_CYCLES / SIMD-cycles; SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_* count quad-cycles), share of wave time
issuing VALU / any instruction, waiting for issue, parked (s_waitcnt / barrier), VALU instructions per SIMD-cycle x 4
(an fp64 VALU instruction occupies its SIMD's issue for 4 cycles, so 1.0 = the fp64 VALU pipe is saturated)."""
import csv
import sys
from collections import defaultdict


def load(path):
    d = defaultdict(dict)
    with open(path, newline="") as fh:
        for r in csv.DictReader(fh):
            k = int(r["Dispatch_Id"])
            d[k]["name"] = r["Kernel_Name"]
            d[k]["us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
            d[k]["grid"] = int(r["Grid_Size"])
            d[k]["vgpr"] = int(r["VGPR_Count"])
            d[k][r["Counter_Name"]] = float(r["Counter_Value"])
    return [d[k] for k in sorted(d)]


A, B = load(sys.argv[1]), load(sys.argv[2])
filt = sys.argv[3:]
assert len(A) == len(B), (len(A), len(B))
best = {}
for a, b in zip(A, B):
    assert a["name"] == b["name"]
    n = a["name"].replace("void wv::", "").split("(")[0]
    if filt and not any(f in n for f in filt):
        continue
    if n not in best or a["grid"] > best[n][0]["grid"]:
        best[n] = (a, b)
print("kernel,grid_threads,vgpr,dur_ms,clock_GHz,waves_per_simd,valu_active,any_active,wait_issue,parked,"
      "valu_inst_x4_per_simd_cycle,lds_idx_active_per_cu_cycle,lds_bank_conflict")
for n, (a, b) in sorted(best.items(), key=lambda kv: -kv[1][0]["us"]):
    gui = b["GRBM_GUI_ACTIVE"] / 8.0                   # cycles of the launch (b's pass)
    simd_cycles = 1024.0 * gui * (a["us"] / b["us"])   # rescaled to pass A's duration
    wc = 4.0 * a["SQ_WAVE_CYCLES"]
    print(f'"{n}",{a["grid"]},{a["vgpr"]},{a["us"] * 1e-3:.3f},{gui / (b["us"] * 1e3):.3f},{wc / simd_cycles:.2f},'
          f'{4 * a["SQ_ACTIVE_INST_VALU"] / wc:.3f},{4 * a["SQ_ACTIVE_INST_ANY"] / wc:.3f},'
          f'{4 * a["SQ_WAIT_INST_ANY"] / wc:.3f},{4 * a["SQ_WAIT_ANY"] / wc:.3f},'
          f'{4 * a["SQ_INSTS_VALU"] / simd_cycles:.3f},{b["SQ_LDS_IDX_ACTIVE"] / (256 * gui):.3f},'
          f'{b["SQ_LDS_BANK_CONFLICT"]:.0f}')
