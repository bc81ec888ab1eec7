"""profiles/rNN_traffic.json from two rocprofv3 counter passes of `tools/prof_eval.py <fits> 1`:
   pass A: --pmc FETCH_SIZE      pass B: --pmc WRITE_SIZE      (both in one pass exceeds the TCC counter slots)
usage: python tools/make_traffic_json.py <fetch counter_collection.csv> <write counter_collection.csv> <fits> <out.json>
FETCH_SIZE / WRITE_SIZE are reported in units of 1024 B; on gfx950 FETCH_SIZE counts the 128-byte requests of wide
streaming reads at 64 bytes, so it is doubled (MI355X_MICROARCH.md, HBM section).  Per kernel group the LARGEST
launches of the run are taken (the full-batch evaluation), summed over the symbols of the group."""
import csv
import json
import sys
from collections import defaultdict

GROUPS = {
    "lauum_gemm": ["gemm_f64_kernel<false, false, 128, 2, 0, 1>", "gemm_f64_kernel<false, false, 128, 2, 0, 5>"],
    "chol_trailing_syrk": ["gemm_f64_kernel<false, false, 128, 2, 0, 2>", "gemm_f64_kernel<false, false, 128, 2, 0, 6>"],
    "gram_assembly": ["gram_kernel<double"],   # (prefix: the template arguments changed between rounds)
    "grad_trace": ["trace_kernel<double"],
}


def load(path, counter):
    rows = defaultdict(lambda: [0, 0.0])  # symbol -> [grid, value] of the largest launch
    with open(path, newline="") as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].replace("void wv::", "").split("(")[0]
            grid, val = int(r["Grid_Size"]), float(r["Counter_Value"])
            if grid > rows[name][0] or (grid == rows[name][0] and val > rows[name][1]):
                rows[name] = [grid, val]
    return rows


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
fits = int(sys.argv[3])
N = 2048
alg = {"lauum_gemm": (8.0 * N * (N + 128) / 2, 8.0 * N * (N + 128) / 2),        # lower 128-tiles of M in, of K_y^-1 out
       "chol_trailing_syrk": (8.0 * (N / 2) ** 2 + 8.0 * (N / 2) * (N / 2 + 128) / 2, 8.0 * (N / 2) * (N / 2 + 128) / 2),
       "gram_assembly": (8.0 * N * 3, 8.0 * N * (N + 64) / 2),
       "grad_trace": (8.0 * N * (N + 64) / 2, 0.0)}
out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 "
                 f"tools/prof_eval.py {fits} 1; parsed by tools/make_traffic_json.py",
       "workload": f"headline N=2048, {fits} fits per launch",
       "correction": "FETCH_SIZE x 2 on gfx950 (MI355X_MICROARCH.md, HBM section), WRITE_SIZE as reported, units of 1024 B",
       "kernels": {}}
for g, pats in GROUPS.items():
    syms = sorted({s for s in list(fetch) + list(write) if any(s == p or (p.endswith("<double") and s.startswith(p)) for p in pats)})
    rd = sum(2.0 * fetch[s][1] * 1024.0 for s in syms if s in fetch)
    wr = sum(write[s][1] * 1024.0 for s in syms if s in write)
    out["kernels"][g] = {"symbols": [s for s in syms if s in fetch or s in write], "fits_per_launch": fits,
                         "read_bytes": rd, "write_bytes": wr,
                         "algorithmic_read_bytes": alg[g][0] * fits, "algorithmic_write_bytes": alg[g][1] * fits}
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps(out, indent=1))
