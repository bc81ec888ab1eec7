"""How a job's tasks spread over R ranks under the dealing rules of waveome_amd/parallel.py, from the MEASURED evaluation
counts of a one-GPU run (tools/run_c4.py with WV_C4_TABLE=...: a fit's cost is its number of value+gradient evaluations).

Model of a rank (the batched device path): it holds at most `cap` tasks (queued + in flight), `slots` of them in flight;
every live fit advances one evaluation per ROUND and a round takes a + b x (fits in flight) ms - a = the latency floor of
a nearly empty batch, b = the per-fit cost of a full one (defaults: the N = 1500 figures of profiles/r03_*).
  static   tasks sorted by the a-priori cost proxy (trainable parameters of the structure), dealt round-robin up front
  dynamic  the list, heaviest structures first, is dealt in chunks on demand: a rank draws whenever it holds fewer than
           min(slots + slots / 4, share x fair share) tasks (engine.Problem.fit_stream's intake rule)
Usage: python tools/simulate_dealing.py TABLE.csv [ranks=8] [slots=512] [replicate=1]"""
import sys

import numpy as np
import pandas as pd

tab = pd.read_csv(sys.argv[1])
R = int(sys.argv[2]) if len(sys.argv) > 2 else 8
SLOTS = int(sys.argv[3]) if len(sys.argv) > 3 else 512
REP = int(sys.argv[4]) if len(sys.argv) > 4 else 1
A_MS, B_MS = 0.75, 0.055  # ms per round: floor, and per fit in flight (N = 1500: 512 fits ~ 29 ms per round)
rng = np.random.default_rng(0)
cost = np.tile(tab["nfev"].to_numpy(dtype=np.int64), REP)
proxy = np.tile(tab["structure"].map({"lin": 7, "se": 12, "m12": 12, "m32": 12, "m52": 12, "per": 17}).to_numpy(float), REP)
T = len(cost)
order = np.argsort(-proxy, kind="stable")


class Rank:
    def __init__(self):
        self.t = 0.0
        self.queue = []      # evaluations left of queued tasks (arrival order)
        self.live = []       # evaluations left of the fits in flight
        self.done = 0

    def held(self):
        return len(self.queue) + len(self.live)

    def round(self):
        while self.queue and len(self.live) < SLOTS:
            self.live.append(self.queue.pop(0))
        if not self.live:
            return
        self.t += A_MS + B_MS * len(self.live)
        self.live = [c - 1 for c in self.live if c > 1]


def run(dynamic, chunk=0, share=1.0, strided=False):
    ranks = [Rank() for _ in range(R)]
    nxt = 0
    seq = order
    if strided and chunk:  # chunk j = every n_chunks-th task of the sorted list: each chunk a cross-section of the job
        nch = -(-T // chunk)
        seq = np.concatenate([order[j::nch] for j in range(nch)])
    fair = -(-T // R)
    cap = min(SLOTS + SLOTS // 4, max(chunk, int(np.ceil(share * fair)))) if dynamic else None
    if not dynamic:
        for r in range(R):
            ranks[r].queue = cost[np.sort(order[r::R])].tolist()
    while True:
        # the rank that is furthest behind in time acts next
        r = min(range(R), key=lambda q: ranks[q].t if (ranks[q].held() or (dynamic and nxt < T)) else np.inf)
        rk = ranks[r]
        if not (rk.held() or (dynamic and nxt < T)):
            break
        if dynamic:
            while nxt < T and rk.held() < cap:
                ids = seq[nxt:nxt + chunk]
                nxt += len(ids)
                rk.queue += cost[ids].tolist()
            if not rk.held():
                rk.t = min(q.t for q in ranks if q.held()) if any(q.held() for q in ranks) else rk.t
                continue
        rk.round()
    times = np.array([rk.t for rk in ranks]) / 1e3
    return times


print(f"{T} tasks on {R} ranks x {SLOTS} slots; evaluations per task: mean {cost.mean():.0f}, max {cost.max()}, by structure "
      f"{tab.groupby('structure')['nfev'].mean().round(0).astype(int).to_dict()}")
t = run(False)
print(f"static, sorted by parameter count, round-robin    : rank seconds {np.round(t, 1).tolist()}  wall {t.max():6.1f} s  max/mean {t.max() / t.mean():.3f}")
for strided in (False, True):
    for share in (1.0, 0.75):
        for chunk in (SLOTS // 4, SLOTS // 16):
            t = run(True, chunk, share, strided)
            print(f"dynamic, {'cross-section' if strided else 'contiguous   '} chunks of {chunk:3d}, first fill {share:4.2f} x fair share: "
                  f"rank seconds {np.round(t, 1).tolist()}  wall {t.max():6.1f} s  max/mean {t.max() / t.mean():.3f}")
