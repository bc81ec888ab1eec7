"""Diagnostic: which (A fragment, K chunk) contributions of the M21-shaped product differ between the two K loops."""
import os, sys
sys.path.insert(0, ".")
import numpy as np
import _probes_lib  # noqa
import ctypes as C
from waveome_amd import _lib as L
from waveome_amd.engine import Context
ctx = Context(0)
fn = ctx._probe("wv_dbg_gemm_ex")
dp = C.POINTER(C.c_double)
fn.argtypes = [C.c_void_p, dp, dp, dp, dp] + [C.c_int32] * 4 + [C.c_double] * 2 + [C.c_int32] * 7
fn.restype = C.c_int32
rng = np.random.default_rng(3)
M = N = Kd = 128
nb = 1
for trial, (khi, tout, trimA) in enumerate([(1, 2, True), (1, 0, True), (1, 2, False), (0, 2, True)]):
    A = rng.normal(size=(nb, Kd, M)); B = rng.normal(size=(nb, Kd, N))
    kk, cc = np.arange(Kd)[:, None], np.arange(M)[None, :]
    if trimA: A = A * (kk <= cc)
    C0 = np.zeros((nb, M, N))
    res = {}
    for a in (1, 0):
        os.environ["WV_GEMM_ASM"] = str(a)
        out, outT = C0.copy(), np.zeros((nb, N, M))
        L.check(fn(ctx.handle, L.as_dp(A), L.as_dp(B), L.as_dp(out), L.as_dp(outT), M, N, Kd, 0, 1.0, 0.0, nb, 0, 0, khi, tout, 0, 0), "ex")
        res[a] = out
    D = res[1][0] - res[0][0]
    print(f"khi_mode={khi} tout={tout} trimmed A={trimA}: max |asm - hipcc| = {np.abs(D).max():.3e}")
    if np.abs(D).max() == 0: continue
    for f in range(8):       # 16-row fragments of the tile
        rows = slice(16 * f, 16 * f + 16)
        P = np.stack([A[0][16 * c:16 * c + 16, rows].T @ B[0][16 * c:16 * c + 16, :] for c in range(8)])   # [chunk][16][N]
        w, *_ = np.linalg.lstsq(P.reshape(8, -1).T, D[rows].reshape(-1), rcond=None)
        resid = np.abs(D[rows].reshape(-1) - P.reshape(8, -1).T @ w).max()
        print(f"  rows {16*f:3d}-{16*f+15:3d}: weights of chunks 0..7 in (asm - hipcc): {np.round(w, 3)}  residual {resid:.1e}")

print("---- tout = 2, where do the wrong rows come from?")
A = rng.normal(size=(1, Kd, M)); B = rng.normal(size=(1, Kd, N))
res = {}
for a in (1, 0):
    os.environ["WV_GEMM_ASM"] = str(a)
    out, outT = np.zeros((1, M, N)), np.zeros((1, N, M))
    L.check(fn(ctx.handle, L.as_dp(A), L.as_dp(B), L.as_dp(out), L.as_dp(outT), M, N, Kd, 0, 1.0, 0.0, 1, 0, 0, 0, 2, 0, 0), "ex")
    res[a] = (out[0], outT[0])
Ca, CTa = res[1]; Ch, CTh = res[0]
print("C  rows differing:", sorted(set(np.argwhere(Ca != Ch)[:, 0].tolist())))
print("CT (as rows of C) differing:", sorted(set(np.argwhere(CTa.T != CTh.T)[:, 0].tolist())))
bad = Ca[64:80]
for name, ref in (("hipcc rows 0-15", Ch[0:16]), ("hipcc rows 64-79", Ch[64:80]), ("zeros", np.zeros((16, N)))):
    print(f"   asm C[64:80] == {name}: {np.array_equal(bad, ref)}; equal entries {np.mean(bad == ref):.3f}")
cols = sorted(set(np.argwhere(Ca[64:80] != Ch[64:80])[:, 1].tolist()))
print("   columns differing in rows 64-79:", cols[:40], "..." if len(cols) > 40 else "")
# is each wrong entry some other entry of the correct result?
flat = {v: idx for idx, v in np.ndenumerate(Ch)}
hits = [(r, c, flat.get(Ca[r, c])) for r in range(64, 80) for c in range(0, N, 9)]
print("   sample (row, col) -> position of that value in the correct C:", hits[:24])
print("   full map rows 64-67, cols 0-15 -> position of that value in the correct C:")
for r in range(64, 68):
    print("    row", r, [flat.get(Ca[r, c]) for c in range(16)])
print("   cols 64-79 of row 64:", [flat.get(Ca[64, c]) for c in range(64, 80)])
