"""Bit identity of the hand-scheduled K loop against hipcc's loop (diagnostic build, one process): plain products on
random data in both operand layouts, and a whole evaluation (M = L^-1, K_y^-1, loss, gradient) with the 128-wide
tiles forced, at a size where every partial-chunk stream runs (triangular operands, diagonal launches)."""
import os
import sys

sys.path.insert(0, ".")
SMALL_ONLY = len(sys.argv) > 1 and sys.argv[1] == "small"   # `small`: default tile thresholds - three fits run on the small 64 x 64 tile
if not SMALL_ONLY:
    os.environ.setdefault("WV_BN_FILL", "0")        # 128 x 128 tiles whatever the launch size
    os.environ.setdefault("WV_TRI_SPLIT_MIN", "1")
import numpy as np

import _probes_lib  # noqa: F401
from waveome_amd import ModelSpec
from waveome_amd import kernels as K
from waveome_amd.engine import Context, Problem

ctx = Context(0)
rng = np.random.default_rng(5)
bad = 0
for (nb, M, N, Kd, akc) in [] if SMALL_ONLY else [(9, 256, 384, 208, False), (9, 256, 384, 208, True), (3, 192, 320, 80, False), (2, 128, 128, 16, True),
                            (2, 130, 126, 48, False), (17, 640, 512, 1024, False)]:
    A = rng.normal(size=(nb, M, Kd))
    B = rng.normal(size=(nb, N, Kd))
    C0 = rng.normal(size=(nb, M, N))
    Ain = A if akc else np.ascontiguousarray(A.transpose(0, 2, 1))
    Bin = np.ascontiguousarray(B.transpose(0, 2, 1))
    out = {}
    for a in (1, 0):
        os.environ["WV_GEMM_ASM"] = str(a)
        out[a] = ctx.dbg_gemm(Ain, Bin, C0, a_kcontig=akc, b_kcontig=False, bn=128, alpha=-1.0, beta=1.0)
    ref = C0 - np.einsum("bmk,bnk->bmn", A, B)
    same = np.array_equal(out[1], out[0])
    err = np.max(np.abs(out[1] - ref))
    print(f"plain nb={nb} M={M} N={N} K={Kd} a_kc={akc}: asm == hipcc bitwise: {same}; max |asm - einsum| = {err:.2e}", flush=True)
    bad += (not same) or not (err < 1e-10)

# the structured forms of the pipeline, one launch each (diagnostic entry wv_dbg_gemm_ex): S = A21 M11' (K-contiguous A,
# k <= column, transposed output only), T = S M11 (k >= column, walked downwards), M21 = -M22 T (k <= row, both
# outputs), the trailing update (lower tiles, beta = 1) and K_y^-1 = M'M (lower tiles, k >= row, walked downwards)
import ctypes as C
from waveome_amd import _lib as L
fn = ctx._probe("wv_dbg_gemm_ex")
dp = C.POINTER(C.c_double)
fn.argtypes = [C.c_void_p, dp, dp, dp, dp] + [C.c_int32] * 4 + [C.c_double] * 2 + [C.c_int32] * 7
fn.restype = C.c_int32


def run_ex(A, B, C0, a_kc, alpha, beta, tri, klo, khi, tout, tag, same):
    nb, M, N = C0.shape
    Kd = A.shape[2] if a_kc else A.shape[1]
    out, outT = C0.copy(), np.zeros((nb, N, M))
    L.check(fn(ctx.handle, L.as_dp(A), L.as_dp(B), L.as_dp(out), L.as_dp(outT), M, N, Kd, int(a_kc), alpha, beta, nb, tri,
               klo, khi, tout, tag, int(same)), "wv_dbg_gemm_ex")
    return out, outT


forms = [("S", True, 1.0, 0.0, 0, 0, 2, 1, 0, False), ("T", False, 1.0, 0.0, 0, 2, 0, 0, 0, False),
         ("M21", False, -1.0, 0.0, 0, 0, 1, 2, 0, False), ("M21 plain output", False, -1.0, 0.0, 0, 0, 1, 0, 0, False),
         ("both outputs, full k", False, -1.0, 0.0, 0, 0, 0, 2, 0, False), ("update", False, -1.0, 1.0, 1, 0, 0, 0, 2, True),
         ("M'M", False, 1.0, 0.0, 1, 1, 0, 0, 1, True)]
for (name, a_kc, alpha, beta, tri, klo, khi, tout, tag, same) in ([] if SMALL_ONLY else forms):
    for (nb, M) in ((3, 128), (9, 256), (2, 512), (16, 384)):
        N = Kd = M
        A = rng.normal(size=(nb, M, Kd) if a_kc else (nb, Kd, M))
        B = rng.normal(size=(nb, Kd, N))
        # triangular operands where the product trims its contraction range (zeros exactly where the kernel skips)
        kk, cc = np.arange(Kd)[:, None], np.arange(M)[None, :]
        if name == "S": B = B * (kk <= cc)            # B(n, k) = B[k][n], zero for k > n
        if name == "T": B = B * (kk >= cc)            # zero for k < n
        if name.startswith("M21"): A = A * (kk <= cc)  # A(m, k) = A[k][m], zero for k > m
        if name == "M'M": A = A * (kk >= cc)          # zero for k < m
        C0 = rng.normal(size=(nb, M, N))
        res = {}
        for a in (1, 0):
            os.environ["WV_GEMM_ASM"] = str(a)
            res[a] = run_ex(A, B, C0, a_kc, alpha, beta, tri, klo, khi, tout, tag, same)
        Aop = A if a_kc else A.transpose(0, 2, 1)
        Bop = (A if same else B).transpose(0, 2, 1)
        ref = beta * C0 + alpha * np.einsum("bmk,bnk->bmn", Aop, Bop)
        got = res[1][1].transpose(0, 2, 1) if tout == 1 else res[1][0]
        mask = np.tril(np.ones((M, N), bool)) if tri else np.ones((M, N), bool)
        err = np.max(np.abs((got - ref)[:, mask]))
        # (lower-only products: nobody reads above the diagonal, and the diagonal-tile launch leaves it untouched)
        same_c = np.array_equal(res[1][0][:, mask], res[0][0][:, mask]) and np.array_equal(res[1][1], res[0][1])
        line = f"{name:20s} nb={nb} M={M}: asm == hipcc bitwise: {same_c}; max |asm - einsum| = {err:.2e}"
        if not same_c:
            d = (res[1][1].transpose(0, 2, 1) != res[0][1].transpose(0, 2, 1)) if tout == 1 else ((res[1][0] != res[0][0]) & mask)
            blk = d.reshape(nb, M // 64, 64, N // 64, 64).any(axis=(0, 2, 4))
            line += "\n   differing 64 x 64 blocks (rows down, columns across):\n   " + "\n   ".join("".join("X" if v else "." for v in row) for row in blk)
        print(line, flush=True)
        bad += (not same_c) or not (err < 1e-10)

for N in ((150, 704, 1500, 2048) if SMALL_ONLY else (704, 2048)):
    X = rng.normal(size=(N, 3))
    Y = np.stack([np.sin(X[:, 0] * (1 + 0.3 * p)) + 0.3 * X[:, 1] + 0.1 * rng.normal(size=N) for p in range(3)], axis=1)
    kern = K.Sum([K.SquaredExponential(active_dims=[0]), K.Matern32(active_dims=[1]),
                  K.Product([K.Periodic(K.SquaredExponential(active_dims=[0])), K.Lin(active_dims=[2])])])
    spec = ModelSpec(kern, 3, K.GaussianLikelihood(), K.ConstantMean())
    p = Problem(ctx, X, Y, spec, max_batch=3)
    U = spec.unconstrain()[None, :] + 0.2 * rng.normal(size=(3, spec.n_trainable))
    cols = np.arange(3, dtype=np.int32)
    th = spec.constrain(U[0])
    res = {}
    for a in (1, 0):
        os.environ["WV_GEMM_ASM"] = str(a)
        f, g, st = p.value_grad(cols, U)
        res[a] = dict(f=f, g=g, M=p.dbg_stage(0, th, 1), Kinv=p.dbg_stage(0, th, 2))
    for key in ("f", "g", "M", "Kinv"):
        if key == "Kinv":   # lower-only product: nobody reads above the diagonal (the diagonal-tile launch leaves it alone)
            res[1][key], res[0][key] = np.tril(res[1][key]), np.tril(res[0][key])
        same = np.array_equal(res[1][key], res[0][key])
        d = np.abs(np.asarray(res[1][key], dtype=float) - np.asarray(res[0][key], dtype=float))
        print(f"evaluation N={N}: {key}: asm == hipcc bitwise: {same}; max |diff| = {np.nanmax(d):.3e} of max |value| "
              f"{np.nanmax(np.abs(res[0][key])):.3e}; nan asm/hipcc {np.isnan(res[1][key]).sum()}/{np.isnan(res[0][key]).sum()}", flush=True)
        if not same and np.ndim(d) == 2:
            bad_rc = np.argwhere(d > 1e-9 * np.nanmax(np.abs(res[0][key])))
            if len(bad_rc):
                print("   first / last differing (row, col):", bad_rc[0], bad_rc[-1], "count", len(bad_rc),
                      " rows//64:", sorted(set((bad_rc[:, 0] // 64).tolist()))[:12], " cols//64:", sorted(set((bad_rc[:, 1] // 64).tolist()))[:12])
        bad += not same
    p.close()
print("FAILED" if bad else "all identical")
sys.exit(1 if bad else 0)
