"""Block structure of longitudinal designs (structures of waveome/regularization.py:103-149).

CPU: which structures qualify (`wv_spec_block_column`) and the span tables (`wv_block_spans`).
GPU (`-m gpu`): the zero-block-skipping products give the SAME BITS as the dense path, and the oracle's numbers."""
import ctypes as C
import os

import numpy as np
import pytest

from waveome_amd import ModelSpec, _lib
from waveome_amd import kernels as K


def _spec(kern, D=4):
    return ModelSpec(kern, D, K.GaussianLikelihood(), K.ConstantMean())


def _cat(d):
    k = K.Categorical(active_dims=[d])
    return k


def test_block_column_of_structures():
    unit, other = 3, 2
    se = lambda d: K.SquaredExponential(active_dims=[d])
    # every term carries Categorical[unit]: unit main effect + unit x numeric interactions
    k = K.Sum([_cat(unit), K.Product([_cat(unit), se(0)]), K.Product([_cat(unit), K.Lin(active_dims=[1])])])
    assert _spec(k).block_column() == unit
    # a dense main effect breaks it; so does a different categorical
    assert _spec(K.Sum([_cat(unit), se(0)])).block_column() == -1
    assert _spec(K.Sum([_cat(unit), K.Product([_cat(other), se(0)])])).block_column() == -1
    # terms with both categoricals: the lowest common column is reported
    both = K.Sum([K.Product([_cat(unit), _cat(other), se(0)]), K.Product([_cat(other), _cat(unit)])])
    assert _spec(both).block_column() == other
    # a term that is identically zero (Empty factor) does not count
    assert _spec(K.Sum([K.Product([_cat(unit), se(0)]), K.Empty()])).block_column() == unit
    # a lone product, as the kernel search proposes them
    assert _spec(K.Product([_cat(unit), K.Matern32(active_dims=[1])])).block_column() == unit
    assert _spec(se(0)).block_column() == -1


def _spans(X, col):
    X = np.ascontiguousarray(X, dtype=np.float64)
    N, D = X.shape
    out = np.zeros(2 * ((N + 63) // 64), dtype=np.int32)
    _lib.check(_lib.load().wv_block_spans(_lib.as_dp(X), N, D, col, out.ctypes.data_as(C.POINTER(C.c_int32))),
               "wv_block_spans")
    return out.reshape(-1, 2)


def test_block_spans_sorted_unsorted_and_padding():
    # 5 units of 50 rows, sorted: granule g = rows 64 g .. 64 g + 63
    N = 250
    X = np.zeros((N, 2))
    X[:, 1] = np.repeat(np.arange(5), 50)
    sp = _spans(X, 1)
    assert sp.shape == (4, 2)
    # granule 0 holds units 0 (rows 0-49) and 1 (50-99) -> [0, 100) -> rounded outwards to 16: [0, 112)
    assert tuple(sp[0]) == (0, 112)
    # granule 1 = rows 64..127: units 1 (50-99) and 2 (100-149) -> [50, 150) -> [48, 160)
    assert tuple(sp[1]) == (48, 160)
    # last granule = rows 192..255: unit 3 (150-199), unit 4 (200-249), then 6 padding rows (each its own block)
    assert tuple(sp[3]) == (144, 256)
    # every row's range contains the row itself and all rows of its label
    for i in range(N):
        lab = np.flatnonzero(X[:, 1] == X[i, 1])
        assert sp[i // 64, 0] <= lab.min() and lab.max() < sp[i // 64, 1]
    # scattered labels: wide ranges, still covering
    rng = np.random.default_rng(0)
    Xs = X[rng.permutation(N)]
    sps = _spans(Xs, 1)
    for i in range(N):
        lab = np.flatnonzero(Xs[:, 1] == Xs[i, 1])
        assert sps[i // 64, 0] <= lab.min() and lab.max() < sps[i // 64, 1]
    assert (sps[:, 1] - sps[:, 0]).min() > 150
    # labels are compared after rint, like the Categorical kernel (kernels.py:19-32): 0.9 and 1.1 are one label
    Xr = X.copy()
    Xr[:, 1] += np.where(np.arange(N) % 2, 0.1, -0.1)
    assert np.array_equal(_spans(Xr, 1), sp)


# ------------------------------------------------------------------------------------------------ GPU
def _longitudinal(seed, units, per_unit, sort=True):
    rng = np.random.default_rng(seed)
    N = units * per_unit
    uid = np.repeat(np.arange(units), per_unit).astype(float)
    t = np.tile(np.linspace(-1.5, 1.5, per_unit), units) + 0.05 * rng.normal(size=N)
    z = rng.normal(size=N)
    X = np.column_stack([t, z, uid])
    Y = np.stack([np.sin(2 * t + 0.3 * uid) + 0.3 * z + 0.1 * uid + 0.1 * rng.normal(size=N) for _ in range(3)], axis=1)
    if not sort:
        perm = rng.permutation(N)
        X, Y = X[perm], Y[perm]
    return X, Y


def _block_kernel():
    u = lambda: K.Categorical(active_dims=[2])
    return K.Sum([u(), K.Product([u(), K.SquaredExponential(active_dims=[0])]),
                  K.Product([u(), K.Lin(active_dims=[1])]),
                  K.Product([u(), K.Periodic(K.SquaredExponential(active_dims=[0]))])])


@pytest.mark.gpu
@pytest.mark.parametrize("sort", [True, False])
@pytest.mark.parametrize("units,per_unit", [(12, 40), (7, 150), (16, 64), (5, 128), (6, 192)])
def test_block_skipping_is_bit_identical_to_dense_and_matches_oracle(units, per_unit, sort):
    from oracle import gp_oracle as o
    from waveome_amd.engine import Context, Problem

    # (units of 64 / 128 rows end on the boundaries of the 128-row recursion nodes: whole nodes decouple - the recursion
    #  drops S, the update, T and M21 there and factors runs of independent 128-nodes in one launch; 192-row units
    #  decouple at every third boundary only)
    X, Y = _longitudinal(3, units, per_unit, sort)
    spec = ModelSpec(_block_kernel(), 3, K.GaussianLikelihood(), K.ConstantMean())
    assert spec.block_column() == 2
    rng = np.random.default_rng(1)
    B = 3
    U = spec.unconstrain()[None, :] + 0.3 * rng.normal(size=(B, spec.n_trainable))
    cols = np.arange(B, dtype=np.int32) % Y.shape[1]
    ctx = Context(0)
    out = {}
    for mode in ("1", "0"):
        os.environ["WV_BLOCKS"] = mode
        try:
            p = Problem(ctx, X, Y, spec, max_batch=B)
        finally:
            os.environ.pop("WV_BLOCKS", None)
        f, g, st = p.value_grad(cols, U)
        lml, _ = p.lml(cols, np.stack([spec.constrain(u) for u in U]))
        Minv = p.dbg_stage(0, spec.constrain(U[0]), 1)
        Kinv = p.dbg_stage(0, spec.constrain(U[0]), 2)
        out[mode] = (f, g, lml, Minv, Kinv)
        assert np.all(st == 0)
        p.close()
    ctx.close()
    for a, b in zip(out["1"], out["0"]):
        np.testing.assert_array_equal(a, b)  # skipping exact zeros changes no bit
    d = spec.to_dict()
    for b in range(B):
        fo, go = o.objective(d, U[b], X, Y[:, cols[b]])
        assert abs(out["1"][0][b] - fo) <= 1e-10 * max(1.0, abs(fo))
        assert np.max(np.abs(out["1"][1][b] - go)) <= 1e-7 * max(1.0, np.max(np.abs(go)))


_BLOCK_REGIME_SCRIPT = r"""
import sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import numpy as np
from test_blocks import _longitudinal, _block_kernel
from waveome_amd import ModelSpec, kernels as K
from waveome_amd.engine import Context, Problem
units, per_unit, sort = int(sys.argv[3]), int(sys.argv[4]), sys.argv[5] == "1"
X, Y = _longitudinal(3, units, per_unit, sort)
spec = ModelSpec(_block_kernel(), 3, K.GaussianLikelihood(), K.ConstantMean())
rng = np.random.default_rng(1)
B = 3
U = spec.unconstrain()[None, :] + 0.3 * rng.normal(size=(B, spec.n_trainable))
cols = np.arange(B, dtype=np.int32) % Y.shape[1]
ctx = Context(0)
p = Problem(ctx, X, Y, spec, max_batch=B)
f, g, st = p.value_grad(cols, U)
lml, _ = p.lml(cols, np.stack([spec.constrain(u) for u in U]))
th = spec.constrain(U[0])
np.savez(sys.argv[2], f=f, g=g, st=st, lml=lml, M=p.dbg_stage(0, th, 1), Kinv=np.tril(p.dbg_stage(0, th, 2)))
"""


@pytest.mark.gpu
@pytest.mark.parametrize("units,per_unit,sort", [(7, 150, True), (12, 40, True), (9, 100, False), (16, 64, True), (6, 192, True)])
def test_block_skipping_on_the_wide_hand_scheduled_tiles(tmp_path, units, per_unit, sort):
    """three fits select the small 64 x 64 tile, so the test above never reaches the 128-wide kernels whose
    waves own SPREAD fragments (the union of two 64-row and two 64-column granules' spans, then the trimming of the
    hand-scheduled streams) - the regime of a search with hundreds of fits in flight.  The tile choice is read once per
    process: child processes force the 128-wide tiles (WV_BN_FILL=0, WV_TRI_SPLIT_MIN=1) with and without the span
    tables and are compared with the default regime, bit for bit - also for N that is not a multiple of 128
    (7 x 150 = 1050 -> 1088 padded rows) and for scattered labels."""
    import subprocess
    import sys
    from pathlib import Path

    root = str(Path(__file__).resolve().parents[1])
    script = tmp_path / "block_regime.py"
    script.write_text(_BLOCK_REGIME_SCRIPT)
    wide = {"WV_BN_FILL": "0", "WV_TRI_SPLIT_MIN": "1"}
    variants = (("default", {}), ("default dense", {"WV_BLOCKS": "0"}), ("wide", wide),
                ("wide dense", dict(wide, WV_BLOCKS="0")), ("wide one launch", dict(wide, WV_TRI_SPLIT="0")),
                ("128 x 64", {"WV_SMALL_FILL": "0"}))
    outs = {}
    for name, extra in variants:
        out = tmp_path / (name.replace(" ", "_") + ".npz")
        subprocess.run([sys.executable, str(script), root, str(out), str(units), str(per_unit), "1" if sort else "0"],
                       check=True, env=dict(os.environ, **extra), timeout=600)
        outs[name] = np.load(out)
    a = outs["default"]
    assert np.all(a["st"] == 0)
    for name, _ in variants[1:]:
        for key in ("f", "g", "lml", "M", "Kinv"):
            np.testing.assert_array_equal(a[key], outs[name][key], err_msg=f"{name}: {key}")


@pytest.mark.gpu
@pytest.mark.parametrize("units,per_unit", [(12, 40), (7, 150), (16, 64)])
def test_block_skipping_on_a_float32_context(units, per_unit):
    """float32 contexts used to ignore the span tables.  The float32 GEMM now trims its contraction
    ranges like the float64 one (flush points of its float64 partial sums sit on the absolute chunk grid, so leaving out
    exact zeros changes no bit): value, gradient, M and K_y^-1 equal the dense float32 path bit for bit, and the float64
    oracle to the float32 path's tolerances."""
    from oracle import gp_oracle as o
    from waveome_amd.engine import Context, Problem

    X, Y = _longitudinal(3, units, per_unit, True)
    spec = ModelSpec(_block_kernel(), 3, K.GaussianLikelihood(), K.ConstantMean())
    rng = np.random.default_rng(1)
    B = 3
    U = spec.unconstrain()[None, :] + 0.3 * rng.normal(size=(B, spec.n_trainable))
    cols = np.arange(B, dtype=np.int32) % Y.shape[1]
    ctx = Context(0, dtype="float32")
    out = {}
    for mode in ("1", "0"):
        os.environ["WV_BLOCKS"] = mode
        try:
            p = Problem(ctx, X, Y, spec, max_batch=B)
        finally:
            os.environ.pop("WV_BLOCKS", None)
        f, g, st = p.value_grad(cols, U)
        assert np.all(st == 0)
        out[mode] = (f, g, p.dbg_stage(0, spec.constrain(U[0]), 1), np.tril(p.dbg_stage(0, spec.constrain(U[0]), 2)))
        p.close()
    ctx.close()
    for a, b in zip(out["1"], out["0"]):
        np.testing.assert_array_equal(a, b)
    d = spec.to_dict()
    for b in range(B):
        fo, go = o.objective(d, U[b], X, Y[:, cols[b]])
        assert abs(out["1"][0][b] - fo) <= 1e-6 * max(1.0, abs(fo))
        assert np.max(np.abs(out["1"][1][b] - go)) <= 1e-4 * max(1.0, np.max(np.abs(go)))


@pytest.mark.gpu
def test_block_and_dense_structures_share_a_batch():
    """A multi-structure work queue mixes block-diagonal and dense candidates in one launch (as the kernel search
    does): every slot follows its own structure's table."""
    from waveome_amd.engine import Context, Problem, lbfgs_opts

    X, Y = _longitudinal(5, 10, 60)
    u = lambda: K.Categorical(active_dims=[2])
    kerns = [K.Product([u(), K.SquaredExponential(active_dims=[0])]),            # block diagonal
             K.Sum([K.SquaredExponential(active_dims=[0]), u()]),                   # dense
             K.Sum([u(), K.Product([u(), K.Matern32(active_dims=[1])])])]          # block diagonal
    specs = [ModelSpec(k, 3, K.GaussianLikelihood(), K.ConstantMean()) for k in kerns]
    assert [s.block_column() for s in specs] == [2, -1, 2]
    ctx = Context(0)
    opts = lbfgs_opts(maxiter=25)
    tasks_spec = np.array([0, 1, 2, 0, 1, 2], dtype=np.int32)
    tasks_col = np.array([0, 0, 0, 1, 1, 1], dtype=np.int32)
    width = max(s.n_trainable for s in specs)
    U0 = np.zeros((6, width))
    for t, si in enumerate(tasks_spec):
        U0[t, :specs[si].n_trainable] = specs[si].unconstrain()
    pm = Problem(ctx, X, Y, specs, max_batch=6)
    rm = pm.fit_multi(tasks_col, tasks_spec, U0, opts)
    pm.close()
    for t, si in enumerate(tasks_spec):
        p1 = Problem(ctx, X, Y, specs[si], max_batch=1)
        r1 = p1.fit(tasks_col[t:t + 1], U0[t:t + 1, :specs[si].n_trainable], opts)
        p1.close()
        np.testing.assert_array_equal(rm["loss"][t], r1["loss"][0])
        np.testing.assert_array_equal(rm["u"][t, :specs[si].n_trainable], r1["u"][0])
    ctx.close()


@pytest.mark.gpu
def test_decoupled_recursion_gives_way_when_a_dense_structure_is_added():
    """Round 5: while EVERY structure of a problem is block-diagonal with units ending on the 128-row node boundaries, the
    recursion launches no S / update / T / M21 for the decoupled nodes (M21 stays the zero block it was created as) and
    factors runs of independent 128-nodes in one launch.  The moment a dense structure joins the problem
    (`wv_problem_add_spec`, as the kernel search does) every product runs again, for good - and the fits of the
    block-diagonal structure before and after the switch, and beside the dense one, are the same bits as on a problem of
    their own."""
    from waveome_amd.engine import Context, Problem, lbfgs_opts

    X, Y = _longitudinal(5, 8, 128)                      # 1024 rows: eight units, one per 128-node pair boundary
    u = lambda: K.Categorical(active_dims=[2])
    block = ModelSpec(K.Sum([u(), K.Product([u(), K.SquaredExponential(active_dims=[0])])]), 3, K.GaussianLikelihood(),
                      K.ConstantMean())
    dense = ModelSpec(K.Sum([K.SquaredExponential(active_dims=[0]), u()]), 3, K.GaussianLikelihood(), K.ConstantMean())
    assert block.block_column() == 2 and dense.block_column() == -1
    ctx = Context(0)
    opts = lbfgs_opts(maxiter=12)
    caps = {"max_specs": 2, "max_params": 8, "max_feat": 4, "max_factors": 4, "max_red": 8}
    width = max(block.n_trainable, dense.n_trainable)
    cols = np.array([0, 1, 2], dtype=np.int32)

    def u0(spec, n):
        U = np.zeros((n, width))
        U[:, :spec.n_trainable] = spec.unconstrain()
        return U

    alone = Problem(ctx, X, Y, [block], max_batch=3, capacity=caps)          # decoupled throughout
    ref_block = alone.fit_multi(cols, np.zeros(3, dtype=np.int32), u0(block, 3), opts)
    alone.close()
    both = Problem(ctx, X, Y, [block, dense], max_batch=3, capacity=caps)    # never decoupled
    ref_dense = both.fit_multi(cols, np.ones(3, dtype=np.int32), u0(dense, 3), opts)
    both.close()

    grow = Problem(ctx, X, Y, [block], max_batch=3, capacity=caps)
    before = grow.fit_multi(cols, np.zeros(3, dtype=np.int32), u0(block, 3), opts)        # decoupled schedule
    assert grow.add_spec(dense) == 1                                                      # ... no longer
    mixed = grow.fit_multi(np.array([0, 0, 1, 1, 2, 2], dtype=np.int32), np.array([0, 1, 0, 1, 0, 1], dtype=np.int32),
                           np.stack([u0(block, 1)[0], u0(dense, 1)[0]] * 3), opts)
    grow.close()
    ctx.close()
    for key in ("loss", "lml", "nit", "nfev", "status"):
        np.testing.assert_array_equal(before[key], ref_block[key], err_msg=key)
        np.testing.assert_array_equal(mixed[key][0::2], ref_block[key], err_msg=key)
        np.testing.assert_array_equal(mixed[key][1::2], ref_dense[key], err_msg=key)
    np.testing.assert_array_equal(mixed["u"][0::2, :block.n_trainable], ref_block["u"][:, :block.n_trainable])


def test_shared_inputs_sorted_by_keeps_rows_together():
    from waveome_amd.parallel import SharedInputs

    rng = np.random.default_rng(4)
    X = np.column_stack([rng.normal(size=40), rng.integers(0, 5, 40).astype(float)])
    Y = rng.normal(size=(40, 2))
    sh = SharedInputs(X, Y)
    assert sh.sorted_by(None) is sh
    so = sh.sorted_by(1)
    assert np.all(np.diff(so.X[:, 1]) >= 0)
    # the same (x, y) pairs, units in their original relative order (stable)
    key = lambda A, B: sorted(map(tuple, np.column_stack([A, B]).round(12)))
    assert key(so.X, so.Y) == key(X, Y)
    for u in range(5):
        np.testing.assert_array_equal(so.X[so.X[:, 1] == u, 0], X[X[:, 1] == u, 0])
    assert so.sorted_by(1) is so  # already sorted: nothing to do


@pytest.mark.gpu
def test_kernel_sweep_on_shuffled_longitudinal_data():
    """GPSearch sorts the DEVICE copy of the data by the unit column (block structures gain, nothing else changes):
    a sweep over shuffled rows gives the fits of the sorted rows, and the models keep the caller's row order."""
    import pandas as pd

    from waveome_amd.engine import Context
    from waveome_amd.model_search import GPSearch

    X, Y = _longitudinal(8, 9, 30, sort=True)
    perm = np.random.default_rng(2).permutation(len(X))
    u = lambda: K.Categorical(active_dims=[0])
    structures = {"unit x se": K.Sum([u(), K.Product([u(), K.SquaredExponential(active_dims=[1])])]),
                  "dense": K.Sum([u(), K.SquaredExponential(active_dims=[1])])}
    ctx = Context(0)
    tables = []
    for rows in (np.arange(len(X)), perm):
        Xd = pd.DataFrame({"unit": X[rows, 2], "t": X[rows, 0], "z": X[rows, 1]})
        Yd = pd.DataFrame({f"y{p}": Y[rows, p] for p in range(2)})
        gps = GPSearch(Xd, Yd, unit_col="unit")
        tables.append(gps.kernel_sweep(structures, context=ctx))
        m = gps.sweep_models[("y0", "unit x se")]
        np.testing.assert_array_equal(np.asarray(m.data[0])[:, 0], gps.X["unit"].to_numpy())  # caller's order
    a, b = tables
    np.testing.assert_allclose(a["loss"].to_numpy(), b["loss"].to_numpy(), rtol=1e-7, atol=1e-7)
    ctx.close()


def test_shared_inputs_sorted_by_permutes_the_broadcast_tensors_too():
    """Under torch.distributed the inputs also live in the tensors the broadcast filled (device tensors with RCCL):
    sorting must permute those the same way (X as N x D, Y as P x N)."""
    torch = pytest.importorskip("torch")
    from waveome_amd.parallel import SharedInputs

    rng = np.random.default_rng(9)
    X = np.column_stack([rng.normal(size=30), rng.integers(0, 4, 30).astype(float)])
    Y = rng.normal(size=(30, 3))
    sh = SharedInputs(X, Y, torch.from_numpy(X.copy()), torch.from_numpy(np.ascontiguousarray(Y.T)))
    so = sh.sorted_by(1)
    np.testing.assert_array_equal(so._Xt.numpy(), so.X)
    np.testing.assert_array_equal(so._Yt.numpy(), so.Y.T)
    assert so._Xt.is_contiguous() and so._Yt.is_contiguous()
