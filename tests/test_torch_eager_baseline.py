"""Pin the torch-eager CPU baseline (oracle/torch_eager.py, timed by bench.py's cpu_baseline
leg) to the golden vectors the real reference produced (tests/golden/sym.npz), and check
that it has the reference's cost SHAPE: component-first full-batch element-wise ops,
13 / 60 / 266 of them per call for M = 2 / 3 / 4 in the reference (counted
from the TorchScript graphs).  CPU only."""
import numpy as np
import pytest
import torch
from conftest import TOL, relerr


@pytest.mark.parametrize('dn', ['f32', 'f64'])
@pytest.mark.parametrize('M', [1, 2, 3, 4])
def test_matches_reference_golden(golden_sym, dn, M):
    from oracle import torch_eager as T
    k = f'{dn}_M{M}_'
    for mat_key, out_key in (('mat', 'solve'), ('mat_indef', 'solve_indef')):
        if k + mat_key not in golden_sym.files:
            continue
        x = T.sym_solve(torch.from_numpy(golden_sym[k + mat_key]), torch.from_numpy(golden_sym[k + 'vec'])).numpy()
        ref = golden_sym[k + out_key]
        assert x.dtype == ref.dtype and x.shape == ref.shape
        # same polynomials, different summation order than the reference: north-star tolerance
        assert relerr(x, ref) <= TOL[dn], (M, mat_key)


def test_matches_c_oracle(oracle):
    from oracle import torch_eager as T
    g = torch.Generator().manual_seed(7)
    G = torch.randn(4096, 4, 4, dtype=torch.float64, generator=g)
    A = G @ G.transpose(-1, -2) / 4 + torch.eye(4, dtype=torch.float64)
    mat = torch.stack([A[:, i, i] for i in range(4)] + [A[:, i, j] for i in range(4) for j in range(i + 1, 4)], -1)
    vec = torch.randn(4096, 4, dtype=torch.float64, generator=g)
    assert relerr(T.sym_solve(mat, vec).numpy(), oracle.sym_solve(mat.numpy(), vec.numpy())) <= TOL['f64']
    mat32, vec32 = mat.float(), vec.float()
    assert relerr(T.sym_solve(mat32, vec32).numpy(), oracle.sym_solve(mat32.numpy(), vec32.numpy())) <= TOL['f32']


def test_cost_shape():
    from oracle import torch_eager as T
    # monomial counts of the symmetric determinant: 2, 5, 17 (the reference's _sym_det2/3/4)
    assert [len(T.tables(M)[0]) for M in (2, 3, 4)] == [2, 5, 17]
    ops = {M: T.ops_per_call(M) for M in (2, 3, 4)}
    ref = {2: 13, 3: 60, 4: 266}
    for M in ref:     # same order of magnitude of full-batch passes as the reference (within 10 %)
        assert abs(ops[M] - ref[M]) <= 0.1 * ref[M], ops


def test_broadcast_and_channel_first_views():
    from oracle import torch_eager as T
    mat = torch.rand(6, 10, dtype=torch.float64) + torch.tensor([4.] * 4 + [0.] * 6, dtype=torch.float64)
    vec = torch.rand(5, 6, 4, dtype=torch.float64)
    x = T.sym_solve(mat, vec)
    assert x.shape == (5, 6, 4)
    y = T.sym_solve(mat.expand(5, 6, 10).contiguous(), vec)
    assert torch.equal(x, y)


def test_other_workload_baselines_match_the_oracle(oracle, golden_sym, golden_reduce, golden_batched):
    from oracle import torch_eager as T
    for dn in ('f32', 'f64'):
        k = f'{dn}_M3_'
        assert relerr(T.sym_invert(torch.from_numpy(golden_sym[k + 'mat'])).numpy(), golden_sym[k + 'invert']) <= TOL[dn]
        a = golden_batched[f'{dn}_n8_a']
        assert relerr(T.batch_inv(torch.from_numpy(a)).numpy(), golden_batched[f'{dn}_n8_inv']) <= TOL[dn]
    rng = np.random.default_rng(3)
    x = rng.standard_normal(100_003).astype(np.float32)
    x[rng.random(x.size) < 0.01] = np.nan
    xt = torch.from_numpy(x)
    assert float(T.nanmax(xt)) == float(oracle.reduce('nanmax', x)) == float(np.nanmax(x))
    assert abs(float(T.nansum(xt)) - float(np.nansum(x.astype(np.float64)))) <= 1e-6 * float(np.nansum(np.abs(x)))
    assert np.isnan(x).any() and not torch.isnan(xt[~torch.isnan(xt)]).any()          # the input copy is what gets filled
