"""The cases of the reference's own test-suite for this path (`tests/test_batched.py:9-97`,
`tests/test_qr.py:9-38`), restated against this backend THROUGH THE REFERENCE'S IMPORT PATHS
(`compat/` on sys.path: `from nitorch_fastmath.batched import ...`): same shapes, same independent
check (torch's native op at `allclose` defaults; `torch.symeig` was removed from torch, its
successor `torch.linalg.eigvalsh` stands in, as SURVEY 8c prescribes).

The reference draws unseeded inputs; here every case is seeded (a draw whose 5-term dot product
cancels to 1e-3 of its terms once failed `allclose` by 1 ulp of the terms, on both sides a correctly
rounded summation in a different order), and the mat-vec check states its error model: any
summation order of sum_j a_ij v_j is within n eps sum_j |a_ij v_j| of the exact value."""
import os
import sys
import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'compat'))


@pytest.mark.parametrize('mshape,vshape', [([2, 1, 1], [2, 2, 1]), ([2, 2, 2], [2, 2, 2]), ([2, 3, 3], [2, 2, 3]),
                                           ([2, 4, 5], [2, 2, 5]), ([2, 2, 4, 5], [5])],
                         ids=['1x1', '2x2', '3x3', '4x5', 'mat longer'])
def test_batchmatvec_reference_cases(dev, mshape, vshape):
    from nitorch_fastmath.batched import batchmatvec
    torch.manual_seed(1000 + sum(mshape) + 7 * len(vshape))
    mat, vec = torch.randn(mshape, device=dev), torch.randn(vshape, device=dev)
    got, ref = batchmatvec(mat, vec), mat.matmul(vec.unsqueeze(-1)).squeeze(-1)
    n = mshape[-1]
    bound = 2 * n * torch.finfo(torch.float32).eps * mat.abs().matmul(vec.abs().unsqueeze(-1)).squeeze(-1)
    assert got.shape == ref.shape and bool(((got - ref).abs() <= bound).all())
    exact = mat.double().matmul(vec.double().unsqueeze(-1)).squeeze(-1)
    assert bool(((got.double() - exact).abs() <= bound / 2).all())     # and against the fp64 truth


@pytest.mark.parametrize('n', [1, 2, 3, 4])
def test_batchdet_reference_cases(dev, n):
    from nitorch_fastmath.batched import batchdet
    torch.manual_seed(2000 + n)
    mat = torch.randn([2, n, n], device=dev)
    assert torch.allclose(batchdet(mat), torch.det(mat))


@pytest.mark.parametrize('n', [1, 2, 3, 4])
def test_batchinv_reference_cases(dev, n):
    from nitorch_fastmath.batched import batchinv
    torch.manual_seed(3000 + n)
    mat = torch.randn([2, n, n], device=dev) + 10 * torch.eye(n, device=dev)     # test_batched.py:95-96
    assert torch.allclose(batchinv(mat), torch.inverse(mat))


@pytest.mark.parametrize('n', [1, 2, 3, 4])
def test_symeig_reference_cases(dev, n):
    from nitorch_fastmath.qr import eig_sym
    torch.manual_seed(4000 + n)
    mat = torch.randn([2, n, n], dtype=torch.double, device=dev)
    mat = (mat + mat.transpose(-1, -2)) / 2
    got = eig_sym(mat).sort(-1)[0]
    assert torch.allclose(got, torch.linalg.eigvalsh(mat).sort(-1)[0])


def test_top_level_names_of_the_reference_resolve():
    import nitorch_fastmath as nf                                   # `__init__.py:1-10` star re-exports
    for name in ('sym_matvec', 'sym_addmatvec', 'sym_addmatvec_', 'sym_submatvec', 'sym_submatvec_', 'sym_solve',
                 'sym_solve_', 'sym_invert', 'sym_invert_', 'sym_to_full', 'sym_diag', 'sym_outer', 'sym_det',
                 'sym_matmul', 'batchmatvec', 'batchdet', 'batchinv', 'eig_sym', 'qr_hessenberg', 'rq_hessenberg',
                 'hessenberg', 'hessenberg_sym', 'householder', 'householder_apply', 'givens', 'givens_apply',
                 'nansum', 'nanmax', 'nanmin', 'nanmean', 'nanvar', 'nanstd'):
        assert callable(getattr(nf, name)), name
    from nitorch_fastmath.reduce import nansum, median, var, std, mean, min, max, sum   # noqa: F401
    from nitorch_fastmath.sym import sym_solve                      # noqa: F401
