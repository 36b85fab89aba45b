"""CPU-side checks: the C-ABI library loads and exports every symbol declared in
include/nfm_hip.h, argument validation answers without touching a GPU, and the
host-side batch normalisation does what the kernels assume."""
import ctypes
import os
import re
import numpy as np
import pytest
import torch
from conftest import ROOT


@pytest.fixture(scope='module')
def L():
    import __graft_entry__ as G
    if not os.path.exists(os.path.join(ROOT, 'nitorch_fastmath_amd', 'libnfm_hip.so')):
        G.build()
    from nitorch_fastmath_amd import _lib
    return _lib.lib()


def declared_symbols():
    txt = open(os.path.join(ROOT, 'include', 'nfm_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(nfm_[a-z_0-9]+)\s*\(', txt)))


def test_exports_every_declared_symbol(L):
    syms = declared_symbols()
    assert len(syms) >= 15, syms
    for s in syms:
        assert hasattr(L, s), f'{s} declared in include/nfm_hip.h but not exported'


def test_signature_table_matches_header():
    from nitorch_fastmath_amd import _lib
    declared = set(declared_symbols())
    bound = set(_lib.SIGNATURES) | {'nfm_strerror', 'nfm_version', 'nfm_reduce_workspace_bytes'}
    assert declared == bound, declared ^ bound


def test_version_and_strerror(L):
    assert L.nfm_version() == 5
    assert b'dtype' in L.nfm_strerror(-2)
    assert L.nfm_reduce_workspace_bytes() >= 2048 * 8


def test_argument_validation_without_gpu(L):
    from nitorch_fastmath_amd._lib import Operand
    op = Operand(None, 0, 0, 0, 1)
    r = ctypes.byref(op)
    # bad dtype / bad order / negative batch are rejected before any HIP call
    assert L.nfm_sym_solve(7, 4, 0, 1, 1, r, r, r, None, None) == -2
    assert L.nfm_sym_solve(0, 17, 0, 1, 1, r, r, r, None, None) == -3
    assert L.nfm_sym_solve(0, 4, 0, 1, -1, r, r, r, None, None) == -1
    assert L.nfm_sym_solve(0, 4, 9, 1, 1, r, r, r, None, None) == -1
    assert L.nfm_sym_solve(0, 4, 0, 1, 5, r, r, r, None, None) == -1   # null pointers, n > 0
    assert L.nfm_batch_inv(0, 0, 0, 1, 1, r, r, None) == -3
    assert L.nfm_reduce_all(0, 99, 0, 0, None, 8, 1 << 20, 8, None) == -1
    assert L.nfm_reduce_all(0, 0, 0, 0, None, 8, 16, 8, None) == -5     # workspace too small
    # empty batches succeed without a launch
    assert L.nfm_sym_solve(0, 4, 0, 1, 0, r, r, r, None, None) == 0
    assert L.nfm_sym_invert(1, 3, 0, 0, 0, r, r, None) == 0
    # qr family: order / side / index checks happen before any launch
    assert L.nfm_qr_eig_sym(0, 17, 1, 0, 10, 1e-32, 1, 1, r, 8, None) == -3
    assert L.nfm_qr_eig_sym(0, 3, 1, 0, -1, 1e-32, 1, 1, r, 8, None) == -1
    assert L.nfm_qr_givens_apply(0, 3, 5, 0, 1, 1, 1, r, r, r, None) == -1     # bad side
    assert L.nfm_qr_givens_apply(0, 3, 0, 0, 3, 1, 1, r, r, r, None) == -1     # j out of range
    assert L.nfm_qr_householder(1, 4, 4, 1, 1, r, 8, None) == -1               # basis out of range
    assert L.nfm_qr_householder_apply(1, 4, 5, 0, 1, 1, r, r, None) == -1      # reflector longer than n
    assert L.nfm_qr_hessenberg(0, 3, 0, 1, 0, 1, 0, r, None, None) == 0        # empty batch
    assert L.nfm_reduce_dim(0, 99, 0, 1, 10, 1, 16, None, 0, 16, None, None) == -1       # bad op
    assert L.nfm_reduce_dim(0, 0, 0, 1, 10, 1, 6, None, 0, 16, None, None) == -4        # misaligned input
    assert L.nfm_reduce_dim(0, 0, 0, 0, 10, 1, None, None, 0, None, None, None) == 0    # empty
    # few outputs + long axis needs the chunk workspace; short rows need none
    assert L.nfm_reduce_dim_workspace_bytes(0, 0, 1 << 20, 8, 1, 0) == 0
    need = L.nfm_reduce_dim_workspace_bytes(0, 0, 16, 1 << 26, 1, 0)
    assert need > 0 and L.nfm_reduce_dim_workspace_bytes(0, 1, 16, 1 << 26, 1, 1) == 2 * need
    assert L.nfm_reduce_dim(0, 0, 0, 16, 1 << 26, 1, 16, None, 0, 16, None, None) == -1
    assert L.nfm_reduce_dim(0, 0, 0, 16, 1 << 26, 1, 16, 16, 8, 16, None, None) == -5
    assert L.nfm_reduce_moments_workspace_bytes(0, 1, 1000, 1) >= L.nfm_reduce_workspace_bytes()
    assert L.nfm_reduce_stat(0, 3, 0, 1, 10, 1, 16, None, 0, 16, None) == -1            # bad stat kind
    assert L.nfm_reduce_stat(0, 16, 0, 1, 10, 1, 16, None, 0, 16, None) == -1           # bad flag
    assert L.nfm_reduce_moments(3, 1, 1, 1, None, None, 0, None, None) == -2
    bad = Operand(6, 0, 0, 0, 1)   # misaligned for float
    assert L.nfm_sym_det(0, 3, 1, 1, ctypes.byref(bad), ctypes.byref(bad), None) == -4


def test_facade_refuses_cpu_tensors_and_bad_dtypes():
    import nitorch_fastmath_amd as N
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        N.sym_solve(torch.ones(5, 10), torch.ones(5, 4))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        N.batchinv(torch.eye(3)[None])
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        N.reduce.nansum(torch.ones(4))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        N.eig_sym(torch.eye(3)[None])
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        N.qr.givens(torch.ones(3), torch.ones(3))


def test_product_never_imports_oracle():
    # the oracle is test infrastructure: no product source may reference it
    pkg = os.path.join(ROOT, 'nitorch_fastmath_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.hpp', '.h')):
                src = open(os.path.join(dirpath, f)).read()
                assert 'oracle' not in src.lower(), f'{f} mentions the oracle'


def test_batch_collapse():
    from nitorch_fastmath_amd._dispatch import Batch, expand_batch
    # contiguous (B, X, K) collapses to one level
    mat = torch.zeros(6, 7, 10)
    vec = torch.zeros(6, 7, 4)
    out = torch.zeros(6, 7, 4)
    b = Batch((6, 7), [mat, vec, out], [1, 1, 1])
    assert (b.n_outer, b.n_inner) == (1, 42)
    assert [o.stride_inner for o in b.operands] == [10, 4, 4]
    assert [o.stride_col for o in b.operands] == [1, 1, 1]
    # channel-first field (B, C, X, Y) viewed as (B, X, Y, C): two levels, no copy
    f = torch.zeros(3, 10, 5, 8)
    m = f.movedim(1, -1)
    v = torch.zeros(3, 4, 5, 8).movedim(1, -1)
    o = torch.zeros(3, 5, 8, 4)
    b = Batch((3, 5, 8), [m, v, o], [1, 1, 1])
    assert (b.n_outer, b.n_inner) == (3, 40)
    assert b.operands[0].stride_outer == 400 and b.operands[0].stride_inner == 1
    assert b.operands[0].stride_col == 40 and b.operands[2].stride_inner == 4
    assert b.tensors[0].data_ptr() == f.data_ptr()
    # broadcast matrix: stride 0 at both levels
    m1 = expand_batch((6, 7), torch.zeros(10), 1)
    b = Batch((6, 7), [m1, vec, out], [1, 1, 1])
    assert (b.n_outer, b.n_inner) == (1, 42) and b.operands[0].stride_inner == 0
    # partially broadcast: (1, 7, K) against (6, 7, M) -> two levels, outer stride 0
    m2 = expand_batch((6, 7), torch.zeros(1, 7, 10), 1)
    b = Batch((6, 7), [m2, vec, out], [1, 1, 1])
    assert (b.n_outer, b.n_inner) == (6, 7)
    assert b.operands[0].stride_outer == 0 and b.operands[0].stride_inner == 10
    # three genuine levels -> materialised
    t = torch.zeros(4, 2, 6, 2, 5, 10)[:, 0, :, 0]      # (4, 6, 5, 10), strides (1200, 100, 10, 1)
    t = t[:, ::2, ::2]                                    # (4, 3, 3, 10)
    b = Batch((4, 3, 3), [t, torch.zeros(4, 3, 3, 10)], [1, 1])
    assert (b.n_outer, b.n_inner) == (1, 36)
    # empty batch
    b = Batch((0, 5), [torch.zeros(0, 5, 10), torch.zeros(0, 5, 4)], [1, 1])
    assert b.n_outer == 0 and b.n_inner == 0
    # full matrices carry row/col strides
    a = torch.zeros(9, 3, 3).transpose(-1, -2)
    b = Batch((9,), [a, torch.zeros(9, 3, 3)], [2, 2])
    assert (b.operands[0].stride_row, b.operands[0].stride_col) == (1, 3)


def test_mat_kind_and_utils():
    from nitorch_fastmath_amd import sym, utils, _lib
    assert sym._mat_kind(10, 4) == _lib.MAT_SYM
    assert sym._mat_kind(4, 4) == _lib.MAT_DIAG
    assert sym._mat_kind(1, 4) == _lib.MAT_SCAL
    assert sym._mat_kind(16, 4) == _lib.MAT_FULL
    assert sym._mat_kind(1, 1) == _lib.MAT_SYM
    with pytest.raises(ValueError):
        sym._mat_kind(7, 4)
    assert sym._nb_prm(21) == 6
    with pytest.raises(ValueError):
        sym._nb_prm(7)
    ind = torch.tensor([0, 5, 23])
    sub = utils.ind2sub(ind, [2, 3, 4])
    assert sub.tolist() == np.array(np.unravel_index([0, 5, 23], (2, 3, 4))).tolist()
    assert utils.ensure_list(3, 2) == [3, 3] and utils.ensure_list((1, 2)) == [1, 2]
    assert utils.eps(torch.float32) == 2 ** -23 and utils.eps('float64') == 2 ** -52
    x = torch.zeros(3, 6)
    assert sym.sym_diag(x).shape == (3, 3) and sym.sym_diag(x).data_ptr() == x.data_ptr()


def test_compat_package_resolves_reference_import_paths():
    """`import nitorch_fastmath` + the reference's module paths, served by the backend"""
    import importlib
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'compat'))
    try:
        sys.modules.pop('nitorch_fastmath', None)
        nf = importlib.import_module('nitorch_fastmath')
        from nitorch_fastmath.sym import sym_solve, sym_invert, sym_matvec, sym_addmatvec_, sym_solve_  # noqa: F401
        from nitorch_fastmath.batched import batchinv, batchdet, batchmatvec  # noqa: F401
        from nitorch_fastmath.qr import eig_sym, givens, householder  # noqa: F401
        from nitorch_fastmath.reduce import nansum, nanmax, nanmean  # noqa: F401
        import nitorch_fastmath_amd as N
        assert nf.sym_solve is N.sym.sym_solve and nf.reduce is N.reduce
        # every public name of the reference's hot-path modules exists
        for name in ['sym_to_full', 'sym_diag', 'sym_outer', 'sym_det', 'sym_matmul', 'sym_matvec', 'sym_addmatvec',
                     'sym_addmatvec_', 'sym_submatvec', 'sym_submatvec_', 'sym_solve', 'sym_solve_', 'sym_invert',
                     'sym_invert_']:
            assert callable(getattr(nf.sym, name)), name
        for name in ['eig_sym', 'qr_hessenberg', 'rq_hessenberg', 'hessenberg', 'hessenberg_sym', 'householder',
                     'householder_apply', 'givens', 'givens_apply']:
            assert callable(getattr(nf.qr, name)), name
        for name in ['min', 'max', 'nanmin', 'nanmax', 'median', 'sum', 'nansum', 'mean', 'nanmean', 'var', 'nanvar',
                     'std', 'nanstd']:
            assert callable(getattr(nf.reduce, name)), name
    finally:
        sys.path.remove(os.path.join(ROOT, 'compat'))
        for k in [k for k in sys.modules if k == 'nitorch_fastmath' or k.startswith('nitorch_fastmath.')]:
            sys.modules.pop(k)


def test_signatures_match_the_reference():
    """argument names and order of the public functions (reference file:line in the docstrings)"""
    import inspect
    import nitorch_fastmath_amd as N

    def names(fn):
        return list(inspect.signature(fn).parameters)
    assert names(N.sym.sym_solve)[:3] == ['mat', 'vec', 'eps']                 # _impl/sym.py:327
    assert names(N.sym.sym_invert)[:2] == ['mat', 'diag']                       # _impl/sym.py:455
    assert names(N.sym.sym_matvec)[:2] == ['mat', 'vec']                        # _impl/sym.py:134
    assert names(N.batched.batchinv)[0] == 'a' and names(N.batched.batchmatvec) == ['mat', 'vec']
    # qr.py:30-38; `arithmetic` is a keyword-only extension (float32 sweep arithmetic, qr.py docstring)
    assert names(N.qr.eig_sym) == ['a', 'compute_u', 'upper', 'inplace', 'check_finite', 'max_iter', 'tol', 'arithmetic']
    assert inspect.signature(N.qr.eig_sym).parameters['arithmetic'].kind is inspect.Parameter.KEYWORD_ONLY
    assert names(N.qr.hessenberg_sym) == ['a', 'upper', 'fill', 'inplace', 'check_finite', 'compute_u']        # qr.py:226-233
    assert names(N.qr.householder) == ['x', 'basis', 'inplace', 'check_finite', 'return_alpha']               # qr.py:278-284
    assert names(N.qr.householder_apply) == ['a', 'u', 'k', 'side', 'inverse', 'inplace', 'check_finite']     # qr.py:330-338
    assert names(N.qr.givens_apply) == ['a', 'c', 's', 'i', 'j', 'side', 'inplace', 'check_finite']           # qr.py:375-384
    assert names(N.qr.rq_hessenberg) == ['h', 'u', 'inplace', 'check_finite']                                 # qr.py:103-108
    red = ['input', 'dim', 'keepdim', 'omitnan', 'inplace']
    assert names(N.reduce.max) == red + ['return_indices', 'out']                                              # reduce.py:145-153
    assert names(N.reduce.nanmax) == ['input', 'dim', 'keepdim', 'inplace', 'return_indices', 'out']          # reduce.py:267-274
    assert names(N.reduce.sum) == red + ['dtype', 'out']                                                       # reduce.py:431-439
    assert names(N.reduce.nansum) == ['input', 'dim', 'keepdim', 'inplace', 'dtype', 'out']                   # reduce.py:471-478
    assert names(N.reduce.var) == ['input', 'dim', 'keepdim', 'unbiased', 'omitnan', 'inplace', 'dtype', 'out']  # reduce.py:597-606
    assert names(N.reduce.nanstd) == ['input', 'dim', 'keepdim', 'unbiased', 'inplace', 'dtype', 'out']       # reduce.py:729-737


def test_no_scratch_anywhere():
    """Code-object facts (scripts/kernel_resources.py reads the `amdhsa.kernels` notes of the built
    objects): NO kernel of the library has a private segment -- "one matrix per lane held entirely in
    registers", and where a matrix does not fit a lane's 512 registers: float64 orders 13..16 of the
    sym / batched ops take the one-matrix-per-16-lanes kernels of nfm_rowwave.hip, and the QR family at
    orders 9..16 keeps its matrices in LDS ([element][lane] images, `qr_lds_kernel`) where they do not
    fit (round 2 had 23 kernels with up to 11 KB of scratch per lane there)."""
    import glob
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'scripts'))
    try:
        import kernel_resources as KR
    finally:
        sys.path.pop(0)
    objs = sorted(glob.glob(os.path.join(ROOT, 'nitorch_fastmath_amd', 'csrc', '*.o')))
    if not objs:
        pytest.skip('objects not built in this checkout (the .so alone travels to the GPU box)')
    rows = KR.collect(objs)
    assert len(rows) > 1000
    bad = [(k['kernel'], k['scratch']) for k in rows if k['scratch']]
    assert not bad, bad[:5]
    by = {k['kernel']: k for k in rows}
    # the bench kernels: registers as quoted in DESIGN.md section 4
    assert by['rec_kernel<float, SolveOp<float, 4, 0>, 1>']['vgpr'] <= 64          # 8 waves / SIMD
    assert by['rec_kernel<float, SolveOp<float, 6, 0>, 1>']['vgpr'] <= 64
    assert by['rec_kernel<double, BatchInvOp<double, 8>, 1>']['vgpr'] <= 256       # 2 waves / SIMD
    rw = [k for k in rows if 'roww_kernel' in k['kernel']]
    # (the 4-rows-per-lane float64 forms at orders 14..16 -- measurement forms, never the dispatch table's choice --
    # park up to 30 values in accumulation registers: 262..286)
    assert rw and max(k['vgpr'] for k in rw) <= 300 and not any(k['scratch'] for k in rw)
    # the positive-definite-first kernels (nfm_spd.hip): every (dtype, order, op)
    sp = [k for k in rows if 'spd_kernel' in k['kernel']]
    assert len(sp) == 2 * 8 * 4 and not any(k['scratch'] for k in sp)
