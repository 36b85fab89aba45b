#!/usr/bin/env python
"""Generate golden input/output vectors from the REAL reference (build container only).

Runs the reference's in-repo hot-path modules on CPU and stores inputs +
outputs as small .npz fixtures under tests/golden/.  The reference lives at
/root/reference and never travels to the GPU box; only the data written here
does.  `import nitorch_fastmath` itself fails offline (its `sym.py:37` pulls the
absent, un-pinned `jitfields` dependency), so the package `__init__` is
bypassed with a namespace shim and the `_impl` modules are imported directly
(SURVEY.md section 8c).

    python tests/golden/make_golden.py        # rewrites tests/golden/*.npz

Fixtures are data only (inputs and expected outputs).
"""
import sys
import os
import types
import importlib
import warnings
import numpy as np
import torch

warnings.filterwarnings('ignore')
HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/nitorch_fastmath'


def load_ref():
    pkg = types.ModuleType('nitorch_fastmath')
    pkg.__path__ = [REF]
    sys.modules['nitorch_fastmath'] = pkg
    names = ['_impl.sym', '_impl.batched', '_impl.qr', 'reduce', 'qr', 'utils']
    return {n: importlib.import_module('nitorch_fastmath.' + n) for n in names}


def pack_sym(full):
    """(..., M, M) symmetric -> (..., K) compact: diagonal, then upper rows."""
    M = full.shape[-1]
    cols = [full[..., i, i] for i in range(M)]
    cols += [full[..., i, j] for i in range(M) for j in range(i + 1, M)]
    return torch.stack(cols, -1)


def spd(n, M, dtype, gen):
    G = torch.randn(n, M, M, dtype=torch.float64, generator=gen)
    A = G @ G.transpose(-1, -2) / M + torch.eye(M, dtype=torch.float64)
    return pack_sym(A).to(dtype)


def indefinite(n, M, dtype, gen):
    """Symmetric, well conditioned, NOT positive definite, with small or zero
    leading entries so that an unpivoted factorisation would break."""
    Q, _ = torch.linalg.qr(torch.randn(n, M, M, dtype=torch.float64, generator=gen))
    lam = 1 + torch.rand(n, M, dtype=torch.float64, generator=gen)
    sign = torch.ones(M, dtype=torch.float64)
    sign[::2] = -1
    A = (Q * (lam * sign)[:, None, :]) @ Q.transpose(-1, -2)
    A = (A + A.transpose(-1, -2)) / 2
    return pack_sym(A).to(dtype)


def npy(x):
    return x.detach().contiguous().cpu().numpy()


def gen_sym(ref):
    S = ref['_impl.sym']
    out = {}
    gen = torch.Generator().manual_seed(20261003)
    for dtype, dname in ((torch.float32, 'f32'), (torch.float64, 'f64')):
        for M in (1, 2, 3, 4, 5, 6, 7, 8, 12, 16):
            n = 32
            mat = spd(n, M, dtype, gen)
            vec = torch.randn(n, M, dtype=torch.float64, generator=gen).to(dtype)
            inp = torch.randn(n, M, dtype=torch.float64, generator=gen).to(dtype)
            k = f'{dname}_M{M}_'
            out[k + 'mat'] = npy(mat)
            out[k + 'vec'] = npy(vec)
            out[k + 'inp'] = npy(inp)
            out[k + 'solve'] = npy(S.sym_solve(mat, vec))
            mv = S.sym_matvec(mat, vec)
            out[k + 'matvec'] = npy(mv)
            # jitfields-only entry points are defined from the in-repo matvec
            out[k + 'addmatvec'] = npy(inp + mv)
            out[k + 'submatvec'] = npy(inp - mv)
            out[k + 'invert'] = npy(S.sym_invert(mat))
            out[k + 'invert_diag'] = npy(S.sym_invert(mat, diag=True))
            full = S.sym_to_full(mat)
            out[k + 'to_full'] = npy(full)
            out[k + 'outer'] = npy(S.sym_outer(vec))
            # sym_det (reference `_impl/sym.py:433-434`) derives M from a batch
            # dim (quirk Q2); the closed forms are called directly instead and
            # M > 4 uses the branch it would take: det of the full matrix.
            m0 = mat.movedim(-1, 0)
            if M == 1:
                det = m0[0]
            elif M == 2:
                det = S._sym_det2(m0[:M], m0[M:])
            elif M == 3:
                det = S._sym_det3(m0[:M], m0[M:])
            elif M == 4:
                det = S._sym_det4(m0[:M], m0[M:])
            else:
                det = torch.det(full)
            out[k + 'det'] = npy(det)
            if M >= 2:
                imat = indefinite(n, M, dtype, gen)
                out[k + 'mat_indef'] = npy(imat)
                out[k + 'solve_indef'] = npy(S.sym_solve(imat, vec))
                out[k + 'invert_indef'] = npy(S.sym_invert(imat))
            # NN auto-detect kinds (`sym.py:16-24`): diagonal / scaled identity / full
            dg = mat[:, :M]
            out[k + 'solve_diag'] = npy(vec / dg)
            out[k + 'matvec_diag'] = npy(vec * dg)
            out[k + 'solve_scal'] = npy(vec / dg[:, :1])
            out[k + 'matvec_scal'] = npy(vec * dg[:, :1])
        # sym_matmul: J^T H J
        for (kk, d) in ((1, 1), (2, 2), (3, 3), (3, 2), (4, 4), (2, 3)):
            n = 32
            j = torch.randn(n, kk, d, dtype=torch.float64, generator=gen).to(dtype)
            h = spd(n, kk, dtype, gen)
            k = f'{dname}_k{kk}_d{d}_'
            out[k + 'j'] = npy(j)
            out[k + 'h'] = npy(h)
            out[k + 'matmul'] = npy(S.sym_matmul(j, h))
    np.savez_compressed(os.path.join(HERE, 'sym.npz'), **out)
    return len(out)


def gen_batched(ref):
    B = ref['_impl.batched']
    out = {}
    gen = torch.Generator().manual_seed(20261004)
    for dtype, dname in ((torch.float32, 'f32'), (torch.float64, 'f64')):
        for n_ in (1, 2, 3, 4, 5, 6, 7, 8, 12, 16):
            nb = 32
            a = torch.randn(nb, n_, n_, dtype=torch.float64, generator=gen)
            a = (a + 8 * torch.eye(n_, dtype=torch.float64)).to(dtype)
            v = torch.randn(nb, n_, dtype=torch.float64, generator=gen).to(dtype)
            k = f'{dname}_n{n_}_'
            out[k + 'a'] = npy(a)
            out[k + 'v'] = npy(v)
            # CPU path == torch fallbacks (`_impl/batched.py:53,119,175`)
            out[k + 'inv'] = npy(B.batchinv(a))
            out[k + 'det'] = npy(B.batchdet(a))
            out[k + 'matvec'] = npy(B.batchmatvec(a, v))
            # the TorchScript closed forms themselves (gated on is_cuda upstream)
            am = a.movedim(-1, 0).movedim(-1, 0)
            if n_ == 2:
                out[k + 'inv_ts'] = npy(B.inv2(am).movedim(0, -1).movedim(0, -1))
                out[k + 'det_ts'] = npy(B.det2(am))
            if n_ == 3:
                out[k + 'inv_ts'] = npy(B.inv3(am).movedim(0, -1).movedim(0, -1))
                out[k + 'det_ts'] = npy(B.det3(am))
        # rectangular matvec (reference test_batched.py:31-35 uses 4x5)
        a = torch.randn(32, 4, 5, dtype=torch.float64, generator=gen).to(dtype)
        v = torch.randn(32, 5, dtype=torch.float64, generator=gen).to(dtype)
        out[f'{dname}_rect_a'] = npy(a)
        out[f'{dname}_rect_v'] = npy(v)
        out[f'{dname}_rect_matvec'] = npy(B.batchmatvec(a, v))
    np.savez_compressed(os.path.join(HERE, 'batched.npz'), **out)
    return len(out)


def gen_reduce(ref):
    R = ref['reduce']
    out = {}
    gen = torch.Generator().manual_seed(20261005)
    for dtype, dname in ((torch.float32, 'f32'), (torch.float64, 'f64')):
        for size in (1, 63, 64, 65, 4097, 20011):
            for nanfrac, nname in ((0.0, 'nan0'), (0.01, 'nan1'), (1.0, 'nanall')):
                x = torch.randn(size, dtype=torch.float64, generator=gen).to(dtype)
                m = torch.rand(size, dtype=torch.float64, generator=gen) < nanfrac
                x[m] = float('nan')
                if nname == 'nan1' and size > 64:
                    x[5] = float('inf')
                    x[7] = -float('inf')
                    x[5] = 3.0 if size == 65 else x[5]
                k = f'{dname}_{size}_{nname}_'
                out[k + 'x'] = npy(x)
                out[k + 'nansum'] = npy(R.nansum(x))
                out[k + 'nansum64'] = npy(R.nansum(x, dtype=torch.float64))
                out[k + 'nanmax'] = npy(R.nanmax(x))
                out[k + 'nanmin'] = npy(R.nanmin(x))
                out[k + 'sum'] = npy(R.sum(x))
                out[k + 'max'] = npy(R.max(x))
                out[k + 'min'] = npy(R.min(x))
                out[k + 'mean'] = npy(R.mean(x))
        # dim-wise nansum works in the reference
        x = torch.randn(7, 33, 5, dtype=torch.float64, generator=gen).to(dtype)
        x[torch.rand(7, 33, 5, generator=gen) < 0.1] = float('nan')
        out[f'{dname}_nd_x'] = npy(x)
        for dim, dn in ((0, 'd0'), (1, 'd1'), (2, 'd2'), (-1, 'dm1'), ((0, 2), 'd02'), ((1, 2), 'd12')):
            out[f'{dname}_nd_nansum_{dn}'] = npy(R.nansum(x, dim=dim))
            out[f'{dname}_nd_nansum_keep_{dn}'] = npy(R.nansum(x, dim=dim, keepdim=True))
            out[f'{dname}_nd_sum_{dn}'] = npy(R.sum(x, dim=dim))
            out[f'{dname}_nd_mean_{dn}'] = npy(R.mean(x, dim=dim))
    # empty input
    out['f32_empty_nansum'] = npy(R.nansum(torch.zeros(0)))
    np.savez_compressed(os.path.join(HERE, 'reduce.npz'), **out)
    return len(out)


def gen_qr(ref):
    Q = ref['qr']
    out = {}
    gen = torch.Generator().manual_seed(20261006)
    for dtype, dname in ((torch.float32, 'f32'), (torch.float64, 'f64')):
        nb = 16
        x = torch.randn(nb, dtype=torch.float64, generator=gen).to(dtype)
        y = torch.randn(nb, dtype=torch.float64, generator=gen).to(dtype)
        x[0] = 0
        y[0] = 0
        x[1] = 0
        y[2] = 0
        c, s = Q.givens(x, y)
        out[f'{dname}_givens_x'] = npy(x)
        out[f'{dname}_givens_y'] = npy(y)
        out[f'{dname}_givens_c'] = npy(c)
        out[f'{dname}_givens_s'] = npy(s)
        for n_ in (1, 2, 3, 4, 5, 6, 8, 12):
            k = f'{dname}_n{n_}_'
            a = torch.randn(nb, n_, n_, dtype=torch.float64, generator=gen).to(dtype)
            out[k + 'a'] = npy(a)
            # householder (+ projection alpha) on every basis worth testing
            v = torch.randn(nb, n_, dtype=torch.float64, generator=gen).to(dtype)
            v[0] = 0            # zero vector: reflector must come out as zeros, not NaN
            if n_ > 1:
                v[1, 0] = 0     # zero pivot component: sign(0) -> +1
            out[k + 'hh_x'] = npy(v)
            for basis in sorted({0, n_ - 1}):
                u, alpha = Q.householder(v, basis=basis, return_alpha=True)
                out[k + f'hh_u_b{basis}'] = npy(u)
                out[k + f'hh_alpha_b{basis}'] = npy(alpha)
            u = Q.householder(v)
            for side in ('left', 'right', 'both'):
                out[k + f'hh_apply_{side}'] = npy(Q.householder_apply(a, u, side=side))
            if n_ >= 3:     # a shorter reflector acts on the trailing block only
                u2 = Q.householder(v[:, 1:])
                out[k + 'hh_apply_short'] = npy(Q.householder_apply(a, u2, side='both'))
                out[k + 'hh_apply_two_inv'] = npy(Q.householder_apply(a, [u, u2], side='left', inverse=True))
            # givens_apply
            if n_ >= 2:
                cc, ss = Q.givens(a[:, 0, 0], a[:, 1, 0])
                out[k + 'ga_c'] = npy(cc)
                out[k + 'ga_s'] = npy(ss)
                for side in ('left', 'right', 'both'):
                    out[k + f'givens_apply_{side}'] = npy(Q.givens_apply(a, cc[:, None], ss[:, None], 0, n_ - 1, side=side))
                out[k + 'givens_apply_default_j'] = npy(Q.givens_apply(a, cc[:, None], ss[:, None], 0, side='left'))
            # hessenberg (general) and QR of a Hessenberg matrix
            h, us = Q.hessenberg(a, compute_u=True)
            out[k + 'hess'] = npy(h)
            for i, ui in enumerate(us):
                out[k + f'hess_u{i}'] = npy(ui)
            hz = torch.triu(a, -1)
            out[k + 'hz'] = npy(hz)
            qq, rr = Q.qr_hessenberg(hz)
            out[k + 'qrh_q'] = npy(qq)
            out[k + 'qrh_r'] = npy(rr)
            # the reference's rq_hessenberg is only right for n <= 3 or tridiagonal input
            # (quirk Q8); golden for the general case = r @ q of its own (correct) QR
            out[k + 'rq_true'] = npy(rr @ qq)
            if n_ <= 3:
                out[k + 'rq_ref'] = npy(Q.rq_hessenberg(hz))
            # symmetric: tridiagonalisation + eigenvalues (reference works for n <= 5, quirk Q7)
            sym = (a + a.transpose(-1, -2)) / 2
            out[k + 'sym'] = npy(sym)
            out[k + 'eigvalsh'] = npy(torch.linalg.eigvalsh(sym.double()))
            if n_ <= 5:
                for upper in (True, False):
                    t, us = Q.hessenberg_sym(sym, upper=upper, fill=True, compute_u=True)
                    out[k + f'hess_sym_{int(upper)}'] = npy(t)
                    for i, ui in enumerate(us):
                        out[k + f'hess_sym_{int(upper)}_u{i}'] = npy(ui)
                    # un-symmetrised input: only one triangle may be read
                    t2 = Q.hessenberg_sym(a, upper=upper, fill=True)
                    out[k + f'hess_nonsym_{int(upper)}'] = npy(t2)
                    # The reference judges convergence on batch-wide sums (quirk Q9), so the
                    # ORDER of its (unsorted) eigenvalues depends on what else is in the batch.
                    # Per-matrix goldens = the reference called on one matrix at a time.
                    out[k + f'eig_{int(upper)}'] = npy(torch.cat([Q.eig_sym(a[i:i + 1], upper=upper) for i in range(nb)]))
                out[k + 'eig'] = npy(torch.cat([Q.eig_sym(sym[i:i + 1]) for i in range(nb)]))
                out[k + 'eig_batched'] = npy(Q.eig_sym(sym))
                pairs = [Q.eig_sym(sym[i:i + 1], compute_u=True) for i in range(nb)]
                out[k + 'eig_u_val'] = npy(torch.cat([p[0] for p in pairs]))
                out[k + 'eig_u_vec'] = npy(torch.cat([p[1] for p in pairs]))
                t3 = Q.hessenberg_sym(sym, upper=True, fill=True)
                out[k + 'tri'] = npy(t3)
                out[k + 'rq_tri'] = npy(Q.rq_hessenberg(t3))
                uu = torch.eye(n_, dtype=dtype).expand(nb, n_, n_).clone()
                r_h, r_u = Q.rq_hessenberg(t3, uu)
                out[k + 'rq_tri_u'] = npy(r_u)
    np.savez_compressed(os.path.join(HERE, 'qr.npz'), **out)
    return len(out)


if __name__ == '__main__':
    torch.set_num_threads(1)
    ref = load_ref()
    which = sys.argv[1:] or ['sym', 'batched', 'reduce', 'qr']
    for w in which:
        n = globals()['gen_' + w](ref)
        print(w, n, 'arrays')
