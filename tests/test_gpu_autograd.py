"""Autograd through the HIP kernels, checked against torch's own autograd of dense fp64
equivalents on the CPU (independent implementation)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def N():
    import nitorch_fastmath_amd as N_
    return N_


def to_full(c, M):
    """differentiable compact -> full in plain torch (CPU reference)"""
    rows = [[None] * M for _ in range(M)]
    k = M
    for i in range(M):
        rows[i][i] = c[..., i]
    for i in range(M):
        for j in range(i + 1, M):
            rows[i][j] = rows[j][i] = c[..., k]
            k += 1
    return torch.stack([torch.stack(r, -1) for r in rows], -2)


def spd(n, M, seed):
    g = torch.Generator().manual_seed(seed)
    G = torch.randn(n, M, M, dtype=torch.float64, generator=g)
    A = G @ G.transpose(-1, -2) / M + torch.eye(M, dtype=torch.float64)
    cols = [A[:, i, i] for i in range(M)] + [A[:, i, j] for i in range(M) for j in range(i + 1, M)]
    return torch.stack(cols, -1), torch.randn(n, M, dtype=torch.float64, generator=g), torch.randn(n, M, dtype=torch.float64, generator=g)


@pytest.mark.parametrize('M', [2, 3, 4, 6, 12])
def test_sym_solve_and_matvec_backward(dev, M):
    n = 50
    mat, vec, w = spd(n, M, M)
    # CPU reference
    mc, vc = mat.clone().requires_grad_(), vec.clone().requires_grad_()
    x = torch.linalg.solve(to_full(mc, M), vc.unsqueeze(-1)).squeeze(-1)
    (x * w).sum().backward()
    md, vd = mat.to(dev).requires_grad_(), vec.to(dev).requires_grad_()
    xd = N().sym_solve(md, vd)
    assert xd.requires_grad
    (xd * w.to(dev)).sum().backward()
    assert torch.allclose(xd.detach().cpu(), x.detach(), rtol=1e-10, atol=1e-12)
    assert torch.allclose(vd.grad.cpu(), vc.grad, rtol=1e-9, atol=1e-11)
    assert torch.allclose(md.grad.cpu(), mc.grad, rtol=1e-9, atol=1e-11)
    # matvec / addmatvec / submatvec
    for mode, fn in ((0, N().sym_matvec), (1, N().sym_addmatvec), (-1, N().sym_submatvec)):
        mc, vc, ic = mat.clone().requires_grad_(), vec.clone().requires_grad_(), w.clone().requires_grad_()
        y = (to_full(mc, M) @ vc.unsqueeze(-1)).squeeze(-1)
        y = y if mode == 0 else ic + mode * y
        (y * y).sum().backward()
        md, vd, idd = mat.to(dev).requires_grad_(), vec.to(dev).requires_grad_(), w.to(dev).requires_grad_()
        yd = fn(md, vd) if mode == 0 else fn(idd, md, vd)
        (yd * yd).sum().backward()
        assert torch.allclose(md.grad.cpu(), mc.grad, rtol=1e-9, atol=1e-11)
        assert torch.allclose(vd.grad.cpu(), vc.grad, rtol=1e-9, atol=1e-11)
        if mode:
            assert torch.allclose(idd.grad.cpu(), ic.grad, rtol=1e-9, atol=1e-11)


def test_sym_solve_backward_broadcast_and_kinds(dev):
    M, n = 3, 20
    mat, vec, w = spd(n, M, 9)
    # one matrix broadcast against many vectors: the matrix gradient is summed over the batch
    mc, vc = mat[:1].clone().requires_grad_(), vec.clone().requires_grad_()
    x = torch.linalg.solve(to_full(mc, M).expand(n, M, M), vc.unsqueeze(-1)).squeeze(-1)
    (x * w).sum().backward()
    md, vd = mat[:1].to(dev).requires_grad_(), vec.to(dev).requires_grad_()
    (N().sym_solve(md, vd) * w.to(dev)).sum().backward()
    assert md.grad.shape == (1, 6)
    assert torch.allclose(md.grad.cpu(), mc.grad, rtol=1e-9, atol=1e-11)
    assert torch.allclose(vd.grad.cpu(), vc.grad, rtol=1e-9, atol=1e-11)
    # diagonal and full matrix kinds
    dc, vc = mat[:, :M].clone().requires_grad_(), vec.clone().requires_grad_()
    ((vc / dc) * w).sum().backward()
    dd, vd = mat[:, :M].to(dev).contiguous().requires_grad_(), vec.to(dev).requires_grad_()
    (N().sym_solve(dd, vd) * w.to(dev)).sum().backward()
    assert torch.allclose(dd.grad.cpu(), dc.grad, rtol=1e-9) and torch.allclose(vd.grad.cpu(), vc.grad, rtol=1e-9)
    g = torch.Generator().manual_seed(1)
    F = torch.randn(n, M, M, dtype=torch.float64, generator=g) + 4 * torch.eye(M, dtype=torch.float64)
    fc, vc = F.clone().requires_grad_(), vec.clone().requires_grad_()
    (torch.linalg.solve(fc, vc.unsqueeze(-1)).squeeze(-1) * w).sum().backward()
    fd, vd = F.reshape(n, M * M).to(dev).requires_grad_(), vec.to(dev).requires_grad_()
    (N().sym_solve(fd, vd) * w.to(dev)).sum().backward()
    assert torch.allclose(fd.grad.cpu().reshape(n, M, M), fc.grad, rtol=1e-9, atol=1e-11)
    assert torch.allclose(vd.grad.cpu(), vc.grad, rtol=1e-9, atol=1e-11)
    with pytest.raises(RuntimeError, match='out='):
        N().sym_solve(md, vd, out=torch.empty(n, M, dtype=torch.float64, device=dev))
    with pytest.raises(NotImplementedError):
        N().qr.householder(vd)     # still forward-only


@pytest.mark.parametrize('n', [2, 3, 4, 6])
def test_eig_sym_backward(dev, n):
    nb = 30
    g = torch.Generator().manual_seed(n)
    a = torch.randn(nb, n, n, dtype=torch.float64, generator=g)
    a = (a + a.transpose(-1, -2)) / 2
    C = torch.randn(nb, n, n, dtype=torch.float64, generator=g)
    C = (C + C.transpose(-1, -2)) / 2

    def loss_vals(lam):
        return (lam.exp() + lam ** 3).sum()       # symmetric in the eigenvalues: order-free

    def loss_full(lam, U):                          # order- and sign-invariant
        return ((U * torch.tanh(lam).unsqueeze(-2)) @ U.transpose(-1, -2) * C).sum() + loss_vals(lam)

    ac = a.clone().requires_grad_()
    lam, U = torch.linalg.eigh((ac + ac.transpose(-1, -2)) / 2)
    loss_vals(lam).backward()
    g_vals = ac.grad.clone()
    ac.grad = None
    lam, U = torch.linalg.eigh((ac + ac.transpose(-1, -2)) / 2)
    loss_full(lam, U).backward()
    g_full = ac.grad.clone()

    ad = a.to(dev).requires_grad_()
    lam_d = N().eig_sym((ad + ad.transpose(-1, -2)) / 2)
    loss_vals(lam_d).backward()
    assert torch.allclose(ad.grad.cpu(), g_vals, rtol=1e-8, atol=1e-9)
    ad.grad = None
    lam_d, U_d = N().eig_sym((ad + ad.transpose(-1, -2)) / 2, compute_u=True)
    Cd = C.to(dev)
    (((U_d * torch.tanh(lam_d).unsqueeze(-2)) @ U_d.transpose(-1, -2) * Cd).sum() + loss_vals(lam_d)).backward()
    assert torch.allclose(ad.grad.cpu(), g_full, rtol=1e-7, atol=1e-8)


def test_reduce_backward(dev):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(5, 33, 7, dtype=torch.float64, generator=g)
    x[torch.rand(x.shape, generator=g) < 0.1] = float('nan')
    w = torch.randn(5, 7, dtype=torch.float64, generator=g)
    R = N().reduce
    for ours, ref, dim in ((R.nansum, torch.nansum, 1), (R.nanmean, torch.nanmean, 1), (R.nansum, torch.nansum, None),
                           (R.nanmean, torch.nanmean, (0, 2))):
        xc = x.clone().requires_grad_()
        xd = x.to(dev).requires_grad_()
        if dim is None:
            (ref(xc) * 3).backward()
            (ours(xd) * 3).backward()
        elif dim == 1:
            (ref(xc, dim=dim) * w).sum().backward()
            (ours(xd, dim=dim) * w.to(dev)).sum().backward()
        else:
            v = torch.arange(33, dtype=torch.float64)
            (ref(xc, dim=dim) * v).sum().backward()
            (ours(xd, dim=dim) * v.to(dev)).sum().backward()
        assert torch.equal(torch.isnan(xd.grad.cpu()), torch.isnan(xc.grad))
        assert torch.allclose(torch.nan_to_num(xd.grad.cpu()), torch.nan_to_num(xc.grad), rtol=1e-12, atol=1e-14)
    y = torch.randn(4, 9, dtype=torch.float64, generator=g)
    yc, yd = y.clone().requires_grad_(), y.to(dev).requires_grad_()
    (torch.sum(yc, 1) ** 2).sum().backward()
    (R.sum(yd, 1, keepdim=True) ** 2).sum().backward()
    assert torch.allclose(yd.grad.cpu(), yc.grad)
    yc.grad = None; yd.grad = None
    torch.mean(yc).backward()
    R.mean(yd).backward()
    assert torch.allclose(yd.grad.cpu(), yc.grad)
    with pytest.raises(NotImplementedError):
        R.median(yd, dim=1) if False else N().qr.hessenberg(torch.randn(3, 3, device=dev, requires_grad=True))


@pytest.mark.parametrize('n', [1, 2, 3, 4, 6])
def test_batched_backward(dev, n):
    """batchmatvec / batchinv / batchdet against torch's autograd of the dense fp64 ops on the CPU"""
    g = torch.Generator().manual_seed(n)
    a = torch.randn(7, n, n, dtype=torch.float64, generator=g) + 4 * torch.eye(n, dtype=torch.float64)
    v = torch.randn(7, n, dtype=torch.float64, generator=g)
    w = torch.randn(7, n, n, dtype=torch.float64, generator=g)
    B = N().batched
    ac, vc = a.clone().requires_grad_(), v.clone().requires_grad_()
    ad, vd = a.to(dev).requires_grad_(), v.to(dev).requires_grad_()
    (torch.matmul(ac, vc.unsqueeze(-1)).squeeze(-1) * w[:, 0]).sum().backward()
    (B.batchmatvec(ad, vd) * w[:, 0].to(dev)).sum().backward()
    assert torch.allclose(ad.grad.cpu(), ac.grad, rtol=1e-11, atol=1e-12)
    assert torch.allclose(vd.grad.cpu(), vc.grad, rtol=1e-11, atol=1e-12)
    # broadcast vector: one vector for every matrix
    v1c, v1d = v[0].clone().requires_grad_(), v[0].to(dev).requires_grad_()
    (torch.matmul(a, v1c) * w[:, 0]).sum().backward()
    (B.batchmatvec(a.to(dev), v1d) * w[:, 0].to(dev)).sum().backward()
    assert torch.allclose(v1d.grad.cpu(), v1c.grad, rtol=1e-11, atol=1e-12)
    for ours, ref, wt in ((B.batchinv, torch.linalg.inv, w), (B.batchdet, torch.linalg.det, w[:, 0, 0])):
        ac, ad = a.clone().requires_grad_(), a.to(dev).requires_grad_()
        (ref(ac) * wt).sum().backward()
        (ours(ad) * wt.to(dev)).sum().backward()
        assert torch.allclose(ad.grad.cpu(), ac.grad, rtol=1e-9, atol=1e-10), ours.__name__


@pytest.mark.parametrize('M', [1, 2, 3, 4, 6])
def test_sym_invert_and_det_backward(dev, M):
    mat, v, w = spd(9, M, 40 + M)
    S = N().sym
    K = M * (M + 1) // 2
    wk = torch.randn(9, K, dtype=torch.float64, generator=torch.Generator().manual_seed(M))

    def compact(full):
        cols = [full[..., i, i] for i in range(M)] + [full[..., i, j] for i in range(M) for j in range(i + 1, M)]
        return torch.stack(cols, -1)
    mc, md = mat.clone().requires_grad_(), mat.to(dev).requires_grad_()
    (compact(torch.linalg.inv(to_full(mc, M))) * wk).sum().backward()
    (S.sym_invert(md) * wk.to(dev)).sum().backward()
    assert torch.allclose(md.grad.cpu(), mc.grad, rtol=1e-9, atol=1e-10)
    mc, md = mat.clone().requires_grad_(), mat.to(dev).requires_grad_()
    (torch.linalg.inv(to_full(mc, M)).diagonal(dim1=-2, dim2=-1) * w).sum().backward()
    (S.sym_invert(md, diag=True) * w.to(dev)).sum().backward()
    assert torch.allclose(md.grad.cpu(), mc.grad, rtol=1e-9, atol=1e-10)
    mc, md = mat.clone().requires_grad_(), mat.to(dev).requires_grad_()
    (torch.linalg.det(to_full(mc, M)) * w[:, 0]).sum().backward()
    (S.sym_det(md) * w[:, 0].to(dev)).sum().backward()
    assert torch.allclose(md.grad.cpu(), mc.grad, rtol=1e-9, atol=1e-10)


def test_pick_and_var_backward(dev):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(4, 9, 5, dtype=torch.float64, generator=g)
    R = N().reduce
    w = torch.randn(4, 5, dtype=torch.float64, generator=g)
    for name, ref in (('max', lambda t: t.amax(1)), ('min', lambda t: t.amin(1))):
        xc, xd = x.clone().requires_grad_(), x.to(dev).requires_grad_()
        (ref(xc) * w).sum().backward()
        (getattr(R, name)(xd, dim=1) * w.to(dev)).sum().backward()
        assert torch.equal(xd.grad.cpu(), xc.grad), name
        xc, xd = x.clone().requires_grad_(), x.to(dev).requires_grad_()
        ref(xc).sum().backward()
        val, idx = getattr(R, name)(xd, dim=1, keepdim=True, return_indices=True)
        val.sum().backward()
        assert val.shape == (4, 1, 5) and torch.equal(xd.grad.cpu(), xc.grad)
    # several dims, NaNs skipped, full reduction
    xn = x.clone()
    xn[torch.rand(xn.shape, generator=g) < 0.2] = float('nan')
    xc, xd = xn.clone().requires_grad_(), xn.to(dev).requires_grad_()
    torch.nan_to_num(xc, nan=-float('inf')).amax((0, 2)).sum().backward()
    R.nanmax(xd, dim=(0, 2)).sum().backward()
    assert torch.equal(xd.grad.cpu(), xc.grad)
    xc, xd = x.clone().requires_grad_(), x.to(dev).requires_grad_()
    (xc.max() * 2).backward()
    (R.max(xd) * 2).backward()
    assert torch.equal(xd.grad.cpu(), xc.grad)
    # var / std, biased and unbiased, one and several dims
    for dim in (1, (0, 2), None):
        for unb in (True, False):
            for ours, ref in ((R.var, torch.var), (R.std, torch.std)):
                xc, xd = x.clone().requires_grad_(), x.to(dev).requires_grad_()
                kw = {} if dim is None else {'dim': dim}
                r = ref(xc, unbiased=unb, **kw)
                o = ours(xd, dim=dim, unbiased=unb)
                assert torch.allclose(o.detach().cpu(), r.detach(), rtol=1e-12)
                (r ** 2).sum().backward()
                (o ** 2).sum().backward()
                assert torch.allclose(xd.grad.cpu(), xc.grad, rtol=1e-10, atol=1e-12), (dim, unb, ours.__name__)
    # nanvar: gradient only on the counted elements
    xd = xn.to(dev).requires_grad_()
    R.nanvar(xd, dim=1).sum().backward()
    mask = (~torch.isnan(xn)).double()
    xc = torch.nan_to_num(xn).requires_grad_()           # NaN-free leaf: masked arithmetic instead
    cnt = mask.sum(1, keepdim=True)
    mean = (xc * mask).sum(1, keepdim=True) / cnt
    ((((xc - mean) ** 2) * mask).sum(1) / (cnt.squeeze(1) - 1)).sum().backward()
    assert torch.allclose(xd.grad.cpu(), xc.grad * mask, rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize('M', [1, 2, 3, 4, 5])
def test_sym_to_full_outer_matmul_backward(dev, M):
    mat, v, w = spd(6, M, 70 + M)
    S = N().sym
    g = torch.Generator().manual_seed(M)
    wf = torch.randn(6, M, M, dtype=torch.float64, generator=g)
    K = M * (M + 1) // 2
    wk = torch.randn(6, K, dtype=torch.float64, generator=g)

    def compact(full, n):
        cols = [full[..., i, i] for i in range(n)] + [full[..., i, j] for i in range(n) for j in range(i + 1, n)]
        return torch.stack(cols, -1)
    mc, md = mat.clone().requires_grad_(), mat.to(dev).requires_grad_()
    (to_full(mc, M) * wf).sum().backward()
    (S.sym_to_full(md) * wf.to(dev)).sum().backward()
    assert torch.allclose(md.grad.cpu(), mc.grad, rtol=1e-12, atol=1e-13)
    vc, vd = v.clone().requires_grad_(), v.to(dev).requires_grad_()
    (compact(vc.unsqueeze(-1) * vc.unsqueeze(-2), M) * wk).sum().backward()
    (S.sym_outer(vd) * wk.to(dev)).sum().backward()
    assert torch.allclose(vd.grad.cpu(), vc.grad, rtol=1e-12, atol=1e-13)
    for d in (1, 2, 3, 4):
        if M > 4:
            continue
        J = torch.randn(6, M, d, dtype=torch.float64, generator=g)
        Kd = d * (d + 1) // 2
        wd = torch.randn(6, Kd, dtype=torch.float64, generator=g)
        flip = M == d and M in (2, 3)                        # quirk Q16: the reference computes J H J^T
        jc, hc = J.clone().requires_grad_(), mat.clone().requires_grad_()
        jd, hd = J.to(dev).requires_grad_(), mat.to(dev).requires_grad_()
        Hf = to_full(hc, M)
        full = jc @ Hf @ jc.transpose(-1, -2) if flip else jc.transpose(-1, -2) @ Hf @ jc
        (compact(full, d) * wd).sum().backward()
        out = S.sym_matmul(jd, hd)
        assert torch.allclose(out.detach().cpu(), compact(full, d).detach(), rtol=1e-12, atol=1e-12)
        (out * wd.to(dev)).sum().backward()
        assert torch.allclose(jd.grad.cpu(), jc.grad, rtol=1e-10, atol=1e-11), (M, d)
        assert torch.allclose(hd.grad.cpu(), hc.grad, rtol=1e-10, atol=1e-11), (M, d)
        # diagonal hessian, one hessian for every jacobian (broadcast)
        dc, dd = mat[0, :M].clone().requires_grad_(), mat[0, :M].to(dev).requires_grad_()
        jc, jd = J.clone().requires_grad_(), J.to(dev).requires_grad_()
        (compact(jc.transpose(-1, -2) @ torch.diag_embed(dc) @ jc, d) * wd).sum().backward()
        (S.sym_matmul(jd, dd) * wd.to(dev)).sum().backward()
        if M > 1:       # (M == 1: a 1-vector is read as compact 1x1, same thing)
            assert torch.allclose(dd.grad.cpu(), dc.grad, rtol=1e-10, atol=1e-11), (M, d)
        assert torch.allclose(jd.grad.cpu(), jc.grad, rtol=1e-10, atol=1e-11), (M, d)


def test_median_backward(dev):
    """the gradient of a median goes to the element that was picked (what torch.median's own
    backward does on the reference's path)"""
    import nitorch_fastmath_amd as N_
    torch.manual_seed(4)
    x = torch.randn(6, 9, 5, device=dev, dtype=torch.float64, requires_grad=True)
    v, i = N_.reduce.median(x, dim=1, return_indices=True)
    v.sum().backward()
    ref = torch.zeros_like(x)
    ref.scatter_(1, i.unsqueeze(1), 1.0)
    assert torch.equal(x.grad, ref) and x.grad.sum() == 30
    xr = x.detach().clone().requires_grad_(True)
    torch.median(xr, dim=1).values.sum().backward()
    assert torch.equal(x.grad, xr.grad)                     # no duplicates in a random draw: same element
    y = torch.randn(1000, device=dev, requires_grad=True)
    N_.reduce.median(y).backward()
    assert y.grad.sum() == 1 and y.grad[y.detach().argsort()[499]] == 1
