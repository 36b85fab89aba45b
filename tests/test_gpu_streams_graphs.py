"""The entry points launch on the caller's current HIP stream, never synchronise and never
allocate: they can be issued on side streams and captured into HIP graphs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def N():
    import nitorch_fastmath_amd as N_
    return N_


def spd(n, M, dev, seed=0):
    from bench import spd_compact
    return spd_compact(n, M, torch.float32, dev, seed)


def test_side_stream_ordering(dev):
    n, M = 2_000_000, 4
    mat, vec = spd(n, M, dev)
    ref = N().sym_solve(mat, vec)
    torch.cuda.synchronize()
    s = torch.cuda.Stream(device=dev)
    out = torch.empty_like(vec)
    v2 = torch.empty_like(vec)
    s.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s):
        v2.copy_(vec, non_blocking=True)         # producer on the side stream ...
        N().sym_solve(mat, v2, out=out)           # ... consumed by our kernel on the SAME stream
        back = N().sym_matvec(mat, out)
    torch.cuda.current_stream(dev).wait_stream(s)
    assert torch.equal(out, ref)
    assert ((back - vec).abs().amax() / vec.abs().amax()).item() < 2e-5


def test_graph_capture_and_replay(dev):
    n, M = 500_000, 6
    mat, vec = spd(n, M, dev, 3)
    out = torch.empty_like(vec)
    total = torch.empty((), device=dev, dtype=torch.float64)
    N().sym_solve(mat, vec, out=out)             # warm-up outside the capture (module load)
    N().reduce.nansum(out, dtype=torch.float64, out=total)
    torch.cuda.synchronize()
    ref = out.clone()
    ref_total = total.clone()
    out.zero_()
    total.zero_()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        N().sym_solve(mat, vec, out=out)
        N().reduce.nansum(out, dtype=torch.float64, out=total)
    assert float(total) == 0.0                   # capture records, it does not execute
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, ref) and torch.equal(total, ref_total)
    # new inputs in the same buffers, replay again
    vec.mul_(2)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, 2 * ref)


def test_graphed_helper_replays_a_sequence(dev):
    """utils.graphed: a Gauss-Newton-like chain (J^T H J -> solve -> update) captured once and
    replayed on new data, equal to the eager chain bit for bit, with one launch per replay"""
    import time
    from nitorch_fastmath_amd.utils import graphed
    S = N()
    n = 5000
    g = torch.Generator(device=dev).manual_seed(3)
    hess, grad = spd(n, 3, dev, 5)
    jac = torch.randn(n, 3, 3, device=dev, generator=g)

    def chain(h, j, b):
        a = S.sym_matmul(j, h)
        x = S.sym_solve(a, b, eps=1e-3)
        return x - S.sym_matvec(a, x) * 0.5

    step = graphed(chain, hess, jac, grad)
    for seed in (7, 8):
        h2, b2 = spd(n, 3, dev, seed)
        j2 = torch.randn(n, 3, 3, device=dev, generator=g)
        assert torch.equal(step(h2, j2, b2), chain(h2, j2, b2))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        step(hess, jac, grad)
    torch.cuda.synchronize()
    t_graph = (time.perf_counter() - t0) / 200
    t0 = time.perf_counter()
    for _ in range(200):
        chain(hess, jac, grad)
    torch.cuda.synchronize()
    t_eager = (time.perf_counter() - t0) / 200
    assert t_graph < t_eager, (t_graph, t_eager)


def test_qr_functions_capture_with_check_finite_off(dev):
    """the qr facade's default `check_finite=True` reads `isfinite(a).all()` on the host and so cannot
    be captured; with `check_finite=False` an eig_sym call is one launch and replays (utils.graphed
    docstring)"""
    from nitorch_fastmath_amd.utils import graphed
    g = torch.Generator(device=dev).manual_seed(1)
    a = torch.randn(4096, 3, 3, device=dev, generator=g)
    a = a + a.transpose(-1, -2)
    step = graphed(lambda x: N().eig_sym(x, check_finite=False), a)
    b = torch.randn(4096, 3, 3, device=dev, generator=g)
    b = b + b.transpose(-1, -2)
    assert torch.equal(step(b), N().eig_sym(b, check_finite=False))
    with pytest.raises(Exception):
        graphed(lambda x: N().eig_sym(x), a)        # the host read inside a capture is an error
    torch.cuda.synchronize()


def test_fast_accessors_agree_with_the_public_api(dev):
    """the facade reads the current stream and device through private torch accessors (cheaper per call);
    they must mean what the public API means: inside a non-default stream context, and `on_device` must
    leave the current device as it found it"""
    from nitorch_fastmath_amd import _dispatch as D
    assert D.stream_ptr(dev) == torch.cuda.current_stream(dev).cuda_stream
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        assert D.stream_ptr(dev) == side.cuda_stream == torch.cuda.current_stream(dev).cuda_stream
    assert D.stream_ptr(dev) == torch.cuda.current_stream(dev).cuda_stream
    before = torch.cuda.current_device()
    with D.on_device(dev):
        assert torch.cuda.current_device() == dev.index
    assert torch.cuda.current_device() == before
    # a broken private accessor falls back to the public API instead of raising
    saved = D._raw_stream, D._cur_device
    try:
        D._raw_stream = lambda *a: (_ for _ in ()).throw(TypeError('signature changed'))
        D._cur_device = lambda *a: (_ for _ in ()).throw(TypeError('signature changed'))
        assert D.stream_ptr(dev) == torch.cuda.current_stream(dev).cuda_stream
        with D.on_device(dev):
            assert torch.cuda.current_device() == dev.index
    finally:
        D._raw_stream, D._cur_device = saved
