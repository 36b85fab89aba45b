"""Host-side logic of the facade that needs no GPU: how reductions map tensors onto the
(outer, red, inner) view of the C ABI, the permutation / staging decisions, the broadcast
helper and the batch-collapse rules."""
import numpy as np
import pytest
import torch


def test_view3_maps_dims_without_copies():
    from nitorch_fastmath_amd.reduce import _view3
    x = torch.zeros(3, 4, 5, 6)
    v, outer, red, inner, dims, kept, redshape = _view3(x, 1)
    assert v is x and (outer, red, inner) == (3, 4, 30) and kept == [0, 2, 3] and redshape == [4]
    v, outer, red, inner, *_ = _view3(x, (1, 2))
    assert v is x and (outer, red, inner) == (3, 20, 6)
    v, outer, red, inner, *_ = _view3(x, -1)
    assert v is x and (outer, red, inner) == (60, 6, 1)
    v, outer, red, inner, *_ = _view3(x, None)
    assert (outer, red, inner) == (1, 360, 1)
    # dims given out of order: one permuting copy, reduced dims last, in the order given
    v, outer, red, inner, dims, kept, redshape = _view3(x, (2, 1))
    assert v is not x and (outer, red, inner) == (18, 20, 1) and redshape == [5, 4] and v.shape == (3, 6, 5, 4)
    with pytest.raises(IndexError):
        _view3(x, 4)
    with pytest.raises(IndexError):
        _view3(x, (1, 1))


def test_canon_recognises_permuted_contiguous_tensors():
    from nitorch_fastmath_amd.reduce import _canon, _dim_groups
    base = torch.arange(2 * 3 * 4 * 5.).reshape(2, 3, 4, 5)
    cl = base.permute(0, 2, 3, 1)                       # channel-last view of a channel-first tensor
    xp, mapped, inv, dims = _canon(cl, -1)
    assert xp.is_contiguous() and xp.data_ptr() == base.data_ptr() and mapped == 1 and dims == [3]
    xp, mapped, inv, dims = _canon(cl, (1, 2))
    assert mapped == [2, 3]
    assert _canon(base, 1) is None                      # already contiguous
    assert _canon(base[:, ::2], 1) is None              # a genuine strided view: copy path
    assert _dim_groups(base, (0, 2, 3)) == [[0], [2, 3]]
    assert _dim_groups(base, (1, 2)) is None and _dim_groups(base, None) is None
    assert _dim_groups(base[:, ::2], (0, 2)) is None


def test_broadcast_shapes_matches_torch():
    from nitorch_fastmath_amd._dispatch import broadcast_shapes
    rng = np.random.default_rng(0)
    for _ in range(200):
        nd = rng.integers(0, 5)
        full = [int(rng.integers(1, 5)) for _ in range(nd)]
        shapes = []
        for _ in range(int(rng.integers(1, 4))):
            k = int(rng.integers(0, nd + 1))
            shapes.append(tuple(1 if rng.random() < 0.3 else s for s in full[nd - k:]))
        assert broadcast_shapes(*shapes) == torch.broadcast_shapes(*shapes), shapes
    with pytest.raises(RuntimeError):
        broadcast_shapes((2, 3), (4, 3))


def test_batch_pack_leaves_broadcasts_alone():
    from nitorch_fastmath_amd._dispatch import Batch, expand_batch
    mat = torch.zeros(78, 50).t()                        # channel-first 12x12 compact field
    one = torch.zeros(1, 12)                             # one vector for every matrix
    out = torch.zeros(50, 12)
    # pack=True (round 3): component-major fields stay as they are (the strided kernels of orders 9..16 read them
    # with consecutive lanes on consecutive addresses), records strided along the batch are packed
    b = Batch((50,), [mat, expand_batch((50,), one, 1), out], [1, 1, 1], pack=True)
    assert b.tensors[0] is mat and b.operands[0].stride_inner == 1 and b.operands[0].stride_col == 50
    assert b.tensors[1].stride(0) == 0 and b.operands[1].stride_inner == 0  # broadcast kept
    every_other = torch.zeros(100, 78)[::2]
    b = Batch((50,), [every_other, expand_batch((50,), one, 1), out], [1, 1, 1], pack=True)
    assert b.tensors[0].is_contiguous() and b.tensors[0] is not every_other and b.operands[0].stride_inner == 78
    # pack='all' (ops without a strided kernel): component-major fields are packed too
    b = Batch((50,), [mat, expand_batch((50,), one, 1), out], [1, 1, 1], pack='all')
    assert b.tensors[0].is_contiguous() and b.tensors[0] is not mat         # packed
    assert b.tensors[1].stride(0) == 0                                       # broadcast kept
    assert b.operands[1].stride_inner == 0 and b.operands[0].stride_inner == 78
    # a strided user `out=` is written through a temporary and copied back
    big = torch.zeros(50, 24)
    b = Batch((50,), [mat, expand_batch((50,), one, 1), big[:, ::2]], [1, 1, 1], pack=True)
    b.tensors[-1].fill_(3.0)
    b.finish()
    assert bool((big[:, ::2] == 3).all()) and bool((big[:, 1::2] == 0).all())


def test_output_allocation_policy_needs_no_gpu_to_decide():
    from nitorch_fastmath_amd.sym import _alloc_out
    cpu = torch.device('cpu')
    vec_cf = torch.zeros(4, 1000).t()                    # channel-first
    out, _ = _alloc_out(None, (1000, 4), torch.float32, cpu, like=vec_cf)
    assert out.stride() == vec_cf.stride()               # layout handed on
    out, _ = _alloc_out(None, (1000, 4), torch.float32, cpu, like=torch.zeros(1000, 4))
    assert out.is_contiguous()
    sl = torch.zeros(1001, 6)[1:]                        # rows 1.. : starts 24 B into an aligned buffer
    out, _ = _alloc_out(None, (1000, 6), torch.float32, cpu, like=sl)
    assert out.is_contiguous()        # no phase games any more: wide global accesses take any element offset
    with pytest.raises(ValueError):
        _alloc_out(torch.zeros(3, 4), (1000, 4), torch.float32, cpu)


def test_utils_only_holds_hot_path_helpers():
    """only `ensure_list`, `ind2sub`, `eps` are reached from the path"""
    from nitorch_fastmath_amd import utils as U
    assert sorted(U.__all__) == ['ensure_list', 'eps', 'graphed', 'ind2sub']
    for gone in ('fast_slice_tensor', 'slice_tensor', 'cumprod', 'broadcast_backward', 'sub2ind'):
        assert not hasattr(U, gone)
    # ensure_list contract (`utils.py:11-28`)
    assert U.ensure_list(3) == [3] and U.ensure_list(3, 2) == [3, 3]
    assert U.ensure_list((1, 2)) == [1, 2] and U.ensure_list(range(3)) == [0, 1, 2]
    assert U.ensure_list(x for x in 'ab') == ['a', 'b']
    assert U.ensure_list([1, 2], 4) == [1, 2, 2, 2] and U.ensure_list([1, 2], 4, default=0) == [1, 2, 0, 0]
    assert U.ensure_list([1, 2, 3], 2) == [1, 2] and U.ensure_list([1, 2, 3], 2, crop=False) == [1, 2, 3]
    assert U.ensure_list('ab') == ['ab'] and U.ensure_list(None) == [None]
    lst = [1]
    assert U.ensure_list(lst, 3) is not None and lst == [1, 1, 1]      # lists are padded in place, like upstream



def test_batch_fast_path_equals_the_general_collapse():
    """contiguous full-shape operands skip the stride collapse (launch-bound small batches): the
    operand descriptors must be the ones the general path computes"""
    from nitorch_fastmath_amd._dispatch import Batch
    from nitorch_fastmath_amd import _dispatch as D

    def fields(b):
        return [(o.stride_outer, o.stride_inner, o.stride_row, o.stride_col) for o in b.operands], b.n_outer, b.n_inner
    for batch, comps, ncomp in (((7, 5), [(6,), (3,), ()], [1, 1, 0]), ((11,), [(4, 4), (4,)], [2, 1]),
                                ((1,), [(10,), (4,)], [1, 1]), ((2, 1, 3), [(8, 8), (8, 8)], [2, 2])):
        ts = [torch.zeros(tuple(batch) + c) for c in comps]
        fast = fields(Batch(batch, ts, ncomp))
        saved = torch.Tensor.is_contiguous
        try:                                       # force the general path on the same tensors
            torch.Tensor.is_contiguous = lambda self, *a, **k: False
            slow = fields(Batch(batch, ts, ncomp))
        finally:
            torch.Tensor.is_contiguous = saved
        assert fast == slow, (batch, fast, slow)
