"""Orders above 16 (outside the register / row-wave kernels, `NFM_MAX_DIM`): the reference's own
route for large orders, on the device -- densify the compact storage and call `torch.linalg`
(`_impl/sym.py:392-396` solve, `:455-493` invert, `:401-452` det; `_impl/batched.py:119-120`,
`:53-54`).  rocSOLVER's batched LU is the right tool once a matrix no longer fits a wavefront's
registers; nothing here touches the CPU.  float32 / float64 GPU tensors only, like the kernels."""
from functools import lru_cache
import torch


@lru_cache(maxsize=64)
def _maps(M, device):
    """(full <- compact gather map (M*M,), compact <- full gather map (K,)) for order M"""
    idx = torch.empty(M, M, dtype=torch.long)
    c = M
    for i in range(M):
        idx[i, i] = i
    for i in range(M):
        for j in range(i + 1, M):
            idx[i, j] = idx[j, i] = c
            c += 1
    rows = list(range(M)) + [i for i in range(M) for j in range(i + 1, M)]
    cols = list(range(M)) + [j for i in range(M) for j in range(i + 1, M)]
    back = torch.tensor([r * M + c_ for r, c_ in zip(rows, cols)], dtype=torch.long)
    return idx.reshape(-1).to(device), back.to(device)


def to_full(mat, M):
    return mat[..., _maps(M, mat.device)[0]].unflatten(-1, (M, M))


def to_compact(full, M):
    return full.flatten(-2)[..., _maps(M, full.device)[1]]


def _eps_vec(eps, M, dtype, device):
    e = torch.as_tensor(eps, dtype=torch.float64).flatten().tolist()
    if not e:
        raise ValueError('eps is empty')
    e = (e + [e[-1]] * M)[:M]
    return torch.tensor(e, dtype=dtype, device=device)


def _dense(mat, M, kind_full):
    return mat.unflatten(-1, (M, M)) if kind_full else to_full(mat, M)


def _deliver(val, out):
    if out is None:
        return val.contiguous()
    out.copy_(val)
    return out


def sym_solve(mat, vec, eps, out, NN):
    M = vec.shape[-1]
    if NN == M:                                   # diagonal
        d = mat if eps is None else mat + _eps_vec(eps, M, mat.dtype, mat.device)
        return _deliver(vec / d, out)
    if NN == 1:                                   # scaled identity
        d = mat if eps is None else mat + _eps_vec(eps, M, mat.dtype, mat.device)
        return _deliver(vec / d, out)
    full = _dense(mat, M, NN == M * M)
    if eps is not None:
        full = full + torch.diag_embed(_eps_vec(eps, M, mat.dtype, mat.device))
    batch = torch.broadcast_shapes(full.shape[:-2], vec.shape[:-1])
    x = torch.linalg.solve(full.expand(*batch, M, M), vec.expand(*batch, M).unsqueeze(-1)).squeeze(-1)
    return _deliver(x, out)


def sym_matvec(mode, inp, mat, vec, out, NN):
    M = vec.shape[-1]
    if NN == M or NN == 1:
        y = mat * vec
    else:
        y = (_dense(mat, M, NN == M * M) @ vec.unsqueeze(-1)).squeeze(-1)
    if mode > 0:
        y = inp + y
    elif mode < 0:
        y = inp - y
    return _deliver(y, out)


def sym_invert(mat, M, diag, out):
    inv = torch.linalg.inv(to_full(mat, M))
    return _deliver(torch.diagonal(inv, dim1=-2, dim2=-1) if diag else to_compact(inv, M), out)


def sym_det(mat, M, out):
    return _deliver(torch.linalg.det(to_full(mat, M)), out)
