"""Second device in ONE process (the facade accepts any `cuda:k`; the one-process-per-GPU bench
never exercises this).  The kernels that need more than 64 KiB of dynamic LDS opt in per kernel
AND per device (`lds_opt_in`, nfm_common.hpp: an atomic per-kernel mask of the devices that have
the attribute, keyed on hipGetDevice()); round 1 kept one process-wide flag, so the first call on
a second device launched without the opt-in.  Needs >= 2 GPUs: skipped on the 1-GPU test box."""
import numpy as np
import pytest
import torch
from conftest import TOL, relerr

@pytest.mark.gpu
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two GPUs in one process')
def test_large_lds_kernels_on_a_second_device(oracle):
    import nitorch_fastmath_amd as N
    rng = np.random.default_rng(0)
    a = rng.standard_normal((3000, 8, 8)) + 8 * np.eye(8)               # 128-lane tiles: ~135 KiB of LDS
    G = rng.standard_normal((2000, 12, 12))
    A = G @ G.transpose(0, 2, 1) / 12 + np.eye(12)
    iu = [(i, j) for i in range(12) for j in range(i + 1, 12)]
    mat = np.concatenate([np.stack([A[:, i, i] for i in range(12)], -1), np.stack([A[:, i, j] for i, j in iu], -1)], -1)
    vec = rng.standard_normal((2000, 12))
    ref_inv, ref_x = oracle.batch_inv(a), oracle.sym_solve(mat, vec)
    for dev in ('cuda:0', 'cuda:1', 'cuda:0'):
        inv = N.batchinv(torch.from_numpy(a).to(dev))
        assert inv.device == torch.device(dev) and relerr(inv.cpu().numpy(), ref_inv) <= TOL['f64']
        x = N.sym_solve(torch.from_numpy(mat).to(dev), torch.from_numpy(vec).to(dev))
        assert relerr(x.cpu().numpy(), ref_x) <= TOL['f64']
        # strided operands of a large order: the LDS-resident fallback (147 KiB)
        ms = torch.from_numpy(mat).to(dev).t().contiguous().t()
        assert relerr(N.sym_solve(ms, torch.from_numpy(vec).to(dev)).cpu().numpy(), ref_x) <= TOL['f64']


def test_lds_opt_in_is_keyed_on_the_device():
    """host-side: the launchers no longer hold a process-wide `static bool` for the opt-in"""
    import os
    from conftest import ROOT
    src = ''
    for f in ('nfm_record_kernel.hpp', 'nfm_big.hpp', 'nfm_common.hpp'):
        src += open(os.path.join(ROOT, 'nitorch_fastmath_amd', 'csrc', f)).read()
    assert 'static bool attr' not in src
    assert 'hipGetDevice' in src and 'std::atomic<uint64_t>' in src and 'lds_opt_in(' in src
    # every hipFuncSetAttribute goes through lds_opt_in (which propagates its status)
    assert src.count('hipFuncSetAttribute(') == 1
