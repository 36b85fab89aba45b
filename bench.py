#!/usr/bin/env python
"""bench.py -- headline benchmark of the hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

Default workload (BASELINE.json configs[1], the configuration the metric is quoted on):
`sym_solve` on 1e8 random 4x4 SPD compact-sym fp32 systems per GPU (AoS layout, the
default torch layout), inputs resident in HBM before the timed region.  One "step" =
one pass of the hot path over the whole batch = ONE kernel launch.  For N > 1 the
driver launches one process per GPU (torch.distributed, backend nccl = RCCL); the batch
shards embarrassingly, so every rank owns 1e8 systems (weak scaling) and there is no
collective on the data path -- only the barrier and the max-over-ranks of the wall time.

Prints ONE JSON line (rank 0) with the contract fields plus
  "roofline":     HBM roofline of the dominant kernel: algorithmic bytes per launch /
                  average launch duration (HIP events on the launch stream),
  "cpu_baseline": the CPU oracle (a C port of the reference's algorithm, OpenMP) timed on
                  this box's host cores on a bounded sample (rank 0, N = 1 only).
Other workloads (parity-test configs, for profiling): --workload sym_solve6 | batchinv8 |
nansum | nanmax | sym_invert3 | eig3.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable copy)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=100)
    p.add_argument('--warmup', type=int, default=10)
    p.add_argument('--workload', default='sym_solve4')
    p.add_argument('--n', type=float, default=None, help='batch per GPU (default: the config size)')
    p.add_argument('--no-cpu', action='store_true', help='skip the cpu_baseline leg')
    p.add_argument('--layout', default='aos', choices=['aos', 'soa'])
    p.add_argument('--gather', action='store_true',
                   help='also time the OPTIONAL epilogue: all-gather of the per-rank outputs over xGMI (reported '
                        'separately as gather_ms; never part of value)')
    return p.parse_args()


def spd_compact(n, M, dtype, device, seed, chunk=1 << 22):
    """A = G G^T / M + I packed diagonal-first, generated on device in chunks
    with element-wise ops only (rocBLAS batched GEMM faults on batch counts of ~1e7)."""
    import torch
    K = M * (M + 1) // 2
    g = torch.Generator(device=device).manual_seed(seed)
    mat = torch.empty(n, K, dtype=dtype, device=device)
    pairs = [(i, i) for i in range(M)] + [(i, j) for i in range(M) for j in range(i + 1, M)]
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        G = torch.randn(M, M, hi - lo, device=device, generator=g)
        for c, (i, j) in enumerate(pairs):
            a = (G[i] * G[j]).sum(0) / M
            if i == j:
                a += 1
            mat[lo:hi, c] = a
        del G
    vec = torch.randn(n, M, device=device, generator=g).to(dtype)
    return mat, vec


class Workload:
    """name, unit count per step, algorithmic bytes per unit, the step closure, the cpu leg."""


def make_workload(name, n_arg, device, rank, layout):
    import torch
    import nitorch_fastmath_amd as N
    w = Workload()
    seed = 1234 + rank
    if name in ('sym_solve4', 'sym_solve6'):
        M = 4 if name == 'sym_solve4' else 6
        n = int(n_arg or 1e8)
        K = M * (M + 1) // 2
        mat, vec = spd_compact(n, M, torch.float32, device, seed)
        if layout == 'soa':
            mat = mat.t().contiguous().t()
            vec = vec.t().contiguous().t()
            out = torch.empty(M, n, device=device).t()
        else:
            out = torch.empty(n, M, device=device)
        w.units, w.bytes_per_unit, w.dtype = n, (K + 2 * M) * 4, 'f32'
        w.desc = f'sym_solve {M}x{M} SPD compact-sym fp32, batch {n:.0e} per GPU, {layout.upper()} layout'
        w.metric, w.unit = f'{M}x{M} compact-sym solves/sec', 'solves/s'
        w.step = lambda: N.sym_solve(mat, vec, out=out)
        w.output = out
        w.kernel = f'rec_kernel<float, SolveOp<float, {M}, 0>>'

        def check():
            import numpy as np
            import oracle as O
            idx = torch.cat([torch.arange(0, 50000), torch.arange(n - 50000, n)]).to(device)
            ref = O.sym_solve(mat[idx].cpu().numpy(), vec[idx].cpu().numpy())
            got = out[idx].cpu().numpy()
            return float(np.abs(got - ref).max() / np.abs(ref).max()), bool(np.array_equal(got, ref))
        w.check = check

        def cpu(budget_s):
            import numpy as np
            import oracle as O
            ns = min(n, 20_000_000)
            m_h, v_h = mat[:ns].contiguous().cpu().numpy(), vec[:ns].contiguous().cpu().numpy()
            return time_cpu(lambda: O.sym_solve(m_h, v_h), ns, budget_s), f'first {ns:.0e} systems of the GPU batch'
        w.cpu = cpu
    elif name == 'sym_invert3':
        n = int(n_arg or 1e5)
        mat, _ = spd_compact(n, 3, torch.float64, device, seed)
        out = torch.empty_like(mat)
        w.units, w.bytes_per_unit, w.dtype = n, 2 * 6 * 8, 'f64'
        w.desc = f'sym_invert 3x3 SPD compact-sym fp64, batch {n:.0e}'
        w.metric, w.unit = '3x3 compact-sym inversions/sec', 'inversions/s'
        w.step = lambda: N.sym_invert(mat, out=out)
        w.kernel = 'rec_kernel<double, InvertOp<double, 3, false>>'
        w.check = lambda: (0.0, True)

        def cpu(budget_s):
            import oracle as O
            m_h = mat.cpu().numpy()
            return time_cpu(lambda: O.sym_invert(m_h), n, budget_s), 'the whole batch'
        w.cpu = cpu
    elif name == 'eig3':
        n = int(n_arg or 5e7)
        g = torch.Generator(device=device).manual_seed(seed)
        a = torch.randn(n, 3, 3, device=device, generator=g)
        a = a + a.transpose(-1, -2)          # symmetric (Hessian-filter shaped workload)
        w.units, w.bytes_per_unit, w.dtype = n, (9 + 3) * 4, 'f32'
        w.desc = f'eig_sym 3x3 symmetric fp32 (eigenvalues), batch {n:.0e}'
        w.metric, w.unit = '3x3 symmetric eigenvalue problems/sec', 'matrices/s'
        w.step = lambda: N.eig_sym(a, check_finite=False)
        w.kernel = 'rec_kernel<float, EigSymOp<float, 3, false>>'
        w.check = lambda: (0.0, True)

        def cpu(budget_s):
            import oracle as O
            ns = min(n, 2_000_000)
            a_h = a[:ns].cpu().numpy()
            return time_cpu(lambda: O.eig_sym(a_h), ns, budget_s), f'first {ns:.0e} matrices'
        w.cpu = cpu
    elif name == 'batchinv8':
        n = int(n_arg or 1e7)
        g = torch.Generator(device=device).manual_seed(seed)
        a = torch.randn(n, 8, 8, device=device, dtype=torch.float64, generator=g)
        a += 8 * torch.eye(8, device=device, dtype=torch.float64)
        w.units, w.bytes_per_unit, w.dtype = n, 2 * 64 * 8, 'f64'
        w.desc = f'batchinv 8x8 general fp64, batch {n:.0e}'
        w.metric, w.unit = '8x8 fp64 inversions/sec', 'inversions/s'
        w.step = lambda: N.batchinv(a)
        w.kernel = 'rec_kernel<double, BatchInvOp<double, 8>>'
        w.check = lambda: (0.0, True)

        def cpu(budget_s):
            import oracle as O
            ns = min(n, 2_000_000)
            a_h = a[:ns].cpu().numpy()
            return time_cpu(lambda: O.batch_inv(a_h), ns, budget_s), f'first {ns:.0e} matrices'
        w.cpu = cpu
    elif name in ('nansum', 'nanmax'):
        n = int(n_arg or 2 ** 33)       # 32 GiB fp32
        g = torch.Generator(device=device).manual_seed(seed)
        x = torch.empty(n, device=device)
        chunk = 1 << 28
        for lo in range(0, n, chunk):
            hi = min(n, lo + chunk)
            x[lo:hi].normal_(generator=g)
            x[lo:hi].masked_fill_(torch.rand(hi - lo, device=device, generator=g) < 0.01, float('nan'))
        w.units, w.bytes_per_unit, w.dtype = n, 4, 'f32'
        w.desc = f'reduce.{name} over {n * 4 / 2 ** 30:.0f} GiB fp32, 1% NaN'
        w.metric, w.unit = f'{name} elements/sec', 'elements/s'
        fn = N.reduce.nansum if name == 'nansum' else N.reduce.nanmax
        w.step = lambda: fn(x)
        w.kernel = f'reduce_all_k1<float, {0 if name == "nansum" else 1}>'
        w.check = lambda: (0.0, True)

        def cpu(budget_s):
            import oracle as O
            ns = min(n, 1 << 28)
            x_h = x[:ns].cpu().numpy()
            return time_cpu(lambda: O.reduce(name, x_h), ns, budget_s), f'first {ns * 4 / 2 ** 30:.0f} GiB'
        w.cpu = cpu
    else:
        raise SystemExit(f'unknown workload {name}')
    return w


def host_cores():
    """CPU threads this process may actually use: the affinity mask, capped by the cgroup
    CPU quota when there is one (a GPU box gives each GPU a share of the host)."""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            txt = open(path).read().split()
            if path.endswith('cpu.max'):
                if txt[0] != 'max':
                    cores = min(cores, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
                if q > 0:
                    cores = min(cores, max(1, int(q / per + 0.5)))
            break
        except Exception:
            continue
    return cores


def time_cpu(fn, units, budget_s):
    """best-of repeated runs of the oracle within ~budget_s seconds"""
    fn()
    best, t_end = float('inf'), time.time() + budget_s
    reps = 0
    while time.time() < t_end or reps < 2:
        t0 = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t0)
        reps += 1
    return units / best


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    from nitorch_fastmath_amd.shard import max_over_ranks
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', rank=rank, world_size=world,
                                device_id=torch.device('cuda', local_rank))
    if a.gpus != world and rank == 0 and world > 1:
        print(f'warning: --gpus {a.gpus} but WORLD_SIZE {world}', file=sys.stderr)
    device = torch.device('cuda', local_rank if world > 1 else 0)
    torch.cuda.set_device(device)

    w = make_workload(a.workload, a.n, device, rank, a.layout)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        w.step()
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()                       # same stream the kernels are launched on
    for _ in range(a.steps):
        w.step()
    ev1.record()
    barrier()
    wall = time.perf_counter() - t0
    wall = max_over_ranks(wall, device=device)
    kern_ms = ev0.elapsed_time(ev1) / a.steps          # average launch duration on the stream
    gather_ms = None
    if a.gather and getattr(w, 'output', None) is not None and world > 1:
        from nitorch_fastmath_amd.shard import gather_outputs
        barrier()
        tg = time.perf_counter()
        full = gather_outputs(w.output.contiguous(), world * w.output.shape[0])
        barrier()
        gather_ms = max_over_ranks(time.perf_counter() - tg, device=device) * 1e3
        del full

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    value = world * w.units * a.steps / wall
    achieved = w.units * w.bytes_per_unit / (kern_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, 'profiles', f'traffic_{a.workload}.json')
    if os.path.exists(tpath):          # PMC pass result (separate rocprofv3 --pmc runs), per launch
        try:
            traffic = json.load(open(tpath)).get('hbm_bytes_per_launch')
        except Exception:
            traffic = None
    line = {
        'metric': w.metric, 'value': value, 'unit': w.unit, 'n_gpus': world, 'steps': a.steps,
        'warmup': a.warmup, 'ms_per_step': wall / a.steps * 1e3, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': w.dtype, 'data': 'synthetic',
        'config': {'workload': w.desc, 'per_gpu_units': w.units, 'parallelism': f'batch-shard x{world}',
                   'layout': a.layout, 'kernel': w.kernel},
        'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                     'bytes_per_unit': w.bytes_per_unit, 'kernel_ms': kern_ms,
                     'frac_of_achievable_6300': achieved / 6300.0},
    }
    if gather_ms is not None:
        line['gather_ms'] = gather_ms   # optional xGMI all-gather of the outputs, outside `value`
    if world == 1 and not a.no_cpu:
        # the CPU leg (rank 0, one GPU only): the oracle is timed as the baseline and, while it is
        # loaded, checks a sample of the GPU output of the last timed step
        import oracle as O
        O.build()
        err, exact = w.check()
        line['parity'] = {'max_rel_err_vs_oracle': err, 'bit_exact_vs_oracle': exact}
        avail = host_cores()
        best = None
        # a one-GPU box owns a share of the host (16 threads by the pool's rule); try that
        # and everything visible, keep the faster, report the thread count actually used
        for cores in sorted({min(16, avail), avail}):
            O.set_num_threads(cores)
            rate, sample = w.cpu(6.0)
            if best is None or rate > best[0]:
                best = (rate, cores, sample)
        line['cpu_baseline'] = {'value': best[0], 'unit': w.unit, 'cores': best[1], 'kind': 'port',
                                'sample': best[2] + ' (C/OpenMP oracle, best of repeated runs, ~6 s per thread count)',
                                'host_threads_visible': avail}
    print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
