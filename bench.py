#!/usr/bin/env python
"""bench.py -- headline benchmark of the hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

Default workload (BASELINE.json configs[1], the configuration the metric is quoted on):
`sym_solve` on 1e8 random 4x4 SPD compact-sym fp32 systems per GPU (AoS layout, the
default torch layout), inputs resident in HBM before the timed region.  One "step" =
one pass of the hot path over the whole batch = ONE kernel launch.

Multi-GPU: one process per GPU (torch.distributed, backend nccl = RCCL).  Either an
external launcher starts the ranks (`python -m torch.distributed.run --nproc-per-node N
bench.py --gpus N`: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment), or --
when `--gpus N` > 1 is given and WORLD_SIZE is NOT set -- this file starts the N ranks
itself, as child processes, BEFORE anything in the parent touches the GPU (the parent never
imports torch).  `--gpus` must equal WORLD_SIZE; a mismatch is an error, never a silent
1-GPU run.  The batch shards embarrassingly, so every rank owns the full per-GPU batch
(weak scaling) and there is no collective on the data path -- only the barrier, the
max-over-ranks of the wall time and a census of the ranks/devices (`ranks_seen`, `devices`).

Prints ONE JSON line (rank 0) with the contract fields plus
  "roofline":     HBM roofline of the dominant kernel: algorithmic bytes per launch /
                  launch duration (HIP events on the launch stream, one per step: mean,
                  median and min are reported; `achieved` uses the mean);
  "parity":       a REAL comparison of the last timed step's output against the CPU oracle
                  (which units, which tolerance, what error) -- a failed check exits non-zero;
  "cpu_baseline": the reference's CPU path shape timed on this box's host cores on a bounded
                  sample (rank 0, N = 1 only): the C/OpenMP port of the algorithm and, for the
                  4x4 solve, the torch-eager restatement of the reference's op sequence at
                  1 thread and at all threads.
Other workloads (parity-test configs, for profiling): --workload sym_solve6 | batchinv8 |
nansum | nanmax | sym_invert3 | eig3 | eig8.  `--workload null` is the launcher self-test (no
computation; runs on CPU with --backend gloo).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable copy)
TOL = {'f32': 1e-6, 'f64': 1e-12}     # BASELINE.json north_star


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=100)
    p.add_argument('--warmup', type=int, default=10)
    p.add_argument('--workload', default='sym_solve4')
    p.add_argument('--n', type=float, default=None, help='batch per GPU (default: the config size)')
    p.add_argument('--no-cpu', action='store_true', help='skip the parity + cpu_baseline leg')
    p.add_argument('--layout', default='aos', choices=['aos', 'soa'])
    p.add_argument('--settle-ms', type=float, default=300.0,
                   help='after the W warm-up steps keep launching (untimed) for this many ms so that the '
                        'power-state ramp of the first ~10 launches after an idle gap is over before the timed '
                        'region starts; the count is reported as settle_steps (0 disables)')
    p.add_argument('--gather', action='store_true',
                   help='also time the OPTIONAL epilogue: all-gather of the per-rank outputs over xGMI (reported '
                        'separately as gather_ms; never part of value)')
    p.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                   help='gloo = CPU launcher self-test (--workload null), or a rehearsal with --share-gpu')
    p.add_argument('--timeout-s', type=float, default=1500.0,
                   help='self-launched ranks: overall deadline of the job (on expiry the spawned PIDs are terminated, '
                        'then killed, and the exit code is 124); every rank: deadline of the rendezvous and of each '
                        'collective (init_process_group timeout)')
    p.add_argument('--fail-rank', type=int, default=-1,
                   help='TEST HOOK: this rank exits 1 before the rendezvous (tests/test_bench_launcher.py)')
    p.add_argument('--share-gpu', action='store_true',
                   help='REHEARSAL of the N > 1 path on a one-GPU box: every rank runs the real workload on cuda:0 '
                        '(needs --backend gloo: RCCL refuses two ranks on one device).  The line says so '
                        '(config.rehearsal) and is not a scaling measurement.')
    return p.parse_args(argv)


# ----------------------------------------------------------------------------------------------
# self-launch: N ranks as child processes, started before the parent touches the GPU
# ----------------------------------------------------------------------------------------------

def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(a, argv):
    """Start a.gpus children of this script (rank r on GPU r), wait, propagate failure.
    Rank 0 inherits stdout, so its JSON line is this process's output."""
    assert 'torch' not in sys.modules, 'the launcher must not have imported torch'
    port = _free_port()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus),
                   LOCAL_WORLD_SIZE=str(a.gpus), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   NFM_BENCH_SELF_LAUNCHED='1')
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        # rank 0's stdout is the JSON line; what the other ranks print goes to stderr (kept, not discarded)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    pending = dict(enumerate(procs))
    deadline = time.monotonic() + a.timeout_s
    kill_at = None
    while pending:
        for r, p in list(pending.items()):
            code = p.poll()
            if code is None:
                continue
            del pending[r]
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f'bench.py: rank {r} exited with {code}; stopping the other ranks', file=sys.stderr)
                for q in pending.values():      # exactly the PIDs started above
                    q.terminate()
                kill_at = time.monotonic() + 10.0
        now = time.monotonic()
        if pending and now > deadline and kill_at is None:
            print(f'bench.py: ranks {sorted(pending)} still running after --timeout-s {a.timeout_s:.0f}; '
                  f'terminating them', file=sys.stderr)
            rc = rc or 124
            for q in pending.values():
                q.terminate()
            kill_at = now + 10.0
        if pending and kill_at is not None and now > kill_at:
            for q in pending.values():          # a rank that ignored SIGTERM (hung in a collective / on the GPU)
                q.kill()
            kill_at = now + 3600.0
        time.sleep(0.05)
    return rc


# ----------------------------------------------------------------------------------------------
# workloads
# ----------------------------------------------------------------------------------------------

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


def head_tail(n, head=1_000_000, tail=100_000):
    """unit indices checked against the oracle: the first 1e6 of the timed batch
    and the last 1e5 (the ragged last tiles)"""
    import torch
    if n <= head + tail:
        return torch.arange(n)
    return torch.cat([torch.arange(head), torch.arange(n - tail, n)])


def verdict(got, ref, dn, what, exact_expected=False, scale=None):
    """parity record: max-norm relative error against the oracle and whether it is within `tol`"""
    import numpy as np
    got, ref = np.asarray(got), np.asarray(ref)
    den = float(scale) if scale is not None else float(np.abs(ref.astype(np.float64)).max())
    err = float(np.abs(got.astype(np.float64) - ref.astype(np.float64)).max() / (den if den > 0 else 1.0))
    exact = bool(np.array_equal(got, ref))
    ok = exact if exact_expected else bool(err <= TOL[dn])
    return {'max_rel_err_vs_oracle': err, 'bit_exact_vs_oracle': exact, 'tol': 0.0 if exact_expected else TOL[dn],
            'checked': what, 'ok': ok}


class Workload:
    """name, unit count per step, algorithmic bytes per unit, the step closure, the parity check, the cpu leg."""
    output = None
    cpu_eager = None
    eager_sizes = (0, 0)


def make_workload(name, n_arg, device, rank, layout):
    import torch
    w = Workload()
    seed = 1234 + rank
    if name == 'null':
        # launcher / rendezvous / census self-test: no computation, runs anywhere
        w.units, w.bytes_per_unit, w.dtype = 1, 0, 'none'
        w.desc, w.metric, w.unit = 'null (launcher self-test, no computation)', 'null steps/sec', 'steps/s'
        w.step = lambda: None
        w.kernel = None
        w.check = w.cpu = None
        return w
    import nitorch_fastmath_amd as N
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
        kind = {'aos': 1, 'soa': 2}[layout]
        w.kernel = f'rec_kernel<float, SolveOp<float, {M}, 0>, {kind}>'

        def check():
            import oracle as O
            idx = head_tail(n).to(device)
            ref = O.sym_solve(mat[idx].cpu().numpy(), vec[idx].cpu().numpy())
            # M <= 4: reference closed forms in the reference's operation order -> bit-identical
            return verdict(out[idx].cpu().numpy(), ref, 'f32', f'first 1e6 + last 1e5 of {n:.0e} systems',
                           exact_expected=(M <= 4))
        w.check = check

        def cpu(budget_s):
            import oracle as O
            ns = min(n, 20_000_000)
            m_h, v_h = mat[:ns].contiguous().cpu().numpy(), vec[:ns].contiguous().cpu().numpy()
            return time_cpu(lambda: O.sym_solve(m_h, v_h), ns, budget_s), f'first {ns:.0e} systems of the GPU batch'
        w.cpu = cpu
        if M == 4:
            w.cpu_eager = eager_timer(lambda ns: (mat[:ns].contiguous().cpu(), vec[:ns].contiguous().cpu()), 'sym_solve', n)
            w.eager_sizes = (2_000_000, 10_000_000)
    elif name == 'sym_invert3':
        n = int(n_arg or 1e5)
        mat, _ = spd_compact(n, 3, torch.float64, device, seed)
        out = torch.empty_like(mat)
        w.units, w.bytes_per_unit, w.dtype = n, 2 * 6 * 8, 'f64'
        w.desc = f'sym_invert 3x3 SPD compact-sym fp64, batch {n:.0e}'
        w.metric, w.unit = '3x3 compact-sym inversions/sec', 'inversions/s'
        w.step = lambda: N.sym_invert(mat, out=out)
        w.output = out
        w.kernel = 'rec_kernel<double, InvertOp<double, 3, false>, 1>'

        def check():
            import oracle as O
            idx = head_tail(n).to(device)
            return verdict(out[idx].cpu().numpy(), O.sym_invert(mat[idx].cpu().numpy()), 'f64',
                           f'{len(idx):.0e} of {n:.0e} matrices', exact_expected=True)
        w.check = check

        def cpu(budget_s):
            import oracle as O
            m_h = mat.cpu().numpy()
            return time_cpu(lambda: O.sym_invert(m_h), n, budget_s), 'the whole batch'
        w.cpu = cpu
        w.cpu_eager = eager_timer(lambda ns: (mat[:ns].cpu(),), 'sym_invert', n)
        w.eager_sizes = (n, n)
    elif name in ('eig3', 'eig8'):
        E = 3 if name == 'eig3' else 8
        n = int(n_arg or (5e7 if E == 3 else 8e6))
        g = torch.Generator(device=device).manual_seed(seed)
        a = torch.randn(n, E, E, device=device, generator=g)
        a = a + a.transpose(-1, -2)          # symmetric (Hessian-filter shaped workload)
        w.units, w.bytes_per_unit, w.dtype = n, (E * E + E) * 4, 'f32'
        w.desc = f'eig_sym {E}x{E} symmetric fp32 (eigenvalues), batch {n:.0e}'
        w.metric, w.unit = f'{E}x{E} symmetric eigenvalue problems/sec', 'matrices/s'
        last = {}

        def step():
            last['out'] = N.eig_sym(a, check_finite=False)
        w.step = step
        w.kernel = f'rec_kernel<float, EigSymOp<float, {E}, false, false>, 1>'

        def check():
            import numpy as np
            import oracle as O
            idx = head_tail(n).to(device)
            got, ref = last['out'][idx].cpu().numpy(), O.eig_sym(a[idx].cpu().numpy())
            # eigenvalues come unsorted, in deflation order (qr.py:45-46): compared POSITION BY POSITION
            # with the oracle's (the default arithmetic reproduces the reference's order), TOL as is
            v = verdict(got, ref, 'f32', f'first 1e6 + last 1e5 of {n:.0e} matrices (unsorted, deflation order)')
            v['same_deflation_order_frac'] = float(np.mean(np.abs(got - ref).max(-1) <= 8e-6 * np.abs(ref).max()))
            v['bit_exact_frac'] = float(np.mean((got == ref).all(-1)))
            v['ok'] = bool(v['ok'] and v['same_deflation_order_frac'] == 1.0)
            return v
        w.check = check

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
        last = {}

        def step():
            last['out'] = N.batchinv(a)
        w.step = step
        w.kernel = 'rec_kernel<double, BatchInvOp<double, 8>, 1>'

        def check():
            import oracle as O
            idx = head_tail(n).to(device)
            # Gauss-Jordan in registers vs the oracle's LU: same pivots, different summation order
            return verdict(last['out'][idx].cpu().numpy(), O.batch_inv(a[idx].cpu().numpy()), 'f64',
                           f'first 1e6 + last 1e5 of {n:.0e} matrices')
        w.check = check

        def cpu(budget_s):
            import oracle as O
            ns = min(n, 2_000_000)
            a_h = a[:ns].cpu().numpy()
            return time_cpu(lambda: O.batch_inv(a_h), ns, budget_s), f'first {ns:.0e} matrices'
        w.cpu = cpu
        w.cpu_eager = eager_timer(lambda ns: (a[:ns].cpu(),), 'batch_inv', n)
        w.eager_sizes = (200_000, 2_000_000)
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
        last = {}

        def step():
            last['out'] = fn(x)
        w.step = step
        w.kernel = f'reduce_all_k1<float, {0 if name == "nansum" else 1}>'

        def check():
            # the output is ONE scalar over the whole tensor, so the oracle has to see all of it:
            # streamed to the host in 1 GiB chunks, reduced there by the C oracle (float64 accumulation)
            import numpy as np
            import oracle as O
            got = float(last['out'])
            tot, tot_abs, mx = 0.0, 0.0, -np.inf
            for lo in range(0, n, chunk):
                x_h = x[lo:min(n, lo + chunk)].cpu().numpy()
                if name == 'nansum':
                    tot += float(O.reduce('nansum', x_h, out_f64=True))
                    tot_abs += float(O.reduce('nansum', np.abs(x_h), out_f64=True))
                else:
                    mx = max(mx, float(O.reduce('nanmax', x_h)))
            what = f'the whole tensor ({n:.3g} elements, chunked float64 oracle)'
            if name == 'nansum':     # |s - s64| / sum|x| <= 1e-6
                return verdict(np.float64(got), np.float64(tot), 'f32', what, scale=tot_abs)
            return verdict(np.float32(got), np.float32(mx), 'f32', what, exact_expected=True)
        w.check = check

        def cpu(budget_s):
            import oracle as O
            ns = min(n, 1 << 28)
            x_h = x[:ns].cpu().numpy()
            return time_cpu(lambda: O.reduce(name, x_h), ns, budget_s), f'first {ns * 4 / 2 ** 30:.0f} GiB'
        w.cpu = cpu
        w.cpu_eager = eager_timer(lambda ns: (x[:ns].cpu(),), name, n)
        w.eager_sizes = (1 << 26, 1 << 28)
    else:
        raise SystemExit(f'unknown workload {name}')
    return w


def eager_timer(make_inputs, fn_name, n):
    """cpu_baseline.torch_eager leg: the reference's CPU op sequence restated in eager torch
    (oracle/torch_eager.py), timed at a given thread count on the first `ns` units"""
    def run(threads, ns):
        import torch
        from oracle import torch_eager as T
        ns = min(n, ns)
        inputs = make_inputs(ns)
        fn = getattr(T, fn_name)
        old = torch.get_num_threads()
        torch.set_num_threads(threads)
        try:
            rate = time_cpu(lambda: fn(*inputs), ns, 0.0, reps_min=1)
        finally:
            torch.set_num_threads(old)
        return rate, ns
    return run


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


def time_cpu(fn, units, budget_s, reps_min=2):
    """best-of repeated runs within ~budget_s seconds (after one untimed run)"""
    fn()
    best, t_end = float('inf'), time.time() + budget_s
    reps = 0
    while time.time() < t_end or reps < reps_min:
        t0 = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t0)
        reps += 1
    return units / best


def stored_traffic(workload, units, layout):
    """PMC traffic per launch from the committed rocprofv3 --pmc passes (profiles/traffic_*.json).
    It is a STORED measurement, not taken in this run, and is reported only when this run's
    batch and layout are the ones the counters were collected on."""
    path = os.path.join(ROOT, 'profiles', f'traffic_{workload}.json')
    try:
        rec = json.load(open(path))
    except Exception:
        return None, None
    if int(rec.get('units', -1)) != int(units) or rec.get('layout', 'aos') != layout:
        return None, f'profiles/traffic_{workload}.json holds a different batch/layout; not reported'
    return rec.get('hbm_bytes_per_launch'), f'profiles/traffic_{workload}.json (stored rocprofv3 --pmc pass, not this run)'


def census(world, device, dist, comm_device=None):
    """every rank contributes one count and its device identity: proves N ranks on N distinct GPUs"""
    import torch
    if device.type == 'cuda':
        p = torch.cuda.get_device_properties(device)
        ident = {'rank': int(os.environ.get('RANK', '0')), 'index': device.index,
                 'name': p.name, 'uuid': str(getattr(p, 'uuid', '')),
                 'pci': '%04x:%02x:%02x' % (getattr(p, 'pci_domain_id', 0), getattr(p, 'pci_bus_id', 0),
                                            getattr(p, 'pci_device_id', 0))}
    else:
        ident = {'rank': int(os.environ.get('RANK', '0')), 'index': None, 'name': 'cpu', 'uuid': '', 'pci': ''}
    if world == 1:
        return 1, [ident]
    ones = torch.ones(1, dtype=torch.int64, device=comm_device if comm_device is not None else device)
    dist.all_reduce(ones)
    idents = [None] * world
    dist.all_gather_object(idents, ident)
    return int(ones.item()), idents


def run_rank(a):
    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if a.gpus != world:
        print(f'bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: refusing to report a {world}-rank run as '
              f'{a.gpus} GPUs (launch with --nproc-per-node {a.gpus}, or unset WORLD_SIZE and let bench.py '
              f'start the ranks)', file=sys.stderr)
        return 2
    if a.share_gpu and a.backend != 'gloo':
        print('bench.py: --share-gpu needs --backend gloo (RCCL refuses two ranks on one device)', file=sys.stderr)
        return 2
    on_gpu = a.backend == 'nccl' or a.share_gpu
    if not on_gpu and a.workload != 'null':
        print('bench.py: --backend gloo only runs the launcher self-test (--workload null) or a --share-gpu '
              'rehearsal; the product has no CPU path', file=sys.stderr)
        return 2
    if a.fail_rank == rank:
        print(f'bench.py: rank {rank} fails before the rendezvous (--fail-rank test hook)', file=sys.stderr)
        return 1
    from nitorch_fastmath_amd.shard import max_over_ranks
    if on_gpu:
        device = torch.device('cuda', 0 if (a.share_gpu or world == 1) else local_rank)
        torch.cuda.set_device(device)
    else:
        device = torch.device('cpu')
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        import datetime
        # the deadline covers the rendezvous (a rank that never arrives) and every collective after it
        tmo = datetime.timedelta(seconds=max(10.0, a.timeout_s))
        if a.backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device, timeout=tmo)
        else:
            dist.init_process_group('gloo', rank=rank, world_size=world, timeout=tmo)

    w = make_workload(a.workload, a.n, device, rank, a.layout)

    def sync():
        if on_gpu:
            torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()
        sync()

    def event():
        if not on_gpu:
            return time.perf_counter()
        e = torch.cuda.Event(enable_timing=True)
        e.record()                      # torch's current stream = the stream the kernels are launched on
        return e

    def elapsed_ms(e0, e1):
        return (e1 - e0) * 1e3 if not on_gpu else e0.elapsed_time(e1)

    for _ in range(a.warmup):
        w.step()
    # what the same K steps cost WITHOUT the settle phase below (the first launches after the idle gap of
    # input generation run slower: power-state ramp) -- reported next to the headline, never part of `value`
    cold_ms = None
    if a.settle_ms > 0 and on_gpu:
        c0 = event()
        for _ in range(a.steps):
            w.step()
        c1 = event()
        sync()
        cold_ms = elapsed_ms(c0, c1) / a.steps
    settle_steps = 0
    if a.settle_ms > 0 and on_gpu:
        sync()
        t_end = time.perf_counter() + a.settle_ms * 1e-3
        while time.perf_counter() < t_end and settle_steps < 10000:
            for _ in range(4):
                w.step()
            sync()
            settle_steps += 4
    barrier()
    marks = []
    t0 = time.perf_counter()
    for _ in range(a.steps):
        marks.append(event())
        w.step()
    marks.append(event())
    barrier()
    wall = time.perf_counter() - t0
    wall = max_over_ranks(wall, device=device if a.backend == 'nccl' else None)
    per_step = sorted(elapsed_ms(marks[i], marks[i + 1]) for i in range(a.steps))
    kern_ms = elapsed_ms(marks[0], marks[-1]) / a.steps      # average launch duration on the stream
    median_ms, min_ms = per_step[len(per_step) // 2], per_step[0]
    ranks_seen, devices = census(world, device, dist, None if a.backend == 'nccl' else torch.device('cpu'))
    # every rank checks ITS OWN last timed output against the oracle (outside the timed region) and reports its
    # own per-step kernel times, so that an N > 1 line carries parity and a slow point can be attributed
    parity = None
    if not a.no_cpu and w.check is not None:
        import oracle as O
        O.build()
        O.set_num_threads(max(1, host_cores() // world))
        parity = w.check()
    per_rank = [{'rank': rank, 'kernel_ms': kern_ms, 'kernel_ms_median': median_ms, 'kernel_ms_min': min_ms,
                 'parity': parity}]
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, per_rank[0])
        per_rank = gathered
    gather_ms = None
    if a.gather and w.output is not None and world > 1:
        from nitorch_fastmath_amd.shard import gather_outputs
        barrier()
        tg = time.perf_counter()
        full = gather_outputs(w.output.contiguous(), world * w.output.shape[0])
        barrier()
        gather_ms = max_over_ranks(time.perf_counter() - tg, device=device if a.backend == 'nccl' else None) * 1e3
        del full

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return 0
    rc = 0
    distinct = len({(d['uuid'], d['pci'], d['index']) for d in devices})
    value = world * w.units * a.steps / wall
    line = {
        'metric': w.metric, 'value': value, 'unit': w.unit, 'n_gpus': world, 'steps': a.steps,
        'warmup': a.warmup, 'ms_per_step': wall / a.steps * 1e3, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': w.dtype, 'data': 'synthetic',
        'config': {'workload': w.desc, 'per_gpu_units': w.units, 'parallelism': f'batch-shard x{world}',
                   'layout': a.layout, 'kernel': w.kernel, 'settle_steps': settle_steps,
                   'warmup_effective': a.warmup + (a.steps if cold_ms is not None else 0) + settle_steps,
                   'launcher': 'bench.py (self-launched ranks)' if os.environ.get('NFM_BENCH_SELF_LAUNCHED')
                   else ('external (WORLD_SIZE in env)' if world > 1 else 'single process')},
        'ranks_seen': ranks_seen, 'distinct_devices': distinct, 'devices': devices,
    }
    if a.share_gpu:
        line['config']['rehearsal'] = (f'{world} ranks share cuda:0 over gloo: exercises the N > 1 code path on a one-GPU '
                                       'box; NOT a scaling measurement')
    if ranks_seen != world or (on_gpu and not a.share_gpu and distinct != world):
        print(f'bench.py: census mismatch: {ranks_seen} ranks on {distinct} distinct devices for world {world}',
              file=sys.stderr)
        rc = 3
    if w.bytes_per_unit:
        achieved = w.units * w.bytes_per_unit / (kern_ms * 1e-3) / 1e9
        traffic, source = stored_traffic(a.workload, w.units, a.layout)
        line['roofline'] = {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                            'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_source': source,
                            'bytes_per_unit': w.bytes_per_unit, 'kernel_ms': kern_ms,
                            'kernel_ms_median': median_ms, 'kernel_ms_min': min_ms,
                            'frac_at_median': w.units * w.bytes_per_unit / (median_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            'frac_at_min': w.units * w.bytes_per_unit / (min_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            'frac_of_achievable_6300': achieved / 6300.0}
        if cold_ms is not None:
            # the same K steps timed right after the W warm-ups, before the settle phase
            line['roofline']['kernel_ms_without_settle'] = cold_ms
            line['roofline']['frac_without_settle'] = w.units * w.bytes_per_unit / (cold_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        if world > 1:
            line['roofline']['per_rank_kernel_ms_median'] = [r['kernel_ms_median'] for r in per_rank]
            line['roofline']['per_rank_kernel_ms'] = [r['kernel_ms'] for r in per_rank]
    if gather_ms is not None:
        line['gather_ms'] = gather_ms   # optional xGMI all-gather of the outputs, outside `value`
    checked = [r['parity'] for r in per_rank if r['parity'] is not None]
    if checked:
        # the oracle checked the output of the LAST TIMED STEP on every rank: the line carries the worst one
        worst = max(checked, key=lambda v: (not v['ok'], v['max_rel_err_vs_oracle']))
        line['parity'] = dict(worst, ranks_checked=len(checked), ranks_ok=sum(1 for v in checked if v['ok']))
        if not all(v['ok'] for v in checked) or len(checked) != world:
            print('bench.py: PARITY FAILED against the oracle', file=sys.stderr)
            rc = 4
    if world == 1 and not a.no_cpu and w.check is not None:
        # the CPU baseline leg (rank 0, one GPU only)
        import oracle as O
        avail = host_cores()
        best = None
        # a one-GPU box owns a share of the host (16 threads by the pool's rule); try that
        # and everything visible, keep the faster, report the thread count actually used
        for cores in sorted({min(16, avail), avail}):
            O.set_num_threads(cores)
            rate, sample = w.cpu(4.0)
            if best is None or rate > best[0]:
                best = (rate, cores, sample)
        line['cpu_baseline'] = {'value': best[0], 'unit': w.unit, 'cores': best[1], 'kind': 'port',
                                'sample': best[2] + ' (C/OpenMP oracle, best of repeated runs, ~4 s per thread count)',
                                'host_threads_visible': avail}
        if w.cpu_eager is not None:
            # the reference's own cost shape (component-first, ~270 full-batch ATen ops with n-sized
            # temporaries per 4x4 solve), restated in eager torch: 1 thread and all threads
            from oracle import torch_eager as T
            r1, n1 = w.cpu_eager(1, w.eager_sizes[0])
            ra, na = w.cpu_eager(avail, w.eager_sizes[1])
            line['cpu_baseline']['torch_eager'] = {
                'k1': r1, 'kall': ra, 'cores_all': avail, 'unit': w.unit,
                'sample': f'first {n1:.0e} (1 thread) / {na:.0e} ({avail} threads) units, best of 1 run after warm-up',
                'note': 'oracle/torch_eager.py: eager restatement of the reference CPU op sequence, pinned to tests/golden'}
            if a.workload == 'sym_solve4':
                line['cpu_baseline']['torch_eager']['ops_per_call'] = T.ops_per_call(4)
    print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return rc


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    a = parse(argv)
    if a.gpus < 1:
        print('bench.py: --gpus must be >= 1', file=sys.stderr)
        return 2
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        return launch_ranks(a, argv)
    return run_rank(a)


if __name__ == '__main__':
    sys.exit(main())
