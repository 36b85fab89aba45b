"""bench.py prints ONE JSON line with the driver's contract fields plus roofline / cpu_baseline."""
import json
import os
import subprocess
import sys
import pytest
from conftest import ROOT

pytestmark = pytest.mark.gpu


def run_bench(*args):
    env = dict(os.environ)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), *args], capture_output=True, text=True,
                       timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().split('\n') if l.startswith('{')]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_default_workload_contract(dev):
    d = run_bench('--gpus', '1', '--steps', '3', '--warmup', '1', '--n', '2e6')
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 3 and d['warmup'] == 1
    assert d['higher_is_better'] is True and d['scaling'] == 'weak' and d['vs_baseline'] is None
    assert d['dtype'] == 'f32' and d['data'] == 'synthetic' and 'workload' in d['config']
    assert 'model' not in d['config']
    r = d['roofline']
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12 and r['bytes_per_unit'] == 72
    c = d['cpu_baseline']
    assert c['kind'] == 'port' and c['cores'] >= 1 and c['value'] > 0 and c['unit'] == d['unit']
    assert d['parity']['bit_exact_vs_oracle'] is True
    # the settle phase is disclosed: effective warm-up count and what the same steps cost without it
    assert d['config']['warmup_effective'] == d['warmup'] + d['steps'] + d['config']['settle_steps']
    assert r['frac_without_settle'] > 0 and r['kernel_ms_without_settle'] > 0
    assert abs(d['value'] - 2e6 * 3 / (d['ms_per_step'] * 3e-3)) / d['value'] < 1e-6


def test_other_workloads_run(dev):
    for w, n in (('sym_solve6', '1e6'), ('batchinv8', '2e5'), ('nansum', '4e7')):
        d = run_bench('--workload', w, '--steps', '2', '--warmup', '1', '--n', n, '--no-cpu')
        assert d['value'] > 0 and 'cpu_baseline' not in d and d['roofline']['achieved'] > 0


def test_config_c1_at_full_size_against_the_oracle(dev):
    """BASELINE config C1 (1e5 3x3 SPD compact-sym fp64 sym_invert) at its full size WITH the oracle
    check: closed form in the reference's operation order -> every one of the 1e5 results bit-identical"""
    d = run_bench('--workload', 'sym_invert3', '--steps', '2', '--warmup', '1')
    assert d['config']['per_gpu_units'] == 100000 and d['dtype'] == 'f64'
    assert d['parity']['bit_exact_vs_oracle'] is True and d['parity']['ok'] and d['parity']['ranks_checked'] == 1
    assert d['cpu_baseline']['value'] > 0


def test_eig3_default_arithmetic_holds_the_reference_order(dev):
    """`--workload eig3` with the DEFAULT arithmetic: unsorted (deflation-order) parity at TOL, every
    checked matrix in the oracle's order"""
    d = run_bench('--workload', 'eig3', '--steps', '2', '--warmup', '1', '--n', '3e6')
    p = d['parity']
    assert p['ok'] and p['tol'] == 1e-6 and p['max_rel_err_vs_oracle'] <= 1e-6 and p['same_deflation_order_frac'] == 1.0
    assert 'unsorted' in p['checked']


def test_multi_rank_rehearsal_on_one_gpu(dev):
    """the N > 1 path end to end with the real workload: `--gpus 3` self-launches three ranks that
    share cuda:0 over gloo (RCCL refuses two ranks on one device, so this is a rehearsal of the code
    path -- device selection, barrier, max-over-ranks wall time, census, whole-job value -- not a
    scaling measurement; the line says so)"""
    d = run_bench('--gpus', '3', '--steps', '5', '--warmup', '2', '--n', '4e6', '--backend', 'gloo', '--share-gpu',
                  '--settle-ms', '20')
    assert d['n_gpus'] == 3 and d['ranks_seen'] == 3 and d['config']['parallelism'] == 'batch-shard x3'
    assert 'rehearsal' in d['config'] and d['config']['launcher'].startswith('bench.py')
    assert sorted(x['rank'] for x in d['devices']) == [0, 1, 2] and d['distinct_devices'] == 1
    assert abs(d['value'] - 3 * 4e6 * 5 / (d['ms_per_step'] * 5e-3)) / d['value'] < 1e-6
    assert 'cpu_baseline' not in d and d['roofline']['achieved'] > 0
    # every rank checked its own output against the oracle; the line carries the worst and the per-rank times
    assert d['parity']['ranks_checked'] == 3 and d['parity']['ranks_ok'] == 3 and d['parity']['bit_exact_vs_oracle']
    assert len(d['roofline']['per_rank_kernel_ms_median']) == 3 and 'gather_ms' not in d
