#!/usr/bin/env python
"""Condense rocprofv3 --pmc SQ/GRBM passes of one kernel into a JSON summary with the derived
ratios used in DESIGN.md: instructions per wave, VALU busy fraction, lane utilisation of the
VALU instructions (divergence), transcendental share, effective clock.

usage: parse_sq.py <pass_dir> [<pass_dir> ...] <kernel-substring> <out.json>

Units (MI355X_MICROARCH.md, cycle constants): SQ_WAVE_CYCLES / SQ_BUSY_CYCLES / SQ_ACTIVE_INST_* /
SQ_WAIT_* count quad-cycles (x4 = shader cycles), summed over the chip; SQ_INSTS_* count
wave-instructions; SQ_THREAD_CYCLES_VALU = quad... (reported raw, used only as a ratio to
SQ_ACTIVE_INST_VALU x 64); GRBM_GUI_ACTIVE sums the 8 XCDs.
"""
import csv
import glob
import json
import os
import sys


def collect(d, kernel):
    """{counter: mean over dispatches of (sum over the rows of one dispatch)}, durations"""
    per = {}
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel not in r.get('Kernel_Name', ''):
                continue
            key = (r['Counter_Name'], r.get('Dispatch_Id', r.get('Correlation_Id', '0')))
            per[key] = per.get(key, 0.0) + float(r['Counter_Value'])
    out = {}
    for (name, _), v in per.items():
        out.setdefault(name, []).append(v)
    durs = []
    for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel in r.get('Kernel_Name', ''):
                durs.append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    return {k: sum(v) / len(v) for k, v in out.items()}, {k: len(v) for k, v in out.items()}, durs


def main():
    *dirs, kernel, out = sys.argv[1:]
    c, n, durs = {}, {}, []
    for d in dirs:
        ci, ni, di = collect(d, kernel)
        c.update(ci)
        n.update(ni)
        durs += di
    if not c:
        print('no counter rows for', kernel)
        sys.exit(1)
    res = {'kernel': kernel, 'counters_per_launch': c, 'launches_averaged': n,
           'kernel_ns_under_pmc': sorted(durs)[len(durs) // 2] if durs else None}
    g = c.get
    d = {}
    if g('SQ_WAVES') and g('SQ_INSTS_VALU'):
        d['valu_insts_per_wave'] = g('SQ_INSTS_VALU') / g('SQ_WAVES')
        if g('SQ_INSTS_SALU'):
            d['salu_insts_per_wave'] = g('SQ_INSTS_SALU') / g('SQ_WAVES')
    if g('SQ_ACTIVE_INST_VALU') and g('SQ_BUSY_CYCLES'):
        # fraction of the busy time in which a SIMD's VALU is issuing: per-SE busy cycles x SIMDs
        d['valu_active_over_wave_cycles'] = g('SQ_ACTIVE_INST_VALU') / g('SQ_WAVE_CYCLES') if g('SQ_WAVE_CYCLES') else None
    if g('SQ_ACTIVE_INST_ANY') and g('SQ_WAVE_CYCLES'):
        d['active_any_over_wave_cycles'] = g('SQ_ACTIVE_INST_ANY') / g('SQ_WAVE_CYCLES')
    if g('SQ_WAIT_INST_ANY') and g('SQ_WAVE_CYCLES'):
        d['wait_inst_over_wave_cycles'] = g('SQ_WAIT_INST_ANY') / g('SQ_WAVE_CYCLES')
    if g('SQ_WAIT_ANY') and g('SQ_WAVE_CYCLES'):
        d['wait_any_over_wave_cycles'] = g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES')
    if g('SQ_THREAD_CYCLES_VALU') and g('SQ_ACTIVE_INST_VALU'):
        d['valu_lane_utilisation'] = g('SQ_THREAD_CYCLES_VALU') / (64.0 * g('SQ_ACTIVE_INST_VALU'))
    if g('SQ_INSTS_VALU_TRANS_F32') and g('SQ_INSTS_VALU'):
        d['trans_f32_share_of_valu'] = g('SQ_INSTS_VALU_TRANS_F32') / g('SQ_INSTS_VALU')
    if g('GRBM_GUI_ACTIVE') and durs:
        d['effective_clock_GHz'] = g('GRBM_GUI_ACTIVE') / 8.0 / res['kernel_ns_under_pmc']
    res['derived'] = d
    json.dump(res, open(out, 'w'), indent=1)
    print(json.dumps(res['derived']))


if __name__ == '__main__':
    main()
