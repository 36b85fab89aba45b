#!/usr/bin/env python
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of a bench.py run into
profiles/traffic_<workload>.json.

gfx950 corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE / WRITE_SIZE are in KiB;
FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane) coalesced streaming
read, so it is doubled; WRITE_SIZE is exact for 16 B/lane streaming stores.

usage: parse_pmc.py <fetch_dir> <write_dir> <kernel-substring> <out.json> [workload] [units] [layout]
(units / layout = the batch and operand layout of the profiled run; bench.py reports the stored
traffic only for runs of the same batch and layout)
"""
import csv
import glob
import json
import os
import sys


def counter_rows(d, name, kernel):
    vals = []
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get('Counter_Name') == name and kernel in r.get('Kernel_Name', ''):
                vals.append(float(r['Counter_Value']))
    return vals


def main():
    fd, wd, kernel, out = sys.argv[1:5]
    workload = sys.argv[5] if len(sys.argv) > 5 else ''
    units = int(float(sys.argv[6])) if len(sys.argv) > 6 else None
    layout = sys.argv[7] if len(sys.argv) > 7 else 'aos'
    fetch = counter_rows(fd, 'FETCH_SIZE', kernel)
    write = counter_rows(wd, 'WRITE_SIZE', kernel)
    if not fetch or not write:
        print('no counter rows found', len(fetch), len(write))
        sys.exit(1)
    f = sum(fetch) / len(fetch)
    w = sum(write) / len(write)
    res = {
        'workload': workload, 'kernel': kernel, 'units': units, 'layout': layout, 'launches_averaged': [len(fetch), len(write)],
        'FETCH_SIZE_KiB_raw': f, 'WRITE_SIZE_KiB_raw': w,
        'correction': 'FETCH_SIZE x2 (gfx950 tallies 128-B requests of wide coalesced reads at 64 B); WRITE_SIZE exact',
        'hbm_read_bytes_per_launch': 2 * f * 1024, 'hbm_write_bytes_per_launch': w * 1024,
        'hbm_bytes_per_launch': (2 * f + w) * 1024,
    }
    json.dump(res, open(out, 'w'), indent=1)
    print(json.dumps(res))


if __name__ == '__main__':
    main()
