#!/usr/bin/env python
"""Condense a rocprofv3 --kernel-trace --stats CSV directory into a short, readable
summary (kernel names trimmed) for profiles/.  usage: summarize_prof.py <dir> <out.md> [title]"""
import csv
import glob
import os
import re
import sys


def trim(name):
    name = re.sub(r'\(.*', '', name)
    name = name.replace('void ', '').replace('nfm::', '')
    return name[:110]


def main():
    d, out = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else os.path.basename(d)
    stats = glob.glob(os.path.join(d, '**', '*kernel_stats.csv'), recursive=True)
    trace = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)
    lines = [f'# {title}', '', 'source: `rocprofv3 --kernel-trace --stats --output-format csv` '
             '(per-kernel durations in ns; names trimmed)', '']
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        lines += ['| kernel | calls | avg ns | min ns | max ns | % |', '|---|---|---|---|---|---|']
        for r in rows[:12]:
            lines.append(f"| `{trim(r['Name'])}` | {r['Calls']} | {float(r['AverageNs']):.0f} | {r['MinNs']} | "
                         f"{r['MaxNs']} | {float(r['Percentage']):.2f} |")
    if trace:
        rows = [r for r in csv.DictReader(open(trace[0])) if 'nfm::' in r['Kernel_Name']]
        seen = {}
        for r in rows:
            k = trim(r['Kernel_Name'])
            seen.setdefault(k, r)
        lines += ['', '| nfm kernel | workgroup | grid |', '|---|---|---|']
        for k, r in seen.items():
            lines.append(f"| `{k}` | {r['Workgroup_Size_X']} | {r['Grid_Size_X']}x{r['Grid_Size_Y']} |")
        lines += ['', 'registers / LDS / scratch per kernel: profiles/r02/kernel_resources.md (read from the code objects by '
                  'scripts/kernel_resources.py; the VGPR_Count / LDS_Block_Size columns of the rocprofv3 trace are not the '
                  'allocation -- they report 20-32 VGPRs and 0 LDS for kernels that allocate 61 and use dynamic LDS)']
        durs = {}
        for r in rows:
            durs.setdefault(trim(r['Kernel_Name']), []).append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
        lines += ['', 'durations of the nfm kernels in dispatch order, ns (the first launches are warm-up: cold TLB / '
                  'first touch; bench.py times the launches after its warm-up):', '']
        for k, v in durs.items():
            vs = sorted(v)
            lines.append(f"* `{k}`: {v[:24]}{' ...' if len(v) > 24 else ''} -- median {vs[len(vs) // 2]}, min {vs[0]}")
    open(out, 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines[:14]))


if __name__ == '__main__':
    main()
