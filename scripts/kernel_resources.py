#!/usr/bin/env python
"""Per-kernel resources of libnfm_hip.so straight from the gfx950 code objects (the
`amdhsa.kernels` metadata notes): VGPRs, SGPRs, static LDS, SCRATCH (private segment), and the
waves/SIMD the register count allows (granule 8, min(8, 512 // alloc): MI355X_MICROARCH.md).
rocprofv3's kernel-trace columns are not used for this: its VGPR_Count is not the allocation and
its LDS_Block_Size omits dynamic LDS.

usage: kernel_resources.py [--match SUBSTR ...] [--json out.json] [--md out.md] [objects or .so ...]
       (default: every object under nitorch_fastmath_amd/csrc)"""
import glob
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = '/opt/rocm/lib/llvm/bin'


def code_objects(path, tmp):
    """gfx950 code objects embedded in a host object / shared library (.hip_fatbin bundles)"""
    fat = os.path.join(tmp, os.path.basename(path) + '.fatbin')
    r = subprocess.run([f'{LLVM}/llvm-objcopy', f'--dump-section=.hip_fatbin={fat}', path, os.devnull],
                       capture_output=True, text=True)
    if r.returncode != 0 or not os.path.exists(fat):
        return []
    data = open(fat, 'rb').read()
    outs = []
    # a .so concatenates one bundle per translation unit, each starting with the magic string
    starts = [m.start() for m in re.finditer(b'__CLANG_OFFLOAD_BUNDLE__', data)]
    for i, st in enumerate(starts):
        piece = os.path.join(tmp, f'{os.path.basename(path)}.{i}.bundle')
        open(piece, 'wb').write(data[st:starts[i + 1] if i + 1 < len(starts) else len(data)])
        co = piece + '.co'
        r = subprocess.run([f'{LLVM}/clang-offload-bundler', '--type=o', '--unbundle', f'--input={piece}',
                            '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', f'--output={co}'], capture_output=True, text=True)
        if r.returncode == 0 and os.path.exists(co) and os.path.getsize(co) > 0:
            outs.append(co)
    return outs


def kernels_of(co):
    txt = subprocess.run([f'{LLVM}/llvm-readelf', '--notes', co], capture_output=True, text=True).stdout
    out = []
    for blk in re.split(r'\n\s+- \.agpr_count:', txt)[1:]:
        blk = '.agpr_count:' + blk

        def field(name, cast=int):
            m = re.search(r'\.' + name + r':\s+(\S+)', blk)
            return cast(m.group(1)) if m else None
        sym = field('name', str)
        if not sym:
            continue
        out.append({'symbol': sym, 'vgpr': field('vgpr_count'), 'agpr': field('agpr_count'), 'sgpr': field('sgpr_count'),
                    'lds_static': field('group_segment_fixed_size'), 'scratch': field('private_segment_fixed_size'),
                    'max_flat_workgroup_size': field('max_flat_workgroup_size')})
    return out


def demangle(names):
    r = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True)
    return r.stdout.split('\n')[:len(names)]


def waves_per_simd(vgpr, agpr):
    alloc = -(-max((vgpr or 0) + (agpr or 0), 1) // 8) * 8
    return min(8, 512 // alloc)


def collect(paths):
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for p in paths:
            for co in code_objects(p, tmp):
                for k in kernels_of(co):
                    k['object'] = os.path.basename(p)
                    rows.append(k)
    for k, d in zip(rows, demangle([k['symbol'] for k in rows])):
        d = re.sub(r'\(.*', '', d).replace('void ', '').replace('nfm::', '')
        k['kernel'] = d
        k['waves_per_simd'] = waves_per_simd(k['vgpr'], k['agpr'])
    return rows


def main():
    args = sys.argv[1:]
    match, js, md, paths = [], None, None, []
    while args:
        a = args.pop(0)
        if a == '--match':
            match.append(args.pop(0))
        elif a == '--json':
            js = args.pop(0)
        elif a == '--md':
            md = args.pop(0)
        else:
            paths.append(a)
    if not paths:
        paths = sorted(glob.glob(os.path.join(ROOT, 'nitorch_fastmath_amd', 'csrc', '*.o')))
    rows = collect(paths)
    summary = {'kernels': len(rows), 'with_scratch': sum(1 for k in rows if k['scratch']),
               'max_scratch_bytes_per_lane': max([k['scratch'] or 0 for k in rows] or [0]),
               'max_vgpr': max([k['vgpr'] or 0 for k in rows] or [0])}
    sel = [k for k in rows if not match or any(m in k['kernel'] for m in match)]
    lines = ['| kernel | VGPR | SGPR | static LDS B | scratch B/lane | waves/SIMD | object |', '|---|---|---|---|---|---|---|']
    for k in sel:
        lines.append(f"| `{k['kernel'][:100]}` | {k['vgpr']} | {k['sgpr']} | {k['lds_static']} | {k['scratch']} | "
                     f"{k['waves_per_simd']} | {k['object']} |")
    lines += ['', f"library total: {summary['kernels']} kernels, {summary['with_scratch']} with scratch "
              f"(max {summary['max_scratch_bytes_per_lane']} B/lane), max {summary['max_vgpr']} VGPRs"]
    if any(k['scratch'] for k in rows):
        lines += ['', 'kernels with scratch:'] + [f"* `{k['kernel'][:110]}`: {k['scratch']} B/lane, {k['vgpr']} VGPRs"
                                                  for k in rows if k['scratch']]
    text = '\n'.join(lines)
    if md:
        open(md, 'w').write(text + '\n')
    if js:
        json.dump({'summary': summary, 'kernels': sel}, open(js, 'w'), indent=1)
    print(text if len(sel) <= 60 else '\n'.join(lines[-(4 + summary['with_scratch']):]))
    return summary


if __name__ == '__main__':
    main()
