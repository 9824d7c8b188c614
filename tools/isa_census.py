#!/usr/bin/env python3
"""Static ISA census of the hot variant of k_chain_fused (interior window, 5-tap blur, no streak stage).

Compiles vkit_amd/csrc/fused.hip with -DVKX_FUSED_CENSUS (only that variant is instantiated) and -g1, attributes every
instruction to a phase of the tile body through its .loc line, and prints instruction counts per phase and class.
Static counts: phase A's raster loops run a data-dependent number of times; phases C/D (4 iterations of 2 window rows
per wavefront, unrolled or not as the compiler chose) and E (8 output rows, unrolled) have fixed trip counts, stated in
the output.  Dynamic totals come from the PMC runs (profiles/*pmc*).
Usage: tools/isa_census.py [--asm out.s] > profiles/r2_isa_census.md"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'vkit_amd', 'csrc', 'fused.hip')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-fvisibility=hidden',
         '-Wno-unused-function', '-g1', '-DVKX_FUSED_CENSUS=2', '-S', '--cuda-device-only']


def phase_table():
    """(first line, name) from the phase markers in the source."""
    marks = [('// ---- A: clear the ownership plane', 'A prologue + raster'), ('// ---- C + D:', 'C map'),
             ('unsigned long long ta[CGROUP]', 'C gather'), ('static_assert(CGROUP == 2, "phase D', 'D h-blur'),
             ('// ---- E:', 'E setup + noise load'), ('uint32_t P = epx;', 'E v-blur'),
             ('if (hue_on) P = vkd::hue_shift_packed(lsdiv, lhdiv, lsel, hue_delta, r, g, b);', 'E hue'),
             ('if (noise || (STREAK && streak_on))', 'E noise add'),
             ('// 4 adjacent pixels = 12 bytes', 'E pack + store')]
    table = []
    with open(SRC) as f:
        lines = f.readlines()
    start = next(i for i, l in enumerate(lines) if 'void chain_tile(' in l)
    table.append((start + 1, 'prologue'))
    for needle, name in marks:
        idx = next((i for i in range(start, len(lines)) if needle in lines[i]), None)
        if idx is not None:
            table.append((idx + 1, name))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith('template <bool STREAK>'))
    table.append((end + 1, 'kernel wrapper'))
    table.sort()
    return table, (start + 1, end + 1)


def classify(op):
    if op.startswith(('v_fma_f64', 'v_mul_f64', 'v_add_f64', 'v_div_', 'v_rcp_f64', 'v_cvt_f64', 'v_cvt_f32_f64', 'v_cvt_i32_f64',
                      'v_ldexp_f64', 'v_cmp_class_f64', 'v_cmp_eq_f64', 'v_cmp_neq_f64', 'v_cmp_lt_f64')) or '_f64' in op:
        return 'VALU fp64'
    if op.startswith('v_'):
        return 'VALU'
    if op.startswith('s_'):
        if op.startswith(('s_waitcnt', 's_nop', 's_barrier', 's_cbranch', 's_branch', 's_endpgm')):
            return 'SALU ctl'
        return 'SALU'
    if op.startswith('ds_'):
        return 'LDS'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
        return 'VMEM'
    return 'other'


def main():
    asm_out = sys.argv[sys.argv.index('--asm') + 1] if '--asm' in sys.argv else None
    with tempfile.TemporaryDirectory() as tmp:
        out = asm_out or os.path.join(tmp, 'census.s')
        subprocess.run(['/opt/rocm/bin/hipcc'] + FLAGS + ['-o', out, SRC], check=True, capture_output=True)
        text = open(out).read()
    m = re.search(r'^(_ZN12_GLOBAL__N_113k_chain_fusedILb0[^:\n]*):', text, re.M)
    body = text[m.start():]
    body = body[:body.index('.end_amdhsa_kernel')]
    table, (tile_first, tile_last) = phase_table()
    files = {int(n): name for n, name in re.findall(r'\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', text)}
    files[0] = 'fused.hip'

    def phase_of(line):
        name = 'kernel wrapper'
        for first, n in table:
            if line >= first:
                name = n
        return name if tile_first <= line < tile_last else 'kernel wrapper'

    counts = collections.defaultdict(collections.Counter)
    ops = collections.defaultdict(collections.Counter)
    cur = 'kernel wrapper'
    for raw in body.splitlines():
        s = raw.strip()
        loc = re.match(r'\.loc\s+(\d+)\s+(\d+)', s)
        if loc:
            fno, line = int(loc.group(1)), int(loc.group(2))
            fname = os.path.basename(files.get(fno, ''))
            if fname == 'fused.hip':
                # div_pair / mad_u24 / helpers above the tile body keep the phase of their caller
                if tile_first <= line < tile_last:
                    cur = phase_of(line)
            elif fname == 'vkx_color.h':
                cur = 'E hue'
            continue
        if not s or s.startswith(('.', ';', '//')) or s.endswith(':'):
            continue
        op = s.split()[0]
        if not re.match(r'^(v_|s_|ds_|global_|buffer_|flat_|scratch_)', op):
            continue
        cls = classify(op)
        counts[cur][cls] += 1
        ops[cur][op] += 1
    classes = ['VALU', 'VALU fp64', 'SALU', 'SALU ctl', 'LDS', 'VMEM']
    # the kernel's own entry of the amdhsa metadata (entries list .name before .vgpr_count)
    meta = re.search(r'\.name:\s+' + re.escape(m.group(1)) + r'\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)', text)
    print('# ISA census: k_chain_fused, interior window, 5-tap blur, no streak (static instruction counts)\n')
    print('Generated by `tools/isa_census.py` from the compiler\'s gfx950 assembly of `vkit_amd/csrc/fused.hip` '
          '(`-DVKX_FUSED_CENSUS -g1`); phases from the `.loc` line of every instruction.\n')
    print('| phase | ' + ' | '.join(classes) + ' | total |')
    print('|---|' + '---|' * (len(classes) + 1))
    order = [n for _, n in table]
    seen = []
    for n in order:
        if n in counts and n not in seen:
            seen.append(n)
    tot = collections.Counter()
    for n in seen:
        row = counts[n]
        tot.update(row)
        print(f'| {n} | ' + ' | '.join(str(row.get(c, 0)) for c in classes) + f' | {sum(row.values())} |')
    print('| **all** | ' + ' | '.join(str(tot.get(c, 0)) for c in classes) + f' | {sum(tot.values())} |')
    print('\nTrip counts: A -- data dependent (spans: one lane per (cell, scanline), loop over the span; outline: one lane '
          'per (cell, edge), loop over the steps); C / D -- per wavefront 8 window rows, 2 per iteration; E -- 8 output rows, '
          'unrolled.\n')
    for n in seen:
        top = ', '.join(f'{op} x{c}' for op, c in ops[n].most_common(14))
        print(f'* **{n}**: {top}')
    if meta:
        print(f'\nVGPRs of this single-variant build: {meta.group(1)} (the production kernel holds every variant and is compiled to 64 VGPRs, '
              '8 wavefronts per SIMD)')


if __name__ == '__main__':
    main()
