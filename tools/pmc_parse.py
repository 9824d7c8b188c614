#!/usr/bin/env python3
"""Turns the rocprofv3 outputs of tools/record.sh (gpurun_out/{stats,pmc}_<tag>) into the committed summaries:
profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc.md, profiles/<tag>_traffic.json.

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are collected in
separate --pmc passes; both are reported in KiB; on gfx950 FETCH_SIZE tallies 128-byte requests at 64 B, so it is
doubled.  WRITE_SIZE is taken as reported (uncalibrated, stated as such).
Usage: tools/pmc_parse.py <tag> <images-per-launch> [bench.py's --noise-planes value of the record: 0 tiles, 1 planes, 2 late]"""
import csv
import glob
import json
import os
import re
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag, images = sys.argv[1], int(sys.argv[2])
    noise_mode = {0: 'tiles', 1: 'planes', 2: 'late'}[int(sys.argv[3]) if len(sys.argv) > 3 else 0]
    out_dir = os.path.join(ROOT, 'profiles')
    stats = os.path.join(ROOT, 'gpurun_out', f'stats_{tag}', 'stats_kernel_stats.csv')
    if os.path.exists(stats):
        shutil.copy(stats, os.path.join(out_dir, f'{tag}_kernel_stats_batch64.csv'))
    stats256 = os.path.join(ROOT, 'gpurun_out', f'stats256_{tag}', 'stats_kernel_stats.csv')
    if os.path.exists(stats256):
        shutil.copy(stats256, os.path.join(out_dir, f'{tag}_kernel_stats_batch256.csv'))
    sums = defaultdict(lambda: defaultdict(float))    # kernel -> counter -> sum over dispatches
    counts = defaultdict(lambda: defaultdict(int))
    for path in sorted(glob.glob(os.path.join(ROOT, 'gpurun_out', f'pmc_{tag}', 'p*', '*counter_collection.csv'))):
        with open(path) as fin:
            for row in csv.DictReader(fin):
                name = row['Kernel_Name']
                if 'k_chain' not in name and 'k_np_' not in name:
                    continue
                short = re.search(r'k_(np|chain)_[a-z_]+', name).group(0)
                short = {'k_np_draw_compact': 'k_np_draw'}.get(short, short)          # (the name the library times it under)
                sums[short][row['Counter_Name']] += float(row['Counter_Value'])
                counts[short][row['Counter_Name']] += 1
    lines = [f'# rocprofv3 PMC counters, {tag}, bench.py --batch {images} --steps 2 --warmup 1 (per launch, mean over '
             'the dispatches of the run)', '',
             'Separate `--pmc` passes with `--kernel-trace` only (tools/pmc.sh).  One launch covers the whole batch.', '']
    traffic = {}
    for kernel in sorted(sums):
        lines += [f'## {kernel}', '', '| counter | per launch | per image |', '|---|---|---|']
        per_launch = {c: sums[kernel][c] / counts[kernel][c] for c in sums[kernel]}
        for c in sorted(per_launch):
            lines.append(f'| {c} | {per_launch[c]:.6g} | {per_launch[c] / images:.6g} |')
        lines.append('')
        if 'FETCH_SIZE' in per_launch and 'WRITE_SIZE' in per_launch:
            fetch = per_launch['FETCH_SIZE'] * 1024 * 2.0
            write = per_launch['WRITE_SIZE'] * 1024
            traffic[kernel] = {'fetch_bytes_per_image': fetch / images, 'write_bytes_per_image': write / images,
                               'hbm_bytes_per_image': (fetch + write) / images}
            lines += [f'HBM traffic per image: fetch {fetch / images / 1e6:.2f} MB (FETCH_SIZE KiB x 1024 x 2, the gfx950 '
                      f'correction) + write {write / images / 1e6:.2f} MB (WRITE_SIZE KiB x 1024) = '
                      f'{(fetch + write) / images / 1e6:.2f} MB', '']
        d = per_launch
        if 'SQ_INSTS_VALU' in d and 'SQ_WAVES' in d:
            lines += [f'Per wavefront: VALU {d["SQ_INSTS_VALU"] / d["SQ_WAVES"]:.0f}, SALU '
                      f'{d.get("SQ_INSTS_SALU", 0) / d["SQ_WAVES"]:.0f}, LDS {d.get("SQ_INSTS_LDS", 0) / d["SQ_WAVES"]:.0f}, '
                      f'VMEM rd {d.get("SQ_INSTS_VMEM_RD", 0) / d["SQ_WAVES"]:.1f}, wr '
                      f'{d.get("SQ_INSTS_VMEM_WR", 0) / d["SQ_WAVES"]:.1f} instructions', '']
        if 'SQ_WAIT_ANY' in d and 'SQ_WAVE_CYCLES' in sums[kernel]:
            pass
        if 'SQ_LDS_BANK_CONFLICT' in d and 'SQ_LDS_IDX_ACTIVE' in d and d['SQ_LDS_IDX_ACTIVE']:
            lines += [f'LDS bank-conflict cycles / LDS active cycles: '
                      f'{d["SQ_LDS_BANK_CONFLICT"] / d["SQ_LDS_IDX_ACTIVE"]:.3f}', '']
    with open(os.path.join(out_dir, f'{tag}_pmc_batch{images}.md'), 'w') as fout:
        fout.write('\n'.join(lines))
    # what bench.py reads (bench.TRAFFIC_FILE): per kernel of the step, HBM bytes and VALU instructions per image
    kernels = {}
    for kernel, t in traffic.items():
        d = {c: sums[kernel][c] / counts[kernel][c] for c in sums[kernel]}
        entry = dict(t)
        if 'SQ_INSTS_VALU' in d and 'SQ_WAVES' in d:
            entry.update(valu_insts_per_image=d['SQ_INSTS_VALU'] / images, valu_insts_per_wavefront=d['SQ_INSTS_VALU'] / d['SQ_WAVES'])
            kernels[kernel] = entry
    if kernels:
        out = {'kernels': kernels, 'images': images, 'noise_mode': noise_mode,
               'source': f'rocprofv3 --pmc, separate passes, batch {images} (profiles/{tag}_pmc_batch{images}.md); FETCH_SIZE doubled '
                         f'(gfx950), WRITE_SIZE as reported'}
        # the digest of the kernel sources the GPU box profiled (tools/record.sh writes it next to the counters)
        dpath = os.path.join(ROOT, 'gpurun_out', f'digest_{tag}.txt')
        if os.path.exists(dpath):
            out.update(kernel_source_digest=open(dpath).read().strip(), profile=f'profiles/{tag}_pmc_batch{images}.md')
        with open(os.path.join(out_dir, f'{tag}_traffic.json'), 'w') as fout:
            json.dump(out, fout, indent=1)
        if 'kernel_source_digest' in out:
            with open(os.path.join(out_dir, 'current_traffic.json'), 'w') as fout:
                json.dump(out, fout, indent=1)
    print('\n'.join(lines))


if __name__ == '__main__':
    main()
