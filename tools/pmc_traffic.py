"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of tools/bench_kernels.py --iters 1) into
profiles/traffic.json (read by bench.py for roofline.traffic) and a text summary.

    python tools/pmc_traffic.py fetch_counter_collection.csv write_counter_collection.csv --json profiles/traffic.json --txt profiles/r01_pmc_hbm.txt

Correction (MI355X_MICROARCH.md, HBM section): on gfx950 FETCH_SIZE reports half the bytes of a wide (16 B / lane)
streaming read -- doubled here; WRITE_SIZE is exact.  Both are in KiB.
"""
import argparse
import csv
import json
import re


def short(name):
    name = name.replace('(anonymous namespace)::', '').replace('void ', '')
    return re.sub(r'\(.*$', '', name)


def read(path, counter):
    out = {}
    for row in csv.DictReader(open(path)):
        if row['Counter_Name'] == counter:
            out.setdefault(short(row['Kernel_Name']), []).append(float(row['Counter_Value']))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('fetch_csv')
    ap.add_argument('write_csv')
    ap.add_argument('--json', required=True)
    ap.add_argument('--txt', required=True)
    ap.add_argument('--source', default=None, help='what the passes were taken of (stored as _source, quoted by bench.py)')
    a = ap.parse_args()
    f, w = read(a.fetch_csv, 'FETCH_SIZE'), read(a.write_csv, 'WRITE_SIZE')
    res, lines = {}, ['# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/collect_profiles.sh) of `python bench.py --plain --steps 3 --warmup 2`',
                      '# values: KiB per dispatch as reported; hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 -- on gfx950 FETCH_SIZE',
                      '# reports half the bytes of a wide (16 B / lane) streaming read (MI355X_MICROARCH.md, HBM section).',
                      '# Per kernel the launches are listed in order (title shape first, then body shape; warm-up launch included).']
    for k in sorted(set(f) & set(w)):
        if not k.startswith(('gemm', 'token_attn', 'mean_pool', 'embed', 'ffn_bf16', 'inproj_bf16', 'sage', 'gate_ln', 'interest_match', 'attn_')):
            continue
        n = min(len(f[k]), len(w[k]))
        per = [(2 * f[k][i] + w[k][i]) * 1024 for i in range(n)]
        res[k] = {'hbm_bytes_per_launch': int(sum(per) / n), 'launches_sampled': n, 'fetch_kib': f[k][:n], 'write_kib': w[k][:n],
                  'hbm_bytes_each': [int(x) for x in per]}
        lines += [k, '    FETCH_SIZE KiB %s' % [int(x) for x in f[k][:n]], '    WRITE_SIZE KiB %s' % [int(x) for x in w[k][:n]],
                  '    corrected HBM bytes / launch: mean %.3e, largest (body shape) %.3e' % (sum(per) / n, max(per))]
    if a.source:
        res['_source'] = a.source
    json.dump(res, open(a.json, 'w'), indent=1)
    open(a.txt, 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines))


if __name__ == '__main__':
    main()
