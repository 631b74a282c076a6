#!/bin/bash
# SQ counters of the bf16 encoder-block kernels on the body-chunk shape of config 3 (tools/bench_ffn.py): MFMA busy cycles, LDS bank
# conflicts, wait shares.  Counters only (rocprofv3 --pmc, one list per pass), no trace domains.  Run from the repo root through gpurun:
#   tools/pmc_bf16_kernels.sh r02  ->  gpurun_out/r02_bf16_kernel_counters.txt
set -u
R=${1:-r02}
O=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES \
    -d /tmp/pmc_sq1 -o a --output-format csv -- python tools/bench_ffn.py > $O/${R}_pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE \
    -d /tmp/pmc_sq2 -o b --output-format csv -- python tools/bench_ffn.py > $O/${R}_pmc_sq2.log 2>&1
python - <<'PY' > $O/${R}_bf16_kernel_counters.txt
import csv, collections, re
def short(n):
    return re.sub(r'\(.*$', '', n.replace('(anonymous namespace)::', '').replace('void ', ''))[:60]
agg = collections.OrderedDict()
for f in ('/tmp/pmc_sq1/a_counter_collection.csv', '/tmp/pmc_sq2/b_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = short(r['Kernel_Name'])
        if not (k.startswith('ffn_bf16') or k.startswith('inproj_bf16') or k.startswith('gemm_pp_kernel')):
            continue
        d = agg.setdefault(k, collections.OrderedDict())
        e = d.setdefault(r['Counter_Name'], [0.0, 0])
        e[0] += float(r['Counter_Value']); e[1] += 1
print('# rocprofv3 --pmc (two passes) of tools/bench_ffn.py: mean per dispatch; SQ_* cycle counters are summed over the waves / SIMDs of the chip')
for k, d in agg.items():
    m = {c: v[0] / v[1] for c, v in d.items()}
    print(k)
    print('    ' + '  '.join('%s %.4g' % (c, v) for c, v in m.items()))
    if 'SQ_BUSY_CYCLES' in m and 'SQ_VALU_MFMA_BUSY_CYCLES' in m and m['SQ_BUSY_CYCLES']:
        print('    MFMA busy / SQ busy = %.3f' % (m['SQ_VALU_MFMA_BUSY_CYCLES'] / m['SQ_BUSY_CYCLES']))
    if 'SQ_LDS_BANK_CONFLICT' in m and m.get('SQ_LDS_IDX_ACTIVE'):
        print('    LDS bank conflict cycles / LDS active cycles = %.3f' % (m['SQ_LDS_BANK_CONFLICT'] / m['SQ_LDS_IDX_ACTIVE']))
PY
cat $O/${R}_bf16_kernel_counters.txt
