#!/bin/bash
# SQ counters of the split-product kernels (gemm_sp / wgrad_sp / token_attn_sp / the blocked attention backward) inside one training
# step of config 2b and of the configs[3] shape: matrix-pipe busy share, LDS bank conflicts, VALU instructions per wave.  Counters only
# (rocprofv3 --pmc, one list per pass), no trace domains.  Run from the repo root through gpurun:
#   tools/pmc_sp_kernels.sh r03  ->  gpurun_out/r03_sp_kernel_counters.txt
set -u
R=${1:-r03}
O=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for W in train2b train4; do
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES \
        -d /tmp/pmc_sq1_$W -o a --output-format csv -- python bench.py --plain --workload $W --steps 2 --warmup 1 > $O/${R}_pmc_sq1_$W.log 2>&1
    rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE \
        -d /tmp/pmc_sq2_$W -o b --output-format csv -- python bench.py --plain --workload $W --steps 2 --warmup 1 > $O/${R}_pmc_sq2_$W.log 2>&1
done
python - <<'PY' > $O/${R}_sp_kernel_counters.txt
import csv, collections, re
def short(n):
    return re.sub(r'\(.*$', '', n.replace('(anonymous namespace)::', '').replace('void ', ''))[:70]
KEEP = ('gemm_sp_kernel', 'wgrad_sp_kernel', 'token_attn_sp', 'attn_bwd_long_sp', 'attn_bwd_sp_kernel', 'attn_stats', 'token_attn_bwd_kernel', 'gemm_pp_kernel', 'wgrad_dma')
print('# rocprofv3 --pmc (two passes per workload) of bench.py --plain --workload train2b / train4: mean per dispatch of every launch of the kernel in the run;')
print('# SQ_* cycle counters are summed over the waves / SIMDs of the chip.  MFMA pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs /')
print('# (GRBM_GUI_ACTIVE / 8 XCDs); LDS conflicts = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.')
for W in ('train2b', 'train4'):
    agg = collections.OrderedDict()
    for f in ('/tmp/pmc_sq1_%s/a_counter_collection.csv' % W, '/tmp/pmc_sq2_%s/b_counter_collection.csv' % W):
        for r in csv.DictReader(open(f)):
            k = short(r['Kernel_Name'])
            if not k.startswith(KEEP):
                continue
            d = agg.setdefault(k, collections.OrderedDict())
            e = d.setdefault(r['Counter_Name'], [0.0, 0])
            e[0] += float(r['Counter_Value']); e[1] += 1
    print('== %s' % W)
    for k, d in agg.items():
        m = {c: v[0] / v[1] for c, v in d.items()}
        n = max(v[1] for v in d.values())
        util = m['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0 / (m['GRBM_GUI_ACTIVE'] / 8.0) if m.get('GRBM_GUI_ACTIVE') and 'SQ_VALU_MFMA_BUSY_CYCLES' in m else float('nan')
        conf = m['SQ_LDS_BANK_CONFLICT'] / m['SQ_LDS_IDX_ACTIVE'] if m.get('SQ_LDS_IDX_ACTIVE') else float('nan')
        vpw = m['SQ_INSTS_VALU'] / m['SQ_WAVES'] if m.get('SQ_WAVES') else float('nan')
        print('%-72s launches %3d  cycles/dispatch %9.0f  MFMA pipe utilisation %.3f  LDS conflicts %.3f  VALU instructions / wave %7.0f' % (
            k, n, m.get('GRBM_GUI_ACTIVE', 0) / 8.0, util, conf, vpw))
        print('    ' + '  '.join('%s %.4g' % (c, v) for c, v in m.items()))
PY
tail -40 $O/${R}_sp_kernel_counters.txt | cut -c1-200
