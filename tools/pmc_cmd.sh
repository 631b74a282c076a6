#!/bin/bash
# SQ counters (two rocprofv3 --pmc passes, no trace domains) of the kernels a python command launches, mean per dispatch:
#   tools/pmc_cmd.sh TAG 'kernel name prefix|prefix' tools/exp/attn_bwd_bench.py [args]   ->  gpurun_out/TAG_counters.txt
set -u
TAG=$1; KEEP=$2; shift 2
O=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES \
    -d /tmp/pmc1_$TAG -o a --output-format csv -- python "$@" > $O/${TAG}_pmc1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE \
    -d /tmp/pmc2_$TAG -o b --output-format csv -- python "$@" > $O/${TAG}_pmc2.log 2>&1
TAG=$TAG KEEP="$KEEP" python - <<'PY' > $O/${TAG}_counters.txt
import csv, collections, re, os
TAG, KEEP = os.environ['TAG'], tuple(os.environ['KEEP'].split('|'))
def short(n):
    return re.sub(r'\(.*$', '', n.replace('(anonymous namespace)::', '').replace('void ', ''))[:70]
agg = collections.OrderedDict()
for f in ('/tmp/pmc1_%s/a_counter_collection.csv' % TAG, '/tmp/pmc2_%s/b_counter_collection.csv' % TAG):
    for r in csv.DictReader(open(f)):
        k = short(r['Kernel_Name'])
        if not k.startswith(KEEP):
            continue
        k = '%s grid %s' % (k, r.get('Grid_Size', '?'))
        d = agg.setdefault(k, collections.OrderedDict())
        e = d.setdefault(r['Counter_Name'], [0.0, 0])
        e[0] += float(r['Counter_Value']); e[1] += 1
for k, d in agg.items():
    m = {c: v[0] / v[1] for c, v in d.items()}
    n = max(v[1] for v in d.values())
    cyc = m.get('GRBM_GUI_ACTIVE', 0) / 8.0
    util = m['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0 / cyc if cyc and 'SQ_VALU_MFMA_BUSY_CYCLES' in m else float('nan')
    conf = m['SQ_LDS_BANK_CONFLICT'] / m['SQ_LDS_IDX_ACTIVE'] if m.get('SQ_LDS_IDX_ACTIVE') else float('nan')
    print('%-60s launches %3d  cycles/dispatch %9.0f  MFMA pipe %.3f  LDS conflicts %.3f  LDS active / cycle / CU %.3f  VALU / wave %7.0f' % (
        k, n, cyc, util, conf, m.get('SQ_LDS_IDX_ACTIVE', 0) / 256.0 / cyc if cyc else 0, m.get('SQ_INSTS_VALU', 0) / max(m.get('SQ_WAVES', 1), 1)))
    print('    ' + '  '.join('%s %.4g' % (c, v) for c, v in m.items()))
PY
cat $O/${TAG}_counters.txt | cut -c1-260
