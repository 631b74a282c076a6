#!/bin/bash
# HBM traffic of a config-3 forward (bf16 encoders) before / after the fused encoder block: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
# separate passes of `python bench.py --plain --workload cfg3 --steps 2 --warmup 1`; run from the repo root through gpurun.
#   tools/pmc_cfg3.sh r02  ->  gpurun_out/r02_cfg3_traffic.txt
set -u
R=${1:-r02}
O=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for arm in unfused fused; do
    if [ $arm = unfused ]; then export LIME_BF16_FUSED_FFN=0; else unset LIME_BF16_FUSED_FFN; fi
    rocprofv3 --pmc FETCH_SIZE -d /tmp/pmc3_${arm}_f -o f --output-format csv -- python bench.py --plain --workload cfg3 --steps 2 --warmup 1 > $O/${R}_cfg3_pmc_${arm}_f.log 2>&1
    rocprofv3 --pmc WRITE_SIZE -d /tmp/pmc3_${arm}_w -o w --output-format csv -- python bench.py --plain --workload cfg3 --steps 2 --warmup 1 > $O/${R}_cfg3_pmc_${arm}_w.log 2>&1
done
python tools/pmc_cfg3_summary.py /tmp/pmc3_unfused_f/f_counter_collection.csv /tmp/pmc3_unfused_w/w_counter_collection.csv \
    /tmp/pmc3_fused_f/f_counter_collection.csv /tmp/pmc3_fused_w/w_counter_collection.csv 3 > $O/${R}_cfg3_traffic.txt
cat $O/${R}_cfg3_traffic.txt
