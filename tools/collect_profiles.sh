#!/bin/bash
# Round profiles on the GPU box (run from the repo root through gpurun): the driver's bench line, rocprofv3 kernel stats of the
# plain timed loop for every configuration, the one-stream forward timeline and the two PMC passes behind profiles/traffic.json.
#   tools/collect_profiles.sh r02      -> gpurun_out/r02_*   (copy what is to be kept into profiles/; .db files stay on the box)
set -u
R=${1:-r03}
O=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
T0=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${R}_bench.json 2> $O/${R}_bench.err
echo "bench.py wall seconds: $(( $(date +%s) - T0 ))" | tee $O/${R}_bench_time.txt
prof() {   # name steps extra-args...
    local name=$1 steps=$2; shift 2
    rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o p -- python bench.py --plain --steps $steps --warmup 3 "$@" > $O/${R}_prof_$name.log 2>&1
    python tools/prof_summary.py /tmp/prof_$name/p_results.db --csv $O/${R}_kernel_stats_$name.csv > /dev/null 2>&1
    tail -1 $O/${R}_prof_$name.log | cut -c1-160
}
prof cfg2b 40
prof cfg2a 40 --workload cfg2a
prof cfg3 20 --workload cfg3
prof cfg5 5 --workload cfg5
prof train2b 20 --workload train2b
prof train4 10 --workload train4
prof train2b_dropout 20 --workload train2b_dropout
prof train2a 20 --workload train2a
prof cfg4fwd 20 --workload cfg4fwd
# every branch on ONE stream: per-kernel durations undisturbed by the forks (what bench.py's HIP-event pass measures as well)
LIME_OVERLAP_STREAMS=0 rocprofv3 --kernel-trace --stats -d /tmp/prof_tl -o p -- python bench.py --plain --steps 40 --warmup 10 > $O/${R}_prof_timeline.log 2>&1
python tools/prof_summary.py /tmp/prof_tl/p_results.db --csv $O/${R}_kernel_stats_onestream.csv --timeline > $O/${R}_forward_timeline.txt 2>&1
rocprofv3 --pmc FETCH_SIZE -d /tmp/pmc_f -o f --output-format csv -- python bench.py --plain --steps 3 --warmup 2 > $O/${R}_pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/pmc_w -o w --output-format csv -- python bench.py --plain --steps 3 --warmup 2 > $O/${R}_pmc_w.log 2>&1
cp /tmp/pmc_f/f_counter_collection.csv $O/${R}_pmc_fetch.csv
cp /tmp/pmc_w/w_counter_collection.csv $O/${R}_pmc_write.csv
python tools/pmc_traffic.py $O/${R}_pmc_fetch.csv $O/${R}_pmc_write.csv --json $O/${R}_traffic.json --txt $O/${R}_pmc_hbm.txt \
    --source "profiles/${R}_pmc_hbm.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py --plain, round ${R})" > /dev/null
# the same two passes for configs[2] (bf16) and configs[4] (1024 x 100): per-workload traffic for their bench lines
for W in cfg3 cfg5; do
    rocprofv3 --pmc FETCH_SIZE -d /tmp/pmc_f_$W -o f --output-format csv -- python bench.py --plain --workload $W --steps 2 --warmup 1 > $O/${R}_pmc_f_$W.log 2>&1
    rocprofv3 --pmc WRITE_SIZE -d /tmp/pmc_w_$W -o w --output-format csv -- python bench.py --plain --workload $W --steps 2 --warmup 1 > $O/${R}_pmc_w_$W.log 2>&1
    python tools/pmc_traffic.py /tmp/pmc_f_$W/f_counter_collection.csv /tmp/pmc_w_$W/w_counter_collection.csv --json $O/${R}_traffic_$W.json \
        --txt $O/${R}_pmc_hbm_$W.txt --source "profiles/${R}_pmc_hbm_$W.txt (rocprofv3 --pmc passes of bench.py --plain --workload $W, round ${R})" > /dev/null
done
echo collected
