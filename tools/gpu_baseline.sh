#!/bin/bash
# One GPU-box session: GPU tests, the plain timed loop, an overlapped (default forks) timeline of one forward.
#   tools/gpu_baseline.sh TAG   -> gpurun_out/TAG_*
set -u
R=${1:-base}
O=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/${R}_pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/${R}_pytest.log
tail -3 $O/${R}_pytest.log
timeout -k 10 300 python bench.py --plain --steps 300 --warmup 20 > $O/${R}_plain.log 2>&1 && tail -1 $O/${R}_plain.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_ov -o p -- python bench.py --plain --steps 40 --warmup 10 > $O/${R}_prof_ov.log 2>&1
python tools/prof_overlap.py /tmp/prof_ov/p_results.db > $O/${R}_overlap_timeline.txt 2>&1
tail -2 $O/${R}_overlap_timeline.txt
