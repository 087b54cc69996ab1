#!/bin/bash
# Round-5 baseline: bench line, kernel trace of the eager default step and of the graph-replayed step (host out of the picture),
# MFMA-idle analysis of both.  Usage (gpurun): bash tools/prof_r05_base.sh <tag>
TAG=${1:-r05_base}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG
R=/tmp/prof_raw_$TAG
rm -rf $O $R; mkdir -p $O $R
trap "rm -rf $R" EXIT
db() { find $1 -name "*.db" | head -1; }
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs > $O/bench.json 2> $O/bench.err
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs"
RPNET_BENCH_GRAPH=0 timeout 200 rocprofv3 --kernel-trace --stats -d $R/eager -o t -- $B > $R/eager.log 2>&1
python tools/rocpd_stats.py $(db $R/eager) $O/kernel_stats_eager.csv
python tools/mfma_idle.py $(db $R/eager) $O/mfma_idle_eager.txt; rm -rf $R/eager
RPNET_BENCH_GRAPH=1 timeout 200 rocprofv3 --kernel-trace --stats -d $R/graph -o t -- $B > $R/graph.log 2>&1
python tools/rocpd_stats.py $(db $R/graph) $O/kernel_stats_graph.csv
python tools/mfma_idle.py $(db $R/graph) $O/mfma_idle_graph.txt; rm -rf $R/graph
tail -3 $R/*.log | cut -c1-200
ls -la $O
