#!/bin/bash
# kernel trace of the headline step with the streams serialised (every launch owns the GPU): per-kernel durations -> $1 (csv)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=/tmp/tr_$$; rm -rf $R
RPNET_BENCH_GRAPH=0 RPNET_ASYNC_WGRAD=0 RPNET_CRE_STREAMS_TRAIN=0 RPNET_ENC_STREAMS=0 timeout 300 rocprofv3 --kernel-trace --stats -d $R -o t -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs ${@:2} > $R.log 2>&1
python tools/rocpd_stats.py $(find $R -name "*.db" | head -1) $1
rm -rf $R
