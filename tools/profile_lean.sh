#!/bin/bash
# Lean version of profile_all.sh (round 4, after a full pass ran into its time limit): the kernel traces of the headline step
# (serialised / default streams), where the matrix pipe idles, the PMC traffic of configs[4]'s conv launches, host overhead, the
# round's A/B block.  Every command under its own timeout; raw traces deleted as soon as their summary exists.
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_$TAG
R=/tmp/prof_raw_$TAG
rm -rf $O $R; mkdir -p $O $R
trap "rm -rf $R" EXIT
B="env RPNET_BENCH_GRAPH=0 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs"
db() { find $1 -name "*.db" | head -1; }
csvc() { find $1 -name "*counter_collection.csv" | head -1; }
RPNET_ASYNC_WGRAD=0 RPNET_CRE_STREAMS_TRAIN=0 RPNET_ENC_STREAMS=0 timeout 200 rocprofv3 --kernel-trace --stats -d $R/trace_serial -o t -- $B > $R/trace_serial.log 2>&1
python tools/rocpd_stats.py $(db $R/trace_serial) $O/${TAG}_bench_kernel_stats.csv; rm -rf $R/trace_serial
timeout 200 rocprofv3 --kernel-trace --stats -d $R/trace_async -o t -- $B > $R/trace_async.log 2>&1
python tools/rocpd_stats.py $(db $R/trace_async) $O/${TAG}_bench_kernel_stats_async_wgrad.csv
python tools/mfma_idle.py $(db $R/trace_async) $O/${TAG}_mfma_idle.txt; rm -rf $R/trace_async
S5="env RPNET_BENCH_GRAPH=0 python bench.py --size 512 --iters 10 --ways 2 --batch 4 --conv-math f16 --steps 2 --warmup 1 --no-cpu-baseline"
RPNET_ASYNC_WGRAD=0 RPNET_CRE_STREAMS_TRAIN=0 RPNET_ENC_STREAMS=0 timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/pmc_fetch5 -o p --output-format csv -- $S5 > $R/pmc_fetch5.log 2>&1
RPNET_ASYNC_WGRAD=0 RPNET_CRE_STREAMS_TRAIN=0 RPNET_ENC_STREAMS=0 timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/pmc_write5 -o p --output-format csv -- $S5 > $R/pmc_write5.log 2>&1
python tools/pmc_traffic.py $(csvc $R/pmc_fetch5) $(csvc $R/pmc_write5) $O/${TAG}_pmc_traffic_f16_512.json > $O/${TAG}_pmc_traffic_f16_512.txt; rm -rf $R/pmc_fetch5 $R/pmc_write5
timeout 120 python tools/cpu_overhead.py 2>/dev/null | grep -v Warning > $O/${TAG}_cpu_overhead.txt
( timeout 100 python tools/ab_overlap.py | tail -1
  RPNET_MASK_SKIP=0 timeout 100 python tools/ab_overlap.py | tail -1
  RPNET_MASK_SKIP=0 RPNET_WGRAD_KEEPALIVE=0 RPNET_PACK_STREAM=0 timeout 100 python tools/ab_overlap.py | tail -1
  RPNET_MASK_SKIP=0 RPNET_BN_POOL_ALONE=0 RPNET_BN_POOL_DRAIN=0 timeout 100 python tools/ab_overlap.py | tail -1
  RPNET_MASK_SKIP=0 RPNET_ENC_STREAMS=3 timeout 100 python tools/ab_overlap.py | tail -1
  RPNET_MASK_SKIP=0 timeout 100 python tools/ab_overlap.py | tail -1 ) 2>/dev/null > $O/${TAG}_ab_round4.txt
tail -2 $R/*.log | cut -c1-160
ls -la $O
