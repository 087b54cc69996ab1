#!/bin/bash
# Second half of the round-5 profiling pass (the first call's configs[4] PMC pass hung past `timeout 150`: rocprofv3's signal handler
# waited on the application; every command here is under `timeout -k 5`, SIGKILL five seconds after SIGTERM).
TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_$TAG
R=/tmp/prof_raw_$TAG
mkdir -p $O $R
trap "rm -rf $R" EXIT
db() { find $1 -name "*.db" 2>/dev/null | head -1; }
csvc() { find $1 -name "*counter_collection.csv" 2>/dev/null | head -1; }
SER="env RPNET_BENCH_GRAPH=0 RPNET_ASYNC_WGRAD=0 RPNET_CRE_STREAMS_TRAIN=0 RPNET_ENC_STREAMS=0"
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs"
RPNET_BENCH_GRAPH=0 timeout -k 5 250 rocprofv3 --kernel-trace --stats -d $R/ta -o t -- $B > $R/ta.log 2>&1
python tools/mfma_idle.py $(db $R/ta) $O/${TAG}_mfma_idle.txt; rm -rf $R/ta
C4="--size 512 --iters 10 --ways 2 --conv-math f16 --no-cpu-baseline"
$SER timeout -k 5 250 rocprofv3 --kernel-trace --stats -d $R/t5 -o t -- python bench.py $C4 --batch 4 --steps 3 --warmup 2 > $R/t5.log 2>&1; echo "c4 trace rc $?"
python tools/rocpd_stats.py $(db $R/t5) $O/${TAG}_bench_c5_f16_kernel_stats.csv; rm -rf $R/t5
timeout -k 5 120 python tools/cpu_overhead.py 2>/dev/null | grep -v Warning > $O/${TAG}_cpu_overhead.txt
for bt in 1 4; do
  S5="$SER python bench.py $C4 --batch $bt --steps 1 --warmup 1"
  timeout -k 5 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/pf5 -o p --output-format csv -- $S5 > $R/pf5.log 2>&1; rc=$?; echo "c4 batch $bt fetch rc $rc"
  [ $rc -ne 0 ] && { tail -5 $R/pf5.log | cut -c1-300; break; }
  timeout -k 5 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/pw5 -o p --output-format csv -- $S5 > $R/pw5.log 2>&1; rc=$?; echo "c4 batch $bt write rc $rc"
  [ $rc -ne 0 ] && { tail -5 $R/pw5.log | cut -c1-300; break; }
  sfx=$([ $bt = 4 ] && echo "" || echo "_batch1")
  python tools/pmc_traffic.py $(csvc $R/pf5) $(csvc $R/pw5) $O/${TAG}_pmc_traffic_f16_512$sfx.json > $O/${TAG}_pmc_traffic_f16_512$sfx.txt
  rm -rf $R/pf5 $R/pw5
done
ls -la $O
