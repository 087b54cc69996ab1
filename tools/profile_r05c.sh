#!/bin/bash
# Refresh of the headline step's kernel traces and MFMA-busy / traffic PMC passes on the FINAL tree of round 5 (after the SGPR pinning and
# the gradient fan-in changes): same commands as tools/profile_r05.sh, every one under `timeout -k 5`; files r05_final_*.
TAG=${1:-r05_final}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_$TAG; R=/tmp/prof_raw_$TAG
rm -rf $O $R; mkdir -p $O $R
trap "rm -rf $R" EXIT
db() { find $1 -name "*.db" 2>/dev/null | head -1; }
csvc() { find $1 -name "*counter_collection.csv" 2>/dev/null | head -1; }
SER="env RPNET_BENCH_GRAPH=0 RPNET_ASYNC_WGRAD=0 RPNET_CRE_STREAMS_TRAIN=0 RPNET_ENC_STREAMS=0"
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs"
$SER timeout -k 5 250 rocprofv3 --kernel-trace --stats -d $R/ts -o t -- $B > $R/ts.log 2>&1
python tools/rocpd_stats.py $(db $R/ts) $O/${TAG}_bench_kernel_stats.csv; rm -rf $R/ts
RPNET_BENCH_GRAPH=0 timeout -k 5 250 rocprofv3 --kernel-trace --stats -d $R/ta -o t -- $B > $R/ta.log 2>&1
python tools/rocpd_stats.py $(db $R/ta) $O/${TAG}_bench_kernel_stats_async_wgrad.csv
python tools/mfma_idle.py $(db $R/ta) $O/${TAG}_mfma_idle.txt; rm -rf $R/ta
S="$SER python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs"
timeout -k 5 150 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $R/pm -o p --output-format csv -- $S > $R/pm.log 2>&1; echo "mfma rc $?"
[ -n "$(csvc $R/pm)" ] && python tools/pmc_mfma.py $(csvc $R/pm) $O/${TAG}_pmc_mfma_busy.json > $O/${TAG}_pmc_mfma_busy.txt
rm -rf $R/pm
timeout -k 5 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/pf -o p --output-format csv -- $S > $R/pf.log 2>&1; echo "fetch rc $?"
timeout -k 5 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/pw -o p --output-format csv -- $S > $R/pw.log 2>&1; echo "write rc $?"
[ -n "$(csvc $R/pf)" ] && [ -n "$(csvc $R/pw)" ] && python tools/pmc_traffic.py $(csvc $R/pf) $(csvc $R/pw) $O/${TAG}_pmc_traffic.json > $O/${TAG}_pmc_traffic.txt
rm -rf $R/pf $R/pw
timeout -k 5 120 python tools/cpu_overhead.py 2>/dev/null | grep -v Warning > $O/${TAG}_cpu_overhead.txt
ls -la $O
