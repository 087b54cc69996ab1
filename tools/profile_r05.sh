#!/bin/bash
# Round-5 profiling pass on one MI355X (one gpurun call): kernel traces of the headline step (streams serialised / default), where the
# matrix pipe idles, the headline step's PMC passes (HBM traffic, MFMA busy cycles, SQ wave cycles), the PMC traffic of configs[4] at
# batch 1 AND at its batch of 4 (the latter died with a segmentation fault in round 4: each pass under its own timeout), host overhead.
# Raw traces are deleted as soon as their summary exists.  Usage: bash tools/profile_r05.sh [tag]
TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_$TAG
R=/tmp/prof_raw_$TAG
rm -rf $O $R; mkdir -p $O $R
trap "rm -rf $R" EXIT
db() { find $1 -name "*.db" 2>/dev/null | head -1; }
csvc() { find $1 -name "*counter_collection.csv" 2>/dev/null | head -1; }
SER="env RPNET_BENCH_GRAPH=0 RPNET_ASYNC_WGRAD=0 RPNET_CRE_STREAMS_TRAIN=0 RPNET_ENC_STREAMS=0"
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs"
$SER timeout 250 rocprofv3 --kernel-trace --stats -d $R/ts -o t -- $B > $R/ts.log 2>&1
python tools/rocpd_stats.py $(db $R/ts) $O/${TAG}_bench_kernel_stats.csv; rm -rf $R/ts
RPNET_BENCH_GRAPH=0 timeout 250 rocprofv3 --kernel-trace --stats -d $R/ta -o t -- $B > $R/ta.log 2>&1
python tools/rocpd_stats.py $(db $R/ta) $O/${TAG}_bench_kernel_stats_async_wgrad.csv
python tools/mfma_idle.py $(db $R/ta) $O/${TAG}_mfma_idle.txt; rm -rf $R/ta
S="$SER python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs"
timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/pf -o p --output-format csv -- $S > $R/pf.log 2>&1; echo "fetch rc $?"
timeout 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/pw -o p --output-format csv -- $S > $R/pw.log 2>&1; echo "write rc $?"
[ -n "$(csvc $R/pf)" ] && [ -n "$(csvc $R/pw)" ] && python tools/pmc_traffic.py $(csvc $R/pf) $(csvc $R/pw) $O/${TAG}_pmc_traffic.json > $O/${TAG}_pmc_traffic.txt
rm -rf $R/pf $R/pw
timeout 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $R/pm -o p --output-format csv -- $S > $R/pm.log 2>&1; echo "mfma rc $?"
[ -n "$(csvc $R/pm)" ] && python tools/pmc_mfma.py $(csvc $R/pm) $O/${TAG}_pmc_mfma_busy.json > $O/${TAG}_pmc_mfma_busy.txt
rm -rf $R/pm
timeout 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace -d $R/ps -o p --output-format csv -- $S > $R/ps.log 2>&1; echo "sq rc $?"
[ -n "$(csvc $R/ps)" ] && python tools/pmc_sq.py $(csvc $R/ps) $O/${TAG}_pmc_sq_wave_cycles.json > $O/${TAG}_pmc_sq_wave_cycles.txt
rm -rf $R/ps
# configs[4]: 2-way 512^2, T = 10, one fp16 plane — at batch 1 (same kernels and tiles per image; the per-launch figures scale with the batch)
# and at its own batch of 4
for bt in 1 4; do
  S5="$SER python bench.py --size 512 --iters 10 --ways 2 --batch $bt --conv-math f16 --steps 2 --warmup 1 --no-cpu-baseline"
  timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/pf5 -o p --output-format csv -- $S5 > $R/pf5.log 2>&1; echo "c4 batch $bt fetch rc $?"
  timeout 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/pw5 -o p --output-format csv -- $S5 > $R/pw5.log 2>&1; echo "c4 batch $bt write rc $?"
  sfx=$([ $bt = 4 ] && echo "" || echo "_batch1")
  [ -n "$(csvc $R/pf5)" ] && [ -n "$(csvc $R/pw5)" ] && python tools/pmc_traffic.py $(csvc $R/pf5) $(csvc $R/pw5) $O/${TAG}_pmc_traffic_f16_512$sfx.json > $O/${TAG}_pmc_traffic_f16_512$sfx.txt
  rm -rf $R/pf5 $R/pw5
done
$SER timeout 250 rocprofv3 --kernel-trace --stats -d $R/t5 -o t -- python bench.py --size 512 --iters 10 --ways 2 --batch 4 --conv-math f16 --steps 3 --warmup 2 --no-cpu-baseline > $R/t5.log 2>&1
python tools/rocpd_stats.py $(db $R/t5) $O/${TAG}_bench_c5_f16_kernel_stats.csv; rm -rf $R/t5
timeout 120 python tools/cpu_overhead.py 2>/dev/null | grep -v Warning > $O/${TAG}_cpu_overhead.txt
tail -2 $R/*.log 2>/dev/null | cut -c1-200
ls -la $O
