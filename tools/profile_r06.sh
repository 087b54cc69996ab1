#!/bin/bash
# Round 6 profile set, DENSE and reproducible: every pass traces tools/one_step.py — the training step of BASELINE
# configs[1] and nothing else in the process (no zero-tile skip, no other legs) — so `avg x launches` of a kernel group is the step's.
#   r06_kernel_stats_serial.csv   rocprofv3 --kernel-trace, streams serialised (every launch owns the GPU): per-kernel durations
#   r06_kernel_stats_async.csv    the same in bench.py's default schedule (weight gradients / CRE branch on side streams)
#   r06_mfma_idle.txt, r06_timeline.txt   where the matrix pipe idles in the default schedule; the step's kernels in start order
#   r06_pmc_mfma_busy.*, r06_pmc_traffic.*  SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE and FETCH_SIZE / WRITE_SIZE, separate passes, serialised
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_$TAG; R=/tmp/prof_raw_$TAG
rm -rf $O $R; mkdir -p $O $R
trap "rm -rf $R" EXIT
db() { find $1 -name "*.db" 2>/dev/null | head -1; }
csvc() { find $1 -name "*counter_collection.csv" 2>/dev/null | head -1; }
timeout -k 5 250 rocprofv3 --kernel-trace --stats -d $R/ts -o t -- python tools/one_step.py --steps 5 --serial > $R/ts.log 2>&1
python tools/rocpd_stats.py $(db $R/ts) $O/${TAG}_kernel_stats_serial.csv; rm -rf $R/ts
timeout -k 5 250 rocprofv3 --kernel-trace --stats -d $R/ta -o t -- python tools/one_step.py --steps 5 > $R/ta.log 2>&1
python tools/rocpd_stats.py $(db $R/ta) $O/${TAG}_kernel_stats_async.csv
python tools/mfma_idle.py $(db $R/ta) $O/${TAG}_mfma_idle.txt
python tools/step_timeline.py $(db $R/ta) $O/${TAG}_timeline.txt; rm -rf $R/ta
S="python tools/one_step.py --steps 2 --serial"
timeout -k 5 150 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $R/pm -o p --output-format csv -- $S > $R/pm.log 2>&1; echo "mfma rc $?"
[ -n "$(csvc $R/pm)" ] && python tools/pmc_mfma.py $(csvc $R/pm) $O/${TAG}_pmc_mfma_busy.json > $O/${TAG}_pmc_mfma_busy.txt
rm -rf $R/pm
timeout -k 5 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/pf -o p --output-format csv -- $S > $R/pf.log 2>&1; echo "fetch rc $?"
timeout -k 5 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/pw -o p --output-format csv -- $S > $R/pw.log 2>&1; echo "write rc $?"
[ -n "$(csvc $R/pf)" ] && [ -n "$(csvc $R/pw)" ] && python tools/pmc_traffic.py $(csvc $R/pf) $(csvc $R/pw) $O/${TAG}_pmc_traffic.json > $O/${TAG}_pmc_traffic.txt
rm -rf $R/pf $R/pw
python bench.py --steps 20 --warmup 3 --no-other-configs > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
ls -la $O; head -30 $O/${TAG}_pmc_traffic.txt; head -20 $O/${TAG}_pmc_mfma_busy.txt; grep "^step" $O/${TAG}_mfma_idle.txt; head -8 $O/${TAG}_timeline.txt
