#!/bin/bash
# HBM traffic of configs[4] (2-way, 512^2, T = 10, one fp16 plane) per kernel and launch: separate --pmc FETCH_SIZE / WRITE_SIZE passes
# over tools/one_step.py (the training step alone, streams serialised; bench.py itself under --pmc dies in one of its other legs on
# this ROCm: profiles/r05_pmc_c4_bisect.txt) at batch 4 (the configuration's own) and batch 1.
TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_$TAG; R=/tmp/pmcc4; rm -rf $R; mkdir -p $O $R
csvc() { find $1 -name "*counter_collection.csv" 2>/dev/null | head -1; }
for bt in 4 1; do
  S="python tools/one_step.py --size 512 --ways 2 --iters 10 --batch $bt --conv-math f16 --steps 2 --serial"
  timeout -k 5 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/pf -o p --output-format csv -- $S > $R/pf.log 2>&1; echo "batch $bt fetch rc $?"
  timeout -k 5 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/pw -o p --output-format csv -- $S > $R/pw.log 2>&1; echo "batch $bt write rc $?"
  sfx=$([ $bt = 4 ] && echo "" || echo "_batch1")
  [ -n "$(csvc $R/pf)" ] && [ -n "$(csvc $R/pw)" ] && python tools/pmc_traffic.py $(csvc $R/pf) $(csvc $R/pw) $O/${TAG}_pmc_traffic_f16_512$sfx.json > $O/${TAG}_pmc_traffic_f16_512$sfx.txt
  rm -rf $R/pf $R/pw
done
head -12 $O/${TAG}_pmc_traffic_f16_512.txt
