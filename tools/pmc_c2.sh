#!/bin/bash
# HBM traffic of configs[2] (1-way 5-shot, 256^2, T = 5, batch 16, f16x2) per kernel and launch, as tools/pmc_c4.sh: separate --pmc FETCH_SIZE /
# WRITE_SIZE passes over tools/one_step.py (the training step alone, streams serialised) -> profiles/rNN_pmc_traffic_f16x2_256.json, which
# bench.py reads for other_configs.configs[2].roofline.traffic.
TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_$TAG; R=/tmp/pmcc2; rm -rf $R; mkdir -p $O $R
csvc() { find $1 -name "*counter_collection.csv" 2>/dev/null | head -1; }
S="python tools/one_step.py --size 256 --ways 1 --shots 5 --iters 5 --batch 16 --conv-math f16x2 --steps 2 --serial"
timeout -k 5 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/pf -o p --output-format csv -- $S > $R/pf.log 2>&1; echo "fetch rc $?"
timeout -k 5 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/pw -o p --output-format csv -- $S > $R/pw.log 2>&1; echo "write rc $?"
[ -n "$(csvc $R/pf)" ] && [ -n "$(csvc $R/pw)" ] && python tools/pmc_traffic.py $(csvc $R/pf) $(csvc $R/pw) $O/${TAG}_pmc_traffic_f16x2_256.json > $O/${TAG}_pmc_traffic_f16x2_256.txt
head -8 $O/${TAG}_pmc_traffic_f16x2_256.txt
