#!/bin/bash
# Where does rocprofv3 --pmc die on configs[4] (2-way, 512^2, T = 10, one fp16 plane)?  One small command per variant, rc + the
# first frames of the crash.  (bench.py under --pmc FETCH_SIZE: segmentation fault in rounds 4 and 5, a hang once.)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=/tmp/pmcb; rm -rf $R; mkdir -p $R
try() {
  tag=$1; shift
  timeout -k 5 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/$tag -o p --output-format csv -- python tools/one_step.py "$@" > $R/$tag.log 2>&1; rc=$?
  n=$(find $R/$tag -name "*counter_collection.csv" | head -1 | xargs -r wc -l | cut -d" " -f1)
  echo "== $tag ($*): rc $rc, counter rows ${n:-none}; last app line: $(grep -E '^step|^OK' $R/$tag.log | tail -1)"
  [ $rc -ne 0 ] && grep -E "^\*\*\*|    @ " $R/$tag.log | head -${FRAMES:-14} | cut -c1-160
  rm -rf $R/$tag
}
try a --size 256 --ways 1 --iters 1 --batch 1 --conv-math f16 --serial
try b --size 256 --ways 2 --iters 1 --batch 1 --conv-math f16 --serial
try c --size 512 --ways 2 --iters 2 --batch 1 --conv-math f16 --serial
try d --size 512 --ways 2 --iters 2 --batch 1 --conv-math f16x2 --serial
try e --size 512 --ways 2 --iters 10 --batch 1 --conv-math f16 --serial
try f --size 512 --ways 2 --iters 10 --batch 1 --conv-math f16
