#!/bin/bash
# each order N times, a fresh process each; prints one line per order: successes / crashes by exit code
cd $GRAFT_REPO_ROOT
N=${1:-12}
for order in before after; do
  ok=0; codes=""
  for i in $(seq $N); do
    timeout 180 python tools/graph_capture_order.py $order > /tmp/gco.log 2>&1; rc=$?
    if [ $rc -eq 0 ] && grep -q "^OK" /tmp/gco.log; then ok=$((ok+1)); else codes="$codes $rc"; fi
  done
  echo "order=$order: $ok of $N runs completed; exit codes of the others:${codes:- none}"
  grep "^OK" /tmp/gco.log | tail -1
done
