#!/bin/bash
# A/B of two builds of the library on ONE box: the default bench line per library (RPNET_LIB_PATH), interleaved, N rounds
# usage: bash tools/ab_lib.sh <old.so> [rounds] [extra bench args...]
cd $GRAFT_REPO_ROOT
OLD=$1; N=${2:-2}; shift; shift
run() { env RPNET_LIB_PATH=$2 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs "${@:3}" 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline'];print('$1', d['value'], d['step_ms']['median'], 'conv_frac', r['frac'], 'avg_launch_ms', r['avg_launch_ms'], 'wgrad_tf', r.get('wgrad_tflops'))"; }
for i in $(seq $N); do
  run old $OLD "$@"
  run new $GRAFT_REPO_ROOT/rpnet_amd/librpnet_hip.so "$@"
done
