#!/bin/bash
# A/B of the round-5 switches on ONE box: the default bench line (20 steps) per variant, back to back; prints pairs/s, median ms
cd $GRAFT_REPO_ROOT
run() { env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline'];print('$*', d['value'], d['step_ms']['median'], 'conv_frac', r['frac'], 'avg_launch_ms', r['avg_launch_ms'], 'wgrad_tf', r.get('wgrad_tflops'), 'calls', r['abi_calls_per_step'])"; }
for v in "$@"; do run $v; done
