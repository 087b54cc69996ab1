cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
db() { find $1 -name "*.db" 2>/dev/null | head -1; }
OLD=$GRAFT_REPO_ROOT/rpnet_amd/librpnet_hip_r05.so
# serial-mode A/B: nothing overlaps, so a faster kernel must show
run() { env RPNET_LIB_PATH=$2 RPNET_ASYNC_WGRAD=0 RPNET_CRE_STREAMS_TRAIN=0 RPNET_ENC_STREAMS=0 RPNET_BENCH_GRAPH=0 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline'];print('serial $1', d['value'], d['step_ms']['median'], 'conv_frac', r['frac'], 'wgrad_tf', r.get('wgrad_tflops'))"; }
for i in 1 2; do run old $OLD; run new $GRAFT_REPO_ROOT/rpnet_amd/librpnet_hip.so; done > gpurun_out/r06/ab_ring_serial.txt 2>&1
cat gpurun_out/r06/ab_ring_serial.txt
for w in old new; do
  L=$GRAFT_REPO_ROOT/rpnet_amd/librpnet_hip.so; [ $w = old ] && L=$OLD
  R=/tmp/tl_$w; rm -rf $R
  RPNET_LIB_PATH=$L timeout -k 5 200 rocprofv3 --kernel-trace -d $R -o t -- python tools/one_step.py --steps 6 > $R.log 2>&1
  python tools/step_timeline.py $(db $R) gpurun_out/r06/timeline_$w.txt
  python tools/mfma_idle.py $(db $R) gpurun_out/r06/mfma_idle_$w.txt
  head -8 gpurun_out/r06/timeline_$w.txt; grep "^step" gpurun_out/r06/mfma_idle_$w.txt
  rm -rf $R
done
