cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 600 python -m pytest tests/test_gpu_model.py -x -q -k "two_models" 2>&1 | tail -12
bash tools/profile_r06.sh r06 > gpurun_out/r06/profile_r06.log 2>&1; tail -22 gpurun_out/r06/profile_r06.log
for i in 1 2 3; do python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline'];print('bench', d['value'], d['step_ms']['median'], d['step_ms']['host_enqueue_median'], 'conv_frac', r['frac'], 'wgrad_tf', r.get('wgrad_tflops'))"; done
