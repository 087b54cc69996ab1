cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_gpu_dist.py -x -q 2>&1 | tail -4
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs 2>gpurun_out/r06/bench4.err | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline'];print('bench', d['value'], d['step_ms']['median'], d['step_ms']['host_enqueue_median'], 'conv_frac', r['frac'], 'wgrad_tf', r.get('wgrad_tflops'))"
echo "warnings in stderr: $(grep -c 'AccumulateGrad' gpurun_out/r06/bench4.err)"; grep -v "amdgpu.ids" gpurun_out/r06/bench4.err | head -5
