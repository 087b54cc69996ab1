cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 1700 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > gpurun_out/r06/t_gpu_all2.txt
cat gpurun_out/r06/t_gpu_all2.txt
timeout 200 python tools/aten_ops.py 2>/dev/null > gpurun_out/r06/aten_ops3.txt; head -16 gpurun_out/r06/aten_ops3.txt
timeout 300 python tools/layer_table.py gpurun_out/r06/layer_table.txt 2> gpurun_out/r06/layer_table.err; tail -3 gpurun_out/r06/layer_table.err; cat gpurun_out/r06/layer_table.txt
for i in 1 2; do python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs 2>gpurun_out/r06/bench3.err | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline'];print('bench', d['value'], d['step_ms']['median'], d['step_ms']['host_enqueue_median'], 'conv_frac', r['frac'], 'wgrad_tf', r.get('wgrad_tflops'))"; done
grep -c "AccumulateGrad" gpurun_out/r06/bench3.err
