cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "objective or align_loss or dice" 2>&1 | tail -8
timeout 1200 python -m pytest tests/test_gpu_model.py -x -q 2>&1 | tail -8
timeout 200 python tools/aten_ops.py 2>/dev/null > gpurun_out/r06/aten_ops2.txt; head -30 gpurun_out/r06/aten_ops2.txt
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs 2>gpurun_out/r06/bench2.err | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline'];print('bench', d['value'], d['step_ms'], 'conv_frac', r['frac'], 'wgrad_tf', r.get('wgrad_tflops')); print(d['config'].get('launches'), d.get('launches'))"
grep -i "warn" gpurun_out/r06/bench2.err | head -5
