cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 1700 python -m pytest tests -m gpu -q 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/profile_r06.sh r06 > gpurun_out/r06/profile_r06.log 2>&1; grep -E "^step|kernels, total|rc " gpurun_out/r06/profile_r06.log
( time python bench.py > gpurun_out/r06/bench_final.json 2> gpurun_out/r06/bench_final.err ) 2>&1 | tail -3
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06/bench_final.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('wgrad_tflops'), d['step_ms']['host_enqueue_median'])
for k,v in d.get('other_configs',{}).items():
    print(k, v['value'], v['ms_per_step'], v['roofline']['frac'], v['roofline'].get('wgrad_tflops'))
e=d['eval']; print('eval', e['calls'][0]['ms_per_call_graph_replay'], e['calls'][1]['ms_per_call_graph_replay'], e['fp16_scale_prediction'])
print(d['cpu_baseline']['value'], d['cpu_baseline']['gpu_over_cpu'], d.get('mask_tile_skip',{}).get('value'), d.get('graph_replay',{}).get('value'))
PY
grep -c AccumulateGrad gpurun_out/r06/bench_final.err
