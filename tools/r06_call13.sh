cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
( time python bench.py > gpurun_out/r06/bench_full.json 2> gpurun_out/r06/bench_full.err ) 2>&1 | tail -3
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06/bench_full.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('wgrad_tflops'))
for k,v in d.get('other_configs',{}).items():
    print(k, {kk:vv for kk,vv in v.items() if kk in ('value','ms_per_step','conv_frac','roofline','what')} if isinstance(v,dict) else v)
print(d['cpu_baseline'])
print({k:d[k] for k in d if k in ('graph_replay','mask_tile_skip')})
PY
