cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_f16.py -x -q -k "eval" 2>&1 | tail -5
timeout 300 python tools/eval_layers.py 2>/dev/null | grep -E "ms per call|up 1|rpnet_conv_up4|rpnet_conv_fwd  " | head
timeout 300 python tools/eval_layers.py 8 2>/dev/null | grep -E "ms per call" | head -2
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());e=d['eval'];print('eval replayed', e['calls'][0]['ms_per_call_graph_replay'], e['calls'][1]['ms_per_call_graph_replay'], 'eager', e['calls'][0]['ms_per_call'], e['calls'][1]['ms_per_call'], e['fp16_scale_prediction'])"
