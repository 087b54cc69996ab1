cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_ops.py -x -q -k "corr" 2>&1 | tail -4
timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_f16.py -x -q -k "eval" 2>&1 | tail -4
for b in 2 8; do for u in 1 1; do EVAL_UP4=$u python tools/bench_eval.py $b 2>/dev/null; done; done
