cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for b in 2 8; do for u in 0 1 0 1; do EVAL_CORR_PLANES=$u python tools/bench_eval.py $b 2>/dev/null | sed "s/^/CORR_PLANES=$u /"; done; done
