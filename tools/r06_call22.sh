cd $GRAFT_REPO_ROOT
timeout 300 python tools/eval_layers.py 2>/dev/null | tail -90
