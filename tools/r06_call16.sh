cd $GRAFT_REPO_ROOT
for x in 0 0; do RPNET_CORR_BWD_XCD=$x python tools/bench_corr_bwd.py 2>/dev/null | sed "s/^/ABL=$((x/16)) /"; done
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "corr" 2>&1 | tail -4
