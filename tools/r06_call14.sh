cd $GRAFT_REPO_ROOT
for x in 0 1 0 1; do RPNET_CORR_BWD_XCD=$x python tools/bench_corr_bwd.py 2>/dev/null | sed "s/^/XCD=$x /"; done
