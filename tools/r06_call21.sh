cd $GRAFT_REPO_ROOT
for x in 0 16 32 64 128 240; do RPNET_CORR_BWD_XCD=$x python tools/bench_corr_bwd.py 2>/dev/null | head -1 | sed "s/^/ABL=$((x/16)) /"; done
