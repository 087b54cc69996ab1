cd $GRAFT_REPO_ROOT
run() { ok=0; for i in 1 2 3 4 5 6; do env "$@" RPNET_BENCH_FORCE_DIST=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 100 python bench.py --gpus 1 --steps 2 --warmup 1 --batch 2 --size 128 --iters 2 --no-cpu-baseline > /tmp/o.txt 2>/tmp/e.txt; rc=$?; [ $rc -eq 0 ] && grep -q hip_graph_replay /tmp/o.txt && ok=$((ok+1)) || { echo "  rc=$rc: $(grep -E "Segmentation|not permitted|Error" /tmp/e.txt | head -1 | cut -c1-160)"; }; done; echo "$*: $ok of 6 ok"; }
run RPNET_GRAPH_CAPTURE_MODE=quiesce
run RPNET_GRAPH_CAPTURE_MODE=thread_local
