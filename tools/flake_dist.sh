# how often does the 2-rank bucket test fail under a kernel selection (env passes to the spawned ranks)
run() { f=0; for i in $(seq 1 $1); do timeout 300 python -m pytest tests/test_gpu_dist.py::test_two_rank_rp_net_bucket_on_one_gpu -x -q 2>&1 | grep -q "1 passed" || f=$((f+1)); done; echo "$2: $f failures of $1"; }
run ${1:-30} "default (DMA conv + DMA wgrad)"
