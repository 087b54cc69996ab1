# scratch batch for one gpurun call (edited per call; outputs under gpurun_out/)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_dist_debug4.txt
run() { echo "=== $*"; env "$@" RPNET_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 300 python bench.py --gpus 2 --steps 2 --warmup 1 --batch 2 --size 128 --iters 2 --no-cpu-baseline 2>&1 | grep -E "SIGSEGV|^\{" | cut -c1-120 | head -3; }
( for i in 1 2 3 4 5; do run A=$i; done ) > $O 2>&1
cat $O
python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r04_gputest_e.log
tail -15 gpurun_out/r04_gputest_e.log
