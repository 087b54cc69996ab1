#!/bin/bash
# One process per GPU on ONE node over RCCL (torch.distributed backend "nccl" on ROCm), the launch line the driver uses:
#   tools/launch_ddp.sh 8 bench [bench.py arguments]      BASELINE configs[3]: the headline step on 8 GPUs, gradients all-reduced
#   tools/launch_ddp.sh 8 train [train_rpnet.py arguments]
# RPNET_DIST_BACKEND=gloo runs the same code path without RCCL (plumbing tests: several ranks on one GPU).
# The bench line of N > 1 carries `distributed`: backend, rccl_ranks_seen (an all-reduce of ones: did RCCL see N ranks),
# allreduce_exposed_ms (the part of the gradient exchange not hidden under backward), bucket segment sizes.
set -e
N=${1:?number of GPUs}; WHAT=${2:?bench or train}; shift 2
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
export HSA_ENABLE_IPC_MODE_LEGACY=0          # dmabuf IPC: required by RCCL / device-memory sharing on this host driver
PORT=${MASTER_PORT:-29517}
case "$WHAT" in
  bench) SCRIPT="$ROOT/bench.py --gpus $N" ;;
  train) SCRIPT="$ROOT/train_rpnet.py" ;;
  *) echo "second argument: bench or train" >&2; exit 2 ;;
esac
exec python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "$PORT" $SCRIPT "$@"
