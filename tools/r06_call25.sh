cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/ab_lib.sh $GRAFT_REPO_ROOT/rpnet_amd/librpnet_hip_prev.so 4
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "glue or refine or pool" 2>&1 | tail -3
