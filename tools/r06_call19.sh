cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 600 python tools/lds_canary.py 20 2>&1 | grep -v amdgpu.ids > gpurun_out/r06/lds_canary.txt; cat gpurun_out/r06/lds_canary.txt
timeout 1700 python -m pytest tests -m gpu -q -x 2>&1 | tail -5
