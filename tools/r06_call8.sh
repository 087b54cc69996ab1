cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 1700 python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/r06/t_gpu_all.txt
cat gpurun_out/r06/t_gpu_all.txt
timeout 200 python tools/aten_ops.py 2>/dev/null > gpurun_out/r06/aten_ops.txt; head -50 gpurun_out/r06/aten_ops.txt
