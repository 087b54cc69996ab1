cd $GRAFT_REPO_ROOT
python -W "error:The AccumulateGrad:UserWarning" bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs 2>&1 | grep -v amdgpu.ids | tail -25 | cut -c1-600
