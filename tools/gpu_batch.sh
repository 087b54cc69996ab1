# scratch batch for one gpurun call (edited per call; outputs under gpurun_out/)
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_dist.py tests/test_gpu_model.py -m gpu -q --timeout 600 --tb=short -k "forced_rccl or reduction_in_dgrad or maxpool_in_one_pass or (extension_rows and bf16x3)" 2>&1 | grep -v "UserWarning\|run_backward" | tail -60 > gpurun_out/r04_gputest_g.log
tail -60 gpurun_out/r04_gputest_g.log
( RPNET_BN_POOL_ALONE=0 python tools/ab_overlap.py | tail -1
  python tools/ab_overlap.py | tail -1
  RPNET_BN_POOL_ALONE=0 python tools/ab_overlap.py | tail -1
  python tools/ab_overlap.py | tail -1
  AB_CONFIG=c5 RPNET_BNBWD_FUSE=1 python tools/ab_overlap.py 10 | tail -1
  AB_CONFIG=c5 python tools/ab_overlap.py 10 | tail -1 ) 2>&1 | grep -v "UserWarning\|run_backward\|amdgpu.ids" > gpurun_out/r04_ab2.txt
cat gpurun_out/r04_ab2.txt
