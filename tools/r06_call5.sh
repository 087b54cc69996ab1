cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests/test_gpu_conv_dma.py -x -q 2>&1 | tail -5 > gpurun_out/r06/t_convdma.txt
cat gpurun_out/r06/t_convdma.txt
bash tools/ab_lib.sh $GRAFT_REPO_ROOT/rpnet_amd/librpnet_hip_r05.so 3 > gpurun_out/r06/ab_ring.txt 2>&1
cat gpurun_out/r06/ab_ring.txt
