cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_conv_dma.py -x -q -k "weight_gradient" 2>&1 | tail -15 > gpurun_out/r06/t_wgrad.txt
cat gpurun_out/r06/t_wgrad.txt
export WG_ONLY=1 FWD_ONLY=0 WG_GEMM_ONLY=1
export SHAPES="16,64,64,256,256;16,256,256,64,64;16,128,128,128,128;16,64,64,256,256;16,32,32,512,512;16,32,32,1024,512;8,64,64,256,256"
for z in "" xw; do for tune in 16 17 0 16 0; do ZERO=$z WG_TUNE=$tune python tools/bench_conv_split.py 2>/dev/null | grep "^wgrad" | tail -6 | sed "s/^/ZERO='$z' tune=$tune /"; done; done > gpurun_out/r06/wgrad_ring_vs_row.txt
cat gpurun_out/r06/wgrad_ring_vs_row.txt
