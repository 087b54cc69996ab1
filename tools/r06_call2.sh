cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_conv_dma.py -x -q -k "weight_gradient" 2>&1 | tail -15 > gpurun_out/r06/t_wgrad.txt
cat gpurun_out/r06/t_wgrad.txt
export WG_ONLY=1 FWD_ONLY=0 WG_GEMM_ONLY=1
for tune in 16 0 16 0; do WG_TUNE=$tune python tools/bench_conv_split.py 2>/dev/null | grep "^wgrad" | sed "s/^/tune=$tune /"; done > gpurun_out/r06/wgrad_ring_vs_row.txt
cat gpurun_out/r06/wgrad_ring_vs_row.txt
export SHAPES="16,64,64,256,256;16,128,128,128,128;16,32,32,512,512"
for z in "" xw; do for a in 0 2 3 4; do ZERO=$z WG_ABL=$a python tools/bench_conv_split.py 2>/dev/null | grep "^wgrad" | sed "s/^/ring ZERO='$z' ABL=$a /"; done; done > gpurun_out/r06/wgrad_anatomy_ring.txt
cat gpurun_out/r06/wgrad_anatomy_ring.txt
