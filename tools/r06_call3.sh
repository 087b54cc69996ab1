cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
export WG_ONLY=1 FWD_ONLY=0 WG_GEMM_ONLY=1
# (the first shape of a process warms the clocks up: listed twice, read the later lines)
export SHAPES="16,64,64,256,256;16,128,128,128,128;16,64,64,256,256;16,32,32,512,512;8,64,64,256,256"
for z in xw ""; do for a in 0 5 6 2 3 4; do ZERO=$z WG_ABL=$a python tools/bench_conv_split.py 2>/dev/null | grep "^wgrad" | tail -4 | sed "s/^/ring ZERO='$z' ABL=$a /"; done; 
  ZERO=$z WG_TUNE=16 python tools/bench_conv_split.py 2>/dev/null | grep "^wgrad" | tail -4 | sed "s/^/rowmajor ZERO='$z' ABL=0 /"; done > gpurun_out/r06/wgrad_anatomy_ring2.txt
cat gpurun_out/r06/wgrad_anatomy_ring2.txt
