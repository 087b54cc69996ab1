#!/bin/bash
# Where does conv_wgrad9_dma_kernel (f16x2) lose its time?  The same launches with parts of the K loop removed (template parameter ABL,
# rpnet_conv_desc.tune bits 8-10; results are then meaningless): 1 = only the centre x strip is fetched (the DMA count a strip ring would
# have), 2 = no DMA, 3 = no fragment reads, 4 = neither — on dense operands and on operands that toggle nothing in the matrix pipe.
cd $GRAFT_REPO_ROOT
export WG_ONLY=1 FWD_ONLY=0 WG_GEMM_ONLY=1 SHAPES="${SHAPES:-16,64,64,256,256;8,64,64,256,256;16,128,128,128,128;16,32,32,512,512}"
for z in "" xw; do for a in 0 1 2 3 4; do
  ZERO=$z WG_ABL=$a python tools/bench_conv_split.py 2>/dev/null | grep "^wgrad" | sed "s/^/ZERO='$z' ABL=$a /"
done; done
