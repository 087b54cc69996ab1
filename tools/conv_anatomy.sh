#!/bin/bash
# Where a launch of the LDS-DMA convolution kernel (tile variant 11: 256 x 128, f16x2) spends its time: the kernel's ablation bits
# (rpnet_conv_desc.tune bits 8..: 1 = only the first channel chunk, 2 = no epilogue) on the layer shapes of the headline step.
cd $GRAFT_REPO_ROOT
export FWD_ONLY=1 PLANES=2 TILE=11 SHAPES="8,64,64,256,256;16,64,64,256,256;16,32,32,512,512;16,128,128,128,128;16,16,16,1024,1024"
for dbg in 0 1 2 3; do echo "== DBG=$dbg (1: first chunk only, 2: no epilogue)"; DBG=$dbg python tools/bench_conv_split.py 2>/dev/null | grep "^M="; done
