#!/bin/bash
# Is the K loop of the LDS-DMA convolution kernel stalled or power-limited?  The same launches on dense operands, on all-zero activations and
# on all-zero activations AND weights (identical instruction stream, DMAs, LDS reads and barriers; only the toggling in the matrix pipe changes).
cd $GRAFT_REPO_ROOT
export FWD_ONLY=1 PLANES=2 TILE=11 SHAPES="8,64,64,256,256;16,64,64,256,256;16,32,32,512,512"
for z in "" x xw; do echo "== ZERO='$z'"; ZERO=$z DBG=2 python tools/bench_conv_split.py 2>/dev/null | grep "^M="; done
echo "== RELU=1"; RELU=1 DBG=2 python tools/bench_conv_split.py 2>/dev/null | grep "^M="
echo "== weight gradient (conv_wgrad9_dma_kernel, f16x2): dense / constant x / constant x and dy"
for z in "" x xw; do ZERO=$z WG_ONLY=1 FWD_ONLY=0 python tools/bench_conv_split.py 2>/dev/null | grep "^wgrad" | sed "s/^/ZERO='$z' /"; done
