# A/B of the split conv tile variants on one box (tools/bench_conv_split.py forward table): TILE = variant, DBG = ablation
# bits of variant 11 (1: first channel chunk only, 2: no epilogue), RELU = half of the activations zero
mkdir -p gpurun_out/ab
run() { PLANES=${P:-2} FWD_ONLY=1 timeout 120 python tools/bench_conv_split.py 2>&1 | grep "^M=" | cut -c1-52; }
for cfg in ${CFGS:-"-1,0,1 11,0,1 12,0,1"}; do IFS=, read T D R <<< "$cfg"; echo "== TILE=$T DBG=$D RELU=$R"; TILE=$T DBG=$D RELU=$R run; done > gpurun_out/ab/ab_${TAG:-x}.log 2>&1
cat gpurun_out/ab/ab_${TAG:-x}.log
