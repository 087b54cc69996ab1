cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "bn or batchnorm or BatchNorm or blocks or chain" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "golden or canary or repeats" 2>&1 | tail -4
echo "== 64-channel layers, tile variants (TILE = variant; default policy first)"
export FWD_ONLY=1 PLANES=2 SHAPES="16,128,128,128,128;16,256,256,64,64;16,128,128,64,128;16,128,128,128,64;16,256,256,64,64"
python tools/bench_conv_split.py 2>/dev/null | grep "^M=" | sed "s/^/policy /"
for t in 12 9 8 7; do TILE=$t python tools/bench_conv_split.py 2>/dev/null | grep "^M=" | sed "s/^/TILE=$t /"; done
bash tools/ab_lib.sh $GRAFT_REPO_ROOT/rpnet_amd/librpnet_hip_prev.so 3
