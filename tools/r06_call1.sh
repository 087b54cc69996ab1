cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
python bench.py --steps 20 --warmup 3 --no-other-configs > gpurun_out/r06/base_bench.json 2> gpurun_out/r06/base_bench.err
bash tools/wgrad_anatomy.sh > gpurun_out/r06/wgrad_anatomy.txt 2>&1
WG_ONLY=1 python tools/bench_conv_split.py 2>/dev/null | grep "^wgrad" > gpurun_out/r06/wgrad_with_reduce.txt
for ka in 0 1; do
  echo "== CANARY_KEEPALIVE=$ka guards off" >> gpurun_out/r06/keepalive.txt
  CANARY_KEEPALIVE=$ka RPNET_BN_LDS=big RPNET_BN_POOL_DRAIN=0 RPNET_BN_POOL_ALONE=0 timeout 600 python tools/canary_two_chains.py 12 2>&1 | grep -v amdgpu.ids | tail -3 >> gpurun_out/r06/keepalive.txt
done
echo "== PYTORCH_NO_CUDA_MEMORY_CACHING=1 guards off" >> gpurun_out/r06/keepalive.txt
PYTORCH_NO_CUDA_MEMORY_CACHING=1 RPNET_BN_LDS=big RPNET_BN_POOL_DRAIN=0 RPNET_BN_POOL_ALONE=0 timeout 900 python tools/canary_two_chains.py 8 2>&1 | grep -v amdgpu.ids | tail -3 >> gpurun_out/r06/keepalive.txt
tail -c 600 gpurun_out/r06/base_bench.json; cat gpurun_out/r06/wgrad_anatomy.txt gpurun_out/r06/keepalive.txt
