cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_gpu_dist.py -x -q -k "rccl" 2>&1 | tail -6
timeout 600 python -m pytest tests/test_gpu_conv_dma.py -x -q -k "weight_gradient" 2>&1 | tail -3
S="python tools/one_step.py --steps 2 --serial"
R=/tmp/pf12; rm -rf $R; mkdir -p $R
csvc() { find $1 -name "*counter_collection.csv" 2>/dev/null | head -1; }
timeout -k 5 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/pf -o p --output-format csv -- $S > $R/pf.log 2>&1; echo "fetch rc $?"
timeout -k 5 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/pw -o p --output-format csv -- $S > $R/pw.log 2>&1; echo "write rc $?"
python tools/pmc_traffic.py $(csvc $R/pf) $(csvc $R/pw) gpurun_out/r06/pmc_traffic_xcd.json | grep -i "wgrad9\|total"
