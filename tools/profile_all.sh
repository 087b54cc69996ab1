#!/bin/bash
# rocprofv3 evidence for profiles/: kernel trace (serialised + async default), PMC traffic (two passes), MFMA busy, and
# the bench lines of the same tree.  Run on the GPU box through gpurun:   bash tools/profile_all.sh r02
# Budget: ~25 GPU-minutes when every pass works; every profiler pass runs under `timeout 150` (round 4 lost two calls to PMC passes
# that hung until a 300 s timeout, six in a row).  tools/profile_lean.sh is the ten-minute subset.
# Raw outputs stay in gpurun_out/prof_<tag>/raw (scratch, deleted at the end); the summaries made by tools/*.py land
# in gpurun_out/prof_<tag>/ and are copied into profiles/<tag>_* by hand.
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_$TAG
R=$O/raw
rm -rf $O; mkdir -p $R
trap "rm -rf $R" EXIT      # a killed pass must not leave raw traces behind (gpurun copies back at most 64 MiB)
# (under the tracer the host is slow enough for bench.py's probe to choose graph replays: the traced runs force eager steps)
B="env RPNET_BENCH_GRAPH=0 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs"
S="env RPNET_BENCH_GRAPH=0 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs"
db() { find $1 -name "*.db" | head -1; }
csvc() { find $1 -name "*counter_collection.csv" | head -1; }
RPNET_ASYNC_WGRAD=0 RPNET_CRE_STREAMS_TRAIN=0 RPNET_ENC_STREAMS=0 timeout 150 rocprofv3 --kernel-trace --stats -d $R/trace_serial -o t -- $B > $R/trace_serial.log 2>&1
python tools/rocpd_stats.py $(db $R/trace_serial) $O/${TAG}_bench_kernel_stats.csv
timeout 150 rocprofv3 --kernel-trace --stats -d $R/trace_async -o t -- $B > $R/trace_async.log 2>&1
python tools/rocpd_stats.py $(db $R/trace_async) $O/${TAG}_bench_kernel_stats_async_wgrad.csv
# where the matrix pipe idles inside the default (multi-stream) step: union of the MFMA-bound kernels' intervals per step
python tools/mfma_idle.py $(db $R/trace_async) $O/${TAG}_mfma_idle.txt
RPNET_ASYNC_WGRAD=0 RPNET_CRE_STREAMS_TRAIN=0 RPNET_ENC_STREAMS=0 timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/pmc_fetch -o p --output-format csv -- $S > $R/pmc_fetch.log 2>&1
RPNET_ASYNC_WGRAD=0 RPNET_CRE_STREAMS_TRAIN=0 RPNET_ENC_STREAMS=0 timeout 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/pmc_write -o p --output-format csv -- $S > $R/pmc_write.log 2>&1
python tools/pmc_traffic.py $(csvc $R/pmc_fetch) $(csvc $R/pmc_write) $O/${TAG}_pmc_traffic.json > $O/${TAG}_pmc_traffic.txt
RPNET_ASYNC_WGRAD=0 RPNET_CRE_STREAMS_TRAIN=0 RPNET_ENC_STREAMS=0 timeout 150 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $R/pmc_mfma -o p --output-format csv -- $S > $R/pmc_mfma.log 2>&1
python tools/pmc_mfma.py $(csvc $R/pmc_mfma) $O/${TAG}_pmc_mfma_busy.json > $O/${TAG}_pmc_mfma_busy.txt
RPNET_ASYNC_WGRAD=0 RPNET_CRE_STREAMS_TRAIN=0 RPNET_ENC_STREAMS=0 timeout 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace -d $R/pmc_sq -o p --output-format csv -- $S > $R/pmc_sq.log 2>&1
python tools/pmc_sq.py $(csvc $R/pmc_sq) $O/${TAG}_pmc_sq_wave_cycles.json > $O/${TAG}_pmc_sq_wave_cycles.txt
# the clock a chip-wide MFMA stream holds on zero / dense operands, and whether fragment reads hide under it
[ -x tools/probe/lds_mfma_probe ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/probe/lds_mfma_probe tools/probe/lds_mfma_probe.cpp
./tools/probe/lds_mfma_probe > $O/${TAG}_lds_mfma_probe.txt 2>&1
# host enqueue cost and launch counts of the headline step
python tools/cpu_overhead.py 2>/dev/null | grep -v Warning > $O/${TAG}_cpu_overhead.txt
# configs[4] (one fp16 plane, 2-way 512^2): HBM traffic of its conv launches (two PMC passes; bench.py picks the file up as
# roofline.traffic of other_configs.configs[4])
# (round 4: this command dies with a segmentation fault under `rocprofv3 --pmc` on this ROCm; PROFILE_C5_PMC=1 tries it anyway)
S5="env RPNET_BENCH_GRAPH=0 python bench.py --size 512 --iters 10 --ways 2 --batch 4 --conv-math f16 --steps 2 --warmup 1 --no-cpu-baseline"
[ "$PROFILE_C5_PMC" = "1" ] && {
RPNET_ASYNC_WGRAD=0 RPNET_CRE_STREAMS_TRAIN=0 RPNET_ENC_STREAMS=0 timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/pmc_fetch5 -o p --output-format csv -- $S5 > $R/pmc_fetch5.log 2>&1
RPNET_ASYNC_WGRAD=0 RPNET_CRE_STREAMS_TRAIN=0 RPNET_ENC_STREAMS=0 timeout 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/pmc_write5 -o p --output-format csv -- $S5 > $R/pmc_write5.log 2>&1
python tools/pmc_traffic.py $(csvc $R/pmc_fetch5) $(csvc $R/pmc_write5) $O/${TAG}_pmc_traffic_f16_512.json > $O/${TAG}_pmc_traffic_f16_512.txt
}
# round 4's switches, each against the default, one process per variant, back to back on this box (the tool's step has the
# zero-tile skip at its library default, ON; bench.py's headline is measured with it OFF)
( python tools/ab_overlap.py | tail -1
  RPNET_MASK_SKIP=0 python tools/ab_overlap.py | tail -1
  RPNET_WGRAD_KEEPALIVE=0 python tools/ab_overlap.py | tail -1
  RPNET_PACK_STREAM=0 python tools/ab_overlap.py | tail -1
  RPNET_CONV1_RECOMPUTE=0 python tools/ab_overlap.py | tail -1
  RPNET_BN_POOL_ALONE=0 RPNET_BN_POOL_DRAIN=0 python tools/ab_overlap.py | tail -1
  RPNET_ENC_STREAMS=3 python tools/ab_overlap.py | tail -1
  python tools/ab_overlap.py | tail -1
  AB_CONFIG=c5 python tools/ab_overlap.py 10 | tail -1
  AB_CONFIG=c5 RPNET_MASK_SKIP=0 python tools/ab_overlap.py 10 | tail -1
  AB_CONFIG=c5 RPNET_BNBWD_FUSE=1 python tools/ab_overlap.py 10 | tail -1 ) 2>/dev/null | sed 's/^configs/    configs/' > $O/${TAG}_ab_round4.txt
# A/B of the step's scheduling switches, one process per variant, back to back on this box
( RPNET_WGRAD_DEFER=0 RPNET_CRE_STREAMS_TRAIN=0 RPNET_BN_LDS=big RPNET_DICE_MULTI=0 python tools/ab_overlap.py | tail -1
  RPNET_WGRAD_DEFER=0 python tools/ab_overlap.py | tail -1
  RPNET_CRE_STREAMS_TRAIN=0 python tools/ab_overlap.py | tail -1
  python tools/ab_overlap.py | tail -1 ) > $O/${TAG}_ab_overlap.txt 2>/dev/null
# configs[4] (one fp16 plane, 2-way 512^2 T=10 batch 4): kernel trace of the same command as its bench line
C5="env RPNET_BENCH_GRAPH=0 python bench.py --size 512 --iters 10 --ways 2 --batch 4 --conv-math f16 --steps 4 --warmup 2 --no-cpu-baseline"
RPNET_ASYNC_WGRAD=0 RPNET_CRE_STREAMS_TRAIN=0 RPNET_ENC_STREAMS=0 timeout 150 rocprofv3 --kernel-trace --stats -d $R/trace_c5 -o t -- $C5 > $R/trace_c5.log 2>&1
python tools/rocpd_stats.py $(db $R/trace_c5) $O/${TAG}_bench_c5_f16_kernel_stats.csv
# the reference driver's call (eval mode, 2 slices, 256^2, T = 10): kernel trace of 33 eager + 33 replayed calls
timeout 150 rocprofv3 --kernel-trace --stats -d $R/trace_eval -o t -- python tools/bench_eval.py > $O/${TAG}_eval_call.txt 2>&1
python tools/rocpd_stats.py $(db $R/trace_eval) $O/${TAG}_eval_call_kernel_stats.csv
# the bench lines themselves (default incl. CPU baseline and eval leg; configs[4]; configs[2]; two gloo ranks on the one GPU)
python bench.py > $O/${TAG}_bench_final.json 2> $R/bench_final.err
# (configs[2] / configs[4] are part of the default line since round 3: other_configs)
RPNET_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > $O/${TAG}_bench_2ranks_gloo_one_gpu.json 2> $R/bench_2r.err
tail -2 $R/*.log | cut -c1-160
rm -rf $R
ls -la $O
