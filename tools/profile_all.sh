#!/bin/bash
# rocprofv3 evidence for profiles/: kernel trace (async default + serialised), PMC traffic (two passes), MFMA busy.
# Run on the GPU box through gpurun; raw outputs land in gpurun_out/ (scratch), summaries are made by tools/*.py.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof
rm -rf $O; mkdir -p $O
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline"
S="python bench.py --steps 2 --warmup 1 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_async -o t -- $B > $O/trace_async.log 2>&1
RPNET_ASYNC_WGRAD=0 timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_serial -o t -- $B > $O/trace_serial.log 2>&1
RPNET_ASYNC_WGRAD=0 timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -o p --output-format csv -- $S > $O/pmc_fetch.log 2>&1
RPNET_ASYNC_WGRAD=0 timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -o p --output-format csv -- $S > $O/pmc_write.log 2>&1
RPNET_ASYNC_WGRAD=0 timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_mfma -o p --output-format csv -- $S > $O/pmc_mfma.log 2>&1
for f in trace_async trace_serial; do tail -1 $O/$f.log | cut -c1-120; done
find $O -name "*.db" -o -name "*counter_collection.csv" | head
# keep only what the summaries need (gpurun_out merge limit)
find $O -name "*kernel_trace.csv" -delete 2>/dev/null
du -sh $O
