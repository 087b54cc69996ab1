# the headline step's PMC passes (each under its own short timeout: a pass of configs[4]'s command dies with a segmentation fault
# under `rocprofv3 --pmc` on this tree and ROCm — its roofline.traffic stays null)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r04c; R=/tmp/prof_raw_c; rm -rf $O $R; mkdir -p $O $R
csvc() { find $1 -name "*counter_collection.csv" 2>/dev/null | head -1; }
S="env RPNET_BENCH_GRAPH=0 RPNET_ASYNC_WGRAD=0 RPNET_CRE_STREAMS_TRAIN=0 RPNET_ENC_STREAMS=0 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs"
timeout 90 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/pf -o p --output-format csv -- $S > $R/pf.log 2>&1; echo "fetch rc $?"
timeout 90 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/pw -o p --output-format csv -- $S > $R/pw.log 2>&1; echo "write rc $?"
[ -n "$(csvc $R/pf)" ] && [ -n "$(csvc $R/pw)" ] && python tools/pmc_traffic.py $(csvc $R/pf) $(csvc $R/pw) $O/r04_pmc_traffic.json > $O/r04_pmc_traffic.txt
rm -rf $R/pf $R/pw
timeout 90 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $R/pm -o p --output-format csv -- $S > $R/pm.log 2>&1; echo "mfma rc $?"
[ -n "$(csvc $R/pm)" ] && python tools/pmc_mfma.py $(csvc $R/pm) $O/r04_pmc_mfma_busy.json > $O/r04_pmc_mfma_busy.txt
rm -rf $R/pm
timeout 90 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace -d $R/ps -o p --output-format csv -- $S > $R/ps.log 2>&1; echo "sq rc $?"
[ -n "$(csvc $R/ps)" ] && python tools/pmc_sq.py $(csvc $R/ps) $O/r04_pmc_sq_wave_cycles.json > $O/r04_pmc_sq_wave_cycles.txt
rm -rf $R; ls -la $O; head -6 $O/r04_pmc_traffic.txt; head -8 $O/r04_pmc_mfma_busy.txt
