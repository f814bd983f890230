# round 2: counters of the FINAL tile GEMM, each group in its own pass (no trace domains besides --kernel-trace), the default
# bench command (60 ms pre-warm + 500 warm-up + 200 timed launches): the summary takes the last 200 dispatches
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace -d gpurun_out/r02pmc_$tag -o out --output-format csv -- python3 bench.py --no-extra --no-cpu > gpurun_out/r02pmc_$tag.log 2>&1 || exit 1
  python3 tools/pmc_summarize.py gpurun_out/r02pmc_$tag gemm_tile_kernel 200 > gpurun_out/r02pmc_$tag.json
  rm -rf gpurun_out/r02pmc_$tag
done
cat gpurun_out/r02pmc_*.json
