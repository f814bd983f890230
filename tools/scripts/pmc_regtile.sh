# PMC passes over gemm_regtile.hip (each counter group in its own pass, --kernel-trace only): L2 hit / miss, fetch bytes, wave-state shares, at
# M = 256 and M = 512 (N = K = 4096) and at config[1].  Usage (gpurun): bash tools/scripts/pmc_regtile.sh <tag>
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for shape in "1 4096 4096 64" "256 4096 4096 0" "512 4096 4096 0"; do
  s=$(echo $shape | tr ' ' 'x')
  for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE"; do
    g=$(echo $grp | cut -d' ' -f1)
    rocprofv3 --pmc $grp --kernel-trace -d gpurun_out/${tag}pmcr_${s}_$g -o out --output-format csv -- python3 tools/pmc_regtile_run.py $shape > gpurun_out/${tag}pmcr_${s}_$g.log 2>&1 || { echo "failed $s $g"; tail -3 gpurun_out/${tag}pmcr_${s}_$g.log; continue; }
    python3 tools/pmc_summarize.py gpurun_out/${tag}pmcr_${s}_$g gemm_regtile_kernel 20 > gpurun_out/${tag}_pmc_regtile_${s}_$g.json
    rm -rf gpurun_out/${tag}pmcr_${s}_$g
  done
done
ls gpurun_out/${tag}_pmc_regtile_*
