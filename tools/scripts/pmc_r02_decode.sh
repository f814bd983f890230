# round 2: PMC counters of the decode GEMMs on the Qwen2.5-7B gate|up shape, each group in its own pass (--kernel-trace only)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVES SQ_LDS_BANK_CONFLICT"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace -d gpurun_out/r02pmcd_$tag -o out --output-format csv -- python3 tools/pmc_decode_run.py > gpurun_out/r02pmcd_$tag.log 2>&1 || exit 1
  for k in gemm_decode_kernel gemm_rowblock_kernel gemm_stream_kernel; do
    python3 tools/pmc_summarize.py gpurun_out/r02pmcd_$tag $k 20 > gpurun_out/r02pmcd_${tag}_$k.json
  done
  rm -rf gpurun_out/r02pmcd_$tag
done
ls gpurun_out/r02pmcd_*
