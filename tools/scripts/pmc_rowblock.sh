cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVES SQ_LDS_BANK_CONFLICT"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace -d gpurun_out/pmcrb_$tag -o out --output-format csv -- python tools/repacked_bench.py 4,37888,3584 > gpurun_out/pmcrb_$tag.log 2>&1 || exit 1
done
ls gpurun_out/pmcrb_*
