cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace -d gpurun_out/pmcf_$tag -o out --output-format csv -- python bench.py --steps 20 --warmup 5 --no-extra --no-cpu > gpurun_out/pmcf_$tag.log 2>&1 || exit 1
done
ls gpurun_out | grep pmcf_
