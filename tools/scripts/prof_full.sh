cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r1e_full -o out --output-format csv -- python bench.py --no-cpu > gpurun_out/r01e_bench_full_profiled_stdout.json 2> gpurun_out/r01e_prof_full.log || exit 1
head -12 gpurun_out/prof_r1e_full/out_kernel_stats.csv | cut -c1-150
