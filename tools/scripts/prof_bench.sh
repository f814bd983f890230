cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r01e_bench_stdout.json 2> gpurun_out/r01e_bench_stderr.log || exit 1
tail -c 600 gpurun_out/r01e_bench_stdout.json
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r1e -o out --output-format csv -- python bench.py --no-extra > gpurun_out/r01e_bench_profiled_stdout.json 2> gpurun_out/r01e_prof.log || exit 1
head -5 gpurun_out/prof_r1e/out_kernel_stats.csv | cut -c1-200
