# Round-2 final evidence: full bench line, the same command under rocprofv3 --kernel-trace --stats, a whole bench run with extras
# under the profiler, the e2e decode step per kernel.  Usage (gpurun): bash tools/scripts/prof_r02.sh <tag>
tag=${1:-r02z}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py > gpurun_out/${tag}_bench_full.json 2> gpurun_out/${tag}_bench_full.err || exit 1
echo "bench done"
rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_prof -o out --output-format csv -- python3 bench.py --no-extra --no-cpu > gpurun_out/${tag}_bench_profiled_stdout.json 2> gpurun_out/${tag}_prof.log || exit 1
python tools/summarize_kernel_trace.py gpurun_out/${tag}_prof 12 --last 200 gemm_tile_kernel > gpurun_out/${tag}_bench_kernel_trace_summary.txt 2>&1
find gpurun_out/${tag}_prof -name '*kernel_stats.csv' -exec cp {} gpurun_out/${tag}_bench_kernel_stats.csv \;
rm -rf gpurun_out/${tag}_prof
echo "profiled bench done"
for att in cache current; do
  rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_prof_e2e_$att -o out --output-format csv -- python3 tools/e2e_profile.py 28 $att > gpurun_out/${tag}_e2e_$att.json 2> gpurun_out/${tag}_e2e_$att.log || exit 1
  python tools/summarize_kernel_trace.py gpurun_out/${tag}_prof_e2e_$att 30 > gpurun_out/${tag}_e2e_${att}_kernel_summary.txt 2>&1
  rm -rf gpurun_out/${tag}_prof_e2e_$att
done
echo "e2e profiles done"
