cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_e2e3 -o out --output-format csv -- python tools/e2e_profile.py 28 fused > gpurun_out/prof_e2e3.log 2>&1
tail -2 gpurun_out/prof_e2e3.log
