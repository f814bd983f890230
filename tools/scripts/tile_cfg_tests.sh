# The GEMM parity tests forced through the tile configurations named on the command line (one pytest process each).
cd $GRAFT_REPO_ROOT
out=gpurun_out/${OUT:-r03t}/tile_cfg_tests.txt
mkdir -p $(dirname $out); : > $out
rc=0
for ov in "$@"; do
  echo "== $ov" >> $out
  env $ov python -m pytest tests/test_gpu_parity.py tests/test_baseline_configs_gpu.py -q -m gpu -k "gemm or baseline" 2>&1 | tail -2 >> $out || rc=1
done
cat $out
exit $rc
