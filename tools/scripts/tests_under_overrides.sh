# The GEMM parity tests forced through every kernel variant (tuning switches of csrc/*.hip): one pytest process per override.
# usage: tests_under_overrides.sh [tag] [first] [last]   (41 overrides x ~40 s: run it in halves under gpurun's 20-minute limit; results append)
cd $GRAFT_REPO_ROOT
out=gpurun_out/${1:-r03}_gemm_tests_under_every_kernel_override.txt
first=${2:-1}; last=${3:-999}
[ "$first" = 1 ] && : > $out
i=0
for ov in ARCQ_DECODE=1 ARCQ_DECODE=2 ARCQ_TILE_CFG=1 ARCQ_TILE_CFG=3 ARCQ_TILE_CFG=4 ARCQ_TILE_CFG=5 ARCQ_TILE_CFG=6 ARCQ_TILE_CFG=7 ARCQ_TILE_CFG=8 ARCQ_TILE_CFG=10 ARCQ_TILE_CFG=11 ARCQ_TILE_CFG=12 ARCQ_TILE_CFG=13 ARCQ_TILE_CFG=14 ARCQ_TILE_CFG=15 ARCQ_TILE_CFG=16 ARCQ_TILE_CFG=17 ARCQ_REGTILE_CFG=-1 ARCQ_REGTILE_CFG=1 ARCQ_REGTILE_CFG=2 ARCQ_REGTILE_CFG=3 ARCQ_REGTILE_CFG=4 ARCQ_REGTILE_CFG=5 ARCQ_REGTILE_CFG=6 ARCQ_REGTILE_CFG=7 ARCQ_REGTILE_CFG=8 ARCQ_REGTILE_CFG=9 ARCQ_REGTILE_CFG=10 ARCQ_REGTILE_CFG=11 ARCQ_REGTILE_CFG=12 ARCQ_REGTILE_CFG=13 ARCQ_TILE_PIPE=0 ARCQ_TILE_STAGGER=1 ARCQ_SKINNY_WAVES=8 ARCQ_REPACKED_STREAM=1 ARCQ_ROWBLOCK_SLICES=2 ARCQ_ROWBLOCK_SLICES=8 ARCQ_ROWBLOCK_DIRECT=0 ARCQ_ROWBLOCK_DIRECT=1 ARCQ_ROWTOK=0 ARCQ_ROWTOK=1 ARCQ_ROWMID_SLICES=2 ARCQ_ROWMID_SLICES=8; do
  i=$((i+1))
  [ $i -lt $first ] && continue
  [ $i -gt $last ] && break
  echo "== $ov" >> $out
  env $ov python -m pytest tests/test_gpu_parity.py tests/test_baseline_configs_gpu.py -q -m gpu -k "gemm or baseline or repacked or fused or silu" 2>&1 | tail -2 >> $out || true
done
cat $out
