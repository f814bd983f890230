# Round-3 evidence: full bench line, the same command under rocprofv3 --kernel-trace --stats, the e2e decode step per kernel, PMC
# passes of the tile GEMM (traffic) and of the decode GEMMs, in-kernel stamps of the stream kernel, decode-batch sweep, the
# multi-GPU paths rehearsed on one device.  Usage (gpurun): bash tools/scripts/prof_r03.sh <tag>
tag=${1:-r03z}
part=${2:-all}     # 1 = bench + profiles + tile PMC, 2 = decode PMC + stamps + sweeps + multi-GPU rehearsal (each fits one 20-minute gpurun call)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
if [ $part != 2 ]; then
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
# PMC: the tile GEMM (roofline.traffic), each counter group in its own pass, --kernel-trace only
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  g=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace -d gpurun_out/${tag}pmc_$g -o out --output-format csv -- python3 bench.py --no-extra --no-cpu > gpurun_out/${tag}pmc_$g.log 2>&1 || exit 1
  python3 tools/pmc_summarize.py gpurun_out/${tag}pmc_$g gemm_tile_kernel 200 > gpurun_out/${tag}_pmc_tile_$g.json
  rm -rf gpurun_out/${tag}pmc_$g
done
echo "pmc tile done"
fi
if [ $part = 1 ]; then ls gpurun_out/${tag}_*; exit 0; fi
# PMC: the decode GEMMs on the Qwen2.5-7B gate|up shape
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVES SQ_LDS_BANK_CONFLICT"; do
  g=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace -d gpurun_out/${tag}pmcd_$g -o out --output-format csv -- python3 tools/pmc_decode_run.py > gpurun_out/${tag}pmcd_$g.log 2>&1 || exit 1
  for k in gemm_decode_kernel gemm_rowblock_kernel gemm_stream_kernel; do
    python3 tools/pmc_summarize.py gpurun_out/${tag}pmcd_$g $k 20 > gpurun_out/${tag}_pmcd_${g}_$k.json
  done
  rm -rf gpurun_out/${tag}pmcd_$g
done
echo "pmc decode done"
ARCQ_HIP_LIB=arcquant_amd/lib/libarcq_hip_diag.so python tools/stream_stamps.py > gpurun_out/${tag}_stream_kernel_stamps.txt 2>&1
python tools/midm_bench.py > gpurun_out/${tag}_decode_batch_sweep.jsonl 2> gpurun_out/${tag}_midm.err
python tools/mall_probe.py > gpurun_out/${tag}_weights_hbm_cold_vs_cache_hot.jsonl 2> gpurun_out/${tag}_mall.err
python tools/e2e_models.py > gpurun_out/${tag}_e2e_other_models.jsonl 2> gpurun_out/${tag}_e2e_models.err
python tools/direct_ab.py > gpurun_out/${tag}_rowblock_direct_vs_image_ab.jsonl 2> gpurun_out/${tag}_direct_ab.err
echo "stamps / sweeps done"
# the multi-GPU paths on this ONE-GPU box: both ranks on cuda:0, gloo instead of RCCL (tests/test_bench_launch_gpu.py runs the same)
ARCQ_BENCH_ONE_DEVICE=1 ARCQ_BENCH_BACKEND=gloo python bench.py --gpus 2 --no-cpu > gpurun_out/${tag}_bench_gpus2_one_device_gloo.json 2> gpurun_out/${tag}_bench_gpus2.err
ARCQ_BENCH_ONE_DEVICE=1 ARCQ_BENCH_BACKEND=gloo python -m arcquant_amd.e2e --tp 2 > gpurun_out/${tag}_e2e_tp2_one_device_gloo.json 2> gpurun_out/${tag}_e2e_tp2.err
python -m arcquant_amd.e2e --tp 1 > gpurun_out/${tag}_e2e_tp1_full_layer.json 2> gpurun_out/${tag}_e2e_tp1.err
echo "multi-gpu rehearsal done"
ls gpurun_out/${tag}_*
