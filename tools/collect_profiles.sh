#!/bin/bash
# Round profiles on the GPU box (one gpurun call):  tools/collect_profiles.sh r03
#   kernel-trace stats of the bench and of the dominant kernel alone, PMC passes (HBM bytes, MFMA busy) in their own runs,
#   GEMM sweep, attention shapes.  Raw rocprofv3 output stays under gpurun_out/ (scratch); the summaries land in
#   gpurun_out/profiles_<round>/ -- copy them into profiles/ (tracked).
set -e
R=${1:-r04}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/profiles_$R; mkdir -p $OUT
P=gpurun_out/prof_raw; rm -rf $P; mkdir -p $P
echo "[1] bench kernel stats"
rocprofv3 --kernel-trace --stats --output-format csv -d $P/bench -o bench -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $P/bench.log 2>&1
cp $(find $P/bench -name "*kernel_stats.csv" | head -1) $OUT/${R}_rocprof_kernel_stats.csv
echo "[2] dominant kernel alone"
rocprofv3 --kernel-trace --stats --output-format csv -d $P/konly -o konly -- python3 bench.py --kernel-only --micro-batch 64 > $P/konly.log 2>&1
cp $(find $P/konly -name "*kernel_stats.csv" | head -1) $OUT/${R}_rocprof_kernel_only_stats.csv
grep '^{' $P/konly.log | tail -1 > $OUT/${R}_kernel_only_hip_events.json
echo "[3] PMC passes (each in its own run)"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $P/pmc_fetch -o f -- python3 bench.py --kernel-only --micro-batch 64 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $P/pmc_write -o w -- python3 bench.py --kernel-only --micro-batch 64 > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $P/pmc_sq -o s -- python3 bench.py --kernel-only --micro-batch 64 > /dev/null 2>&1
python3 tools/probes/pmc_summary.py $OUT/${R}_dominant_kernel_pmc_raw.json \
    fetch=$(find $P/pmc_fetch -name "*counter_collection.csv" | head -1) \
    write=$(find $P/pmc_write -name "*counter_collection.csv" | head -1) \
    sq=$(find $P/pmc_sq -name "*counter_collection.csv" | head -1)
python3 tools/probes/pmc_derive.py $OUT/${R}_dominant_kernel_pmc_raw.json $OUT/${R}_dominant_kernel_pmc_all.json images=64 note="bench.py --kernel-only --micro-batch 64: conv3x3 192->192 @256x256 forward (conv3x3_halo) and its weight + bias gradient (wgrad_kx3), 40 dispatches each incl. warm-up"
echo "[4] attention: per-shape rates and MFMA busy"
python3 tools/attn_bench.py 64 > $OUT/${R}_attention_shapes.txt 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $P/pmc_attn -o a -- python3 tools/attn_bench.py 64 > /dev/null 2>&1
python3 tools/probes/pmc_summary.py $OUT/${R}_attention_pmc_raw.json sq=$(find $P/pmc_attn -name "*counter_collection.csv" | head -1)
rocprofv3 --kernel-trace --stats --output-format csv -d $P/attn -o attn -- python3 tools/attn_bench.py 64 > /dev/null 2>&1
cp $(find $P/attn -name "*kernel_stats.csv" | head -1) $OUT/${R}_attention_kernel_stats.csv
echo "[5] GEMM sweep"
python3 tools/gemm_sweep.py --mb 64 > $OUT/${R}_gemm_sweep_mb64.txt 2>&1
echo "[6] GEMM kernels PMC (res192@256: conv + wgrad)"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $P/pmc_gemm -o g -- python3 tools/gemm_sweep.py --mb 64 --only res192@256 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $P/pmc_gemm_f -o g -- python3 tools/gemm_sweep.py --mb 64 --only res192@256 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $P/pmc_gemm_w -o g -- python3 tools/gemm_sweep.py --mb 64 --only res192@256 > /dev/null 2>&1
python3 tools/probes/pmc_summary.py $OUT/${R}_gemm_kernels_pmc_raw.json \
    sq=$(find $P/pmc_gemm -name "*counter_collection.csv" | head -1) \
    fetch=$(find $P/pmc_gemm_f -name "*counter_collection.csv" | head -1) \
    write=$(find $P/pmc_gemm_w -name "*counter_collection.csv" | head -1)
python3 tools/probes/pmc_derive.py $OUT/${R}_gemm_kernels_pmc_raw.json $OUT/${R}_gemm_kernels_pmc.json images=64 note="tools/gemm_sweep.py --mb 64 --only res192@256: forward + data gradient (conv3x3_halo), weight gradient without bias (wgrad_kx3)"
python3 tools/probes/pmc_derive.py $OUT/${R}_attention_pmc_raw.json $OUT/${R}_attention_pmc.json images=64
echo "[7] eight-phase linear kernel alone: 1536 -> 6144 on 16384 rows (the kernel of bench.py's third roofline entry), 40 launches"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $P/pmc_p8_f -o f -- python3 tools/probes/p8_pmc_run.py 16384 1536 6144 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $P/pmc_p8_w -o w -- python3 tools/probes/p8_pmc_run.py 16384 1536 6144 > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $P/pmc_p8_s -o s -- python3 tools/probes/p8_pmc_run.py 16384 1536 6144 > /dev/null 2>&1
python3 tools/probes/pmc_summary.py $OUT/${R}_p8_gemm_pmc_raw.json \
    fetch=$(find $P/pmc_p8_f -name "*counter_collection.csv" | head -1) \
    write=$(find $P/pmc_p8_w -name "*counter_collection.csv" | head -1) \
    sq=$(find $P/pmc_p8_s -name "*counter_collection.csv" | head -1)
python3 tools/probes/pmc_derive.py $OUT/${R}_p8_gemm_pmc_raw.json $OUT/${R}_p8_gemm_pmc.json images=64 note="tools/probes/p8_pmc_run.py 16384 1536 6144: igemm_nt_kernel<256,256,...> on the eight-phase loop, linear 1536->6144 forward (bias), 16384 rows = 64 images x 256 tokens, super-tile block order 4 x 8, 40 launches"
echo "[8] block traces and launch census"
python3 tools/probes/block_trace.py 384 64 64 > $OUT/${R}_block_trace_stage2.txt 2>&1
python3 tools/probes/block_trace.py 1536 16 64 > $OUT/${R}_block_trace_stage4.txt 2>&1
python3 tools/probes/launch_census.py large > $OUT/${R}_launch_census.txt 2>&1
rm -rf $P
ls -la $OUT
