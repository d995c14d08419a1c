#!/bin/bash
# All profiling passes of one round on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh r03
# writes gpurun_out/prof_<tag>/: kernel stats of the bench under rocprofv3 (--kernel-trace --stats), the bench line printed by that run,
# and three separate --pmc passes (FETCH_SIZE, WRITE_SIZE, MFMA / wait counters; never combined with a trace domain) with their
# summaries.  Afterwards, in the container: copy the summaries to profiles/<tag>_* and stamp them (tools/stamp_profile.py).
set -o pipefail
tag=${1:-rXX}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
common="--steps 1 --warmup 1 --no-extras --no-cpu-baseline"
echo "[1/4] kernel trace" && rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks -o ks -- python3 bench.py $common > $out/bench_under_rocprof.json 2> $out/ks.err || exit 1
cp $(find $out/ks -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
pmc="--nfe 5 --steps 1 --warmup 0 --no-extras --no-cpu-baseline --no-profile"
echo "[2/4] FETCH_SIZE" && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o f -- python3 bench.py $pmc > /dev/null 2> $out/fetch.err || exit 1
echo "[3/4] WRITE_SIZE" && rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o w -- python3 bench.py $pmc > /dev/null 2> $out/write.err || exit 1
python3 tools/summarize_pmc.py $(find $out/fetch -name "*counter_collection.csv" | head -1) $(find $out/write -name "*counter_collection.csv" | head -1) > $out/hbm_traffic.json || exit 1
echo "[4/4] MFMA counters" && rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv \
    -d $out/mfma -o m -- python3 bench.py $pmc > /dev/null 2> $out/mfma.err || exit 1
python3 tools/summarize_mfma.py $(find $out/mfma -name "*counter_collection.csv" | head -1) > $out/mfma_util.json || exit 1
if [ -n "$2" ]; then      # optional: the same kernel trace for one sampler run in a split GEMM mode (tools/shape_breakdown.py warms up, then 10 NFE)
  echo "[+] kernel trace, mode $2" && rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks_$2 -o ks -- python3 tools/shape_breakdown.py 16 --mode $2 > $out/shape_$2.txt 2> $out/ks_$2.err || exit 1
  cp $(find $out/ks_$2 -name "*kernel_stats.csv" | head -1) $out/kernel_stats_$2.csv
  rm -rf $out/ks_$2
fi
rm -rf $out/ks $out/fetch $out/write $out/mfma      # raw traces are large; the summaries are what is kept
ls -la $out
