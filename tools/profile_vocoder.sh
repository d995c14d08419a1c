#!/bin/bash
# Vocoder per-kernel evidence (run through gpurun from the repo root): kernel stats of one 16 x 512-frame decode under rocprofv3 and the
# matrix-pipe counters of the same run in a separate --pmc pass -> gpurun_out/prof_<tag>_voc/
set -o pipefail
tag=${1:-rXX}
out=gpurun_out/prof_${tag}_voc
mkdir -p $out
export TMPDIR=/tmp
echo "[1/2] kernel trace" && rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks -o ks -- python3 tools/voc_breakdown.py > $out/voc_breakdown.txt 2> $out/ks.err || exit 1
cp $(find $out/ks -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
echo "[2/2] MFMA counters" && rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv \
    -d $out/mfma -o m -- python3 tools/voc_breakdown.py > /dev/null 2> $out/mfma.err || exit 1
python3 tools/summarize_mfma.py $(find $out/mfma -name "*counter_collection.csv" | head -1) > $out/mfma_util.json || exit 1
rm -rf $out/ks $out/mfma
ls -la $out
