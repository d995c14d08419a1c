#!/bin/bash
# Counters of the fused vocoder step kernels (run through gpurun from the repo root): two separate --pmc passes of tools/voc_breakdown.py
set -o pipefail
out=gpurun_out/pmc_voc_pair
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv \
    -d $out/a -o m -- python3 tools/voc_breakdown.py > /dev/null 2> $out/a.err || exit 1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_LDS --output-format csv \
    -d $out/b -o m -- python3 tools/voc_breakdown.py > /dev/null 2> $out/b.err || exit 1
cp $(find $out/a -name "*counter_collection.csv" | head -1) $out/a.csv
cp $(find $out/b -name "*counter_collection.csv" | head -1) $out/b.csv
rm -rf $out/a $out/b
ls -la $out
