set -o pipefail
out=gpurun_out/pmc_unet
mkdir -p $out
export TMPDIR=/tmp
pmc="--nfe 5 --steps 1 --warmup 0 --no-extras --no-cpu-baseline --no-profile"
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $out/a -o m -- python3 bench.py $pmc > /dev/null 2> $out/a.err || { tail -5 $out/a.err; exit 1; }
cp $(find $out/a -name "*counter_collection.csv" | head -1) $out/a.csv
rm -rf $out/a
ls -la $out
