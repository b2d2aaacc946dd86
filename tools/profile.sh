#!/bin/bash
# Profile `python bench.py` (defaults: 512^3 map, batch 8) on the GPU box: kernel-trace stats, then PMC passes in their own runs
# (SQ + GRBM; FETCH_SIZE; WRITE_SIZE - MI355X_MICROARCH.md "rocprofv3 PMC slots").  usage: tools/profile.sh <tag> [what...]
#   what: stats sq fetch write  (default: all)
set -o pipefail
TAG=${1:-r02}; shift
WHAT=${*:-stats sq fetch write}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# every summary is stamped with the hash of the library sources it was taken on: bench.py quotes counter figures only from a profile
# whose stamp equals the running tree's (mica_amd/_cabi.py::source_hash)
HASH=$(python3 -c "import sys; sys.path.insert(0, '.'); from mica_amd._cabi import source_hash; print(source_hash())")
echo "library_source_hash: $HASH" > $OUT/source_hash.txt
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-alt-tiling --no-whole-map --af-coverage 0"
for w in $WHAT; do
  case $w in
    stats) timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o r -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-alt-tiling --no-whole-map --af-coverage 0 > $OUT/stats.log 2>&1 || exit 1
           python tools/prof_summary.py $(find $OUT/stats -name "*kernel_stats.csv" | head -1) 72 30 > $OUT/kernel_stats.txt; echo "library_source_hash: $HASH" >> $OUT/kernel_stats.txt; cat $OUT/kernel_stats.txt ;;
    sq)    timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU GRBM_GUI_ACTIVE -d $OUT/sq -o r -- python3 bench.py $ARGS > $OUT/sq.log 2>&1 || exit 1
           python tools/pmc_conv_summary.py $OUT/sq > $OUT/pmc_sq_summary.txt; echo "library_source_hash: $HASH" >> $OUT/pmc_sq_summary.txt; cat $OUT/pmc_sq_summary.txt ;;
    fetch) timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/fetch -o r -- python3 bench.py $ARGS > $OUT/fetch.log 2>&1 || exit 1 ;;
    write) timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/write -o r -- python3 bench.py $ARGS > $OUT/write.log 2>&1 || exit 1
           python tools/pmc_traffic.py $(find $OUT/fetch -name "*counter_collection.csv" | head -1) $(find $OUT/write -name "*counter_collection.csv" | head -1) $OUT/pmc_traffic.json "$TAG: python bench.py defaults (512^3 map, batch 8)" $HASH > /dev/null
           cat $OUT/pmc_traffic.json ;;
  esac
done
# keep the merged-back artefacts small: the per-dispatch CSVs are summarised above
find $OUT -name "*counter_collection.csv" -size +8M -delete
find $OUT -name "*kernel_trace.csv" -size +8M -delete
