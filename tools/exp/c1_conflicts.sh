#!/bin/bash
# Round 6, review item 6: what the LDS bank conflicts of conv1x1_kernel's epilogue cost.  The shipped kernel against a build whose epilogue
# reads the staging tile conflict-free (make -C mica_amd/csrc exp_c1lin: neighbouring lanes read neighbouring rows - garbage results, the
# shipped instruction mix), both under the SQ counters over `python bench.py` (512^3 map, batch 8): bank-conflict share and time per launch.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/c1cf
for v in normal c1lin normal c1lin; do
  if [ $v = normal ]; then L=$PWD/mica_amd/lib/libmica_hip.so; else L=$PWD/tools/exp/libmica_$v.so; fi
  rm -rf gpurun_out/c1cf/$v
  MICA_HIP_LIB=$L timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE -d gpurun_out/c1cf/$v -o r -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt-tiling --no-whole-map --af-coverage 0 > gpurun_out/c1cf/$v.log 2>&1 || { tail -5 gpurun_out/c1cf/$v.log; exit 1; }
  echo "== $v"; python tools/pmc_conv_summary.py gpurun_out/c1cf/$v | grep -E "^kernel|conv1x1"
done
find gpurun_out/c1cf -name "*.csv" -size +2M -delete
