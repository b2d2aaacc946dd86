#!/bin/bash
# round 5: the in-process hand-off and the multi-GPU readiness items on one box.  usage: tools/gpu_r5b.sh <tag>
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
T=${1:-r5b}
mkdir -p gpurun_out/$T
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_refgold.py tests/test_gpu_volume.py -x -q -m gpu > gpurun_out/$T/t_sel.log 2>&1; rc=$?
echo "selected gpu tests rc=$rc"; tail -5 gpurun_out/$T/t_sel.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python tools/file_predictor_bench.py 256 > gpurun_out/$T/file_predictor.txt 2>&1; rc=$?; echo "file predictor rc=$rc"; tail -8 gpurun_out/$T/file_predictor.txt
[ $rc -eq 0 ] || exit $rc
for n in 256 512; do
  timeout -k 10 300 python bench.py --strong --map $n --grid 48 --pad 8 > gpurun_out/$T/strong_$n.json 2> gpurun_out/$T/strong_$n.err; echo "strong $n rc=$?"; head -c 250 gpurun_out/$T/strong_$n.json; echo
done
timeout -k 10 300 python bench.py --strong --map 256 --grid 48 --pad 8 --backend nccl --force-exchange > gpurun_out/$T/strong_256_rccl.json 2> gpurun_out/$T/strong_256_rccl.err; echo "strong rccl rc=$?"; head -c 250 gpurun_out/$T/strong_256_rccl.json; echo
timeout -k 10 300 python bench.py --strong --map 256 --grid 48 --pad 8 --backend nccl --force-exchange --gather-to-root > gpurun_out/$T/strong_256_rccl_root.json 2> gpurun_out/$T/strong_256_rccl_root.err; echo "strong rccl root rc=$?"; head -c 250 gpurun_out/$T/strong_256_rccl_root.json; echo
timeout -k 10 300 python bench.py --backend nccl --force-exchange --no-cpu-baseline --no-alt-tiling --no-whole-map > gpurun_out/$T/weak_rccl.json 2> gpurun_out/$T/weak_rccl.err; echo "weak rccl rc=$?"; head -c 200 gpurun_out/$T/weak_rccl.json; echo
timeout -k 10 300 python bench.py --backend nccl --force-exchange --gather-to-root --no-cpu-baseline --no-alt-tiling --no-whole-map > gpurun_out/$T/weak_rccl_root.json 2> gpurun_out/$T/weak_rccl_root.err; echo "weak rccl root rc=$?"; head -c 200 gpurun_out/$T/weak_rccl_root.json; echo
