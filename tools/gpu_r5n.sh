#!/bin/bash
# round 5: rehearse the N > 1 code paths of bench.py on ONE GPU (two gloo ranks sharing cuda:0, records staged through the host: a
# functional rehearsal, not a scaling number): weak mode with the all-gather and with gather-to-root, strong mode.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
T=${1:-r5n}
mkdir -p gpurun_out/$T
timeout -k 10 400 python bench.py --gpus 2 --backend gloo --single-device --steps 6 --warmup 2 --map 256 > gpurun_out/$T/weak2.json 2> gpurun_out/$T/weak2.err; echo "weak 2-rank rc=$?"; head -c 400 gpurun_out/$T/weak2.json; echo
timeout -k 10 400 python bench.py --gpus 2 --backend gloo --single-device --steps 6 --warmup 2 --map 256 --gather-to-root > gpurun_out/$T/weak2_root.json 2> gpurun_out/$T/weak2_root.err; echo "weak 2-rank root rc=$?"; head -c 400 gpurun_out/$T/weak2_root.json; echo
timeout -k 10 400 python bench.py --gpus 2 --backend gloo --single-device --strong --map 256 --grid 48 --pad 8 > gpurun_out/$T/strong2.json 2> gpurun_out/$T/strong2.err; echo "strong 2-rank rc=$?"; head -c 400 gpurun_out/$T/strong2.json; echo
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --backend gloo --single-device --steps 4 --warmup 1 --map 256 > gpurun_out/$T/weak2_torchrun.json 2> gpurun_out/$T/weak2_torchrun.err; echo "torchrun 2-rank rc=$?"; head -c 300 gpurun_out/$T/weak2_torchrun.json; echo
