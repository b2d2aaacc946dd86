#!/bin/bash
# Every record of a round that profiles/README.md lists, one named step per record.  On the GPU box (one gpurun call = a few steps that
# fit its time limit):      gpurun --timeout 1200 -- 'tools/records.sh r06 tests bench'
# afterwards, in the build container:                tools/records.sh r06 copy      (gpurun_out/r06/* -> profiles/r06_*)
#
#   tests      all GPU tests                                      -> t_all.log
#   profile    rocprofv3 passes of bench.py (tools/profile.sh)    -> kernel_stats.txt/.csv, pmc_sq_summary.txt, pmc_traffic.json (stamped
#                                                                    with the library source hash bench.py checks)
#   bench      the default bench line                             -> bench_default.json
#   parityA, parityB   tools/parity_full_tile.py (every logit of the sixteen 64^3 tiles vs the oracle, f32 and f64; ~75 s of CPU per tile: two
#              calls of eight tiles)                               -> parity_full_tile_a.txt, _b.txt (copy joins them: parity_full_tile.txt)
#   margins    tools/parity_margins.py 0,1,3 (every fixture, lattice samples)                                -> parity_margins.txt
#   files      getData + nnPred through the mirrors, 256^3        -> file_predictor.txt          files512: the same at 512^3
#   ranks      rank start-up of the multi-GPU product entry       -> rank_startup.txt
#   strong     bench.py --strong: 1 rank 256^3 / 512^3, RCCL all-gather and gather-to-root forced in a group of one
#   rehearse   bench.py --gpus 2 --backend gloo --single-device (weak, root, strong): two ranks SHARING one GPU - functional, not scaling
#   e2e        tools/e2e_bench.py 384 4 (configs[4], host -> host) -> e2e_streamed.txt
#   soak       tools/soak.py 1500 8                               -> soak.txt
#   fuzz       tools/fuzz_ops.py (random shapes through the single-op entries, 300 s + 200 s big) -> fuzz_ops.txt
#   fuzzvol    tests/test_gpu_fuzz_volume.py for 90 s per family  -> fuzz_volume.txt
#   mixed      the A/B of the one-pass forward on a map with atoms in 30 % of the box (MICA_TRUNK_PER_RUN=1 | 0) -> mixed_af_ab.txt
set -o pipefail
R=${1:?usage: tools/records.sh <rNN> step...}; shift
if [ "$1" = copy ]; then
  cd "$(dirname "$0")/.." || exit 1
  T=gpurun_out/$R
  cpy() { [ -f "$T/$1" ] && grep -v amdgpu.ids "$T/$1" > "profiles/${R}_$2" && echo "profiles/${R}_$2"; }
  cpy bench_default.json bench_default.json; cpy kernel_stats.txt kernel_stats.txt; [ -f $T/stats/r_kernel_stats.csv ] && cp $T/stats/r_kernel_stats.csv profiles/${R}_kernel_stats.csv
  cpy pmc_sq_summary.txt pmc_sq_summary.txt; [ -f $T/pmc_traffic.json ] && cp $T/pmc_traffic.json profiles/${R}_pmc_traffic.json
  [ -f $T/parity_full_tile_a.txt ] && [ -f $T/parity_full_tile_b.txt ] && cat $T/parity_full_tile_a.txt $T/parity_full_tile_b.txt | grep -v amdgpu.ids > profiles/${R}_parity_full_tile.txt && echo profiles/${R}_parity_full_tile.txt
  for f in fuzz_ops.txt fuzz_volume.txt parity_margins.txt file_predictor.txt file_predictor_512.txt rank_startup.txt e2e_streamed.txt soak.txt mixed_af_ab.txt; do cpy $f $f; done
  for f in $T/bench_strong_*.json $T/bench_2rank_*.json; do [ -f "$f" ] && cp $f profiles/${R}_$(basename $f) && echo profiles/${R}_$(basename $f); done
  tail -1 $T/t_all.log 2>/dev/null; cat $T/source_hash.txt 2>/dev/null
  exit 0
fi
cd "$GRAFT_REPO_ROOT" || exit 1
T=gpurun_out/$R
mkdir -p $T
Q="--no-cpu-baseline --no-alt-tiling --no-whole-map --af-coverage 0"
rc=0
run() { local name=$1 out=$2 lim=$3; shift 3; timeout -k 10 $lim "$@" > $T/$out 2> $T/$name.err; local r=$?; echo "$name rc=$r"; [ $r -eq 0 ] || { tail -5 $T/$name.err; rc=$r; }; return $r; }
for step in "$@"; do
  case $step in
    tests)    timeout -k 10 1000 python -m pytest tests -x -q -m gpu -s > $T/t_all.log 2>&1; r=$?; echo "gpu tests rc=$r"; tail -3 $T/t_all.log; [ $r -eq 0 ] || exit $r ;;
    profile)  bash tools/profile.sh $R stats sq fetch write > $T/profile.log 2>&1; echo "profile rc=$?"; head -14 $T/pmc_sq_summary.txt
              # a `bench` step later in this call quotes the counters of THIS tree (bench.py reads the newest profiles/rNN_pmc_*)
              cp $T/pmc_traffic.json profiles/${R}_pmc_traffic.json 2>/dev/null; cp $T/pmc_sq_summary.txt profiles/${R}_pmc_sq_summary.txt 2>/dev/null ;;
    bench)    run bench bench_default.json 400 python bench.py && head -c 400 $T/bench_default.json && echo ;;
    parity|parityA)  run parityA parity_full_tile_a.txt 1100 python tools/parity_full_tile.py --cases w2022g6,w7g3,w99g10,zeroaf_w2022g6,blob,heavy,w99g10_s101,w99g10_s102 && tail -12 $T/parity_full_tile_a.txt ;;
    parityB)  run parityB parity_full_tile_b.txt 1100 python tools/parity_full_tile.py --cases w99g10_s103,w99g10_s104,w7g3_s201,w7g3_s202,w2022g6_s201,w2022g6_s202,w31g6_s301,w57g10_s302 && tail -12 $T/parity_full_tile_b.txt ;;
    margins)  run margins parity_margins.txt 900 python tools/parity_margins.py 0,1,3 && tail -4 $T/parity_margins.txt ;;
    files)    run files file_predictor.txt 400 python tools/file_predictor_bench.py 256 && cat $T/file_predictor.txt ;;
    files512) run files512 file_predictor_512.txt 900 python tools/file_predictor_bench.py 512 && cat $T/file_predictor_512.txt ;;
    ranks)    run ranks rank_startup.txt 300 python tools/rank_startup.py 256 2 gloo && cat $T/rank_startup.txt ;;
    strong)   for n in 256 512; do run strong$n bench_strong_1rank_$n.json 300 python bench.py --strong --map $n --grid 48 --pad 8; done
              run strong_rccl bench_strong_rccl_allgather_1rank_256.json 300 python bench.py --strong --map 256 --grid 48 --pad 8 --backend nccl --force-exchange
              run strong_root bench_strong_rccl_gather_1rank_256.json 300 python bench.py --strong --map 256 --grid 48 --pad 8 --backend nccl --force-exchange --gather-to-root
              for f in $T/bench_strong_*.json; do echo "$(basename $f): $(head -c 120 $f)"; done ;;
    rehearse) run weak2 bench_2rank_gloo_one_gpu_256.json 400 python bench.py --gpus 2 --backend gloo --single-device --steps 6 --warmup 2 --map 256
              run weak2root bench_2rank_gloo_root_one_gpu_256.json 400 python bench.py --gpus 2 --backend gloo --single-device --steps 6 --warmup 2 --map 256 --gather-to-root
              run strong2 bench_strong_2rank_gloo_one_gpu_256.json 400 python bench.py --gpus 2 --backend gloo --single-device --strong --map 256 --grid 48 --pad 8 ;;
    e2e)      run e2e e2e_streamed.txt 300 python tools/e2e_bench.py 384 4 && tail -3 $T/e2e_streamed.txt ;;
    soak)     run soak soak.txt 600 python tools/soak.py 1500 8 && tail -2 $T/soak.txt ;;
    fuzz)     timeout -k 10 420 python tools/fuzz_ops.py 300 6 > $T/fuzz_a.log 2> $T/fuzz.err; echo "fuzz rc=$?"
              timeout -k 10 300 python tools/fuzz_ops.py 200 7 big > $T/fuzz_b.log 2>> $T/fuzz.err; echo "fuzz big rc=$?"
              { echo "# tools/fuzz_ops.py 300 6:"; tail -12 $T/fuzz_a.log; echo "# tools/fuzz_ops.py 200 7 big:"; tail -12 $T/fuzz_b.log; } > $T/fuzz_ops.txt; tail -3 $T/fuzz_a.log; tail -3 $T/fuzz_b.log ;;
    fuzzvol)  MICA_FUZZ_SECONDS=90 timeout -k 10 600 python -m pytest tests/test_gpu_fuzz_volume.py -q -m gpu -s > $T/fuzz_volume.txt 2>&1; echo "fuzz volume rc=$?"; tail -3 $T/fuzz_volume.txt ;;
    mixed)    : > $T/mixed_af_ab.txt
              for v in 1 0 1 0; do
                MICA_TRUNK_PER_RUN=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-alt-tiling --no-whole-map > $T/mixed_$v.json 2> $T/mixed.err || { rc=1; tail -3 $T/mixed.err; }
                python - <<PY | tee -a $T/mixed_af_ab.txt
import json
d = json.load(open("$T/mixed_$v.json")); m = d["mixed_af"]
print("MICA_TRUNK_PER_RUN=$v (%s): all-AF map %.2f sub-grids/s; atoms in %.0f %% of the box: %.2f sub-grids/s = %.3f of it (%.2f runs of equal gate per batch, %.0f %% of the tiles with atoms)"
      % ("the whole network once per run of equal gate, rounds 1-5" if $v else "MultiScaleInput per run, the rest once per batch", d["value"], 100 * m["af_coverage"], m["value"], m["ratio_to_value"], m["input_runs_per_batch"], 100 * m["tiles_with_atoms"]))
PY
              done ;;
    *) echo "unknown step $step"; exit 2 ;;
  esac
done
find $T -name "*counter_collection.csv" -size +8M -delete 2>/dev/null
exit $rc
