#!/bin/bash
# One measurement pass on the GPU box (run through gpurun): the two PMC passes FIRST (so that the bench line quotes the
# traffic of this very build), bench lines for every BASELINE config that fits one GPU, the rocprofv3 kernel-trace summary
# of the default bench command, and a kernel-trace + marker-trace run split by roctx phase (prepare / denoise_loop /
# vae_decode / d2h).
#   gpurun --timeout 1150 -- "bash tools/measure_round.sh r2 $(git rev-parse --short HEAD)"
# Outputs under gpurun_out/final/; copy what should be judged into profiles/.
set -o pipefail
TAG=${1:-r2}
COMMIT=${2:-unknown}
OUT=gpurun_out/final
mkdir -p $OUT
export TMPDIR=/tmp
T="timeout -k 10"
$T 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 tools/pmc_step.py > $OUT/pmc_fetch.log 2>&1 || exit 3
$T 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 tools/pmc_step.py > $OUT/pmc_write.log 2>&1 || exit 4
python tools/pmc_summarize.py $OUT/pmc_fetch $OUT/pmc_write $OUT/${TAG}_pmc_traffic.json "$COMMIT" > $OUT/pmc_sum.log 2>&1 || exit 5
cp $OUT/${TAG}_pmc_traffic.json profiles/${TAG}_pmc_traffic.json   # (this box's scratch copy of the repo: bench.py reads it from there)
rm -rf $OUT/pmc_fetch $OUT/pmc_write
echo "pmc done"
$T 400 python bench.py > $OUT/${TAG}_bench_n1.json 2> $OUT/bench_n1.err || exit 1
echo "bench n1 done"; head -c 330 $OUT/${TAG}_bench_n1.json; echo
$T 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/prof_bench.log 2>&1 || exit 2
ST=$(ls $OUT/prof/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$ST" ] && cp $ST $OUT/${TAG}_bench_kernel_stats.csv
rm -rf $OUT/prof
echo "rocprof stats done"
$T 300 rocprofv3 --kernel-trace --marker-trace --output-format csv -d $OUT/prof_phase -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --sync-phases > $OUT/prof_phase.log 2>&1 || exit 6
python tools/trace_summary.py $OUT/prof_phase $OUT/${TAG}_phase_summary.md "round ${TAG#r} (commit $COMMIT), python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --sync-phases" > /dev/null 2> $OUT/phase_sum.err || echo "phase summary failed (see phase_sum.err)"
rm -rf $OUT/prof_phase
echo "phase trace done"
$T 200 python bench.py --batch-per-gpu 2 --steps 3 --no-cpu-baseline --no-roofline > $OUT/bench_b2.json 2>> $OUT/bench_cfg.err
$T 200 python bench.py --batch-per-gpu 4 --steps 3 --no-cpu-baseline --no-roofline > $OUT/bench_b4.json 2>> $OUT/bench_cfg.err
$T 300 python bench.py --size 768 --denoise-steps 50 --steps 2 --no-cpu-baseline --no-roofline > $OUT/bench_768.json 2>> $OUT/bench_cfg.err
$T 300 python bench.py --controlnet --steps 3 --no-cpu-baseline --no-roofline > $OUT/bench_controlnet.json 2>> $OUT/bench_cfg.err
echo "configs done"
