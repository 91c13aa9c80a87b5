#!/bin/bash
# One measurement pass on the GPU box (run through gpurun), per BASELINE configuration that fits one GPU:
#   the two PMC passes FIRST (their summary goes to profiles/; the secondary configurations' bench lines quote it with
#   --quoted-traffic, the headline bench line makes its own live passes, as the driver's run does), then the bench line WITH its
#   roofline block; for the headline configuration also the rocprofv3 kernel-trace summary of the default bench command and a
#   kernel-trace + marker-trace run split by roctx phase (prepare / denoise_loop / vae_decode / d2h).
#   gpurun --timeout 1150 -- "bash tools/measure_round.sh r3 $(git rev-parse --short HEAD) [headline|b4|768|controlnet|all]"
# Outputs under gpurun_out/final/; copy what should be judged into profiles/.
set -o pipefail
TAG=${1:-r5}
COMMIT=${2:-unknown}
WHAT=${3:-all}
OUT=gpurun_out/final
mkdir -p $OUT
export TMPDIR=/tmp
T="timeout -k 10"

# pmc <suffix> <pmc_step args...>: FETCH_SIZE and WRITE_SIZE in separate passes (they do not fit one), summarised per entry point
pmc() {
    local sfx=$1; shift
    $T 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 tools/pmc_step.py "$@" > $OUT/pmc_fetch$sfx.log 2>&1 || return 3
    $T 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 tools/pmc_step.py "$@" > $OUT/pmc_write$sfx.log 2>&1 || return 4
    python tools/pmc_summarize.py $OUT/pmc_fetch $OUT/pmc_write $OUT/${TAG}_pmc_traffic$sfx.json "$COMMIT" > $OUT/pmc_sum$sfx.log 2>&1 || return 5
    cp $OUT/${TAG}_pmc_traffic$sfx.json profiles/${TAG}_pmc_traffic$sfx.json   # (this box's scratch copy of the repo: bench.py reads it from there)
    rm -rf $OUT/pmc_fetch $OUT/pmc_write
    echo "pmc$sfx done"
}

# mfma: matrix-pipe counter against the FLOP-derived fraction (tools/pmc_mfma.py): PMC pass, un-profiled trace, calibration loop
if [ "$WHAT" = mfma ]; then
    rm -rf $OUT/mfma_pmc $OUT/mfma_plain $OUT/mfma_calib
    $T 300 rocprofv3 --pmc MfmaUtil MfmaFlopsBF16 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/mfma_pmc -- python3 tools/pmc_step.py --calls-json $OUT/mfma_pmc_calls.json > $OUT/mfma_pmc.log 2>&1 \
        || $T 300 rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d $OUT/mfma_pmc -- python3 tools/pmc_step.py --calls-json $OUT/mfma_pmc_calls.json > $OUT/mfma_pmc.log 2>&1 || exit 3
    mv $OUT/mfma_pmc_calls.json $OUT/mfma_pmc/calls.json
    $T 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/mfma_plain -- python3 tools/pmc_step.py --calls-json $OUT/mfma_plain_calls.json > $OUT/mfma_plain.log 2>&1 || exit 4
    mv $OUT/mfma_plain_calls.json $OUT/mfma_plain/calls.json
    $T 200 rocprofv3 --pmc MfmaUtil MfmaFlopsBF16 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/mfma_calib -- python3 tools/mfma_rate.py > $OUT/mfma_calib.log 2>&1 || echo "calibration pass failed (see mfma_calib.log)"
    python tools/pmc_mfma.py $OUT/mfma_pmc $OUT/mfma_plain $OUT/${TAG}_pmc_mfma.json "$COMMIT" $( [ -d $OUT/mfma_calib ] && echo $OUT/mfma_calib ) > $OUT/mfma_sum.log 2>&1 || { echo "pmc_mfma failed"; tail -5 $OUT/mfma_sum.log; exit 5; }
    rm -rf $OUT/mfma_pmc/*/*.db $OUT/mfma_plain/*/*.db 2>/dev/null
    tail -40 $OUT/mfma_sum.log
    echo "mfma done"
    exit 0
fi
if [ "$WHAT" = all ] || [ "$WHAT" = headline ]; then
    pmc "" || exit $?
    MSD_BENCH_KEEP_TRACE=$OUT/${TAG}_graph $T 500 python bench.py > $OUT/${TAG}_bench_n1.json 2> $OUT/bench_n1.err || exit 1
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
    $T 200 python bench.py --batch-per-gpu 2 --steps 3 --no-cpu-baseline --no-roofline > $OUT/${TAG}_bench_b2.json 2>> $OUT/bench_cfg.err
fi
if [ "$WHAT" = all ] || [ "$WHAT" = b4 ]; then      # C3's per-GPU shape: batch 4 (fused cond+uncond batch 8)
    pmc _b4 --batch 4 || exit $?
    MSD_BENCH_KEEP_TRACE=$OUT/${TAG}_graph $T 400 python bench.py --batch-per-gpu 4 --steps 3 --no-cpu-baseline --quoted-traffic > $OUT/${TAG}_bench_b4.json 2>> $OUT/bench_cfg.err
    echo "b4 done"
    # the serving choice for the GroupNorm form (MSD_GN_ROWS=4096: DESIGN.md 4.3), its own file
    MSD_GN_ROWS=4096 $T 300 python bench.py --batch-per-gpu 4 --steps 3 --no-cpu-baseline --no-roofline > $OUT/${TAG}_bench_b4_gn_rows4096.json 2>> $OUT/bench_cfg.err
    echo "b4 gn_rows 4096 done"
fi
if [ "$WHAT" = all ] || [ "$WHAT" = 768 ]; then     # C4: 768x768, 50 steps
    pmc _768 --size 768 || exit $?
    MSD_BENCH_KEEP_TRACE=$OUT/${TAG}_graph $T 600 python bench.py --size 768 --denoise-steps 50 --steps 2 --no-cpu-baseline --quoted-traffic > $OUT/${TAG}_bench_768.json 2>> $OUT/bench_cfg.err
    echo "768 done"
fi
if [ "$WHAT" = all ] || [ "$WHAT" = controlnet ]; then   # C5's per-GPU shape: ControlNet, batch 1
    pmc _controlnet --controlnet || exit $?
    MSD_BENCH_KEEP_TRACE=$OUT/${TAG}_graph $T 400 python bench.py --controlnet --steps 3 --no-cpu-baseline --quoted-traffic > $OUT/${TAG}_bench_controlnet.json 2>> $OUT/bench_cfg.err
    echo "controlnet done"
fi
echo "configs done"
