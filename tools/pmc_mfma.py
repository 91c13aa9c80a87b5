"""Reconcile the matrix-pipe counter with the FLOP-derived fraction, per kernel of the conv / dense family and attention.

Inputs (all taken on ONE box, one after the other; tools/measure_round.sh `mfma` runs them):

    rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d PMC   -- python3 tools/pmc_step.py --calls-json PMC/calls.json
    rocprofv3               --kernel-trace --output-format csv -d PLAIN -- python3 tools/pmc_step.py --calls-json PLAIN/calls.json
    rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d CALIB -- python3 tools/mfma_rate.py      (optional)
    python tools/pmc_mfma.py PMC PLAIN profiles/r4_pmc_mfma.json [commit] [CALIB]

Per kernel it writes: dispatches; mean duration in the PMC pass and in the un-profiled trace (counter collection serialises
dispatches and runs them cold, so the two differ — and MfmaUtil's denominator is the PMC pass's duration); MfmaUtil as the
counter reports it (percent of SIMD-cycles of a dispatch with the matrix pipe busy, gfx94x formula: ROCm 7.2 has no gfx950
section), TIME-weighted over the dispatches instead of averaged per dispatch; and, for the conv / dense kernels, the
FLOP-derived busy fraction = algorithmic FLOP of the call / (duration x 1,048,576 FLOP per cycle x clock) — the chip issues
256 CUs x 4 SIMDs x 1024 FLOP per cycle with `v_mfma_f32_16x16x32_bf16` back to back — at 2.4 GHz (the peak bench.py prices
against) and at the clock actually held (sysfs sclk sampled while the steps ran).  The algorithmic FLOP come from the call log
tools/pmc_step.py writes (launch order = dispatch order on the one stream; a split-K call is its main kernel followed by its
`splitk_finalize` launch).  CALIB: the counter on a bare `16x16x32` / `32x32x16` loop on every SIMD, which must read ~100 %."""
import collections
import csv
import glob
import json
import os
import sys

FLOP_PER_CYCLE = 256 * 4 * 1024.0   # dense bf16 MFMA, whole chip
XCDS = 8                            # rocprofv3 reports the raw GRBM_GUI_ACTIVE summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back);
                                    # MfmaUtil's own formula takes the per-XCD maximum
CONV = ("conv_gemm_dma", "conv3x3_halo", "dense_rowpanel", "conv_wreg", "conv_big")   # (conv_big_kernel and conv_bighalo_kernel)


def find(d, pat):
    f = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return f[0] if f else None


def trace(d):
    """[(kernel name, start ns, end ns)] in dispatch order."""
    kt = find(d, "*kernel_trace.csv")
    if not kt:
        raise SystemExit(f"no kernel_trace.csv under {d}")
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Dispatch_Id", 0) or 0)) for r in csv.DictReader(open(kt))]
    rows.sort()
    return rows


def counters(d, name="MfmaUtil"):
    """dispatch id -> (kernel name, value of counter `name`)."""
    f = find(d, "*counter_collection.csv")
    if not f:
        raise SystemExit(f"no counter_collection.csv under {d}")
    out = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name:
            out[int(r["Dispatch_Id"])] = (r["Kernel_Name"], float(r["Counter_Value"]))
    return out


def is_conv(k):
    return any(c in k for c in CONV)


def join_calls(rows, calls):
    """Dispatch index -> algorithmic FLOP for the main kernel of every conv / dense call (launch order = dispatch order)."""
    mains = [i for i, r in enumerate(rows) if is_conv(r[2])]
    if len(mains) != len(calls):
        print(f"warning: {len(mains)} conv / dense main kernels in the trace, {len(calls)} calls logged: joining the last "
              f"{min(len(mains), len(calls))}", file=sys.stderr)
    n = min(len(mains), len(calls))
    return {mains[len(mains) - n + j]: calls[len(calls) - n + j] for j in range(n)}


def main():
    pmc_dir, plain_dir, out = sys.argv[1:4]
    commit = sys.argv[4] if len(sys.argv) > 4 else None
    calib_dir = sys.argv[5] if len(sys.argv) > 5 else None
    pmc_rows, plain_rows = trace(pmc_dir), trace(plain_dir)
    util = counters(pmc_dir)
    # optional, when the pass also collected them: executed MFMA FLOP (SQ_INSTS_VALU_MFMA_MOPS_BF16 x 512) and GRBM_GUI_ACTIVE (cycles of
    # the busiest XCD): executed FLOP / (1,048,576 x GUI cycles) has MfmaUtil's own denominator, so the two must agree if the busy
    # counter is right, whatever the clock; GUI cycles / duration is the clock the dispatch ran at (reads high on dispatches < 0.3 ms)
    xflop = {k: v[1] for k, v in counters(pmc_dir, "MfmaFlopsBF16").items()}
    gui = {k: v[1] / XCDS for k, v in counters(pmc_dir, "GRBM_GUI_ACTIVE").items()}
    pmc_calls = json.load(open(os.path.join(pmc_dir, "calls.json")))
    plain_calls = json.load(open(os.path.join(plain_dir, "calls.json")))
    flop_pmc = join_calls(pmc_rows, pmc_calls["conv_calls"])
    flop_plain = join_calls(plain_rows, plain_calls["conv_calls"])
    clock_ghz = (plain_calls.get("sclk_mhz_median") or 2400.0) / 1e3

    per = collections.defaultdict(lambda: {"n": 0, "pmc_us": 0.0, "util_x_us": 0.0, "flop": 0.0, "flop_us": 0.0, "xflop": 0.0, "gui": 0.0, "util_x_gui": 0.0})
    for i, (s, e, k, did) in enumerate(pmc_rows):
        if did not in util:
            continue
        us = (e - s) / 1e3
        p = per[k]
        p["n"] += 1
        p["pmc_us"] += us
        p["util_x_us"] += util[did][1] * us
        p["xflop"] += xflop.get(did, 0.0)
        p["gui"] += gui.get(did, 0.0)
        p["util_x_gui"] += util[did][1] * gui.get(did, 0.0)
        if i in flop_pmc:
            p["flop"] += flop_pmc[i]["flop"]
    plain = collections.defaultdict(lambda: {"n": 0, "us": 0.0, "flop": 0.0})
    for i, (s, e, k, _did) in enumerate(plain_rows):
        q = plain[k]
        q["n"] += 1
        q["us"] += (e - s) / 1e3
        if i in flop_plain:
            q["flop"] += flop_plain[i]["flop"]

    def frac(flop, us, ghz):
        return 100.0 * flop / (us * 1e-6 * FLOP_PER_CYCLE * ghz * 1e9) if us > 0 else 0.0

    kernels = {}
    for k, p in per.items():
        q = plain.get(k, {"n": 0, "us": 0.0, "flop": 0.0})
        row = {"dispatches": p["n"], "pmc_us_mean": round(p["pmc_us"] / p["n"], 2),
               "plain_us_mean": round(q["us"] / q["n"], 2) if q["n"] else None,
               "mfma_util_pct_time_weighted": round(p["util_x_us"] / p["pmc_us"], 2)}
        if p["flop"] > 0:
            row["flop_busy_pct_pmc_pass_at_2.4GHz"] = round(frac(p["flop"], p["pmc_us"], 2.4), 2)
            if q["flop"] > 0:
                row["flop_busy_pct_plain_at_2.4GHz"] = round(frac(q["flop"], q["us"], 2.4), 2)
                row[f"flop_busy_pct_plain_at_sclk_{clock_ghz:.2f}GHz"] = round(frac(q["flop"], q["us"], clock_ghz), 2)
        if p["gui"] > 0:
            row["executed_flop_busy_pct_over_gui_cycles"] = round(100.0 * p["xflop"] / (FLOP_PER_CYCLE * p["gui"]), 2)
            row["mfma_util_pct_gui_weighted"] = round(p["util_x_gui"] / p["gui"], 2)
            row["clock_ghz_gui_cycles_over_pmc_duration"] = round(p["gui"] / (p["pmc_us"] * 1e3), 3)
            if p["flop"] > 0:
                row["executed_over_algorithmic_flop"] = round(p["xflop"] / p["flop"], 3)
        if row["mfma_util_pct_time_weighted"] > 0.3 or p["flop"] > 0:
            kernels[k] = row

    def family(pred, with_finalize=False):
        sel = [k for k in per if pred(k)]
        pmc_us = sum(per[k]["pmc_us"] for k in sel)
        ux = sum(per[k]["util_x_us"] for k in sel)
        fl = sum(per[k]["flop"] for k in sel)
        pl_us = sum(plain[k]["us"] for k in sel if k in plain)
        pl_fl = sum(plain[k]["flop"] for k in sel if k in plain)
        fin_pmc = sum(per[k]["pmc_us"] for k in per if "splitk_finalize" in k) if with_finalize else 0.0
        fin_pl = sum(plain[k]["us"] for k in plain if "splitk_finalize" in k) if with_finalize else 0.0
        r = {"dispatches": sum(per[k]["n"] for k in sel), "pmc_ms": round((pmc_us + fin_pmc) / 1e3, 3), "plain_ms": round((pl_us + fin_pl) / 1e3, 3),
             "pmc_over_plain_duration": round((pmc_us + fin_pmc) / (pl_us + fin_pl), 3) if pl_us else None,
             "mfma_util_pct_time_weighted": round(ux / (pmc_us + fin_pmc), 2) if pmc_us else None}
        g_ = sum(per[k]["gui"] for k in sel)
        if g_ > 0:
            xf = sum(per[k]["xflop"] for k in sel)
            r["executed_flop_busy_pct_over_gui_cycles"] = round(100.0 * xf / (FLOP_PER_CYCLE * g_), 2)
            r["mfma_util_pct_gui_weighted"] = round(sum(per[k]["util_x_gui"] for k in sel) / g_, 2)
            r["clock_ghz_gui_cycles_over_pmc_duration"] = round(g_ / (pmc_us * 1e3), 3)
            if fl:
                r["executed_over_algorithmic_flop"] = round(xf / fl, 3)
            # RECONCILED: GRBM_GUI_ACTIVE keeps counting for a few microseconds around each dispatch (on dispatches of 60 us and more
            # GUI cycles / duration equals the sampled shader clock; on 20-us dispatches it reads 3+ "GHz"), so MfmaUtil's denominator is
            # too large on short dispatches.  Busy cycles over the cycles of the kernel's own start -> end interval at the sampled clock:
            pmc_clock = (pmc_calls.get("sclk_mhz_max") or 2400.0) / 1e3
            run_clock = clock_ghz   # (median sclk of the un-profiled pass: the PMC pass idles between its serialised dispatches)
            r["mfma_busy_pct_over_kernel_interval_at_sampled_clock"] = round(r["mfma_util_pct_gui_weighted"] * g_ / (pmc_us * 1e3 * run_clock), 2)
            r["sampled_clock_ghz"] = round(run_clock, 3)
        if fl:
            r["flop_busy_pct_pmc_pass_at_2.4GHz"] = round(frac(fl, pmc_us + fin_pmc, 2.4), 2)
            r["flop_busy_pct_plain_at_2.4GHz"] = round(frac(pl_fl, pl_us + fin_pl, 2.4), 2)
            r[f"flop_busy_pct_plain_at_sclk_{clock_ghz:.2f}GHz"] = round(frac(pl_fl, pl_us + fin_pl, clock_ghz), 2)
            # the like-for-like pair: counter and FLOP-derived figure over the SAME (PMC-pass) durations, FLOP priced at the sampled clock
            pmc_clock = (pmc_calls.get("sclk_mhz_median") or 2400.0) / 1e3
            r[f"flop_busy_pct_pmc_pass_at_sclk_{pmc_clock:.2f}GHz"] = round(frac(fl, pmc_us + fin_pmc, pmc_clock), 2)
        return r

    res = {"_note": "rocprofv3 --pmc MfmaUtil over tools/pmc_step.py (2 eager denoise steps, 512x512, batch 1) joined with the un-profiled "
                    "kernel trace of the same program and the call log's algorithmic FLOP; see tools/pmc_mfma.py",
           "sclk_mhz": {"plain": {k: plain_calls.get(k) for k in ("sclk_mhz_median", "sclk_mhz_min", "sclk_mhz_max", "sclk_mhz_samples")},
                        "pmc_pass": {k: pmc_calls.get(k) for k in ("sclk_mhz_median", "sclk_mhz_min", "sclk_mhz_max", "sclk_mhz_samples")}},
           "conv / dense family (main kernels)": family(is_conv),
           "conv / dense family (with splitk_finalize time in the denominator)": family(is_conv, True),
           "attention (all forms)": family(lambda k: "attention" in k or "xattn" in k),
           "self-attention S=4096 (attention32_kernel<40)": family(lambda k: "attention32_kernel<40" in k)}
    if calib_dir:
        cu = counters(calib_dir)
        cal = collections.defaultdict(list)
        for _did, (k, v) in cu.items():
            cal[k].append(v)
        cx = counters(calib_dir, "MfmaFlopsBF16")
        cg = counters(calib_dir, "GRBM_GUI_ACTIVE")
        calx = collections.defaultdict(list)
        for did, (k, v) in cx.items():
            if did in cg and cg[did][1] > 0:
                calx[k].append(100.0 * v / (FLOP_PER_CYCLE * cg[did][1] / XCDS))
        # tools/mfma_rate.py launches every mode on 1 block and on 256 blocks (one wave per SIMD, every SIMD): the 256-block
        # dispatches are the larger readings
        res["calibration (bare MFMA loops, tools/mfma_rate.py)"] = {
            k: {"dispatches": len(v), "mfma_util_pct_max": round(max(v), 2),
                "executed_flop_busy_pct_over_gui_cycles_max": round(max(calx[k]), 2) if calx.get(k) else None}
            for k, v in cal.items() if max(v) > 1.0}
    if commit:
        res["commit"] = commit
    res["kernels"] = dict(sorted(kernels.items(), key=lambda kv: -kv[1]["mfma_util_pct_time_weighted"]))
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "kernels"}, indent=1))
    for k, v in list(res["kernels"].items())[:16]:
        print(f"{v['mfma_util_pct_time_weighted']:6.2f} %  x{v['dispatches']:4d}  pmc {v['pmc_us_mean']:7.2f} us  plain {v['plain_us_mean']} us  "
              f"flop-derived {v.get('flop_busy_pct_plain_at_2.4GHz')}  {k[:90]}")


if __name__ == "__main__":
    main()
