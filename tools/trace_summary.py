"""Summarise a rocprofv3 --kernel-trace (CSV) of a run with roctx phase ranges (bench.py --sync-phases, rocprofv3
--marker-trace) into per-phase and per-kernel tables.

    rocprofv3 --kernel-trace --marker-trace --output-format csv -d gpurun_out/prof -- python3 bench.py --sync-phases ...
    python tools/trace_summary.py gpurun_out/prof profiles/r2_phase_summary.md ["title"]

Attribution is by TIME: a kernel dispatch belongs to the innermost phase range (prepare / denoise_loop / vae_decode / d2h)
whose host interval contains the dispatch's start; with --sync-phases the device is drained at every range end, so host
ranges and device work line up.  Dispatches outside every range (warm-up, roofline passes) are listed as "(outside)".
"""
import collections
import csv
import glob
import os
import sys


def find(d, pat):
    f = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return f[0] if f else None


def main():
    d, out = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else ""
    kt = find(d, "*kernel_trace.csv")
    mk = find(d, "*marker_api_trace.csv")
    if not kt:
        raise SystemExit(f"no kernel_trace.csv under {d}")
    ranges = []
    if mk:
        for r in csv.DictReader(open(mk)):
            name = r.get("Function") or r.get("Name") or ""
            try:
                ranges.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
            except (KeyError, ValueError):
                continue
    ranges.sort()
    per_phase = collections.defaultdict(lambda: [0, 0.0])
    per_kernel = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for r in csv.DictReader(open(kt)):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        phase = "(outside)"
        for (a, b, name) in ranges:
            if a <= s <= b:
                phase = name   # innermost = the last containing range in start order
        per_phase[phase][0] += 1
        per_phase[phase][1] += (e - s) / 1e3
        k = r["Kernel_Name"]
        per_kernel[phase][k][0] += 1
        per_kernel[phase][k][1] += (e - s) / 1e3
    njobs = max(1, sum(1 for x in ranges if x[2] == "denoise_loop"))
    with open(out, "w") as f:
        f.write(f"# rocprofv3 --kernel-trace + --marker-trace by phase — {title}\n\n")
        f.write(f"{njobs} profiled job(s); kernel time per job and phase (device time of the dispatches whose start lies in the phase's host range):\n\n")
        f.write("| phase | dispatches per job | kernel ms per job |\n|---|---|---|\n")
        for ph, (n, us) in sorted(per_phase.items(), key=lambda kv: -kv[1][1]):
            div = njobs if ph != "(outside)" else 1
            f.write(f"| {ph} | {n / div:.0f} | {us / div / 1e3:.3f} |\n")
        for ph in sorted(per_kernel, key=lambda k: -per_phase[k][1]):
            if ph == "(outside)":
                continue
            f.write(f"\n## {ph}: top kernels (per job)\n\n| kernel | calls | total ms | avg us |\n|---|---|---|---|\n")
            for k, (n, us) in sorted(per_kernel[ph].items(), key=lambda kv: -kv[1][1])[:14]:
                kk = k.replace("|", "\\|")
                kk = kk if len(kk) <= 100 else kk[:97] + "..."
                f.write(f"| `{kk}` | {n / njobs:.0f} | {us / njobs / 1e3:.3f} | {us / n:.2f} |\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
