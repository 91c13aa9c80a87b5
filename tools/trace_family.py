"""Per-kernel and per-family durations of a `rocprofv3 --kernel-trace --output-format csv` run, for the roofline figures.

    python tools/trace_family.py DIR [--steps N] [--md OUT.md --title "..."] [--csv OUT.csv] [--json OUT.json]

DIR holds *kernel_trace.csv.  Families are the C-ABI entry points (tools/pmc_summarize.klass): `conv_gemm` = conv_gemm_dma /
conv3x3_halo / dense_rowpanel / conv_wreg / conv_big + their splitk_finalize reductions — the dominant family of bench.py's
roofline block.  `--steps N`: the trace covers N fused denoise steps (tools/pmc_step.py --graph-loops L at S steps: (L + 1) S), so
family microseconds per step = total / N; that is the figure `roofline.achieved_in_graph` is computed from."""
import argparse
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summarize import klass   # noqa: E402


def load(dirname):
    files = glob.glob(os.path.join(dirname, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no kernel_trace.csv under {dirname}")
    per = collections.OrderedDict()
    for r in csv.DictReader(open(files[0])):
        ns = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        e = per.setdefault(r["Kernel_Name"], [0, 0, 1 << 62, 0])
        e[0] += 1
        e[1] += ns
        e[2] = min(e[2], ns)
        e[3] = max(e[3], ns)
    return per


def summarise(per, steps=None):
    fam = collections.defaultdict(lambda: {"dispatches": 0, "calls": 0, "us": 0.0})
    for name, (n, ns, _lo, _hi) in per.items():
        if name.startswith("void at::") or name.startswith("at::") or "rocclr" in name:
            k, is_call = "torch / runtime", True
        else:
            k, is_call = klass(name)
        f = fam[k]
        f["dispatches"] += n
        f["calls"] += n if is_call else 0
        f["us"] += ns / 1e3
    out = {k: {"dispatches": v["dispatches"], "calls": v["calls"], "total_us": round(v["us"], 1)} for k, v in fam.items()}
    if steps:
        for v in out.values():
            v["us_per_step"] = round(v["total_us"] / steps, 2)
            v["dispatches_per_step"] = round(v["dispatches"] / steps, 2)
        out["_steps"] = steps
    return out


def write_md(per, path, title, fam=None):
    tot = sum(v[1] for v in per.values()) or 1
    with open(path, "w") as f:
        f.write(f"# rocprofv3 --kernel-trace — {title}\n\n")
        if fam:
            f.write("| family (C-ABI entry point) | dispatches | total ms |" + (" us per fused step |" if "_steps" in fam else "") + "\n|---|---|---|" + ("---|" if "_steps" in fam else "") + "\n")
            for k, v in sorted(((k, v) for k, v in fam.items() if isinstance(v, dict)), key=lambda kv: -kv[1]["total_us"]):
                f.write(f"| {k} | {v['dispatches']} | {v['total_us'] / 1e3:.3f} |" + (f" {v['us_per_step']:.2f} |" if "_steps" in fam else "") + "\n")
            f.write("\n")
        f.write("| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|\n")
        for name, (n, ns, lo, hi) in sorted(per.items(), key=lambda kv: -kv[1][1]):
            nm = name.replace("|", "\\|")
            nm = nm if len(nm) <= 110 else nm[:107] + "..."
            f.write(f"| `{nm}` | {n} | {ns / 1e6:.2f} | {ns / n / 1e3:.2f} | {lo / 1e3:.2f} | {hi / 1e3:.2f} | {100.0 * ns / tot:.2f} |\n")


def write_csv(per, path):
    tot = sum(v[1] for v in per.values()) or 1
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for name, (n, ns, lo, hi) in sorted(per.items(), key=lambda kv: -kv[1][1]):
            w.writerow([name, n, ns, round(ns / n, 1), round(100.0 * ns / tot, 3), lo, hi])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--steps", type=int, default=0)
    ap.add_argument("--md", default=None)
    ap.add_argument("--csv", default=None)
    ap.add_argument("--json", default=None)
    ap.add_argument("--title", default="")
    args = ap.parse_args()
    per = load(args.dir)
    fam = summarise(per, args.steps or None)
    if args.md:
        write_md(per, args.md, args.title, fam)
    if args.csv:
        write_csv(per, args.csv)
    if args.json:
        with open(args.json, "w") as f:
            json.dump(fam, f, indent=1, sort_keys=True)
    print(json.dumps(fam, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
