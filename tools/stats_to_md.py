"""rocprofv3 --kernel-trace --stats kernel_stats.csv -> a markdown table for profiles/.

    python tools/stats_to_md.py gpurun_out/prof/<pid>_kernel_stats.csv "title / command" > profiles/rN_bench_kernel_stats.md
"""
import csv
import sys


def main():
    path, title = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    rows = list(csv.DictReader(open(path)))
    print(f"# rocprofv3 --kernel-trace --stats — {title}\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|")
    for r in rows:
        name = r["Name"].replace("|", "\\|")
        if len(name) > 110:
            name = name[:107] + "..."
        print(f"| `{name}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.2f} | "
              f"{float(r['MinNs']) / 1e3:.2f} | {float(r['MaxNs']) / 1e3:.2f} | {float(r['Percentage']):.2f} |")


if __name__ == "__main__":
    main()
