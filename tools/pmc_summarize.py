"""Summarise the two rocprofv3 PMC passes over tools/pmc_step.py into profiles/<tag>_pmc_traffic.json.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 tools/pmc_step.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 tools/pmc_step.py
    python tools/pmc_summarize.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r1_pmc_traffic.json

Units and corrections as prescribed by MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and
WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts a 128-byte request as 64 bytes, so read bytes =
FETCH_SIZE x 1024 x 2; Infinity-Cache hits are counted (these are L2-miss / fabric bytes, an upper bound
on HBM bytes).  Kernels are grouped per C-ABI entry point; for msd_conv_gemm a "launch" is one call
(main kernel + its split-K reduction when there is one), which is how bench.py counts launches."""
import csv
import glob
import json
import os
import sys


def klass(name):
    if "splitk_finalize" in name:
        return "conv_gemm", False
    if "conv_gemm" in name or "conv3x3_halo" in name or "dense_rowpanel" in name or "conv_wreg" in name or "conv_big" in name:
        return "conv_gemm", True
    if name.startswith("void gn_") or name.startswith("gn_"):
        return "group_norm", "stats" not in name and "finalize" not in name
    if "layer_norm" in name:
        return "layer_norm", True
    if "attention" in name or "xattn_q" in name:
        return "attention", True
    return "other", True


def load(dirname, counter):
    files = glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {dirname}")
    acc = {}
    for row in csv.DictReader(open(files[0])):
        if row["Counter_Name"] != counter:
            continue
        k, is_call = klass(row["Kernel_Name"])
        e = acc.setdefault(k, {"kernels": 0, "calls": 0, "kib": 0.0})
        e["kernels"] += 1
        e["calls"] += 1 if is_call else 0
        e["kib"] += float(row["Counter_Value"])
    return acc


def summarise(rd, wr):
    """Per entry point: launches, KiB sums and corrected bytes per launch of the two passes (also called by bench.py on its own passes)."""
    res = {"_note": "rocprofv3 --pmc over tools/pmc_step.py (2 eager denoise steps, 512x512, batch 1 = fused cond+uncond batch 2); "
                    "bytes = FETCH_SIZE KiB x 1024 x 2 (gfx950 correction) + WRITE_SIZE KiB x 1024; L2-miss bytes incl. Infinity-Cache hits"}
    for k in sorted(set(rd) | set(wr)):
        r, w = rd.get(k, {"kernels": 0, "calls": 0, "kib": 0.0}), wr.get(k, {"kernels": 0, "calls": 0, "kib": 0.0})
        calls = max(r["calls"], w["calls"], 1)
        read_b, write_b = r["kib"] * 1024.0 * 2.0, w["kib"] * 1024.0
        res[k] = {"kernel_launches": max(r["kernels"], w["kernels"]), "calls": calls, "fetch_size_kib_sum": round(r["kib"], 1),
                  "write_size_kib_sum": round(w["kib"], 1), "read_bytes_per_launch": round(read_b / calls),
                  "write_bytes_per_launch": round(write_b / calls), "hbm_bytes_per_launch": round((read_b + write_b) / calls)}
    return res


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    commit = sys.argv[4] if len(sys.argv) > 4 else None   # the build the counters were taken on (bench.py quotes it)
    res = summarise(load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE"))
    if commit:
        res["commit"] = commit
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
