"""Summarise a rocprofv3 `--pmc MfmaUtil` pass over tools/pmc_step.py: matrix-pipe utilisation per kernel (the derived counter
MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES summed over SIMDs / (GRBM_GUI_ACTIVE x SIMD count) x 100, i.e. the share of SIMD-cycles of a
dispatch in which the MFMA pipe was busy — padding MFMAs included, unlike the algorithmic TFLOP/s of bench.py).

    rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d OUT -- python3 tools/pmc_step.py
    python tools/pmc_mfma.py OUT profiles/r3_pmc_mfma.json [commit]
"""
import collections
import csv
import glob
import json
import os
import sys


def main():
    d, out = sys.argv[1:3]
    commit = sys.argv[3] if len(sys.argv) > 3 else None
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {d}")
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(files[0])):
        if row["Counter_Name"] != "MfmaUtil":
            continue
        acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    res = {"_note": "rocprofv3 --pmc MfmaUtil over tools/pmc_step.py (2 eager denoise steps, 512x512, batch 1): percent of SIMD-cycles with "
                    "the matrix pipe busy, mean over the dispatches of each kernel (PMC collection serialises dispatches)"}
    rows = {k: {"dispatches": len(v), "mfma_util_pct": round(sum(v) / len(v), 2), "min": round(min(v), 2), "max": round(max(v), 2)}
            for k, v in acc.items() if max(v) > 0.5}
    res["kernels"] = dict(sorted(rows.items(), key=lambda kv: -kv[1]["mfma_util_pct"]))

    def klass(pred):
        v = [x for k, vs in acc.items() if pred(k) for x in vs]
        return {"dispatches": len(v), "mfma_util_pct": round(sum(v) / max(1, len(v)), 2)}
    res["attention (all forms)"] = klass(lambda k: "attention" in k or "xattn" in k)
    res["self-attention S=4096 (attention32_kernel<40"] = klass(lambda k: "attention32_kernel<40" in k)
    res["conv / dense (conv_gemm_dma, conv3x3_halo, dense_rowpanel)"] = klass(lambda k: "conv_gemm_dma" in k or "conv3x3_halo" in k or "dense_rowpanel" in k)
    if commit:
        res["commit"] = commit
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "kernels"}, indent=1))
    for k, v in list(res["kernels"].items())[:14]:
        print(f"{v['mfma_util_pct']:6.2f} %  x{v['dispatches']:4d}  {k[:100]}")


if __name__ == "__main__":
    main()
