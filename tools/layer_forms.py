"""Which kernel form runs which layer: the launch configuration of every msd_conv_gemm call of the UNet (one and four images
per GPU, cond + uncond fused) and of the VAE decoder at 512x512, grouped by resolution level and layer kind, as the tuning
table + tuning.shape_config resolve them today.  No GPU needed (the emitters walk the topology with tensor-less weights).

    python tools/layer_forms.py > profiles/r6_layer_forms.md
"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def form_name(tm, tn, sk, stg):
    from minsdtf_amd import tuning as t

    if t.is_halo(tm):
        kind = {1128: "halo 8x16", 2128: "halo 8x16 (8 waves)", 1256: "halo 16x16"}[tm]
        extra = " 3 taps/step" if 30 <= stg < 60 else " 3 taps + loaders" if 60 <= stg < 90 else " rotated 3 taps" if 90 <= stg < 150 else " rotated 1 tap" if stg >= 150 else ""
        s = f"{kind} x{tn}{extra}"
    elif t.is_rowpanel(tm):
        s = f"row panel x{tn}"
    elif t.is_wreg(tm):
        s = f"wreg {tm - 4000}x{tn}" + (" 2K/stage" if stg >= 20 else "")
    elif t.is_big(tm):
        s = (f"staged halo 16x16 x{tn}" if stg >= 20 else f"big {tm - 5000}x{tn} chunk-major" if stg >= 10 else f"big {tm - 5000}x{tn}")
    else:
        s = f"tile {tm}x{tn}" + (" (8 waves)" if 10 <= stg < 20 else " (64x64/wave)" if stg >= 20 else "")
    return s + (f", split-K {sk}" if sk > 1 else "")


def kind_of(sh):
    batch, h_in, w_in, cin, N, ks, stride, ups, M, nk, allow_split, cx = sh
    if ks == 3:
        if stride == 2:
            return "3x3 stride 2 (downsample)"
        if ups:
            return "nearest x2 + 3x3 (upsample)"
        if cx:
            return "3x3 conv2 + folded 1x1 shortcut"
        if N <= 8:
            return "3x3 conv_out"
        return "3x3 (ResBlock conv1 / conv2)"
    if not allow_split:
        if w_in == 1 and cin == 768:
            return "context k|v projection"
        return "q|k|v projection" if N == 3 * cin else "GEGLU projection" if N == 8 * cin else "attn2.to_q" if N == cin else "Dense (LayerNorm consumer)"
    if cin == N:
        return "1x1 C->C (proj_in, to_out: LayerNorm-partial producers)"
    if cin == 5 * N:
        return "ff.net.2 + proj_out (K = 5C)"
    return "1x1 shortcut / other"


def main():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_host_cpu import _walk_shapes
    from minsdtf_amd import tuning

    print("# Which kernel form runs which layer (512x512; `python tools/layer_forms.py`)\n")
    print("Resolved from `minsdtf_amd/conv_tuning.json` (+ `tuning.shape_config` for shapes without a row).  One numerics class per layer and row: the forms of a row differ in speed, never in bits.\n")
    for net, nbs, h in (("UNet", (2, 8), 64), ("VAE decoder", (1, 4), 64)):
        rows = collections.OrderedDict()
        for nb in nbs:
            for sh in _walk_shapes(nb, h, h, "unet" if net == "UNet" else "vae"):
                batch, h_in, w_in, cin, N, ks, stride, ups, M, nk, allow_split, cx = sh
                if w_in == 1 and cin == 768:
                    lvl = "text context"
                else:
                    lvl = f"{h_in}x{w_in}"
                key = (lvl, kind_of(sh))
                cfg = tuning.lookup(*sh)
                rows.setdefault(key, {}).setdefault(nb, collections.Counter())[form_name(*cfg)] += 1
        hdr = " | ".join(f"fused batch {nb} ({nb // 2 if net == 'UNet' else nb} image{'s' if (nb // 2 if net == 'UNet' else nb) > 1 else ''} per GPU)" for nb in nbs)
        print(f"## {net}\n\n| level | layer kind | {hdr} |\n|---|---|" + "---|" * len(nbs))
        for (lvl, kind), per in rows.items():
            cells = []
            for nb in nbs:
                c = per.get(nb, {})
                cells.append("; ".join(f"{n} x {f}" if n > 1 else f for f, n in sorted(c.items(), key=lambda kv: -kv[1])))
            print(f"| {lvl} | {kind} | " + " | ".join(cells) + " |")
        print()


if __name__ == "__main__":
    main()
