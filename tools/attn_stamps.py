"""Phase timeline of attention32_kernel (needs `make -C minsdtf_amd/csrc stamps [ASTAMP_MASK=0x..] [STAMP_OUT=..]`).

    python tools/attn_stamps.py [--lib tools/_build/libminsdtf_hip_stamps.so] [--only IDX]

Wave 0 of every workgroup sums the shader-clock time between fixed points of its tile loop: 0 barrier passed, 1 QK^T MFMAs
issued (K fragments read), 2 lane maximum known (= QK^T results back), 3 exponentials done, 4 PV MFMAs issued, 5 next
tile stored to LDS.  Printed: mean cycles per tile of every interval (interval i ends at stamp i), the total per tile, and
the s_memtime : 100 MHz wall-clock ratio (the clock the counts are in)."""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NAMES = ["barrier wait", "K reads + QK issue", "QK done + max chain", "rescale branch + exp", "packs + V reads + PV issue",
         "gload wait + lstore", "-", "-"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.path.join(ROOT, "tools", "_build", "libminsdtf_hip_stamps.so"))
    ap.add_argument("--only", type=int, default=-1)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--form", type=int, default=2, help="1 = plain loop, 2 = software-pipelined loop (stamps: 0 barrier, 1 next tile stored + loads issued, "
                    "2 QK^T(t+1) issued + maximum of tile t, 3 exponentials, 4 PV issued)")
    ap.add_argument("--qf", type=int, default=0, help="0 = automatic, 1 / 2 / 4 = 64 / 128 / 256 queries per workgroup")
    args = ap.parse_args()
    from minsdtf_amd import _lib

    _lib.LIB_PATH = args.lib
    from minsdtf_amd import ops
    from tools.attn_bench import SHAPES

    lib = _lib.load()
    lib.msd_init()
    _lib.check(lib.msd_set_option(b"attn_form", args.form), "attn_form")
    _lib.check(lib.msd_set_option(b"attn_qf", args.qf), "attn_qf")
    lib.msd_debug_stamps_attn.restype = C.c_int
    lib.msd_debug_stamps_attn.argtypes = [C.c_void_p, C.c_int]
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream()
    for idx, (name, B, H, d, S, T) in enumerate(SHAPES):
        if d not in (40, 80) or T < 256 or (args.only >= 0 and idx != args.only):
            continue
        Cc = H * d
        q = torch.randn(B, S, Cc, device=dev).to(torch.bfloat16) * (d ** -0.5 * 1.4427)
        k = torch.randn(B, T, Cc, device=dev).to(torch.bfloat16)
        vt = torch.randn(B, Cc, T, device=dev).to(torch.bfloat16)
        out = torch.empty(B, S, Cc, device=dev, dtype=torch.bfloat16)
        call = ops.attention(q=q, k=k, vt=vt, out=out, batch=B, heads=H, head_dim=d, s=S, t=T, q_ld=Cc, k_ld=Cc, vt_ld=T, o_ld=Cc,
                             scale=d ** -0.5, q_prescaled=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(2):
            call(st.cuda_stream)
        torch.cuda.synchronize()
        e0.record(st)
        for _ in range(args.iters):
            call(st.cuda_stream)
        e1.record(st)
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / args.iters
        qt = 64 * args.qf if args.qf else (64 if B * H * ((S + 127) // 128) < 384 else 128)
        nwg = min(2048, B * H * ((S + qt - 1) // qt))
        buf = np.zeros(16 * 4096, dtype=np.uint64)
        rc = lib.msd_debug_stamps_attn(buf.ctypes.data, buf.size)
        assert rc == 0, rc
        rows = buf.reshape(4096, 16)[:nwg].astype(np.float64)
        nt = rows[:, 11]
        ratio = rows[:, 8].sum() / max(1.0, rows[:, 9].sum())
        print(f"{idx} {name}: {us:.1f} us per launch (instrumented), {nwg} workgroups, {nt.mean():.0f} tiles; s_memtime ticks per 10 ns: {ratio:.3f}")
        tot = 0.0
        for i in range(7):
            v = (rows[:, i] / nt).mean()
            tot += v
            print(f"    {i} {NAMES[i]:28s} {v:8.1f} ticks / tile   (p10 {np.percentile(rows[:, i] / nt, 10):.1f}, p90 {np.percentile(rows[:, i] / nt, 90):.1f})")
        print(f"    sum {tot:.1f} ticks / tile; loop span {(rows[:, 8] / nt).mean():.1f}; wall span of a workgroup {(rows[:, 9]).mean() / 100:.2f} us")
        if args.form == 2:
            print("    loader 0: wait for tile t+1 %.1f, barrier %.1f, issue of tile t+3 %.1f ticks / tile" % tuple((rows[:, 12 + i] / nt).mean() for i in range(3)))
        allr = buf.reshape(4096, 16).astype(np.float64)
        for w in range(1, 8):
            rw = allr[2048 + np.arange(min(nwg, 256)) * 8 + w]
            if rw[:, 11].min() > 0:
                print(f"    wave {w}: " + "  ".join(f"{(rw[:, i] / rw[:, 11]).mean():7.1f}" for i in (0, 2, 3, 4)) + "   (barrier, QK + max, exp, PV)")
        if args.form == 2:
            raw = buf.reshape(4096, 16)[:nwg, 15]
            print(f"    loader 0 prologue: issue of tiles 0-2 {(raw >> np.uint64(32)).astype(np.float64).mean():.0f} ticks, then {(raw & np.uint64(0xFFFFFFFF)).astype(np.float64).mean():.0f} until tiles 0-1 have landed")
        w0 = rows[:, 10]
        print(f"    workgroup entry spread {(w0.max() - w0.min()) / 100:.2f} us")


if __name__ == "__main__":
    main()
