"""Opcode census of a kernel's hot loop from the compiler's own assembly ("commit the census").

    python tools/isa_census.py [--kernel REGEX] [--out profiles/r6_attn32_isa_census.md]

Compiles minsdtf_amd/csrc/attention.hip to gfx950 assembly with the library's flags (hipcc -S --cuda-device-only: no GPU needed),
finds the kernel, splits it into basic blocks, takes the loop whose back edge closes on the block with the most MFMAs, and
counts opcodes on its COMMON path (at a conditional forward branch inside the loop the not-taken fall-through up to the branch
target is booked separately as the rare path: in attention32_kernel that is the reference-maximum move).  Issue-cycle model per
wave64 instruction, stated in the output: full-rate VALU 4, v_exp_f32 / v_rcp_f32 / v_rsq_f32 10 (tools/mfma_rate.py measured
9-12 alone, DESIGN r3), v_permlane 8, ds_read_b128 8 (1 KiB over a 128 B/clk port), MFMA 32x32x16 8 issue / 32 pipe,
16x16x32 8 issue / 16 pipe, scalar 0 (other issue port).  It is a model of ISSUE SLOTS, not a timeline.
"""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-fvisibility=hidden", "-mllvm", "-amdgpu-mfma-vgpr-form=1",
         "-mllvm", "-amdgpu-kernarg-preload-count=16", "-S", "--cuda-device-only"]


def issue_cycles(op):
    if op.startswith("v_mfma"):
        return 8
    if op.startswith(("v_exp", "v_rcp", "v_rsq", "v_log", "v_sqrt", "v_sin", "v_cos")):
        return 10
    if op.startswith("v_permlane"):
        return 8
    if op.startswith("ds_read_b128") or op.startswith("ds_write_b128"):
        return 8
    if op.startswith("ds_"):
        return 4
    if op.startswith(("global_", "buffer_", "flat_")):
        return 4
    if op.startswith("v_"):
        return 4
    return 0   # s_*: scalar unit / branch / waitcnt (no vector issue slot; waits are not priced here)


def pipe_cycles(op):
    if "32x32x16" in op:
        return 32
    if "16x16x32" in op or "16x16x16" in op:
        return 16
    return 0


def blocks_of(lines, start):
    end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks, cur = [], ["entry", []]
    blocks.append(cur)
    for i in range(start + 1, end):
        m = re.match(r"^(\.LBB\d+_\d+):", lines[i])
        if m:
            cur = [m.group(1), []]
            blocks.append(cur)
            continue
        t = lines[i].strip()
        if not t or t.startswith(";") or t.startswith("."):
            continue
        cur[1].append(t.split(";")[0].strip())
    return blocks


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default=r"_Z18attention32_kernelILi40ELi4ELi8ELb1EE")
    ap.add_argument("--source", default=os.path.join(ROOT, "minsdtf_amd", "csrc", "attention.hip"))
    ap.add_argument("--asm", default=None, help="an existing .s file (skips the compile)")
    ap.add_argument("--out", default=None)
    ap.add_argument("--tiles-per-iteration", type=int, default=2, help="key tiles one trip of the loop processes (the loop is unrolled by two)")
    args = ap.parse_args()
    asm = args.asm
    if asm is None:
        asm = os.path.join(tempfile.mkdtemp(prefix="isa_census_"), "k.s")
        subprocess.check_call(["hipcc"] + FLAGS + [args.source, "-o", asm], cwd=os.path.dirname(args.source), stderr=subprocess.DEVNULL)
    lines = open(asm).read().split("\n")
    starts = [i for i, ln in enumerate(lines) if re.match(r"^" + args.kernel + r"[^:]*:", ln)]
    if not starts:
        raise SystemExit(f"no kernel matches {args.kernel}")
    sym = lines[starts[0]].split(":")[0]
    blocks = blocks_of(lines, starts[0])
    idx = {b[0]: i for i, b in enumerate(blocks)}
    # loops: a branch to an EARLIER block.  Among the INNERMOST ones (no other loop's span strictly inside) take the one with the most MFMAs
    loops = []
    for i, (lab, ins) in enumerate(blocks):
        for x in ins:
            if x.startswith(("s_cbranch", "s_branch")):
                tgt = x.split()[-1]
                if tgt in idx and idx[tgt] <= i:
                    loops.append((idx[tgt], i))
    inner = [l for l in loops if not any(o != l and l[0] <= o[0] and o[1] <= l[1] for o in loops)]
    best = None
    for lo_, hi_ in inner:
        n = sum(1 for b in blocks[lo_:hi_ + 1] for y in b[1] if y.startswith("v_mfma"))
        if best is None or n > best[0]:
            best = (n, lo_, hi_)
    if best is None:
        raise SystemExit("no loop found")
    _, lo, hi = best
    common, rare = collections.Counter(), collections.Counter()
    skip_until = None
    for lab, ins in blocks[lo:hi + 1]:
        if skip_until is not None:
            if lab == skip_until:
                skip_until = None
            else:
                for x in ins:
                    rare[x.split()[0]] += 1
                continue
        for k, x in enumerate(ins):
            op = x.split()[0]
            if skip_until is not None:
                rare[op] += 1
                continue
            common[op] += 1
            if op.startswith("s_cbranch"):
                tgt = x.split()[-1]
                if tgt in idx and lo <= idx[tgt] <= hi and idx[tgt] > idx[lab]:   # forward skip inside the loop: its fall-through is the rare path
                    skip_until = tgt
        # (a block that ends inside a rare stretch: the following blocks up to the target are rare too)
    n = args.tiles_per_iteration
    rows = sorted(common.items(), key=lambda kv: -kv[1] * max(issue_cycles(kv[0]), 1))
    tot_issue = sum(c * issue_cycles(op) for op, c in common.items())
    tot_pipe = sum(c * pipe_cycles(op) for op, c in common.items())
    out = []
    out.append(f"# ISA census of the hot loop: `{sym}`\n")
    out.append(f"`python tools/isa_census.py --kernel '{args.kernel}'` - hipcc {' '.join(FLAGS[:3])} ... -S, ROCm 7.2; loop = blocks {blocks[lo][0]} .. {blocks[hi][0]} "
               f"({n} key tiles per trip); counts are PER 64-KEY TILE AND WAVE on the common path.  Issue model (wave64): full-rate VALU 4 cycles, "
               f"v_exp/v_rcp 10, v_permlane 8, ds_read_b128 8, MFMA 8 to issue (32x32x16: 32 on the matrix pipe), scalar / waits 0.\n")
    out.append("| opcode | per tile | issue cycles each | issue cycles per tile | share |")
    out.append("|---|---|---|---|---|")
    for op, c in rows:
        ic = issue_cycles(op)
        out.append(f"| `{op}` | {c / n:g} | {ic} | {c * ic / n:g} | {100.0 * c * ic / max(tot_issue, 1):.1f} % |")
    out.append(f"| **sum, common path** | {sum(common.values()) / n:g} | | **{tot_issue / n:g}** | matrix pipe: **{tot_pipe / n:g}** cycles |")
    cls = collections.Counter()
    for op, c in common.items():
        k = ("MFMA" if op.startswith("v_mfma") else "exponentials (v_exp_f32)" if op.startswith("v_exp") else "bf16 packs (v_cvt_pk_bf16_f32)" if op.startswith("v_cvt_pk")
             else "maximum chain (v_max3 / v_max)" if op.startswith("v_max") else "LDS fragment reads" if op.startswith("ds_") else "other VALU" if op.startswith("v_") else "scalar / waits / barrier")
        cls[k] += c * issue_cycles(op)
    out.append("\n| class | issue cycles per tile | share of the issue slots |")
    out.append("|---|---|---|")
    for k, v in sorted(cls.items(), key=lambda kv: -kv[1]):
        out.append(f"| {k} | {v / n:g} | {100.0 * v / max(tot_issue, 1):.1f} % |")
    if rare:
        r_issue = sum(c * issue_cycles(op) for op, c in rare.items())
        top = ", ".join(f"{c} x `{op}`" for op, c in sorted(rare.items(), key=lambda kv: -kv[1])[:6])
        out.append(f"\nRare path inside the loop (taken when some lane's score passes the reference-maximum threshold; not in the table): "
                   f"{sum(rare.values()) / n:g} instructions, {r_issue / n:g} issue cycles per occurrence (the loop holds {n} copies); over both copies: {top}.")
    text = "\n".join(out) + "\n"
    if args.out:
        with open(args.out, "w") as f:
            f.write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
