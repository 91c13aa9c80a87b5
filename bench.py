#!/usr/bin/env python3
"""Headline benchmark: 512x512 images/sec (whole node), SD1.5 25-step txt2img on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1: either launched by torch.distributed.run (one rank per GPU, RANK / LOCAL_RANK / WORLD_SIZE in the environment), or
run plainly — then this process becomes a launcher that starts the N ranks itself (`launch_ranks`) and relays rank 0's
line.  The line carries `n_ranks_seen` (the process group's world size) and `rank_devices` (what each rank ran on).

One "step" = one full pass of the hot path over this rank's batch: the 25-step cond+uncond denoise
loop (50 UNet forwards per image, CFG 7.5, rescale 0.7) + the VAE decode to uint8, on synthetic
inputs already resident in HBM (random-init SD1.5 weights of the reference's architecture, seeded;
N(0,1) text contexts and noise).  Multi-GPU = batch sharding: per-GPU work is fixed (weak scaling),
the text contexts + global noise are broadcast from rank 0 and the images all-gathered inside the
timed region (RCCL), nothing crosses GPUs inside a step.

Rank 0 prints ONE JSON line with the contract fields plus
  roofline      the dominant kernel (implicit-GEMM conv/dense on MFMA): algorithmic FLOP per launch
                / its average launch duration measured here with HIP events on the launch stream;
                `traffic` = L2-miss bytes per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) that THIS
                run makes as child processes over tools/pmc_step.py before it touches the GPU itself (N = 1 only;
                --quoted-traffic, or a box without rocprofv3, falls back to the newest file under profiles/ and says so)
  cpu_baseline  the oracle (fp32 CPU restatement of the reference path; Keras itself is not
                installable here) timed on this box's host cores on a bounded sample
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

T_PROCESS_START = time.time()
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

UNET_TFLOP = 0.8033   # per UNet forward, 512x512, ctx 77, per sample
VAE_TFLOP = 2.5145    # per decode
UNET_SELF_ATTN_TFLOP = 0.1225   # of which self-attention QK^T + AV: grows with the SQUARE of the image area
VAE_ATTN_TFLOP = 0.0344         # (17.2 GMAC single-head attention of the decoder's mid block)
MFMA_PEAK_TFLOPS = 2500.0  # dense bf16, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0      # HBM3E, MI355X_MICROARCH.md


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch-per-gpu", type=int, default=1)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--denoise-steps", type=int, default=25)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--quoted-traffic", action="store_true",
                    help="roofline.traffic from the newest PMC summary under profiles/ instead of two live rocprofv3 --pmc passes (saves ~2 min)")
    ap.add_argument("--controlnet", action="store_true", help="BASELINE config 5: ControlNet residuals every step")
    ap.add_argument("--sync-phases", action="store_true", help="drain the device at the end of every roctx phase range (profiling)")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=INT",
                    help="library A/B switch passed to msd_set_option (same-box comparisons), e.g. --opt attn_swp=0")
    ap.add_argument("--streams", type=int, default=0, help="1: cond+uncond as one batch-2B forward; 2: two HIP streams; 0: automatic")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="nccl = RCCL (the measured path); gloo: launcher self-test on CPU")
    ap.add_argument("--force-collectives", action="store_true",
                    help="run the packed broadcast and the all-gather through torch.distributed even at world size 1 (a one-rank "
                         "RCCL process group): exercises the multi-GPU exchange path on a one-GPU box")
    ap.add_argument("--fail-rank", type=int, default=-1,
                    help="self-test of the failure path: this rank raises right after init_process_group, in front of its first collective")
    ap.add_argument("--stub-local", action="store_true",
                    help="replace the GPU pipeline by a trivial per-sample generator (launcher / sharding self-test; the "
                         "line is marked stub and is not a measurement)")
    return ap.parse_args(argv)


LIVE_TRAFFIC = None   # summary of this run's own PMC passes (live_traffic), read by kernel_roofline
LIVE_INGRAPH = None   # family durations per fused step inside the replayed whole-loop hipGraph (in_graph_trace), read by kernel_roofline


def in_graph_trace(args):
    """What the dominant family costs INSIDE the captured loop: `rocprofv3 --kernel-trace` (no counters) over tools/pmc_step.py
    --graph-loops 1 - the whole denoise loop as one hipGraph, captured and replayed twice exactly as the timed region replays it -
    started as a child process before this process has touched the GPU; the family's kernel durations (split-K reductions included)
    summed by tools/trace_family.py and divided by the fused steps in the trace.  None (reason on stderr) when it cannot run here."""
    import shutil
    import subprocess
    import tempfile

    if os.environ.get("ROCP_TOOL_LIBRARIES") or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        log("in-graph trace: this process is itself running under a profiler; no in-graph figure")
        return None
    exe = shutil.which("rocprofv3")
    if not exe:
        log("in-graph trace: no rocprofv3 on PATH; no in-graph figure")
        return None
    from tools import trace_family

    loops = 1
    step_args = ["--size", str(args.size), "--batch", str(args.batch_per_gpu), "--denoise-steps", str(args.denoise_steps),
                 "--graph-loops", str(loops)] + (["--controlnet"] if args.controlnet else [])
    for kv in args.opt:
        step_args += ["--opt", kv]
    tmp = tempfile.mkdtemp(prefix="msd_bench_trace_")
    t0 = time.time()
    try:
        cmd = [exe, "--kernel-trace", "--output-format", "csv", "-d", os.path.join(tmp, "kt"), "--",
               "python3", os.path.join(ROOT, "tools", "pmc_step.py")] + step_args
        try:
            r = subprocess.run(cmd, cwd=tmp, env=dict(os.environ, TMPDIR=os.environ.get("TMPDIR", "/tmp")), stdout=subprocess.PIPE,
                               stderr=subprocess.STDOUT, timeout=400)
        except subprocess.TimeoutExpired:
            log("in-graph trace: the pass did not finish in 400 s; no in-graph figure")
            return None
        if r.returncode != 0:
            log(f"in-graph trace: the pass failed (rc {r.returncode}): {r.stdout.decode(errors='replace')[-400:]}")
            return None
        try:
            per = trace_family.load(os.path.join(tmp, "kt"))
        except SystemExit as e:
            log(f"in-graph trace: {e}")
            return None
        # fused steps in the trace = its sampler launches (one per step: loops + 1 replays of the loop, plus the one eager warm-up
        # step the engine runs before it captures)
        steps = sum(v[0] for name, v in per.items() if "cfg_step_kernel" in name) or (loops + 1) * args.denoise_steps
        fam = trace_family.summarise(per, steps)
        fam["seconds"] = round(time.time() - t0, 1)
        keep = os.environ.get("MSD_BENCH_KEEP_TRACE")   # tools/measure_round.sh: the same trace becomes profiles/rN_bench_kernel_stats<tag>.{csv,md}
        if keep:
            try:
                tag = config_tag(args.batch_per_gpu, args.size, args.controlnet)
                title = (f"kernels of the replayed whole-loop hipGraph, {steps} fused steps ({loops + 1} loops x {args.denoise_steps} + the eager warm-up step), "
                         f"{args.size}x{args.size}, batch {args.batch_per_gpu}/GPU{', ControlNet' if args.controlnet else ''}: the child pass of "
                         f"`python bench.py` that roofline.achieved_in_graph is computed from")
                trace_family.write_csv(per, keep + f"_bench_kernel_stats{tag}.csv")
                trace_family.write_md(per, keep + f"_bench_kernel_stats{tag}.md", title, fam)
            except Exception as e:   # (keeping the summary is a convenience of tools/measure_round.sh: never the bench's problem)
                log(f"in-graph trace: could not keep the trace summary: {type(e).__name__}: {e}")
        return fam
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def live_traffic(args):
    """roofline.traffic measured by THIS run: `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes, as
    MI355X_MICROARCH.md prescribes; the program itself after `--`) over tools/pmc_step.py — the same launch list run eagerly for two
    denoise steps — started as CHILD processes before this process has touched the GPU, summarised by tools/pmc_summarize.py (its unit
    and gfx950 corrections).  Returns the summary dict, or None (with the reason on stderr) when the passes cannot run here."""
    import shutil
    import subprocess
    import tempfile

    if os.environ.get("ROCP_TOOL_LIBRARIES") or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        log("live traffic: this process is itself running under a profiler; quoting profiles/ instead")
        return None
    exe = shutil.which("rocprofv3")
    if not exe:
        log("live traffic: no rocprofv3 on PATH; quoting profiles/ instead")
        return None
    from tools import pmc_summarize

    step_args = ["--size", str(args.size), "--batch", str(args.batch_per_gpu)] + (["--controlnet"] if args.controlnet else [])
    for kv in args.opt:
        step_args += ["--opt", kv]
    tmp = tempfile.mkdtemp(prefix="msd_bench_pmc_")
    env = dict(os.environ, TMPDIR=os.environ.get("TMPDIR", "/tmp"))
    acc = {}
    t0 = time.time()
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter.lower())
            cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--",
                   "python3", os.path.join(ROOT, "tools", "pmc_step.py")] + step_args
            try:
                r = subprocess.run(cmd, cwd=tmp, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
            except subprocess.TimeoutExpired:
                log(f"live traffic: the {counter} pass did not finish in 300 s; quoting profiles/ instead")
                return None
            if r.returncode != 0:
                log(f"live traffic: the {counter} pass failed (rc {r.returncode}): {r.stdout.decode(errors='replace')[-400:]}")
                return None
            try:
                acc[counter] = pmc_summarize.load(d, counter)
            except SystemExit as e:
                log(f"live traffic: {e}")
                return None
        res = pmc_summarize.summarise(acc["FETCH_SIZE"], acc["WRITE_SIZE"])
        res["seconds"] = round(time.time() - t0, 1)
        return res
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _free_port() -> int:
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n: int, argv) -> int:
    """`python bench.py --gpus N` outside a torchrun environment: start the N ranks ourselves, one child process per GPU
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the child's environment, exactly what torch.distributed.run would set).
    This parent never initialises the GPU (no torch.cuda call; counting devices is not one), so starting children is safe;
    the children are fresh interpreters, nothing is re-exec'd.  Rank 0's stdout (the ONE JSON line) is relayed; the exit
    code is the worst of the children's; a failed rank ends the others (exact PIDs), there is no retry."""
    import subprocess

    port = _free_port()
    procs = []
    import shutil
    import tempfile

    # rank 0 hands its packed weights to the other ranks through this RAM-backed directory (load_synthetic_shared); ours to remove
    shm = tempfile.mkdtemp(prefix="msd_bench_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MSD_BENCH_LAUNCHER="self", MSD_BENCH_SHARE_DIR=shm)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: the only mode this host driver supports for RCCL
        env.setdefault("NCCL_DEBUG", "WARN")                # RCCL's own reason for a failed init / collective reaches the log verbatim
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))
    log(f"launcher: started {n} ranks (pids {[p.pid for p in procs]}), rendezvous 127.0.0.1:{port}")
    rc = 0
    line = None
    pending = set(range(n))
    out0 = []
    import signal
    import threading

    def pump():   # rank 0's stdout must be drained while we poll, or a full pipe would stall it
        for ln in procs[0].stdout:
            out0.append(ln.decode(errors="replace"))

    def stop_all(grace=10.0):
        """End every rank still running — exact PIDs, terminate, then kill after `grace` seconds — and reap them: a rank left
        alone would sit in an RCCL collective for ever and keep its GPU."""
        live = [p for p in procs if p.poll() is None]
        for p in live:
            p.terminate()
        t_end = time.time() + grace
        for p in live:
            try:
                p.wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()

    class _Stop(Exception):
        pass

    def on_signal(signum, _frame):
        raise _Stop(signum)

    old = {sg: signal.signal(sg, on_signal) for sg in (signal.SIGTERM, signal.SIGINT)}
    deadline = time.time() + float(os.environ.get("MSD_BENCH_LAUNCHER_DEADLINE_S", "3300"))
    th = threading.Thread(target=pump, daemon=True)
    th.start()
    try:
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0:
                    log(f"launcher: rank {r} exited with code {code}")
                    rc = rc or code
                    stop_all()   # the other ranks would wait for it in a collective forever
                    pending.clear()
                    break
            if pending and time.time() > deadline:
                log("launcher: deadline passed, ending the ranks")
                rc = rc or 124
                stop_all()
                pending.clear()
            time.sleep(0.05)
    except _Stop as e:
        log(f"launcher: signal {e.args[0]}, ending the ranks")
        rc = 128 + int(e.args[0])
    finally:
        stop_all()   # (no-op when every rank has exited)
        shutil.rmtree(shm, ignore_errors=True)
        for sg, h in old.items():
            signal.signal(sg, h)
    th.join(timeout=10)
    for ln in out0:
        if ln.lstrip().startswith("{"):
            line = ln.strip()
        else:
            sys.stderr.write(ln)
    if rc == 0 and line is None:
        log("launcher: rank 0 printed no JSON line")
        rc = 1
    if line is not None and rc == 0:
        print(line, flush=True)
    return rc if 0 <= rc < 256 else 1


def share_dir(world: int):
    """Directory (RAM-backed) through which rank 0 hands its PACKED weights to the other ranks of the node, or None (one
    rank, or MSD_BENCH_SHARE_WEIGHTS=0).  The launcher names it ($MSD_BENCH_SHARE_DIR) and removes it at the end; under
    torchrun it is derived from the rendezvous port, and rank 0 removes it after the closing barrier."""
    if world <= 1 or os.environ.get("MSD_BENCH_SHARE_WEIGHTS", "1") == "0":
        return None
    d = os.environ.get("MSD_BENCH_SHARE_DIR")
    if not d:
        base = "/dev/shm" if os.path.isdir("/dev/shm") else os.environ.get("TMPDIR", "/tmp")
        d = os.path.join(base, f"msd_bench_{os.getuid()}_{os.environ.get('MASTER_PORT', '0')}")
    os.makedirs(d, exist_ok=True)
    return d


def load_synthetic_shared(model, rank: int, sdir, seed=0, bias_scale=0.0, wait_s=600.0):
    """model.load_synthetic(seed, bias_scale) on rank 0 - which then writes the packed result into `sdir` - and a map + upload
    of that file on the other ranks (8 ranks generating and packing the same 3.4 GB of fp32 draws is 8x the host work for
    nothing).  Same tensors bit for bit: what the ranks launch on IS rank 0's packing.  Returns the Keras-layout arrays on
    rank 0 (the CPU baseline wants them), None elsewhere.  A rank whose file does not appear within `wait_s` seconds says so
    and generates its own copy (rank 0 died: the launcher is about to end this rank anyway)."""
    if sdir is None:
        return model.load_synthetic(seed=seed, bias_scale=bias_scale)
    path = os.path.join(sdir, f"{model.name}.packed.pt")
    meta = dict(seed=seed, bias_scale=bias_scale)
    if rank == 0:
        arrays = model.load_synthetic(seed=seed, bias_scale=bias_scale)
        model.save_packed(path, **meta)
        return arrays
    t_end = time.time() + wait_s
    while not os.path.exists(path):
        if time.time() > t_end:
            log(f"[rank {rank}] {path} did not appear in {wait_s:.0f} s: generating {model.name}'s weights here")
            model.load_synthetic(seed=seed, bias_scale=bias_scale)
            return None
        time.sleep(0.05)
    model.load_packed(path, **meta)
    return None


def stub_generate(ctx, unc, z, *rest):
    """--stub-local: a per-sample function of the sliced inputs (no cross-sample coupling, like the real pipeline), so the
    gathered batch proves broadcast, slicing and gather order.  uint8 [b, 4, 4, 3]."""
    import torch

    ctx, unc, z = (torch.as_tensor(np.asarray(a.cpu()) if isinstance(a, torch.Tensor) else a).float() for a in (ctx, unc, z))
    v = z.reshape(z.shape[0], -1)[:, :48] * 20 + ctx.mean(dim=(1, 2))[:, None] * 100 + unc.std(dim=(1, 2), unbiased=False)[:, None] * 10
    for e in rest:   # per-sample extras (hint images) move their own sample only
        e = torch.as_tensor(np.asarray(e.cpu()) if isinstance(e, torch.Tensor) else e).float()
        v = v + e.reshape(e.shape[0], -1).mean(dim=1, keepdim=True) * 50
    return torch.clamp(v + 128, 0, 255).to(torch.uint8).reshape(z.shape[0], 4, 4, 3)


def gather_scalars(x, dev, world):
    """One float per rank, rank order (None where a rank has none)."""
    import torch
    import torch.distributed as dist

    v = float("nan") if x is None else float(x)
    if world == 1:
        return [None if v != v else v]
    mine = torch.tensor([v], dtype=torch.float64, device=dev)
    every = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(every, mine)
    return [None if float(t.item()) != float(t.item()) else round(float(t.item()), 2) for t in every]


def rank_devices(dev, world):
    """What every rank ran on, gathered: proves to the reader of the line that the process group saw `world` ranks on
    `world` different devices (hipGetDevice ordinal + PCI bus id where the runtime reports one)."""
    import torch
    import torch.distributed as dist

    me = {"rank": int(os.environ.get("RANK", 0)), "pid": os.getpid(), "device": str(dev)}
    if dev.type == "cuda":
        me["hip_device"] = int(torch.cuda.current_device())
        pr = torch.cuda.get_device_properties(dev)
        me["name"] = pr.name
        for k in ("pci_bus_id", "uuid"):
            if hasattr(pr, k):
                me[k] = str(getattr(pr, k))
    if world == 1:
        return [me]
    got = [None] * world
    dist.all_gather_object(got, me)
    return got


def main(argv=None):
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # the driver's plain `python bench.py --gpus N`: become the launcher BEFORE anything touches a GPU
        sys.exit(launch_ranks(args.gpus, sys.argv[1:] if argv is None else argv))

    if (args.gpus == 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.backend == "nccl" and not args.stub_local
            and not args.no_roofline and not args.quoted_traffic):
        global LIVE_TRAFFIC, LIVE_INGRAPH   # (children first: nothing in this process has initialised the GPU yet)
        LIVE_TRAFFIC = live_traffic(args)
        LIVE_INGRAPH = in_graph_trace(args)
    elif (args.gpus == 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.backend == "nccl" and not args.stub_local and not args.no_roofline):
        LIVE_INGRAPH = in_graph_trace(args)   # (--quoted-traffic: the counters are quoted from profiles/, the in-graph trace is still this run's)

    import torch
    import torch.distributed as dist

    from minsdtf_amd import dist as mdist
    from minsdtf_amd import host as mhost

    rank, local_rank, world = mdist.env_rank()
    # torch sizes its CPU thread pool by os.cpu_count() (256 on these hosts: 128 threads) although the container's cgroup grants
    # a fraction of that (16 CPUs per GPU): 8 runnable threads per CPU made the weight packing and the CPU baseline 3-5x slower
    # than the CPUs allow.  N ranks of one node share the quota.
    cpu_threads = mhost.fit_torch_threads()
    if world > 1 and torch.get_num_threads() > max(1, mhost.effective_cpus() // world):
        torch.set_num_threads(max(1, mhost.effective_cpus() // world))
        cpu_threads = torch.get_num_threads()
    if world != args.gpus:
        log(f"note: WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE")
    stub = args.stub_local
    if args.backend == "nccl":
        # dmabuf IPC is the only mode this pool's host driver supports for RCCL / cross-process device memory; the HSA runtime reads
        # the variable when it starts, i.e. at the first call below that touches the device (ranks started by torch.distributed.run
        # do not pass through launch_ranks, which sets it for its own children)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if torch.cuda.device_count() <= local_rank:
            log(f"[rank {rank}] needs HIP device {local_rank}, {torch.cuda.device_count()} visible")
            sys.exit(2)
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    else:
        if not stub:
            raise SystemExit("--backend gloo is the CPU self-test of the launcher / sharding path: it needs --stub-local")
        dev = torch.device("cpu")
    if args.force_collectives:
        mdist.FORCE_COLLECTIVES = True
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # (as launch_ranks sets it for its children)
        if world == 1:   # a one-rank group needs its own rendezvous: a free local port, nothing inherited
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
    os.environ.setdefault("NCCL_DEBUG", "WARN")   # (torchrun-launched ranks: as launch_ranks sets it)
    try:
        mdist.init(args.backend, force=args.force_collectives)
    except Exception as e:   # the first place a rank meets RCCL: say which rank, with what IPC mode, and die (the launcher ends the others)
        log(f"[rank {rank}] init_process_group({args.backend}) FAILED: {type(e).__name__}: {e} "
            f"(HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}, NCCL_DEBUG={os.environ.get('NCCL_DEBUG')})")
        raise
    n_ranks_seen = mdist.world_size()
    assert n_ranks_seen == world, (n_ranks_seen, world)
    if args.fail_rank >= 0 and rank == args.fail_rank:   # (launcher / failure-path self-test: tests/test_dist_cpu.py)
        raise RuntimeError(f"[rank {rank}] --fail-rank: raising in front of the first collective")

    size, nsteps, b = args.size, args.denoise_steps, args.batch_per_gpu
    h = size // 8
    gb = b * world
    rng = np.random.default_rng(1234)
    ctx = rng.standard_normal((gb, 77, 768)).astype(np.float32)
    unc = rng.standard_normal((gb, 77, 768)).astype(np.float32)
    noise = np.random.default_rng(0).standard_normal((gb, h, h, 4)).astype(np.float32)
    hints = ()
    if args.controlnet:   # one hint image per sample of the GLOBAL batch; it travels in the same broadcast (§8e)
        hints = (np.random.default_rng(7).integers(0, 256, (gb, size, size, 3)).astype(np.float32) / 255.0,)
    if rank != 0:   # only rank 0's inputs count: the others hand in right-shaped garbage
        ctx, unc, noise = np.zeros_like(ctx), np.ones_like(unc), np.full_like(noise, 7.0)
        hints = tuple(np.zeros_like(x) for x in hints)

    sd = unet_arrays = vae_arrays = None
    if stub:
        local = stub_generate
        out_shape = (gb, 4, 4, 3)
    else:
        from minsdtf_amd.stable_diffusion import StableDiffusion

        if args.opt:
            from minsdtf_amd import _lib

            for kv in args.opt:
                k, v = kv.split("=")
                _lib.check(_lib.load().msd_set_option(k.encode(), int(v)), f"msd_set_option({kv})")
        t0 = time.time()
        sd = StableDiffusion(size, size, jit_compile=not args.no_graph, device=dev)
        sd.denoise_streams = args.streams or None
        sdir = share_dir(world)   # N > 1: rank 0 generates + packs, the others map its packed file (None at N = 1)
        unet_arrays = load_synthetic_shared(sd.diffusion_model, rank, sdir)
        vae_arrays = load_synthetic_shared(sd.image_decoder, rank, sdir)
        if args.controlnet:  # zero-convs are NOT zero (bias_scale > 0 also draws non-trivial biases), else the path is vacuous
            load_synthetic_shared(sd.control_net, rank, sdir, bias_scale=0.05)
            load_synthetic_shared(sd.hint_net, rank, sdir, bias_scale=0.05)
        if rank != 0 or args.no_cpu_baseline:
            unet_arrays = vae_arrays = None
        log(f"[rank {rank}] weights {'generated + packed' if rank == 0 or sdir is None else 'mapped from rank 0 (' + sdir + ')'} in {time.time() - t0:.1f}s")
        out_shape = (gb, size, size, 3)

        def local(c, u, z, *hint):
            """This rank's slice: prepare -> 25-step loop -> decode.  c / u / z (/ hint) arrive as device tensors (N > 1:
            views of the one broadcast buffer) or host arrays (N = 1)."""
            # (the engine is built once, by the first warm-up job, OUTSIDE the phase range: building it re-lays weights out for the wreg
            #  form - torch index / copy kernels, 3 per matrix - which a by-phase trace would otherwise book under `prepare` of every job)
            eng = sd._engine(b, c.shape[1], u.shape[1], nsteps, 7.5, 0.7, args.controlnet)
            with phase("prepare", args.sync_phases):   # uploads + context K/V, time-embedding tables (+ HintNet)
                eng.prepare(eng.contexts(u, c), z, sd.scheduler, None, 0, hint[0] if hint else None)
            with phase("denoise_loop", args.sync_phases):
                eng.run_steps(nsteps, None)
            with phase("vae_decode", args.sync_phases):
                return sd.image_decoder.decode_to_uint8(eng.latent)

    if world == 1 and dev.type == "cuda":
        # inputs resident in HBM when the timed region starts (with N > 1 they arrive by the broadcast, which is timed)
        ctx, unc, noise = (torch.from_numpy(a).to(dev) for a in (ctx, unc, noise))
        hints = tuple(torch.from_numpy(a).to(dev) for a in hints)
    if sd is not None:
        sd.scheduler.set_timesteps(nsteps)

    first = [True]
    first_job_s = [None]   # this rank's wall time from interpreter start to the end of its first (warm-up) job

    def one_job():
        if not first[0]:
            return sharded_job(local, ctx, unc, noise, dev, sync_phases=args.sync_phases, per_sample=hints)
        first[0] = False
        try:   # the first broadcast / all-gather: where an IPC or topology problem of RCCL shows
            res = sharded_job(local, ctx, unc, noise, dev, sync_phases=args.sync_phases, per_sample=hints)
            if dev.type == "cuda":
                torch.cuda.synchronize()
            first_job_s[0] = round(time.time() - T_PROCESS_START, 2)
            log(f"[rank {rank}] process start -> first job done: {first_job_s[0]:.1f} s")
            return res
        except Exception as e:
            log(f"[rank {rank}] first job FAILED: {type(e).__name__}: {e} (HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')})")
            raise

    elapsed, img, per_rank_ms = timed_jobs(one_job, args.steps, args.warmup, dev)   # (returns after a device synchronise: img has landed)
    if not stub:
        check_job_flags()
    assert tuple(img.shape) == out_shape and img.dtype == torch.uint8 and (rank != 0 or img.device.type == "cpu")

    images = gb * args.steps
    value = images / elapsed
    tflop_per_image = algorithmic_tflop_per_image(size, nsteps, args.controlnet)
    out = {
        "metric": (f"{size}x{size} images/sec (whole node), SD1.5 {nsteps}-step txt2img" + (" + ControlNet" if args.controlnet else "")),
        "value": round(value, 4), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1000.0 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"SD1.5 {size}x{size} {nsteps}-step txt2img, CFG 7.5 + rescale 0.7, batch {b}/GPU, "
                               f"UNet+VAE{'+ControlNet' if args.controlnet else ''} HIP path, random-init weights", "global_batch": gb,
                   "parallelism": f"batch-shard x{world}", "hipgraph": not args.no_graph,
                   # what a job does NOT redo: uploaded / computed when the schedule changes (DenoiseEngine.prepare), i.e. once for the run;
                   # everything else of the hot path (context K/V, the loop, decode, D2H) runs inside every timed job
                   "cached_across_jobs": ["sampler coefficient table", "time-embedding table (25 x timestep MLP + the 22 ResBlock projections)"],
                   # exact common-subexpression sharing inside a step: the part of the UNet in front of the first cross-attention
                   # is the same in the unconditional and the conditioned forward and is computed once per image (same bits as computing it twice)
                   "cfg_prefix_shared": os.environ.get("MSD_SHARE_CFG_PREFIX", "1") != "0",
                   # the one process-wide arithmetic choice read from the environment (minsdtf_amd/_lib.py); None = the default (9216)
                   "gn_rows": os.environ.get("MSD_GN_ROWS")},
        "n_ranks_seen": n_ranks_seen, "backend": args.backend + (" (RCCL)" if args.backend == "nccl" else ""),
        "launcher": os.environ.get("MSD_BENCH_LAUNCHER", "torchrun" if "TORCHELASTIC_RUN_ID" in os.environ else "external" if world > 1 else "none"),
        "rank_devices": rank_devices(dev, world),
        "per_rank_ms": per_rank_ms,   # each rank's own elapsed over the timed jobs (its last job drained), rank order
        "ipc_mode": {"HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"), "NCCL_DEBUG": os.environ.get("NCCL_DEBUG")},
        "start_to_first_job_s": gather_scalars(first_job_s[0], dev, world),   # per rank: interpreter start -> first job finished (weights, plans, capture)
    }
    if args.force_collectives:
        out["collectives_forced"] = True   # broadcast + all-gather ran through the process group at every job, also at world 1
    if stub:
        # a self-test of launcher + broadcast + slicing + gather: NOT a measurement, and it says so in every field a reader uses
        out.update(metric="STUB launcher self-test (no GPU work; not a measurement)", dtype="none", data="stub", stub=True,
                   config={"workload": "stub per-sample generator", "global_batch": gb, "parallelism": f"batch-shard x{world}"},
                   image_sha1=__import__("hashlib").sha1(img.numpy().tobytes()).hexdigest() if rank == 0 else None)
    else:
        out["config"]["cond_uncond"] = "two HIP streams" if sd._engine(b, 77, 77, nsteps, 7.5, 0.7, args.controlnet).dual else "one fused batch"
        out["tflops_per_gpu"] = round(tflop_per_image * b * args.steps / elapsed, 2)

    if rank == 0 and not stub:
        rank0_extras(out, args, sd, world, b, nsteps, size, ctx, unc, noise, hints, unet_arrays, vae_arrays)
    if rank == 0:
        if args.sync_phases and phase.wall_ms:   # (profiling runs: host wall time per phase, device drained at each end)
            log("phase wall ms per job: " + ", ".join(f"{k} {v[0] / v[1]:.3f}" for k, v in phase.wall_ms.items()))
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and not stub and "MSD_BENCH_SHARE_DIR" not in os.environ and world > 1:   # (torchrun: nobody else will)
        import shutil

        d = share_dir(world)
        if d:
            shutil.rmtree(d, ignore_errors=True)


def rank0_extras(out, args, sd, world, b, nsteps, size, ctx, unc, noise, hints, unet_arrays, vae_arrays):
    """Rank 0, outside the timed region: the two halves of a job timed separately, the roofline block of the dominant
    kernel family, parity against the committed oracle latent, and the CPU baseline."""
    import torch

    # the two halves of the job, timed separately (SURVEY.md §8d timing protocol): hipGraph replays on resident inputs
    eng = sd._engine(b, 77, 77, nsteps, 7.5, 0.7, args.controlnet)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    sd.scheduler.set_timesteps(nsteps)
    eng.prepare(eng.contexts(unc[:b], ctx[:b]), noise[:b], sd.scheduler, None, 0, hints[0][:b] if hints else None)
    torch.cuda.synchronize()
    ev[0].record()
    eng.run_steps(nsteps, None)
    ev[1].record()
    sd.image_decoder.decode_to_uint8(eng.latent)
    ev[2].record()
    torch.cuda.synchronize()
    out["ms_denoise_loop"] = round(ev[0].elapsed_time(ev[1]), 3)
    out["ms_vae_decode"] = round(ev[1].elapsed_time(ev[2]), 3)
    out["launches_per_step"] = len(eng.calls)   # C-ABI calls of one sampler step (split-K reductions / second GroupNorm launches come on top)
    if not args.no_roofline:
        out["roofline"], out["roofline_by_kernel"], extra = kernel_roofline(sd, b, nsteps, args.controlnet, size)
        ig = out["roofline"].get("in_graph")
        if ig is not None:
            # the trace's durations come from the PROFILED child pass, which runs a few per cent slower than the timed region: the
            # wall clock of a fused step of THIS process stands beside the trace's sum, so the two can be reconciled from the line
            ig["wall_us_per_fused_step"] = round(1e3 * out["ms_denoise_loop"] / nsteps, 1)
            tot = sum(ig["all_families_us_per_fused_step"].values())
            ig["trace_sum_over_wall"] = round(tot / ig["wall_us_per_fused_step"], 4)
            ig["note"] = ("all_families_us_per_fused_step sums kernel durations of the rocprofv3 child pass (profiler skew included, no launch "
                          "gaps); wall_us_per_fused_step = ms_denoise_loop / denoise steps of this process, unprofiled: not a budget of one another")
        out["eager_ms_per_unet_step_by_entry_point"] = extra  # event-per-launch pass (includes ~1-2 us of event gap per call)
    if world == 1 and not args.controlnet:
        out["psnr_db_vs_oracle_golden"] = golden_psnr(sd, size, nsteps)
    if world == 1 and not args.no_cpu_baseline:
        host = lambda a: a.cpu().numpy() if isinstance(a, torch.Tensor) else a   # noqa: E731
        out["cpu_baseline"] = cpu_baseline(unet_arrays, vae_arrays, host(ctx[:1]), host(unc[:1]), host(noise[:1]), nsteps)


class phase:
    """roctx range around one phase of a job (rocprofv3 --marker-trace); with `sync` the device is drained at the range
    end so that tools/trace_summary.py can attribute the kernel trace to phases by time (profiling runs only)."""

    wall_ms = {}   # with sync: accumulated host wall time per phase name, and the number of ranges (printed by main)

    def __init__(self, name, sync=False):
        self.name, self.sync = name, sync

    def __enter__(self):
        import torch

        if torch.cuda.is_available():
            torch.cuda.nvtx.range_push(self.name)
        self.t0 = time.perf_counter()

    def __exit__(self, *exc):
        import torch

        if torch.cuda.is_available():
            if self.sync:
                torch.cuda.synchronize()
                e = phase.wall_ms.setdefault(self.name, [0.0, 0])
                e[0] += (time.perf_counter() - self.t0) * 1e3
                e[1] += 1
            torch.cuda.nvtx.range_pop()


def sharded_job(local, ctx, unc, noise, dev, sync_phases=False, per_sample=()):
    """One whole job on this rank: ONE packed broadcast of the global contexts + noise (+ ControlNet hint images) (RCCL,
    device resident), the local generator on this rank's slice, all-gather of the uint8 images, and the copy of the
    gathered batch to host memory on rank 0 (the timed job ends with the uint8 images on the host)."""
    from minsdtf_amd import dist as mdist

    img = mdist.generate_sharded(local, ctx, unc, noise, dev, per_sample=per_sample)
    rank = mdist.rank()
    with phase("d2h", sync_phases):
        return to_host(img) if rank == 0 else img


_host_out = {"bufs": [None, None], "i": 0, "flags": None}


def to_host(img):
    """The gathered uint8 batch -> host memory, as a serving loop does it: an asynchronous copy on the job's stream into one
    of two alternating pinned buffers, so that the host can already queue the next job's uploads and launches while this
    job's last kernels and copy run.  Nothing is skipped: every job's copy is inside the timed region, and the clock stops
    only after a device synchronise (timed_jobs), i.e. after the last image has landed on the host."""
    import torch

    if img.device.type != "cuda":
        return img
    k = _host_out["i"] = _host_out["i"] ^ 1
    buf = _host_out["bufs"][k]
    if buf is None or buf.shape != img.shape:
        buf = _host_out["bufs"][k] = torch.empty(img.shape, dtype=img.dtype, pin_memory=True)
    buf.copy_(img, non_blocking=True)
    # the cluster GroupNorm's give-up words (one 4-byte word per launch plan) are folded into a device accumulator behind
    # every job and read ONCE, after the clock has stopped (check_job_flags): a job that ended with abandoned GroupNorm
    # moments fails the run instead of being counted
    from minsdtf_amd import engine

    flags = engine.gn_sync_flags(img.device)
    if flags is not None:
        acc = _host_out["flags"]
        if acc is None or acc.shape != flags.shape:
            acc = _host_out["flags"] = torch.zeros_like(flags)
        acc.bitwise_or_(flags)
    return buf


def check_job_flags():
    """After the timed region: raise if any job's cluster GroupNorm gave up (see to_host)."""
    from minsdtf_amd import engine

    if _host_out.get("flags") is not None:
        engine.check_gn_sync(_host_out["flags"].cpu(), group_wide=True)   # (every rank gets here once: the verdict is the group's)


def timed_jobs(one_job, steps, warmup, dev):
    """The bench contract's timing: `warmup` untimed jobs, then exactly `steps` jobs bracketed by a barrier + device
    synchronise on both sides; returns (MAX over ranks of the elapsed seconds, last job's result, every rank's own elapsed
    milliseconds in rank order: a straggler shows in the line)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size() if dist.is_initialized() else 1

    def barrier():
        if world > 1:
            dist.barrier()
        if dev.type == "cuda":
            torch.cuda.synchronize()

    img = None
    for _ in range(max(warmup, 1)):
        img = one_job()
    barrier()
    t_start = time.perf_counter()
    for _ in range(steps):
        img = one_job()
    own = time.perf_counter() - t_start    # this rank's own jobs (before the closing barrier makes everyone wait for the slowest)
    if dev.type == "cuda":
        torch.cuda.synchronize()
        own = time.perf_counter() - t_start
    barrier()
    elapsed = time.perf_counter() - t_start
    per_rank = [round(1e3 * own, 3)]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        mine = torch.tensor([own], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [round(1e3 * float(x.item()), 3) for x in every]
    return elapsed, img, per_rank


def algorithmic_tflop_per_image(size, nsteps, controlnet=False):
    """Algorithmic FLOP of the reference graph for one image (42.68 TFLOP at 512x512 x 25 steps, 220.6 at
    768x768 x 50).  Convolutions, dense layers and cross-attention grow with the image area a = (size/512)^2, the
    self-attention products (UNet: 0.1225 of 0.8033 TFLOP per forward; VAE: 0.0344 of 2.5145) with a^2."""
    a = (size / 512.0) ** 2
    unet = (UNET_TFLOP - UNET_SELF_ATTN_TFLOP) * a + UNET_SELF_ATTN_TFLOP * a * a
    vae = (VAE_TFLOP - VAE_ATTN_TFLOP) * a + VAE_ATTN_TFLOP * a * a
    t = 2 * nsteps * unet + vae
    if controlnet:   # ControlNet = the UNet's down + mid path per call (0.2686 TFLOP at 512^2, 0.0490 of it self-attention) + HintNet once
        t += 2 * nsteps * ((0.2686 - 0.0490) * a + 0.0490 * a * a) + 0.015 * a
    return t


def config_tag(b, size, control):
    """Suffix of the per-configuration files under profiles/ ("" = the headline configuration)."""
    return ("_controlnet" if control else "") + (f"_{size}" if size != 512 else "") + (f"_b{b}" if b != 1 else "")


def kernel_roofline(sd, b, nsteps, control=False, size=512):
    """Eager, event-timed pass over one denoise step: per-call durations on the launch stream."""
    import torch

    from minsdtf_amd import _lib

    eng = sd._engine(b, 77, 77, nsteps, 7.5, 0.7, control)
    calls = eng.calls
    st = torch.cuda.current_stream()
    reps = 3
    per_name = {}
    klass = {}   # kernel class -> time, algorithmic FLOP and algorithmic bytes of one denoise step

    def work(c):
        """(class, algorithmic FLOP, algorithmic HBM bytes) of one launch: every operand read once, the
        result written once; attention scores never touch HBM."""
        s = c.keep
        if isinstance(s, _lib.MsdConvGemm):
            M, cin = s.batch * s.h_out * s.w_out, s.c0 + s.c1
            K = s.ksize * s.ksize * cin
            n_out = s.N // 2 if s.act == _lib.ACT_GEGLU else s.N
            byt = 2 * (s.batch * s.h_in * s.w_in * cin + s.N * K) + (4 if s.out_dtype == _lib.OUT_F32 else 2) * M * n_out
            if s.residual:
                byt += 2 * M * n_out
            return ("conv3x3" if s.ksize == 3 else "dense / conv1x1"), 2.0 * M * s.N * K, byt
        if isinstance(s, _lib.MsdAttention):
            fl = 4.0 * s.batch * s.heads * s.s * s.t * s.head_dim
            byt = 2 * s.batch * s.heads * s.head_dim * (2 * s.s + 2 * s.t)
            return ("self-attention" if s.s == s.t else "cross-attention"), fl, byt
        if isinstance(s, _lib.MsdCrossAttnQ):   # attn2.to_q + attention over the text context in one launch
            C = s.heads * s.head_dim
            fl = 2.0 * s.batch * s.s * C * C + 4.0 * s.batch * s.heads * s.s * s.t * s.head_dim
            byt = 2 * (2 * s.batch * s.s * C + C * C + 2 * s.batch * s.t * C)
            return "cross-attention + to_q (fused)", fl, byt
        if isinstance(s, _lib.MsdGroupNorm):
            return "group_norm(+swish)", 0.0, 4 * s.batch * s.hw * (s.c0 + s.c1)
        fn = c.fn.__name__ if hasattr(c.fn, "__name__") else str(c.fn)
        if fn == "msd_layer_norm":
            return "layer_norm", 0.0, 4 * c.args[4] * c.args[5]
        return None, 0.0, 0

    for rep in range(reps + 1):
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(calls) + 1)]
        evs[0].record(st)
        for i, c in enumerate(calls):
            c(st.cuda_stream)
            evs[i + 1].record(st)
        torch.cuda.synchronize()
        if rep == 0:
            continue  # warm
        for i, c in enumerate(calls):
            fn = c.fn.__name__ if hasattr(c.fn, "__name__") else str(c.fn)
            ms = evs[i].elapsed_time(evs[i + 1])
            d = per_name.setdefault(fn, {"ms": 0.0, "n": 0, "flop": 0.0})
            d["ms"] += ms / reps
            d["n"] += 1.0 / reps
            if isinstance(c.keep, _lib.MsdConvGemm):
                s = c.keep
                M = s.batch * s.h_out * s.w_out
                K = s.ksize * s.ksize * (s.c0 + s.c1)
                d["flop"] += 2.0 * M * s.N * K / reps
            k, fl, byt = work(c)
            if k is not None:
                e = klass.setdefault(k, {"ms": 0.0, "n": 0.0, "flop": 0.0, "bytes": 0.0})
                e["ms"] += ms / reps
                e["n"] += 1.0 / reps
                e["flop"] += fl / reps
                e["bytes"] += byt / reps
    eng.step_ptr.zero_()
    dump = os.environ.get("MSD_DUMP_CALLS")
    if dump:  # per-call table of the last repetition (shape, duration, achieved TFLOP/s)
        with open(dump, "w") as f:
            for i, c in enumerate(calls):
                ms = evs[i].elapsed_time(evs[i + 1])
                line = f"{i:4d} {c.name:62s} {ms * 1e3:9.1f} us"
                if isinstance(c.keep, _lib.MsdConvGemm):
                    s = c.keep
                    M, K = s.batch * s.h_out * s.w_out, s.ksize * s.ksize * (s.c0 + s.c1)
                    line += f"  M={M:6d} N={s.N:6d} K={K:6d} splitk={s.splitk:2d} {2.0 * M * s.N * K / (ms * 1e-3) / 1e12:8.1f} TF/s"
                f.write(line + "\n")
    extra = {k: round(v["ms"], 3) for k, v in sorted(per_name.items(), key=lambda kv: -kv[1]["ms"])}
    # The roofline figure uses a second measurement without an event between every launch: all
    # conv_gemm calls of the step issued back to back (as they run inside the replayed graph, where
    # no event packets sit between kernels), bracketed by ONE pair of HIP events, 5 repetitions.
    conv_calls = [c for c in calls if isinstance(c.keep, _lib.MsdConvGemm)]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    clock = SclkSampler()   # the shader clock the chip holds while these launches run (sysfs; the peak below is priced at 2.4 GHz)
    for rep in range(1 + ROOFLINE_REPS):
        if rep == 1:
            e0.record(st)
            clock.start()
        for c in conv_calls:
            c(st.cuda_stream)
    e1.record(st)
    torch.cuda.synchronize()
    sclk = clock.stop()
    eng.step_ptr.zero_()
    g = per_name.get("msd_conv_gemm", {"ms": 1e-9, "n": 1, "flop": 0.0})
    n = max(g["n"], 1.0)
    g["ms"] = e0.elapsed_time(e1) / ROOFLINE_REPS
    avg_ms = g["ms"] / n
    flop_per_launch = g["flop"] / n
    achieved = flop_per_launch / (avg_ms * 1e-3) / 1e12
    # HBM-side bytes per launch come from the PMC pass committed under profiles/ (rocprofv3 --pmc cannot run
    # inside this process); null when the file is missing
    traffic, traffic_src, traffic_rw = None, None, None
    tag = config_tag(b, size, control)
    if LIVE_TRAFFIC and LIVE_TRAFFIC.get("conv_gemm"):
        cg = LIVE_TRAFFIC["conv_gemm"]
        traffic, traffic_rw = cg["hbm_bytes_per_launch"], [cg["read_bytes_per_launch"], cg["write_bytes_per_launch"]]
        traffic_src = (f"live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over tools/pmc_step.py, child processes of this run "
                       f"({cg['calls']} calls over 2 eager steps, {LIVE_TRAFFIC['seconds']} s)")
    for rnd in (range(9, 0, -1) if traffic is None else ()):   # newest committed PMC pass of THIS configuration (tools/measure_round.sh)
        name = f"r{rnd}_pmc_traffic{tag}.json"
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                pm = json.load(f)
            traffic = pm.get("conv_gemm", {}).get("hbm_bytes_per_launch")
            traffic_src = f"profiles/{name}" + (f" (commit {pm['commit']})" if pm.get("commit") else "")
            break
        except (OSError, ValueError):
            continue
    roof = {"bound": "mfma", "kernel": "conv_gemm_dma_kernel<*> / conv3x3_halo_kernel<*> / dense_rowpanel_kernel<*> / conv_wreg_kernel<*> / conv_big_kernel<*> / conv_bighalo_kernel<*> + splitk_finalize (implicit-GEMM conv3x3/1x1/dense, bf16 MFMA 16x16x32)",
            "achieved": round(achieved, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / MFMA_PEAK_TFLOPS, 4),
            "achieved_is": "back to back: the step's conv / dense calls alone under one HIP-event pair (no norm / attention launches between them)",
            "traffic": traffic, "traffic_source": traffic_src, "traffic_read_write": traffic_rw, "launches_per_unet_step": int(round(n)), "avg_launch_us": round(avg_ms * 1e3, 2),
            "gflop_per_launch": round(flop_per_launch / 1e9, 3)}
    if LIVE_INGRAPH and LIVE_INGRAPH.get("conv_gemm"):
        # the same family inside the replayed whole-loop graph (norm / attention launches between its kernels, operands as the
        # producers left them): kernel-trace durations of this run's own child pass, split-K reductions included
        fam = LIVE_INGRAPH["conv_gemm"]
        us_step = fam["us_per_step"]
        flop_step = g["flop"]   # algorithmic FLOP of one fused step's conv / dense calls
        roof["achieved_in_graph"] = round(flop_step / (us_step * 1e-6) / 1e12, 2)
        roof["frac_in_graph"] = round(roof["achieved_in_graph"] / MFMA_PEAK_TFLOPS, 4)
        roof["in_graph"] = {"family_us_per_fused_step": us_step, "dispatches_per_fused_step": fam["dispatches_per_step"],
                            "fused_steps_in_trace": LIVE_INGRAPH["_steps"], "tflop_per_fused_step": round(flop_step / 1e12, 4),
                            "source": f"live: rocprofv3 --kernel-trace over tools/pmc_step.py --graph-loops 1 (child process of this run, {LIVE_INGRAPH['seconds']} s)",
                            "all_families_us_per_fused_step": {k: v["us_per_step"] for k, v in LIVE_INGRAPH.items() if isinstance(v, dict) and "us_per_step" in v}}
    if sclk:
        # profiles/r4_pmc_mfma.json: the matrix-pipe counter and this FLOP-derived figure agree within 5 % once both are taken over the
        # kernels' own intervals at the clock actually held; `frac` stays priced at the 2.4 GHz spec peak
        roof["sclk_mhz"] = sclk
        roof["frac_at_sampled_clock"] = round(achieved / (MFMA_PEAK_TFLOPS * sclk["median"] / 2400.0), 4)
    # per kernel class of one denoise step (event-per-launch pass: each duration includes ~1-2 us of event gap):
    # MFMA utilisation of the contractions, achieved algorithmic HBM rate of everything
    by_kernel = {}
    for k, e in sorted(klass.items(), key=lambda kv: -kv[1]["ms"]):
        sec = e["ms"] * 1e-3
        row = {"launches": int(round(e["n"])), "ms": round(e["ms"], 3), "algorithmic_GBps": round(e["bytes"] / sec / 1e9, 1),
               "hbm_frac": round(e["bytes"] / sec / 1e9 / HBM_PEAK_GBS, 4)}
        if e["flop"]:
            row["TFLOPps"] = round(e["flop"] / sec / 1e12, 1)
            row["mfma_frac"] = round(e["flop"] / sec / 1e12 / MFMA_PEAK_TFLOPS, 4)
        by_kernel[k] = row
    return roof, by_kernel, extra


ROOFLINE_REPS = 20   # back-to-back repetitions of the step's conv / dense launch list under the one event pair (~60 ms: enough clock samples)


class SclkSampler:
    """Current shader-clock level of the GPUs of this box, read from sysfs (`pp_dpm_sclk`, the line marked '*') by a thread while
    a measurement runs.  Best effort: no file, no samples, no field in the line."""

    def __init__(self, period_s=0.0005):
        import glob
        import threading

        self.files = glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")
        self.period, self.samples, self._stop = period_s, [], False
        self.thread = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        while not self._stop:
            for fn in self.files:
                try:
                    with open(fn) as f:
                        for ln in f:
                            if "*" in ln:
                                self.samples.append(float(ln.split(":")[1].lower().replace("mhz", "").replace("*", "").strip()))
                except (OSError, ValueError, IndexError):
                    pass
            time.sleep(self.period)

    def start(self):
        if self.files:
            self.thread.start()

    def stop(self):
        self._stop = True
        if self.thread.is_alive():
            self.thread.join(timeout=1)
        v = sorted(x for x in self.samples if x > 1000.0)   # (a card of the box that idles reads its sleep level)
        if not v:
            return None
        return {"median": v[len(v) // 2], "min": v[0], "max": v[-1], "samples": len(v)}


def golden_psnr(sd, size, nsteps):
    """Final-latent PSNR against the committed fp32-oracle latent for this exact configuration."""
    path = os.path.join(ROOT, "tests", "golden", f"oracle_latent_{size}_{nsteps}.npz")
    if not os.path.exists(path):
        return None
    from oracle import sd_oracle as O

    g = np.load(path)
    rng = np.random.default_rng(int(g["context_seed"]))
    h = size // 8
    ctx = rng.standard_normal((1, 77, 768)).astype(np.float32)
    unc = rng.standard_normal((1, 77, 768)).astype(np.float32)
    noise = np.random.default_rng(int(g["noise_seed"])).standard_normal((1, h, h, 4)).astype(np.float32)
    got = sd.generate_image(ctx[0], negative_prompt=unc[0], batch_size=1, num_steps=nsteps,
                            unconditional_guidance_scale=float(g["guidance"]), diffusion_noise=noise[0],
                            guidance_rescale=float(g["guidance_rescale"]), return_latent=True)
    return round(O.psnr(got, g["latent"]), 2)


def cpu_baseline(unet_arrays, vae_arrays, ctx, unc, noise, nsteps):
    """Oracle ("port": fp32 CPU restatement of the reference path) on the host cores, bounded
    sample: ONE denoise step (uncond + cond UNet forward, B=1) + ONE VAE decode; a full image is
    nsteps such steps + the decode."""
    import torch

    from minsdtf_amd import weights as Wt
    from oracle import sd_oracle as O

    Wu = O.named_weights(Wt.table("civitai_model"), unet_arrays)
    Wv = O.named_weights(Wt.table("decoder"), vae_arrays)
    te = O.timestep_embedding(960, 1)
    t0 = time.perf_counter()
    u = O.unet_forward(Wu, noise, te, unc)
    c = O.unet_forward(Wu, noise, te, ctx)
    t_step = time.perf_counter() - t0
    t0 = time.perf_counter()
    O.decoder_forward(Wv, noise * 0.18215)
    t_dec = time.perf_counter() - t0
    per_image = nsteps * t_step + t_dec
    del u, c
    return {"value": round(1.0 / per_image, 6), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"1 of {nsteps} denoise steps (2 UNet forwards, B=1) = {t_step:.2f}s + 1 VAE decode = {t_dec:.2f}s, "
                      f"extrapolated to {nsteps} steps + decode = {per_image:.1f}s/image; torch fp32 CPU, "
                      f"{torch.get_num_threads()} threads = the CPUs the container is granted (os.cpu_count()={os.cpu_count()})"}


if __name__ == "__main__":
    main()
