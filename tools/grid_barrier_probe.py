"""Driver of tools/grid_barrier_probe.hip: what one dependent phase costs INSIDE a persistent launch (flat and
XCD-hierarchical grid barrier) against the same phase as its own launch inside a replayed hipGraph.  Every result is
checked word by word (a barrier that lets a workgroup read its neighbour's block early shows as a wrong count).

    python tools/grid_barrier_probe.py            # on the GPU box; writes nothing, prints the table
"""
import ctypes as C
import os
import subprocess

import torch

here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "_build", "libgrid_barrier_probe.so")
if not os.path.exists(so):
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(here, "grid_barrier_probe.hip"), "-o", so])
lib = C.CDLL(so)
lib.launch_phases.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
lib.launch_one_phase.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
PH = 200
GRID = torch.cuda.get_device_properties(0).multi_processor_count   # one workgroup per CU


def timed(fn, reps=5):
    best = 1e30
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3)
    return best


def persistent(mode, n16, shift, grid=GRID):
    a = torch.zeros(grid * n16 * 4, dtype=torch.int32, device="cuda")
    b = torch.zeros_like(a)
    sync = torch.zeros(lib.sync_bytes() // 4 + 64, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream

    def run():
        a.zero_(); b.zero_(); sync.zero_()
        assert lib.launch_phases(a.data_ptr(), b.data_ptr(), n16, PH, shift, grid, mode, sync.data_ptr(), st) == 0

    run()
    torch.cuda.synchronize()
    ok = True
    if mode != 2:
        res = a if PH % 2 == 0 else b
        ok = bool((res == PH).all()) and int(sync[lib.sync_bytes() // 4 - 1].item()) == 0   # (last word of Sync: err)
    # time the launch alone (the three memsets are outside the event pair)
    best = 1e30
    for _ in range(5):
        a.zero_(); b.zero_(); sync.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lib.launch_phases(a.data_ptr(), b.data_ptr(), n16, PH, shift, grid, mode, sync.data_ptr(), st)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3)
    return best / PH, ok


def launches(n16, shift, grid=GRID):
    a = torch.zeros(grid * n16 * 4, dtype=torch.int32, device="cuda")
    b = torch.zeros_like(a)
    g = torch.cuda.CUDAGraph()
    lib.launch_one_phase(a.data_ptr(), b.data_ptr(), n16, shift, grid, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    a.zero_(); b.zero_()
    with torch.cuda.graph(g):
        s = torch.cuda.current_stream().cuda_stream
        for p in range(PH):
            src, dst = (b, a) if p & 1 else (a, b)
            assert lib.launch_one_phase(src.data_ptr(), dst.data_ptr(), n16, shift, grid, s) == 0
    g.replay()
    torch.cuda.synchronize()
    ok = bool(((a if PH % 2 == 0 else b) == PH).all())
    return timed(g.replay) / PH, ok


print(f"device: {torch.cuda.get_device_name(0)}, {GRID} CUs -> {GRID} workgroups of 256 threads, {PH} dependent phases", flush=True)
print("| bytes read + written per phase (all workgroups) | reads block of | own launch per phase (graph) | in-launch, flat barrier | in-launch, XCD-hierarchical barrier | in-launch, no barrier (body alone) |")
print("|---|---|---|---|---|---|", flush=True)
for kb in (4, 64, 1024, 8192):
    n16 = kb * 1024 // 16 // GRID
    if n16 < 1:
        n16 = 1
    for shift in (1, 8):
        lu, lok = launches(n16, shift)
        f, fok = persistent(0, n16, shift)
        x, xok = persistent(1, n16, shift)
        nb, _ = persistent(2, n16, shift)
        print(f"| {n16 * 16 * GRID // 1024} KB | workgroup + {shift} | {lu:.2f} us{'' if lok else ' WRONG'} | {f:.2f} us{'' if fok else ' WRONG'} | "
              f"{x:.2f} us{'' if xok else ' WRONG'} | {nb:.2f} us |", flush=True)
