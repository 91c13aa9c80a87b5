"""Driver of tools/launch_floor.hip: per-launch time of empty and of dependent-chain kernels inside a replayed hipGraph."""
import ctypes as C
import os
import subprocess

import torch

here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "_build", "liblaunch_floor.so")
if not os.path.exists(so):
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(here, "launch_floor.hip"), "-o", so])
lib = C.CDLL(so)
lib.launch_empty.argtypes = [C.c_int, C.c_void_p]
lib.launch_chain.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
N = 400


def graph_us(fn):
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        s = torch.cuda.current_stream().cuda_stream
        for i in range(N):
            assert fn(i, s) == 0
    g.replay()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / N)
    return best


torch.zeros(1, device="cuda")
for grid in (1, 256, 1024):
    print(f"empty kernel, {grid:5d} workgroups: {graph_us(lambda i, s: lib.launch_empty(grid, s)):6.2f} us per launch", flush=True)
for kb in (64, 1024, 4096, 8192):
    n16 = kb * 1024 // 16
    a = torch.zeros(n16 * 4, dtype=torch.int32, device="cuda")
    b = torch.zeros_like(a)
    for grid in (256, 1024):
        if grid * 256 > n16 * 4:
            continue
        for shift in (0, 1, 8):
            us = graph_us(lambda i, s: lib.launch_chain((a if i % 2 == 0 else b).data_ptr(), (b if i % 2 == 0 else a).data_ptr(), n16, grid, shift, s))
            print(f"dependent chain, {kb:5d} KB read + written per launch, {grid:5d} workgroups, reads workgroup + {shift}'s stores: {us:6.2f} us per launch ({2 * kb * 1024 / us / 1e6:6.2f} TB/s)", flush=True)
