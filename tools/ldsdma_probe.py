"""Driver of tools/ldsdma_probe.hip: per-CU L2 -> LDS staging rate by loader form (see the .hip header)."""
import ctypes as C
import os
import subprocess

import numpy as np
import torch

here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "_build", "libldsdma_probe.so")
if not os.path.exists(so):
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(here, "ldsdma_probe.hip"), "-o", so])
lib = C.CDLL(so)
lib.ldsdma_probe.restype = C.c_int
lib.ldsdma_probe.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
src = torch.randint(0, 255, (64 << 20,), dtype=torch.uint8, device="cuda")
out = torch.zeros(4 * 1024, dtype=torch.int64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
names = {0: "glds+m0 save/restore", 1: "glds m0 write only", 2: "global_load -> ds_write", 3: "buffer_load lds"}
tiles = 480
print("mode                      thr pieces depth stride grid |  us/loop  us/tile  KB/tile  GB/s per CU  chip TB/s | first-tile issue us")
for grid in (256, 512):
    for row_stride in (2560, 640):
        for threads in (256, 512):
            for pieces, depth in ((4, 3), (4, 6), (2, 4)):
                if grid == 512 and depth * (threads // 64) * pieces * 1024 > 80 * 1024:
                    continue
                for mode in (0, 1, 2, 3):
                    for rep in range(2):
                        rc = lib.ldsdma_probe(src.data_ptr(), src.numel(), row_stride, tiles, mode, pieces, depth, threads, grid, out.data_ptr(), st)
                        assert rc == 0, rc
                        torch.cuda.synchronize()
                    t = out[: 4 * grid].cpu().numpy().reshape(grid, 4)
                    us = np.median(t[:, 0]) / 100.0
                    kb = (threads // 64) * pieces
                    per_wg = kb * 1024 * (tiles + depth) / (us * 1e-6) / 1e9
                    per_cu = per_wg * (grid // 256)
                    print(f"{names[mode]:25s} {threads:3d} {pieces:6d} {depth:5d} {row_stride:6d} {grid:4d} | {us:8.2f} {us / tiles:8.3f} {kb:8d} {per_cu:12.1f} {per_cu * 256 / 1e3:10.2f} | {np.median(t[:, 1]) / 100.0:6.2f}",
                          flush=True)
