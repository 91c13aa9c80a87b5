"""Cycles per instruction of the bf16 MFMA shapes and of v_exp_f32 beside them (tools/mfma_rate.hip): run through gpurun."""
import ctypes
import os
import subprocess

import torch

here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "_build", "libmfma_rate.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(here, "mfma_rate.hip")):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-mllvm", "-amdgpu-mfma-vgpr-form=1",
                           os.path.join(here, "mfma_rate.hip"), "-o", so])
lib = ctypes.CDLL(so)
st = torch.cuda.current_stream().cuda_stream
names = {0: ("16x16x32 bf16", 32), 1: ("16x16x16 bf16 (_1k)", 32), 2: ("32x32x16 bf16", 32), 3: ("32x32x8 bf16 (_1k)", 32),
         4: ("v_exp_f32", 32), 5: ("16x16x32 + 1 v_exp", 32), 6: ("16x16x32 + 2 v_exp", 32), 7: ("32x32x16 + 4 v_exp", 16), 8: ("v_fma_f32", 32)}
iters = 2000
for blocks in (1, 256):
    for mode, (name, per_iter) in names.items():
        cyc = torch.zeros(blocks * 4, dtype=torch.int64, device="cuda")
        sink = torch.zeros(4, device="cuda")
        for _ in range(2):
            rc = lib.mfma_rate(mode, ctypes.c_void_p(cyc.data_ptr()), ctypes.c_void_p(sink.data_ptr()), iters, blocks, ctypes.c_void_p(st))
            assert rc == 0, rc
            torch.cuda.synchronize()
        c = cyc.float().median().item() / (iters * per_iter)
        print(f"blocks {blocks:4d}  {name:24s} {c:7.2f} s_memtime ticks per group-instruction (one wave per SIMD)")
