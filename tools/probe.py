"""GPU toolchain probe: run on the MI355X box via gpurun (see tools/probe.hip)."""
import ctypes, os, sys, subprocess
import torch

here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libprobe.so")
if not os.path.exists(so):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC",
                           os.path.join(here, "probe.hip"), "-o", so])
lib = ctypes.CDLL(so)
print("device:", torch.cuda.get_device_name(0))
p = torch.cuda.get_device_properties(0)
print("CUs:", p.multi_processor_count, "mem GB:", p.total_memory / 2**30)
st = torch.cuda.current_stream().cuda_stream
vp = ctypes.c_void_p

x = torch.arange(1000, device="cuda", dtype=torch.float32)
y = torch.ones(1000, device="cuda")
rc = lib.probe_axpy(vp(x.data_ptr()), vp(y.data_ptr()), ctypes.c_float(2.0), 1000, vp(st))
torch.cuda.synchronize()
print("axpy rc", rc, "ok", bool(torch.allclose(y, 2 * x + 1)))

# same launch captured in a torch CUDA graph on a side stream
g = torch.cuda.CUDAGraph()
y2 = torch.ones(1000, device="cuda")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    with torch.cuda.graph(g, stream=s):
        lib.probe_axpy(vp(x.data_ptr()), vp(y2.data_ptr()), ctypes.c_float(2.0), 1000,
                       vp(torch.cuda.current_stream().cuda_stream))
g.replay(); g.replay()
torch.cuda.synchronize()
print("graph ok", bool(torch.allclose(y2, 4 * x + 1)))

torch.manual_seed(0)
A = torch.randint(-4, 5, (16, 32), device="cuda").float()
B = torch.randint(-4, 5, (32, 16), device="cuda").float()  # asymmetric
Ab = A.bfloat16().contiguous()
Btb = B.t().contiguous().bfloat16()
C = torch.zeros(16, 16, device="cuda")
rc = lib.probe_mfma(vp(Ab.data_ptr()), vp(Btb.data_ptr()), vp(C.data_ptr()), vp(st))
torch.cuda.synchronize()
print("mfma rc", rc, "layout ok", bool(torch.equal(C, A @ B)))

v = torch.randn(64, device="cuda")
o1 = torch.empty(64, device="cuda"); o2 = torch.empty(64, device="cuda"); o3 = torch.empty(64, device="cuda")
rc = lib.probe_xlane(vp(v.data_ptr()), vp(o1.data_ptr()), vp(o2.data_ptr()), vp(o3.data_ptr()), vp(st))
torch.cuda.synchronize()
exp1 = v.view(4, 16).max(dim=1, keepdim=True).values.expand(4, 16).reshape(64)
idx = torch.arange(64, device="cuda")
print("row_ror max ok", bool(torch.equal(o1, exp1)),
      "xor16 ok", bool(torch.equal(o2, v[idx ^ 16])), "xor32 ok", bool(torch.equal(o3, v[idx ^ 32])))
import ctypes.util
with open("/proc/self/maps") as f:
    libs = sorted({ln.split()[-1] for ln in f if "amdhip" in ln or "hsa-runtime" in ln})
print("hip runtimes mapped:", libs)
