"""The RCCL path on the one GPU a test box has (reference batch tiling: stable_diffusion.py:384-397).

No 8-GPU node is available to the builder, and at world size 1 the sharding helpers normally short-circuit — so without
this test librccl would never even be loaded before a driver SCALE run.  A fresh child process creates a ONE-rank process
group with backend `nccl` (= RCCL on ROCm) before anything else touches the GPU and runs the real collectives: the packed
device-resident fp32 broadcast, all_gather_into_tensor of uint8 [b, 512, 512, 3], barrier, destroy, and the pipeline itself
with its exchanges forced through them (bit-identical images)."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_rccl_world1_collectives_and_pipeline(gpu):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    p = subprocess.run([sys.executable, os.path.join(HERE, "_collectives_world1_child.py"), "nccl", "--pipeline"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=570)
    assert p.returncode == 0, f"child failed ({p.returncode}):\n{p.stdout[-2000:]}\n{p.stderr[-4000:]}"
    ok = [ln for ln in p.stdout.splitlines() if ln.startswith("OK ")]
    assert ok, p.stdout[-2000:]
    info = json.loads(ok[-1][3:])
    print("RCCL world-1:", info)
    assert info["backend"] == "nccl" and info["world"] == 1 and info["device"] == "cuda:0" and info["pipeline"] == "bit-identical"
