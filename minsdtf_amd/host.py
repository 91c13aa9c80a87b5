"""Host-side facts the pipeline's CPU work should respect (no GPU, no torch import at module load)."""
import os


def effective_cpus() -> int:
    """CPUs this process may actually use: the minimum of os.cpu_count(), the scheduler affinity mask and the cgroup CPU
    quota (v2 `cpu.max`, v1 `cpu.cfs_quota_us / cpu.cfs_period_us`).  A container on a 256-thread host with a 16-CPU quota
    reports 256 to os.cpu_count(): thread pools sized by that number (torch's intra-op pool: 128) then run 8 threads per
    granted CPU - measured on the MI355X boxes of this pool: an fp32 UNet forward on the CPU 4.2 s with 128 threads, 0.9 s with 16."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:          # cgroup v2: "<quota> <period>" or "max <period>"
            q, p = f.read().split()[:2]
            if q != "max":
                quota = int(q) / int(p)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                p = int(f.read())
            if q > 0 and p > 0:
                quota = q / p
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, int(quota + 0.5)))
    return max(1, n)


def fit_torch_threads() -> int:
    """Lower torch's intra-op thread count to effective_cpus() (never raises it); returns the count in force."""
    import torch

    n = effective_cpus()
    if torch.get_num_threads() > n:
        torch.set_num_threads(n)
    return torch.get_num_threads()
