"""How many host CPUs this process may actually burn.

`os.cpu_count()` (what OpenMP and torch's intra-op pool size themselves by) reports the machine's
hardware threads; inside a container the scheduler affinity and the cgroup CPU quota are what
count. On the MI355X pool's boxes that is 256 hardware threads against a 16-CPU quota: 256 spinning
OpenMP workers exhaust the 100 ms quota period in a few milliseconds and the whole process --
including the thread that waits for the GPU -- is descheduled for the rest of it. Seen from the
pipeline that is a 70-80 ms stall on every third image or so, at any stage (`tools/e2e_probe.py`;
with the pool capped to the quota there is none: `cpu.stat: nr_throttled 0`).
"""
import math
import os


def _cgroup_quota_cpus():
    """CPUs the cgroup's CFS quota allows (v2 `cpu.max`, v1 `cpu.cfs_quota_us`), or None."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            return int(quota) / int(period)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            quota = int(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            period = int(f.read())
        if quota > 0 and period > 0:
            return quota / period
    except (OSError, ValueError):
        pass
    return None


def effective_cpus() -> int:
    """min(scheduler affinity, cgroup quota rounded down), at least 1."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover - non-Linux
        n = os.cpu_count() or 1
    q = _cgroup_quota_cpus()
    if q is not None:
        n = min(n, max(1, int(math.floor(q))))
    return max(1, n)


def cap_torch_threads(reserve: int = 0) -> int:
    """Cap torch's intra-op pool to the CPUs this process may use; returns the pool size in effect.

    `reserve` CPUs are left for the threads that are not in the pool (the HIP runtime's, the
    caller's own). Never raises the pool above what the caller already set."""
    import torch
    n = max(1, effective_cpus() - reserve)
    if torch.get_num_threads() > n:
        torch.set_num_threads(n)
    return torch.get_num_threads()
