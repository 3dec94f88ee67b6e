"""fots_e2e.hostcpus: the CPU count the host side sizes its thread pools by."""
import builtins
import io
import os

import pytest

from fots_e2e import hostcpus


def _fake_open(files):
    real = builtins.open

    def opener(path, *a, **k):
        if path in files:
            if files[path] is None:
                raise FileNotFoundError(path)
            return io.StringIO(files[path])
        if str(path).startswith("/sys/fs/cgroup"):
            raise FileNotFoundError(path)
        return real(path, *a, **k)
    return opener


@pytest.mark.parametrize("text,expect", [("1600000 100000\n", 16), ("max 100000\n", None), ("150000 100000\n", 1),
                                         ("50000 100000\n", 1)])
def test_cgroup_v2_quota(monkeypatch, text, expect):
    monkeypatch.setattr(builtins, "open", _fake_open({"/sys/fs/cgroup/cpu.max": text}))
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(256)))
    assert hostcpus.effective_cpus() == (256 if expect is None else expect)


def test_cgroup_v1_quota_and_affinity(monkeypatch):
    files = {"/sys/fs/cgroup/cpu.max": None, "/sys/fs/cgroup/cpu/cpu.cfs_quota_us": "800000\n",
             "/sys/fs/cgroup/cpu/cpu.cfs_period_us": "100000\n"}
    monkeypatch.setattr(builtins, "open", _fake_open(files))
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(64)))
    assert hostcpus.effective_cpus() == 8
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: {0, 1, 2})   # affinity tighter than the quota
    assert hostcpus.effective_cpus() == 3
    files["/sys/fs/cgroup/cpu/cpu.cfs_quota_us"] = "-1\n"                 # v1 spelling of "no quota"
    assert hostcpus.effective_cpus() == 3


def test_cap_never_raises_the_pool():
    import torch
    before = torch.get_num_threads()
    try:
        torch.set_num_threads(1)
        assert hostcpus.cap_torch_threads() == 1
    finally:
        torch.set_num_threads(before)
    assert 1 <= hostcpus.cap_torch_threads() <= max(1, hostcpus.effective_cpus())
