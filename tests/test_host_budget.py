"""Host-side resource rules of round 5 (no GPU): the CPU budget a rank sizes its thread pools by, and the process-wide stream table."""
import os

import torch

from vitcap_amd import dist_util as D


def test_host_cpu_budget_is_bounded_by_what_the_process_may_use():
    n = D.host_cpu_budget()
    assert 1 <= n <= (os.cpu_count() or 1)
    try:
        assert n <= len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    try:      # a cgroup-v2 bandwidth limit, where one is set, bounds it too
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max':
            assert n <= max(1, int(int(q) / int(per)))
    except (OSError, ValueError):
        pass


def test_cap_host_threads_only_lowers(monkeypatch):
    before = torch.get_num_threads()
    try:
        monkeypatch.delenv('LOCAL_WORLD_SIZE', raising=False)
        monkeypatch.delenv('WORLD_SIZE', raising=False)
        torch.set_num_threads(1)
        assert D.cap_host_threads(reserved=0) == 1                      # never raised
        torch.set_num_threads(max(1, before))
        got = D.cap_host_threads(reserved=10 ** 6)                      # more reserved than there is: one thread is left
        assert got == 1 and torch.get_num_threads() == 1
        torch.set_num_threads(max(1, before))
        monkeypatch.setenv('LOCAL_WORLD_SIZE', '8')                     # eight ranks on the host share the budget
        assert D.cap_host_threads(reserved=0) <= max(1, D.host_cpu_budget() // 8)
    finally:
        torch.set_num_threads(max(1, before))


def test_role_streams_are_created_once_per_device_and_role():
    from vitcap_amd.model import role_stream
    made = []

    def make():
        made.append(object())
        return made[-1]
    a = role_stream('cuda:3', 'test-role-a', make)
    assert role_stream(torch.device('cuda', 3), 'test-role-a', make) is a and len(made) == 1
    b = role_stream('cuda:3', 'test-role-b', make)
    c = role_stream('cuda:4', 'test-role-a', make)
    assert b is not a and c is not a and len(made) == 3
