"""pycollo_amd.hostpin: pin the launching thread before HIP initialises, give the mask back afterwards."""
import os

import pytest

from pycollo_amd.hostpin import (_l3_peers, _parse_cpulist, colocate_runtime_threads, pin_launch_thread,
                                 restore_affinity, tune_launch_core)

pytestmark = pytest.mark.skipif(not hasattr(os, "sched_setaffinity"), reason="no affinity control on this platform")


def test_pin_one_rank_and_restore():
    full = os.sched_getaffinity(0)
    try:
        cpu, prev = pin_launch_thread()
        assert prev == full and cpu in full
        assert os.sched_getaffinity(0) == {cpu}
    finally:
        restore_affinity(full)
    assert os.sched_getaffinity(0) == full


def test_pin_several_ranks_get_disjoint_slices():
    full = os.sched_getaffinity(0)
    world = 2 if len(full) >= 4 else 1
    try:
        slices = []
        for rank in range(world):
            restore_affinity(full)
            first, _ = pin_launch_thread(rank, world)
            mine = os.sched_getaffinity(0)
            assert first in mine and mine <= full
            slices.append(mine)
        if world > 1:
            assert not (slices[0] & slices[1])
    finally:
        restore_affinity(full)


def test_cpulist_parser():
    assert _parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    assert _parse_cpulist("") == set()


def test_busy_helper_thread_is_moved_next_to_the_launcher():
    """A thread that burns CPU during the burst (the stand-in for the runtime's completion thread) is confined to
    the launching core's L3 peers; idle threads and the caller are left alone."""
    import threading
    import time
    full = os.sched_getaffinity(0)
    cpu = sorted(full)[0]
    peers = (_l3_peers(cpu) - {cpu}) & full
    if not peers:
        pytest.skip("no L3 peers visible for this core")
    stop = threading.Event()
    idle_go = threading.Event()
    busy = threading.Thread(target=lambda: [None for _ in iter(stop.is_set, True)])
    idle = threading.Thread(target=idle_go.wait)
    busy.start(); idle.start()
    try:
        os.sched_setaffinity(0, {cpu})
        moved = colocate_runtime_threads(lambda: time.sleep(0.1))
        assert busy.native_id in moved and idle.native_id not in moved
        assert os.sched_getaffinity(busy.native_id) <= _l3_peers(cpu) - {cpu}
        assert os.sched_getaffinity(idle.native_id) == full
        assert os.sched_getaffinity(0) == {cpu}
    finally:
        stop.set(); idle_go.set(); busy.join(); idle.join()
        restore_affinity(full)


def test_launch_core_is_chosen_among_the_probed_ones():
    import time
    full = os.sched_getaffinity(0)
    try:
        cpu, prev = pin_launch_thread()
        best, timings = tune_launch_core(lambda: time.sleep(0.005), prev, tries=4)
        assert best in prev and best in [c for c, _ in timings]
        assert timings[0][0] == cpu                      # the current core is always a candidate
        assert os.sched_getaffinity(0) == {best}
        # candidates come from different last-level-cache domains
        doms = [frozenset(_l3_peers(c) or {c}) for c, _ in timings]
        assert len(set(doms)) == len(doms)
    finally:
        restore_affinity(full)
