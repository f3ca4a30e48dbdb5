"""pycollo_amd.hostpin: pin the launching thread before HIP initialises, give the mask back afterwards."""
import os

import pytest

from pycollo_amd.hostpin import pin_launch_thread, restore_affinity

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
