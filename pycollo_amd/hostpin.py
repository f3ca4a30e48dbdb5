"""Keep the thread that launches kernels (and the HIP runtime's helper threads) on one CPU core.

An evaluation at BASELINE.json's headline size is two ~3 us kernel launches for ~7.5 us of GPU work, so the
host side is on the critical path.  Left to the scheduler on the 256-CPU host of an MI355X box, the launching
thread and the runtime's helper threads migrate between cores and the launch cost drifts between ~3.2 and
~5.8 us per launch from one process to the next (measured: 90k / 121k evals/s unpinned against 125-128k on any
single core, tools/ab_affinity.sh).  Pinning must happen before the first HIP call so that the runtime's own
threads inherit the mask.
"""
from __future__ import annotations

import ctypes
import os


def _current_cpu() -> int:
    try:
        return int(ctypes.CDLL(None).sched_getcpu())
    except (OSError, AttributeError):
        return -1


def pin_launch_thread(local_rank: int = 0, world: int = 1) -> tuple[int, set[int]]:
    """Pin this process to one allowed CPU; returns (cpu, previous affinity mask).

    One rank: the core the thread is running on.  Several ranks on a node: a slice of up to 8 cores each, spread
    evenly over the first half of the allowed list (the physical cores on an SMT host) in rank order, which
    follows the usual GPU-to-NUMA-node order."""
    if not hasattr(os, "sched_setaffinity"):
        return -1, set()
    allowed = sorted(os.sched_getaffinity(0))
    if world <= 1:
        cpu = _current_cpu()
        if cpu not in allowed:
            cpu = allowed[0]
        width = max(1, int(os.environ.get("PYCOLLO_AMD_PIN_WIDTH", "1")))   # experiment knob: cores in the slice
        i = allowed.index(cpu)
        os.sched_setaffinity(0, set(allowed[i:i + width]) or {cpu})
        return cpu, set(allowed)
    # several ranks: each gets its own slice of cores rather than one core -- the collective library's helper
    # threads inherit the mask too and must not queue behind the launching thread
    stride = max(1, len(allowed) // (2 * world))
    lo = (local_rank * stride) % len(allowed)
    mine = set(allowed[lo:lo + min(stride, 8)])
    os.sched_setaffinity(0, mine)
    return allowed[lo], set(allowed)


def restore_affinity(mask: set[int]) -> None:
    """Give the calling thread its full mask back (before CPU-parallel work such as an OpenMP region)."""
    if mask and hasattr(os, "sched_setaffinity"):
        os.sched_setaffinity(0, mask)
