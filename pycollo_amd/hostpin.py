"""Keep the thread that launches kernels (and the HIP runtime's helper threads) on one CPU core.

An evaluation at BASELINE.json's headline size is two ~3 us kernel launches for ~7.5 us of GPU work, so the
host side is on the critical path.  Left to the scheduler on the 256-CPU host of an MI355X box, the launching
thread and the runtime's helper threads migrate between cores and the launch cost drifts between ~3.2 and
~5.8 us per launch from one process to the next (measured: 90k / 121k evals/s unpinned against 125-128k on any
single core, tools/ab_affinity.sh).  Pinning happens before the first HIP call so that the threads the runtime
creates inherit the mask -- except the one that matters most: the runtime's completion thread gets its affinity
reset to every CPU by the runtime itself, and ``colocate_runtime_threads`` moves it next to the launcher afterwards.
"""
from __future__ import annotations

import ctypes
import os


def _current_cpu() -> int:
    try:
        return int(ctypes.CDLL(None).sched_getcpu())
    except (OSError, AttributeError):
        return -1


def spin_seconds(reps: int = 3, n: int = 100_000) -> float:
    """Best-of-``reps`` time of a fixed interpreter loop on the current core: a busy SMT sibling or a clocked-down
    core shows up here before any HIP call is made."""
    import time
    best = float("inf")
    for _ in range(reps):
        t0 = time.perf_counter()
        acc = 0
        for i in range(n):
            acc += i
        best = min(best, time.perf_counter() - t0)
    return best


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
        forced = os.environ.get("PYCOLLO_AMD_PIN_CPU", "")   # experiment knob: start on this core
        if forced.isdigit() and int(forced) in allowed:
            cpu = int(forced)
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


def _parse_cpulist(text: str) -> set[int]:
    out: set[int] = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.update(range(int(lo), int(hi or lo) + 1))
    return out


def _l3_peers(cpu: int) -> set[int]:
    """CPUs that share the last-level cache with ``cpu`` (one CCD on an EPYC host); empty if sysfs does not say."""
    for idx in (3, 2):
        try:
            with open(f"/sys/devices/system/cpu/cpu{cpu}/cache/index{idx}/shared_cpu_list") as f:
                return _parse_cpulist(f.read())
        except (OSError, ValueError):
            continue
    return set()


def _thread_run_ns() -> dict[int, int]:
    """On-CPU time of every thread of this process, in ns (schedstat) or 10 ms ticks scaled to ns (stat)."""
    out = {}
    try:
        tids = os.listdir("/proc/self/task")
    except OSError:
        return out
    for t in tids:
        try:
            with open(f"/proc/self/task/{t}/schedstat") as f:
                out[int(t)] = int(f.read().split()[0])
        except (OSError, ValueError, IndexError):
            try:
                with open(f"/proc/self/task/{t}/stat") as f:
                    st = f.read()
                fld = st[st.rindex(")") + 2:].split()
                out[int(t)] = (int(fld[11]) + int(fld[12])) * 10_000_000
            except (OSError, ValueError, IndexError):
                pass
    return out


def colocate_runtime_threads(burst, min_share: float = 0.03) -> list[int]:
    """Move the HIP runtime's busy helper threads next to the launching thread; returns the thread ids moved.

    The ROCm runtime keeps one thread that wakes on every kernel completion (it runs ~30 % of a core at 250 k
    launches/s) and resets that thread's affinity to every CPU, so pinning the process before HIP initialises does
    not place it.  Where the scheduler leaves it decides the launch cost: measured on the 2-socket MI355X host
    (tools/core_sweep.py, tools/thread_probe.py) the headline loop runs at 128-131 k evals/s with the helper in the
    launcher's L3 domain, 110-123 k on the same socket and 87-99 k across sockets -- the "slow mode" one process
    in three used to land in.  ``burst`` is a callable that launches for a few tens of ms and synchronises; threads
    other than the caller that are on a CPU for more than ``min_share`` of it are the ones moved."""
    if not hasattr(os, "sched_setaffinity"):
        return []
    import threading
    import time
    me = threading.get_native_id()
    mask = os.sched_getaffinity(0)
    if len(mask) == 1:
        cpu = next(iter(mask))
        target = _l3_peers(cpu) - {cpu}
    else:
        target = set(mask)           # a rank's slice of cores: the helper shares it
    if not target:
        return []
    before = _thread_run_ns()
    t0 = time.perf_counter()
    burst()
    wall_ns = (time.perf_counter() - t0) * 1e9
    after = _thread_run_ns()
    moved = []
    for tid, ns in after.items():
        if tid == me or ns - before.get(tid, ns) < min_share * wall_ns:
            continue
        try:
            os.sched_setaffinity(tid, target)
            moved.append(tid)
        except OSError:              # outside the cgroup's cpuset, or the thread has gone
            pass
    return moved


def tune_launch_core(burst, allowed: set[int], tries: int = 4) -> tuple[int, list[tuple[int, float]]]:
    """Pick the launching core by measurement; returns (cpu chosen, [(cpu, seconds per burst), ...]).

    Not every core launches equally fast on the shared 256-CPU host: two of sixteen processes that happened to start
    on one particular CCD ran the headline loop at 87-93 k evals/s against 140-146 k everywhere else
    (tools/core_probe.sh), with the runtime's completion thread correctly placed next to them.  ``burst`` (a few
    thousand launches + a synchronise) is timed on the current core and on ``tries - 1`` cores spread over the other
    L3 domains of ``allowed``, each time with the runtime's busy threads moved alongside; the fastest wins."""
    if not hasattr(os, "sched_setaffinity") or not allowed:
        return -1, []
    import time
    mask = os.sched_getaffinity(0)
    if len(mask) == 1:
        cur = next(iter(mask))
    else:
        # the pin made before HIP initialised did not survive (seen in about one process in eight on the MI355X
        # host: the calling thread comes back from the runtime's start-up with its full mask) -- pin again
        cur = _current_cpu()
        if cur not in allowed:
            cur = sorted(allowed)[0]
        os.sched_setaffinity(0, {cur})
    order = sorted(allowed)
    cands = [cur]
    if cur in order and tries > 1:
        # spread over the machine; a candidate sharing a last-level cache with an earlier one is skipped, and the
        # odd offset keeps the walk from landing on SMT siblings (cpu + n/2) of earlier candidates only
        i0, step = order.index(cur), max(1, len(order) // tries) + 9
        k = 1
        while len(cands) < tries and k < 4 * tries:
            c = order[(i0 + k * step) % len(order)]
            if all(c not in (_l3_peers(x) or {x}) for x in cands):
                cands.append(c)
            k += 1
    helpers = None
    timings = []
    for cpu in cands:
        try:
            os.sched_setaffinity(0, {cpu})
        except OSError:
            continue
        if helpers is None:
            helpers = colocate_runtime_threads(burst)      # finds them (and runs a burst doing so)
        else:
            _confine(helpers, _l3_peers(cpu) - {cpu})
        burst()                                            # settle on the new core
        t0 = time.perf_counter()
        burst()
        timings.append((cpu, time.perf_counter() - t0))
    if not timings:
        return -1, []
    best = min(timings, key=lambda t: t[1])[0]
    os.sched_setaffinity(0, {best})
    _confine(helpers or [], _l3_peers(best) - {best})
    return best, timings


def _confine(tids, target: set[int]) -> None:
    if not target:
        return
    for tid in tids:
        try:
            os.sched_setaffinity(tid, target)
        except OSError:
            pass


def restore_affinity(mask: set[int]) -> None:
    """Give the calling thread its full mask back (before CPU-parallel work such as an OpenMP region)."""
    if mask and hasattr(os, "sched_setaffinity"):
        os.sched_setaffinity(0, mask)
