"""How many host cores may this process actually USE? (TEST INFRASTRUCTURE: bench.py's cpu_baseline leg.)

``os.cpu_count()`` is the machine's core count. A container is usually allowed far less: a CPU affinity mask and / or a
CFS bandwidth quota (cgroup ``cpu.max`` / ``cpu.cfs_quota_us``). A thread pool sized to the machine under a quota of a few
cores burns the whole period's budget in milliseconds and is then frozen until the next 100 ms period — every timed
call then lasts a multiple of the period (round 3's 100.0 ms / 999.3 ms "medians" of a 0.5 MiB copy on 256 torch
threads). ``usable_cores()`` is the smallest of the three limits; ``throttle_counters()`` reads the cgroup's
``nr_throttled`` so a baseline run can SHOW that it was not throttled.
"""
from __future__ import annotations

import math
import os


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def _cgroup_dirs():
    """candidate cgroup directories of this process: v2 unified path and v1 cpu controller path, then the roots"""
    dirs = []
    txt = _read("/proc/self/cgroup") or ""
    for line in txt.splitlines():
        parts = line.split(":", 2)
        if len(parts) != 3:
            continue
        _, ctrl, path = parts
        if ctrl == "":  # cgroup v2
            dirs.append(os.path.join("/sys/fs/cgroup", path.lstrip("/")))
        elif "cpu" in ctrl.split(","):
            dirs.append(os.path.join("/sys/fs/cgroup/cpu", path.lstrip("/")))
            dirs.append(os.path.join("/sys/fs/cgroup", ctrl, path.lstrip("/")))
    dirs += ["/sys/fs/cgroup", "/sys/fs/cgroup/cpu", "/sys/fs/cgroup/cpu,cpuacct"]
    seen, out = set(), []
    for d in dirs:
        if d not in seen:
            seen.add(d)
            out.append(d)
    return out


def quota_cores():
    """CFS quota of the tightest cgroup on the way up, in cores (float), or None when unlimited / unreadable"""
    best = None
    for d in _cgroup_dirs():
        while d.startswith("/sys/fs/cgroup"):
            v2 = _read(os.path.join(d, "cpu.max"))
            if v2:
                q, _, p = v2.partition(" ")
                if q != "max":
                    try:
                        c = float(q) / float(p or 100000)
                        best = c if best is None else min(best, c)
                    except ValueError:
                        pass
            q1, p1 = _read(os.path.join(d, "cpu.cfs_quota_us")), _read(os.path.join(d, "cpu.cfs_period_us"))
            if q1 and p1:
                try:
                    if int(q1) > 0:
                        c = int(q1) / int(p1)
                        best = c if best is None else min(best, c)
                except ValueError:
                    pass
            if d in ("/sys/fs/cgroup", "/"):
                break
            d = os.path.dirname(d)
    return best


def usable_cores() -> dict:
    """{"machine": os.cpu_count(), "affinity": ..., "quota": cores or None, "usable": the smallest of them (>= 1)}"""
    machine = os.cpu_count() or 1
    try:
        aff = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        aff = machine
    q = quota_cores()
    env = os.environ.get("KVQ_CPU_BASELINE_THREADS")
    usable = min(machine, aff, max(1, math.floor(q)) if q else machine)
    if env:
        usable = max(1, int(env))
    return {"machine": machine, "affinity": aff, "quota": None if q is None else round(q, 2), "usable": int(usable),
            "override": env}


def throttle_counters():
    """(nr_periods, nr_throttled, throttled_time) summed over the readable cpu.stat files of this process's cgroups, or None"""
    tot, found = [0, 0, 0], False
    for d in _cgroup_dirs():
        txt = _read(os.path.join(d, "cpu.stat"))
        if not txt:
            continue
        vals = dict(line.split()[:2] for line in txt.splitlines() if len(line.split()) >= 2)
        if "nr_throttled" in vals:
            found = True
            tot[0] += int(vals.get("nr_periods", 0))
            tot[1] += int(vals.get("nr_throttled", 0))
            tot[2] += int(vals.get("throttled_usec", vals.get("throttled_time", 0)))
            break  # the process's own (deepest) cgroup is the one that counts
    return tuple(tot) if found else None
