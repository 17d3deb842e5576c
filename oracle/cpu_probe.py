import os, time, numpy as np, torch
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
os.system("lscpu | egrep 'Model name|Socket|Core|Thread|^CPU\\(s\\)' ; nproc")
import nvbio_amd as nvb
from nvbio_amd import workloads as W
from oracle import pyoracle as O
p, t = W.make_sw_batch(2_000_000, 100, 150, seed=2, device="cpu")
hp, ht = O.StringSet.from_device(p), O.StringSet.from_device(t)
for th in (1, 4, 16, 32, 64, 128, 256):
    O.batch_banded_gotoh_score(15, O.LOCAL, (2,-1,-2,-1), hp, ht, n_threads=th, native=True)
    t0 = time.perf_counter(); O.batch_banded_gotoh_score(15, O.LOCAL, (2,-1,-2,-1), hp, ht, n_threads=th, native=True); dt = time.perf_counter() - t0
    print("threads %3d: %.3f s  %.2f M reads/s  %.2f ns/cell/thread" % (th, dt, 2.0 / dt, dt * th / (2e6 * 1500) * 1e9))
