#!/usr/bin/env python3
"""What does creating an RCCL communicator change for the host thread that issues the step's launches?

Measured on the MI355X box: the eager step issues ~1600 launches in ~33 ms; with a communicator alive the same loop takes
~36 ms although no collective runs.  This prints, before and after communicator creation: the issuing thread's CPU
affinity, the per-launch host cost of a small kernel, and which threads of the process burn CPU while it idles."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist


def launch_cost(n=4000):
    x = torch.zeros(1024, device="cuda")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        x.add_(1.0)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6


def thread_cpu():
    out = {}
    for tid in os.listdir("/proc/self/task"):
        try:
            f = open("/proc/self/task/%s/stat" % tid).read().rsplit(")", 1)[1].split()
            name = open("/proc/self/task/%s/comm" % tid).read().strip()
            out[tid] = (name, int(f[11]) + int(f[12]))          # utime + stime in clock ticks
        except OSError:
            pass
    return out


def report(tag):
    aff = sorted(os.sched_getaffinity(0))
    print("%s: affinity %d cpus [%d..%d], threads %d" % (tag, len(aff), aff[0], aff[-1], len(os.listdir("/proc/self/task"))))
    for _ in range(3):
        print("   launch: host %.2f us, host+device %.2f us" % launch_cost())
    a = thread_cpu()
    time.sleep(2.0)
    b = thread_cpu()
    busy = [(b[t][0], b[t][1] - a[t][1]) for t in b if t in a and b[t][1] - a[t][1] > 2]
    print("   threads busy while the process sleeps 2 s (ticks of 10 ms):", busy)


torch.cuda.set_device(0)
launch_cost(200)
report("before")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29535")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t = torch.ones(1 << 20, device="cuda")
dist.all_reduce(t)
torch.cuda.synchronize()
report("after communicator creation")
dist.destroy_process_group()
report("after destroy_process_group")
