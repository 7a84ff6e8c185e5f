#!/usr/bin/env python3
"""What a command's process looks like while the kernel takes it apart: runs the command, samples /proc/<pid>/statm (resident
pages) and /proc/<pid>/stat (state) every half millisecond, and prints the samples from the moment the resident set starts
to fall (or the state turns Z/X) to the moment wait() returns.  usage: exit_watch.py <command...>"""
import os, subprocess, sys, time
p = subprocess.Popen(sys.argv[1:], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
pid = p.pid
samples = []
zombie_tasks = []
t0 = time.perf_counter()
while p.poll() is None:
    try:
        with open(f"/proc/{pid}/statm") as f:
            rss = int(f.read().split()[1])
        with open(f"/proc/{pid}/stat") as f:
            st = f.read().rsplit(")", 1)[1].split()
        samples.append((time.perf_counter() - t0, rss * 4096 >> 20, st[0], int(st[17])))   # state, num_threads
        if st[0] == "Z" and len(zombie_tasks) < 12:          # the leader is gone: who is left, and where does it wait?
            for tid in os.listdir(f"/proc/{pid}/task"):
                try:
                    comm = open(f"/proc/{pid}/task/{tid}/comm").read().strip()
                    tst = open(f"/proc/{pid}/task/{tid}/stat").read().rsplit(")", 1)[1].split()[0]
                    try:
                        wchan = open(f"/proc/{pid}/task/{tid}/wchan").read().strip()
                    except Exception as e:
                        wchan = f"({type(e).__name__})"
                    try:
                        stack = open(f"/proc/{pid}/task/{tid}/stack").read().strip().replace("\n", " <- ")[:300]
                    except Exception as e:
                        stack = f"({type(e).__name__})"
                    zombie_tasks.append((round(time.perf_counter() - t0, 4), tid, comm, tst, wchan, stack))
                except Exception:
                    pass
    except Exception:
        samples.append((time.perf_counter() - t0, -1, "?", 0))
    time.sleep(0.0005)
t_end = time.perf_counter() - t0
peak = max(s[1] for s in samples)
print(f"wall {t_end:.3f} s; peak resident {peak} MB; the last 350 ms:")
last = None
for s in samples:
    if s[0] < t_end - 0.35:
        continue
    key = (s[1] // 64, s[2], s[3])
    if key != last:                      # a line whenever the resident set moves by 64 MB, or state / thread count change
        print(f"  {s[0]:.4f} s  {s[1]:6d} MB  state {s[2]} threads {s[3]}")
        last = key
for z in zombie_tasks:
    print("  task", z)
