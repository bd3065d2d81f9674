"""Starve the host side of a measurement on purpose: pin THIS process -- and every thread the HIP runtime
starts later -- to ONE core and keep `n` busy-looping sibling processes on that same core.  Must be called before anything
touches the GPU (the siblings are fresh children that never do; they are stopped by PID at exit).  What is left of the
host is a fraction 1 / (n + 1) of one core: a path that needs a launch per kernel shows it, a replayed graph does not."""
import atexit
import os
import signal
import subprocess
import sys


def starve_host(n):
    core = sorted(os.sched_getaffinity(0))[0]
    os.sched_setaffinity(0, {core})
    procs = [subprocess.Popen([sys.executable, "-c", "while True: pass"]) for _ in range(n)]   # (inherit the affinity)

    def stop():
        for p in procs:
            try:
                os.kill(p.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
        for p in procs:
            try:
                p.wait(timeout=5)
            except Exception:
                pass
    atexit.register(stop)
    return {"core": core, "spinners": n}
