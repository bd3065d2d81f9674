"""Starve the host side of a measurement on purpose: pin THIS process -- and every thread the HIP runtime
starts later -- to ONE core and keep `n` busy-looping sibling processes on that same core.  prepare() must be called before
anything touches the GPU (the affinity is inherited by the runtime's threads; the siblings are fresh children that never
touch the GPU and wait on a pipe), release() starts their spinning -- after the untimed set-up, so that only the measured
loop is starved; they are stopped by PID at exit.  What is left of the host is a fraction 1 / (n + 1) of one core: a path
that needs a launch per kernel and polls for every read-back shows it."""
import atexit
import os
import signal
import subprocess
import sys

_procs = []


def prepare(n):
    core = sorted(os.sched_getaffinity(0))[0]
    os.sched_setaffinity(0, {core})
    for _ in range(n):     # (inherit the affinity; block on stdin until released)
        _procs.append(subprocess.Popen([sys.executable, "-c", "import sys\nsys.stdin.buffer.read(1)\nwhile True: pass"],
                                       stdin=subprocess.PIPE))

    def stop():
        for p in _procs:
            try:
                os.kill(p.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
        for p in _procs:
            try:
                p.wait(timeout=5)
            except Exception:
                pass
    atexit.register(stop)
    return {"core": core, "spinners": n}


def release():
    for p in _procs:
        try:
            p.stdin.write(b"x")
            p.stdin.flush()
        except Exception:
            pass


def cpu_seconds():
    """CPU time (user + system, seconds) every spinner has used so far -- proof that they run."""
    out = []
    tick = os.sysconf("SC_CLK_TCK")
    for p in _procs:
        try:
            f = open("/proc/%d/stat" % p.pid).read().rsplit(")", 1)[1].split()
            out.append((int(f[11]) + int(f[12])) / tick)
        except Exception:
            out.append(None)
    return out


def starve_host(n):
    info = prepare(n)
    release()
    return info
