#!/usr/bin/env python3
"""bench.py -- AMM-PGO# outer MM iterations per second on the headline synthetic pose graph.

Workload (BASELINE.json config 4): synthetic SE(3) lattice 50x50x40 = 100 000
poses / 400 000 edges, Huber loss (delta = 0.25, Static rescale), AMM-PGO#, num_nodes = 8 with the
reference's contiguous partition, driver options of C++/examples/dist_pgo.cpp:103-120, centralised
chordal initialisation (untimed set-up).  With N GPUs each rank hosts 8/N nodes, so every N runs the
SAME algorithmic trajectory (scaling = "strong").  One step = one outer iteration of all 8 nodes:
iterate() -> boundary-pose exchange -> update()  (dist_pgo.cpp:496-521).

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Prints ONE JSON line (rank 0) with the throughput, a `roofline` object for the dominant kernel family
(launch durations measured with HIP events on the launch stream in an instrumented pass that repeats the
timed region; a solve sweep's back-to-back launches share one event pair) and, at N = 1, a `cpu_baseline`
object (the C++ CPU restatement timed on a bounded sample).
"""
import argparse
import json
import os
import sys
import time


def _host_cores():
    """CPU cores this process may really use (cgroup quota), for thread-pool sizing."""
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(q) // int(p)))
    except Exception:
        pass
    return n


# keep BLAS / OpenMP pools inside the CPU quota (the GPU boxes show 256 CPUs behind a 16-core quota)
_share = max(1, min(_host_cores(), 32) // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))))
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "DPGO_HOST_THREADS"):
    os.environ.setdefault(_v, str(_share))

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n, argv, rank_timeout=600.0):
    """`python bench.py --gpus N` started directly (no torch.distributed.run around it): this parent -- which has not
    imported torch nor touched HIP, and never does -- starts N fresh children of this same script, one per GPU, with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (a free port) set, relays rank 0's single JSON line,
    and returns non-zero when any rank fails.  The ranks get `rank_timeout` seconds (--rank-timeout): a collective that
    hangs (a peer that never joined, an exchange that never completes) ends with the ranks that were still alive named
    on stderr, their process groups stopped, and exit code 124 -- not with the caller's own, much later, timeout.  Only
    fresh children are ever started or stopped; no process that touched the GPU is re-executed.  The reference analogue
    is one `dist_pgo` process that runs all nodes (C++/examples/dist_pgo.cpp:96-126, 492-531)."""
    import signal
    import subprocess
    port = _free_port()
    share = max(1, min(_host_cores(), 32) // n)
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "DPGO_HOST_THREADS"):
            env[v] = os.environ.get(v + "_PER_RANK", str(share))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, start_new_session=True))
    rc = 0
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.time() + rank_timeout
    try:
        # a rank that dies leaves the others waiting in a collective: poll, and stop everybody when one has failed
        pending = set(range(n))
        while pending:
            for r in list(pending):
                c = procs[r].poll()
                if c is not None:
                    pending.discard(r)
                    if c != 0:
                        sys.stderr.write("[bench] rank %d exited with code %d\n" % (r, c))
                        rc = rc or c
            if rc:
                break
            if pending and time.time() > deadline:
                sys.stderr.write("[bench] ranks %s still running after --rank-timeout %.0f s (finished: %s): stopping them\n"
                                 % (sorted(pending), rank_timeout, sorted(set(range(n)) - pending)))
                rc = 124
                break
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                if rc == 0:
                    p.wait()
                else:
                    try:
                        os.killpg(p.pid, signal.SIGTERM)    # the exact process groups started above
                    except ProcessLookupError:
                        pass
        if rc != 0:
            # a rank stuck inside a device wait may not act on SIGTERM: give it a moment, then SIGKILL its group
            t_end = time.time() + 5.0
            while time.time() < t_end and any(p.poll() is None for p in procs):
                time.sleep(0.05)
            for p in procs:
                if p.poll() is None:
                    try:
                        os.killpg(p.pid, signal.SIGKILL)
                    except ProcessLookupError:
                        pass
    reader.join(timeout=10)
    lines = [l for l in b"".join(chunks).decode(errors="replace").splitlines() if l.startswith("{")]
    if rc == 0 and lines:
        sys.stdout.write(lines[-1] + "\n")
        sys.stdout.flush()
    elif rc == 0:
        rc = 1
    return rc


def measure_traffic(args):
    """HBM bytes per launch of every kernel family from the PMC counters, measured for THIS run: two rocprofv3 passes
    (FETCH_SIZE, WRITE_SIZE -- they do not fit one pass; counters only, with --kernel-trace) over a short child run of this
    same script and workload, summarised as /opt/skills/guides/MI355X_MICROARCH.md prescribes (tools/pmc_summary.py: KiB
    units, read side doubled on gfx950).  Runs before this process touches the GPU; returns {family: bytes per launch} or
    None (no rocprofv3, a failed pass, ...: the caller then quotes the committed summary and says so)."""
    import shutil
    import subprocess
    import tempfile
    if not shutil.which("rocprofv3"):
        return None
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import pmc_summary
    except Exception:
        return None
    tmp = tempfile.mkdtemp(prefix="dpgo_pmc_", dir="/tmp")
    agg = {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, ctr)
            cmd = ["rocprofv3", "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable,
                   os.path.join(ROOT, "bench.py"), "--no-cpu", "--no-prof", "--converge", "0", "--steps", "5", "--warmup", "2",
                   "--traffic", "off", "--grid", args.grid, "--nodes", str(args.nodes), "--loss", args.loss]
            if args.emulate_world:      # (the emulated rank of an N-GPU run: the same counters for its one node per GPU)
                cmd += ["--emulate-world", str(args.emulate_world), "--emulate-rank", str(args.emulate_rank)]
            # (its own session: on a timeout the whole group goes -- rocprofv3 AND the python child under it, which would
            # otherwise keep the GPU busy during the timed run)
            child = subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd="/tmp",
                                     env=dict(os.environ, TMPDIR="/tmp", DPGO_ITER_GRAPH="0"),   # (counters per eager dispatch)
                                     start_new_session=True)
            try:
                child.wait(timeout=150)
            except subprocess.TimeoutExpired:
                import signal
                try:
                    os.killpg(child.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
                child.wait()
                return None
            if child.returncode != 0:
                return None
            agg[ctr] = pmc_summary.collect(d, ctr)
        res = {}
        for k in set(agg["FETCH_SIZE"]) | set(agg["WRITE_SIZE"]):
            nf, sf = agg["FETCH_SIZE"].get(k, [0, 0.0])
            nw, sw = agg["WRITE_SIZE"].get(k, [0, 0.0])
            if nf and nw:
                res[k] = 2.0 * 1024.0 * sf / nf + 1024.0 * sw / nw
        return res or None
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    # started by hand with --gpus N > 1 and no launcher around it: become the launcher (before any torch / HIP import)
    if "WORLD_SIZE" not in os.environ:
        pre = argparse.ArgumentParser(add_help=False)
        pre.add_argument("--gpus", type=int, default=1)
        pre.add_argument("--rank-timeout", type=float, default=600.0)
        known = pre.parse_known_args()[0]
        if known.gpus > 1:
            sys.exit(launch_ranks(known.gpus, sys.argv[1:], known.rank_timeout))

    # The contract is ONE JSON line on stdout.  RCCL prints its banner and warnings to the C-level stdout (the GPU
    # boxes export NCCL_DEBUG=VERSION), so everything this process writes to fd 1 goes to stderr from here on and
    # the JSON line is written to the saved descriptor at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--nodes", type=int, default=8)
    ap.add_argument("--grid", type=str, default="50,50,40,400000")
    ap.add_argument("--loss", type=str, default="huber")
    ap.add_argument("--prof-steps", type=int, default=5)
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-prof", action="store_true")
    ap.add_argument("--backend", type=str, default="nccl",
                    help="nccl (RCCL over xGMI, one GPU per rank) or gloo (host-staged; lets several ranks share one "
                         "GPU, used by the tests to exercise the multi-process path on a 1-GPU box)")
    ap.add_argument("--share-gpu", action="store_true", help="all ranks use GPU 0 (gloo backend only)")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="diagnostic: host only the nodes rank --emulate-rank would own in an N-GPU run, with frozen "
                         "neighbours and no exchange, to see the per-GPU step time of that run on one GPU")
    ap.add_argument("--emulate-rank", type=int, default=0)
    ap.add_argument("--converge", type=int, default=400,
                    help="second half of the metric (N = 1): after the timed region, restart from the chordal "
                         "initialisation, run this many iterations and report when the objective first came within 1e-6 "
                         "(relative) of the lowest one reached (SURVEY 8d: iterations and wall time to the reference "
                         "objective), the whole-run mean ms/iter and the ms/iter + CG steps/iter of the last 20 "
                         "iterations (the interior-step regime); 0 skips it")
    ap.add_argument("--traffic", type=str, default="auto",
                    help="auto: measure roofline.traffic with two rocprofv3 --pmc passes over a short child run before the "
                         "timed run (N = 1, about 40 s); off: quote the committed summary")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="stop after the ranks have met (no GPU needed): checks the launcher")
    ap.add_argument("--rank-timeout", type=float, default=600.0,
                    help="--gpus N > 1 started without a launcher: seconds the ranks get before they are stopped and bench.py "
                         "returns 124 with the ranks that were still alive named on stderr")
    ap.add_argument("--starve-host", type=int, default=-1,
                    help="diagnostic: pin this process (and every runtime thread it starts later) to ONE core and keep this many "
                         "busy-looping sibling processes on the same core (tools/starve.py) -- what a slow or crowded host does to "
                         "the step time; -1: off")
    ap.add_argument("--windows", type=int, default=1,
                    help="diagnostic: repeat the timed region this many times, each from the chordal initialisation again (the "
                         "re-initialisation is not timed); value = all timed steps / all timed seconds -- a run long enough for "
                         "what a crowded host does to it to show (--starve-host)")
    ap.add_argument("--force-exchange", action="store_true",
                    help="diagnostic: run the boundary exchange (pack, all-gather, unpack) even with one rank, to see "
                         "what it adds to a step")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    starved = None
    if args.starve_host >= 0:           # (before anything touches the GPU)
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import starve
        starved = starve.prepare(args.starve_host)      # (the spinners start after the untimed set-up)
    measured_traffic = None
    if (args.traffic == "auto" and world == 1 and args.gpus == 1 and not args.no_prof and args.prof_steps > 0
            and not args.rendezvous_only and args.starve_host < 0):
        measured_traffic = measure_traffic(args)       # (child processes; this one has not touched the GPU yet)

    import torch
    import dpgo_amd
    from dpgo_amd import synthetic

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.nodes % world != 0:
        raise SystemExit("num_nodes must be divisible by the number of GPUs")
    if args.share_gpu:
        local_rank = 0
    dist = None
    host_staged = args.backend == "gloo"
    do_exchange = world > 1 or args.force_exchange
    if do_exchange:
        # control plane (rendezvous, the RCCL unique id, barriers, the max over ranks of the timing): gloo over TCP.
        # data plane: the library's own RCCL communicator (dpgo_comm_*), or -- with --backend gloo -- host-staged
        # gloo collectives, which let several ranks share one GPU (tests on a 1-GPU box)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ["MASTER_ADDR"] in ("127.0.0.1", "localhost"):
            # one node: rendezvous and bootstrap over the loopback interface (the container's hostname may not resolve)
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        if "MASTER_PORT" not in os.environ:
            if world > 1:
                raise SystemExit("WORLD_SIZE > 1 without MASTER_PORT")
            os.environ["MASTER_PORT"] = str(_free_port())
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("gloo")
    if args.rendezvous_only:
        # launcher check (runs without a GPU): every rank arrived with a consistent environment
        seen = [None] * world
        if world > 1:
            dist.all_gather_object(seen, (rank, local_rank, os.environ.get("MASTER_PORT")))
            # (tests of the launcher's deadline and of a rank that dies while the others sit in a collective)
            if os.environ.get("DPGO_BENCH_TEST_DIE_RANK") == str(rank):
                os._exit(3)
            if os.environ.get("DPGO_BENCH_TEST_HANG_RANK") == str(rank):
                time.sleep(3600)
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            os.write(real_stdout, (json.dumps({"rendezvous": "ok", "n_gpus": world, "ranks": seen}) + "\n").encode())
        return
    torch.cuda.set_device(local_rank)
    cdev = "cpu"

    # ---- set-up (untimed): graph, partition, chordal initialisation, operators, factorizations
    t0 = time.time()
    nx, ny, nz, ne = (int(v) for v in args.grid.split(","))
    g = synthetic.grid(nx, ny, nz, ne, seed=synthetic.HEADLINE["seed"])
    G = dpgo_amd.graph_from_edges(3, g["num_poses"], g["I"], g["J"], g["R"], g["t"], g["kappa"], g["tau"], args.nodes)
    loss = dpgo_amd.LOSS_NAMES[args.loss]
    opt = dpgo_amd.Options.driver(loss, True)
    X0 = G.chordal_initialization()
    t_init = time.time() - t0
    per = args.nodes // world
    my_nodes = list(range(rank * per, (rank + 1) * per))
    if args.emulate_world:
        per = args.nodes // args.emulate_world
        my_nodes = list(range(args.emulate_rank * per, (args.emulate_rank + 1) * per))
    t0 = time.time()
    grp = dpgo_amd.NodeGroup(G, my_nodes, opt, device=local_rank)
    t_group = time.time() - t0
    if grp.initialize_global(X0) != 0:
        raise SystemExit("initialize failed")

    RS = (G.d + 1) * G.d
    send = gathered = ext = comm = None
    exchange_fallback = None     # why the exchange of the line is not the one that was asked for (None: it is)
    # Bring-up, at most twice: the exchange that was asked for (RCCL behind the C ABI) through the warm-up iterations -- the
    # first real use of the neighbour-to-neighbour path on a multi-GPU box -- and, if a step fails on ANY rank there (an RCCL
    # error, an exchange that never completes: the library's deadlines turn that into an error return), once more from the
    # initial point with the gloo host-staged exchange, so that the line is still emitted and says which exchange it timed.
    # The ranks vote after the warm-up (gloo): nobody goes on alone.
    released = False
    while True:
        if do_exchange and not host_staged:
            # RCCL behind the C ABI; the ranks agree on the outcome, so that nobody is left waiting in a collective
            def bcast(raw):
                box = [raw]
                dist.broadcast_object_list(box, src=0)
                return box[0]
            try:
                if world == 1 and args.emulate_world:
                    # the emulated rank's neighbours live nowhere: run the neighbour-to-neighbour path against itself, in its
                    # steady state (pack, grouped send / recv of every exported record, unpack into a scratch array)
                    comm = dpgo_amd.Comm.self_exchange_only(grp)
                else:
                    comm = dpgo_amd.Comm(grp, rank, world, bcast if world > 1 else None)
                comm.enable_timing()
                ok = 1
            except Exception as e:
                sys.stderr.write("[bench] rank %d: RCCL communicator failed (%r)\n" % (rank, e))
                ok = 0
            vote = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(vote, op=dist.ReduceOp.MIN)
            if int(vote.item()) == 0:
                sys.stderr.write("[bench] RCCL unavailable on some rank: falling back to the gloo host-staged exchange\n")
                comm = None
                host_staged = True
        if do_exchange and host_staged:
            ext = torch.cuda.ExternalStream(grp.stream())
            keys = grp.sent_keys()
            allkeys = [None] * world
            dist.all_gather_object(allkeys, (keys[0].tolist(), keys[1].tolist()))
            stride = max(max(len(k[0]) for k in allkeys), 1)
            grp.set_recv_layout(stride, [(np.asarray(k[0], np.int32), np.asarray(k[1], np.int32)) for k in allkeys])
            send = torch.zeros(stride * RS, dtype=torch.float64, device="cuda")
            gathered = torch.zeros(world * stride * RS, dtype=torch.float64, device="cuda")
            send_h = torch.zeros(stride * RS, dtype=torch.float64)
            gathered_h = torch.zeros(world * stride * RS, dtype=torch.float64)
            torch.cuda.synchronize()

        def exchange():
            if comm is not None:
                comm.exchange()            # neighbour to neighbour on the group's stream, or pack -> ncclAllGather -> unpack on the communicator's; update() joins it
            grp.communicate_local()
            if do_exchange and host_staged:
                with torch.cuda.stream(ext):
                    grp.pack_sent(send.data_ptr())
                    send_h.copy_(send)
                    dist.all_gather_into_tensor(gathered_h, send_h)
                    gathered.copy_(gathered_h)
                    grp.unpack_recv(gathered.data_ptr())

        native_step = not (do_exchange and host_staged) and os.environ.get("DPGO_BENCH_PY_STEP") != "1"   # (A/B switch)

        def step():
            if native_step:
                # iterate -> exchange (the communicator's own stream) -> communicate -> update in one native call, as the
                # C++ driver's loop does: no interpreter overhead between the launches
                rc = grp.step(comm)
            else:
                rc = grp.iterate()
                exchange()
                rc |= grp.update()
            if rc != 0:
                raise SystemExit("step failed")

        def barrier_local():
            grp.sync()
            torch.cuda.synchronize()

        def barrier():
            barrier_local()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()

        grp.update()
        if starved is not None and not released:
            starve.release()
            released = True
            time.sleep(0.2)
        warm_ok, warm_err = 1, None
        try:
            for _ in range(args.warmup):
                step()
            barrier_local()
        except (SystemExit, Exception) as e:
            warm_ok, warm_err = 0, repr(e)
            sys.stderr.write("[bench] rank %d: the warm-up failed (%s)\n" % (rank, warm_err))
        if do_exchange and dist is not None:
            vote = torch.tensor([warm_ok], dtype=torch.int32)
            dist.all_reduce(vote, op=dist.ReduceOp.MIN)
            warm_ok_all = int(vote.item())
        else:
            warm_ok_all = warm_ok
        if warm_ok_all:
            break
        if not do_exchange or host_staged or exchange_fallback is not None:
            raise SystemExit("step failed")
        # a failed group cannot go on (its stream may hold a collective that was aborted): a new group, the initial point, the
        # host-staged exchange
        exchange_fallback = "the RCCL exchange failed during the warm-up on some rank" + ((": " + warm_err) if warm_err else "")
        sys.stderr.write("[bench] rank %d: falling back to the gloo host-staged exchange\n" % rank)
        try:
            if comm is not None:
                comm.close()
        except Exception:
            pass
        comm = None
        if world == 1:
            do_exchange = False     # (an emulated rank: its neighbours live nowhere, there is nothing to stage through the host)
        else:
            host_staged = True
        grp = dpgo_amd.NodeGroup(G, my_nodes, opt, device=local_rank)
        if grp.initialize_global(X0) != 0:
            raise SystemExit("initialize failed")
    barrier()
    cpu0, wall0, spin0 = time.process_time(), time.perf_counter(), (starve.cpu_seconds() if starved is not None else None)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    window_ms = [1e3 * elapsed / args.steps]
    for _ in range(args.windows - 1):          # (diagnostic: the same window again and again)
        grp.initialize_global(X0)
        grp.update()
        for _ in range(args.warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        window_ms.append(1e3 * (time.perf_counter() - t0) / args.steps)
    if args.windows > 1:
        elapsed = 1e-3 * sum(window_ms) * args.steps / args.windows     # (mean window)
    if starved is not None:
        # how much of the core this process got, and proof that the siblings ran: their CPU seconds over the same span
        wall = time.perf_counter() - wall0
        spin1 = starve.cpu_seconds()
        starved = dict(starved, wall_s=wall, own_cpu_s=time.process_time() - cpu0,
                       spinners_cpu_s=[(b - a) if a is not None and b is not None else None for a, b in zip(spin0, spin1)])
    per_rank = None
    if do_exchange:
        # every rank's own time over the timed region and what it hands to the transport per exchange (the N > 1 line
        # carries both: the max over ranks is `value`'s clock, the spread says whether one rank holds the others up)
        mine = [elapsed, float(comm.bytes_sent()) if comm is not None else float(len(grp.sent_keys()[0]) * RS * 8)]
        if world > 1:
            allr = [None] * world
            dist.all_gather_object(allr, mine)
        else:
            allr = [mine]
        per_rank = {"ms_per_step": [1e3 * a[0] / args.steps for a in allr],
                    "ms_per_step_min": 1e3 * min(a[0] for a in allr) / args.steps,
                    "ms_per_step_max": 1e3 * max(a[0] for a in allr) / args.steps,
                    "bytes_sent_per_exchange": [int(a[1]) for a in allr]}
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    # objective after the timed region: sum_a fobj^a == F
    fsum = sum(grp.results(k).fobj for k in range(len(grp)))
    refined = sum(int(grp.results(k).refined) for k in range(len(grp)))
    inner = sum(int(grp.results(k).tnt_inner_iterations) for k in range(len(grp)))
    if world > 1:
        tt = torch.tensor([fsum, refined, inner], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt)
        fsum, refined, inner = (float(v) for v in tt.tolist())

    # ---- instrumented pass: per-launch kernel durations with HIP events on the launch stream
    roofline = None
    kernels = {}
    if not args.no_prof and args.prof_steps > 0:
        dpgo_amd.prof_enable(True)
        for _ in range(args.prof_steps):
            step()
        barrier()
        stats = dpgo_amd.prof_collect()
        operands = dpgo_amd.prof_collect_operands()
        dpgo_amd.prof_enable(False)
        tot = sum(v[0] for v in stats.values())
        for name, (ms, by, cnt) in sorted(stats.items(), key=lambda kv: -kv[1][0]):
            if cnt:
                kernels[name] = dict(ms_per_step=ms / args.prof_steps, launches_per_step=cnt / args.prof_steps,
                                     avg_us=1e3 * ms / cnt, algo_MB_per_launch=by / cnt / 1e6,
                                     GBps=(by / 1e9) / (ms / 1e3) if ms > 0 else 0.0, share=ms / tot,
                                     frac_of_hbm_peak=((by / 1e9) / (ms / 1e3) / HBM_PEAK_GBS) if ms > 0 else 0.0)
                # the fused passes: SURVEY 8(d)'s formula prices a bare residual / proximal pass; what the kernel has to move,
                # operand by operand (its per-pose blocks, the previous iterate, the halo copy ...), is reported beside it
                if operands.get(name, 0.0) > 0 and ms > 0:
                    kernels[name].update(operand_MB_per_launch=operands[name] / cnt / 1e6,
                                         GBps_operands=(operands[name] / 1e9) / (ms / 1e3),
                                         frac_of_hbm_peak_operands=(operands[name] / 1e9) / (ms / 1e3) / HBM_PEAK_GBS)
        dom = max(stats.items(), key=lambda kv: kv[1][0])
        ms, by, cnt = dom[1]
        ach = (by / 1e9) / (ms / 1e3)
        # HBM bytes per launch from the PMC counters: measured by separate rocprofv3 --pmc passes of this
        # same command (FETCH_SIZE / WRITE_SIZE cannot be read from inside the process); the committed
        # summary (tools/pmc_summary.py) is quoted when it covers this workload and kernel
        traffic = traffic_source = None
        if measured_traffic and dom[0] in measured_traffic:
            traffic = measured_traffic[dom[0]]
            traffic_source = ("measured for this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over a 5-step child run of this "
                              "command (2 x FETCH_SIZE + WRITE_SIZE, KiB, gfx950 correction; tools/pmc_summary.py); the average is over "
                              "every launch of the family in the child run, its set-up (chordal initialisation) and warm-up included")
            for name in kernels:
                if name in measured_traffic:
                    kernels[name]["hbm_MB_per_launch_measured"] = measured_traffic[name] / 1e6
        try:
          if traffic is None:
            import glob
            cand = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic.json")))
            pm = json.load(open(cand[-1]))
            if pm.get("workload") == args.grid and pm.get("n_gpus") == world and not args.emulate_world:
                traffic = pm["kernels"][dom[0]]["hbm_bytes_per_launch"]
                traffic_source = "committed rocprofv3 --pmc passes of this command (%s), not measured in this run" % os.path.basename(cand[-1])
        except Exception:
            traffic = None
        roofline = dict(kernel=dom[0], bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=ach / HBM_PEAK_GBS, traffic=traffic, traffic_source=traffic_source, avg_launch_us=1e3 * ms / cnt,
                        algorithmic_bytes_per_launch=by / cnt, launches_per_step=cnt / args.prof_steps)

    # ---- CPU baseline: the C++ restatement of the path (tools/cpu_baseline) on a bounded sample, rank 0, N = 1 only
    cpu = None
    if world == 1 and not args.no_cpu and args.cpu_steps > 0:
        cpu = cpu_baseline(g, args.nodes, loss, X0, args.cpu_steps)

    convergence = None
    if args.converge > 0 and not args.emulate_world:
        # every rank steps (the exchange included when there are several); each keeps ITS clock and ITS nodes' sums per
        # iteration, and the ranks' traces are combined once after the loop (reduce_trace: objective = sum over the ranks,
        # time = the slowest rank's) -- no collective inside the timed iterations beyond the exchange itself.  The reference
        # driver logs the same pair per iteration for every num_nodes (C++/examples/dist_pgo.cpp:492-531).
        grp.initialize_global(X0)
        grp.update()
        barrier()
        trace, t0 = [], time.perf_counter()
        for _ in range(args.converge):
            step()
            grp.sync()
            trace.append((time.perf_counter() - t0, 2.0 * sum(grp.results(k).fobj for k in range(len(grp))),
                          float(sum(int(grp.results(k).tnt_inner_iterations) for k in range(len(grp)) if grp.results(k).refined)),
                          float(sum(int(grp.results(k).refined) for k in range(len(grp))))))
        trace = reduce_trace(trace, dist if world > 1 else None, args.nodes)
        convergence = summarize_convergence(trace, args.converge, load_cpu_reference(args))
    if rank == 0:
        out = {
            "metric": "AMM-PGO# outer MM iterations/sec, SE(3) PGO, synthetic 100k-pose/400k-edge graph, 8 nodes",
            "value": args.steps / elapsed, "unit": "iters/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "diagnostic_starved_host": starved,
            "diagnostic_windows_ms_per_step": window_ms if args.windows > 1 else None,
            "diagnostic_emulated_rank": ("%d of %d" % (args.emulate_rank, args.emulate_world)) if args.emulate_world else None,
            "config": {"workload": "synthetic SE(3) lattice %dx%dx%d, %d poses / %d edges, %s loss, AMM-PGO#, "
                                   "num_nodes=%d (%d per GPU), chordal init" % (nx, ny, nz, g["num_poses"], len(g["I"]),
                                                                             args.loss, args.nodes, per),
                       "num_nodes": args.nodes, "nodes_per_gpu": per, "loss": args.loss,
                       "iterations_before_timed_region": args.warmup,
                       "refined_nodes_last_step": refined, "tnt_inner_iterations_last_step": inner},
            "objective_2F": 2 * fsum,
            "exchange_fallback": exchange_fallback,
            "exchange": None if not do_exchange else (("gloo, staged through the host" + (" (FALLBACK: %s)" % exchange_fallback if exchange_fallback else "")) if host_staged else
                                                     (("RCCL, neighbour to neighbour (grouped ncclSend / ncclRecv)" + (", THIS RANK AS ITS OWN PEER (measurement mode)" if world == 1 and args.emulate_world else ""))
                                                      + " behind the C ABI (dpgo_comm_exchange) on the group's own stream: pack on the tail of iterate(), unpack inside update()'s inter-edge pass"
                                                      if comm.exchange_kind() == "p2p" else "RCCL all-gather behind the C ABI (dpgo_comm_exchange) on the communicator's stream, joined in update()")),
            "exchange_us_ready_to_done": (lambda t: {"mean_us": t[0], "exchanges": t[1]})(comm.exchange_time()) if comm is not None else None,
            "ranks": per_rank,
            "setup_s": {"graph+chordal_init": t_init, "operators+factorizations": t_group},
            "solver": grp.solver_stats(),
            "graphs": grp.graph_stats(),
            "roofline": roofline, "kernels": kernels, "cpu_baseline": cpu,
        }
        if convergence is not None:
            out["convergence"] = convergence
            # the two halves of BASELINE's metric side by side: `value` times the cheapest regime of the run (one CG step per
            # refinement), this one the whole way to the reference objective
            out["iters_per_s_to_objective"] = convergence["iters_per_s_to_objective"]
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if comm is not None:
        comm.close()
    if do_exchange:
        dist.destroy_process_group()


def reduce_trace(trace, dist, num_nodes):
    """Per-iteration (seconds, 2 F, CG steps, refined nodes) of THIS rank's nodes -> the job's: the objective, the CG steps
    and the refined nodes are sums over the ranks, the time stamp of an iteration is the slowest rank's.  dist: an
    initialised torch.distributed (gloo control plane) or None for one rank.  Returns a list of
    (seconds, 2F, CG steps per node, refined nodes)."""
    import numpy as _np
    a = _np.asarray(trace, dtype=_np.float64).reshape(-1, 4)
    if dist is not None:
        import torch
        sums = torch.from_numpy(_np.ascontiguousarray(a[:, 1:]))
        tmax = torch.from_numpy(_np.ascontiguousarray(a[:, 0]))
        dist.all_reduce(sums)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        a = _np.concatenate([tmax.numpy()[:, None], sums.numpy()], axis=1)
    return [(float(t), float(f), float(c) / num_nodes, float(r)) for t, f, c, r in a]


def load_cpu_reference(args):
    """The objective the CPU path reaches on the headline instance (tools/cpu_convergence.py, committed once per round as
    profiles/rNN_cpu_convergence.json), or None."""
    try:
        import glob
        cand = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_cpu_convergence.json")))
        cj = json.load(open(cand[-1]))
        if args.grid == "50,50,40,400000" and args.nodes == 8 and args.loss == "huber":
            return {"objective_2F": cj["lowest_2F"], "iterations_to_1e-6": cj["iterations_to_1e-6"],
                    "seconds_to_1e-6": cj["seconds_to_1e-6"], "cores": cj["cores"], "cpu_model": cj["cpu_model"],
                    "source": os.path.basename(cand[-1]) + " (tools/cpu_convergence.py: the C++ CPU restatement run to its own 1e-6 on all granted cores; not measured in this run)"}
    except Exception:
        pass
    return None


def summarize_convergence(trace, iterations_run, cpu_ref):
    """The second half of BASELINE's metric from a per-iteration trace of (seconds, 2F, CG steps per node, refined nodes):
    iterations and seconds until the objective first comes within 1e-6 (relative) of the objective the CPU path reaches
    (cpu_ref; without it, or when the run stops short of it: the run's own lowest objective, and the block says so)."""
    best = min(f for _, f, _, _ in trace)
    target = cpu_ref["objective_2F"] if cpu_ref else best
    hit = next((i for i, (_, f, _, _) in enumerate(trace) if f <= target * (1 + 1e-6)), None)
    if hit is None:       # (the run stopped short of the CPU's objective: report against its own lowest, and say so)
        target, cpu_ref = best, dict(cpu_ref or {}, not_reached=True)
        hit = next(i for i, (_, f, _, _) in enumerate(trace) if f <= target * (1 + 1e-6))
    tail = trace[-21:] if len(trace) > 21 else trace
    return {"iterations_run": iterations_run, "lowest_2F": best, "target_2F": target,
            "target": "the objective the CPU path reaches (cpu_reference)" if cpu_ref and not cpu_ref.get("not_reached") else "lowest objective of this run",
            "cpu_reference": cpu_ref, "iterations_to_1e-6": hit + 1,
            "seconds_to_1e-6": trace[hit][0], "mean_ms_per_iter_to_1e-6": 1e3 * trace[hit][0] / (hit + 1),
            "iters_per_s_to_objective": (hit + 1) / trace[hit][0] if trace[hit][0] > 0 else None,
            "mean_ms_per_iter_whole_run": 1e3 * trace[-1][0] / len(trace),
            "last20_ms_per_iter": 1e3 * (tail[-1][0] - tail[0][0]) / max(len(tail) - 1, 1),
            "last20_cg_steps_per_node_per_iter": sum(t[2] for t in tail[1:]) / max(len(tail) - 1, 1),
            "last20_refined_nodes_per_iter": sum(t[3] for t in tail[1:]) / max(len(tail) - 1, 1),
            "objective_2F_after_first_iteration": trace[0][1],
            "objective_2F_at": {str(k): trace[k - 1][1] for k in (1, 10, 50, 100, 200, 400, 800) if k <= len(trace)}}


def cpu_baseline(g, num_nodes, loss, X0, steps):
    """The C++ CPU restatement of the path (tools/cpu_baseline/cpu_dpgo: -O3 -march=native -fopenmp on the library's
    host data structures, see its header) timed on this box's host cores on a bounded sample: all `num_nodes` nodes,
    `steps` outer iterations from the same initial guess, once with one thread and once with the cores the cgroup
    grants.  Timing scope as in the reference driver: sum of iterate() + update() over the nodes, communication
    excluded (dist_pgo.cpp:496-521).  `value` is the all-core rate."""
    import struct
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "tools", "cpu_baseline", "cpu_dpgo")
    if not os.path.exists(exe):
        return {"value": None, "unit": "iters/s", "cores": 0, "kind": "port",
                "sample": "tools/cpu_baseline/cpu_dpgo is not built (make -C tools/cpu_baseline)"}
    d, N, m = 3, g["num_poses"], len(g["I"])
    tmp = tempfile.mkdtemp(prefix="dpgo_cpu_")
    fe, fx = os.path.join(tmp, "edges.bin"), os.path.join(tmp, "X0.bin")
    rec = np.dtype([("i", "<i4"), ("j", "<i4"), ("R", "<f8", (d * d,)), ("t", "<f8", (d,)), ("kappa", "<f8"), ("tau", "<f8")])
    E = np.zeros(m, rec)
    E["i"], E["j"] = g["I"], g["J"]
    E["R"], E["t"] = np.asarray(g["R"]).reshape(m, d * d), g["t"]
    E["kappa"], E["tau"] = g["kappa"], g["tau"]
    with open(fe, "wb") as fh:
        fh.write(struct.pack("<iii", d, N, m))
        fh.write(E.tobytes())
    np.asfortranarray(X0, dtype=np.float64).T.copy().tofile(fx)        # column-major
    cores = _host_cores()
    runs = {}
    try:
        out = subprocess.run([exe, fe, fx, str(num_nodes), str(loss), str(steps), "1,%d" % cores], capture_output=True, text=True,
                             timeout=900)
        res = json.loads(out.stdout.strip().splitlines()[-1])
        for r in res["runs"]:
            runs[r["threads"]] = dict(r, setup_s=res["setup_s"])
    except Exception as e:
        runs = {1: {"error": repr(e)}, cores: {"error": repr(e)}}
    for f in (fe, fx):
        os.remove(f)
    os.rmdir(tmp)
    cpu_model = "unknown"
    try:
        cpu_model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        pass
    rate = lambda r: (1.0 / r["seconds_per_iteration"]) if "seconds_per_iteration" in r else None
    return {"value": rate(runs[cores]), "unit": "iters/s", "cores": min(cores, num_nodes), "kind": "port",
            "value_1_thread": rate(runs[1]), "cpu_model": cpu_model, "host_cores_granted": cores,
            "objective_2F_after_sample": runs[cores].get("objective_2F"),
            "sample": "C++ restatement (tools/cpu_baseline/cpu_dpgo, g++ -O3 -march=native -fopenmp), all %d nodes, %d outer "
                      "iterations (iterate + update, communication excluded) from the chordal initialisation; the nodes "
                      "are dealt to the threads (at most %d run at once); set-up %.1f s untimed"
                      % (num_nodes, steps, num_nodes, runs[cores].get("setup_s", float("nan")))}


if __name__ == "__main__":
    main()
