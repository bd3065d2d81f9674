"""The N > 1 boundary exchange on CPU: two gloo processes, each hosting half of the nodes, run the
plan -> pack -> all_gather -> unpack protocol of dpgo_comm_exchange with the library's own host packing
(dpgo_host_pack_sent / dpgo_host_unpack_recv: same key order and slot lay-out as the device kernels), and must
reproduce the neighbour rows the oracle's in-process communicate() (DPGOHash.h:28-86) produces."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, path, nn, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import dpgo_amd
    from oracle import g2o as og
    from oracle.star import chordal_initialization, DistPGO as ODistPGO
    G = dpgo_amd.read_g2o(path, nn)
    d, N = G.d, G.num_poses
    RS = (d + 1) * d
    per = nn // world
    mine = list(range(rank * per, (rank + 1) * per))
    (sn, sp_), (rn, rp) = G.exchange_plan(mine)
    num_poses, mm = og.read_g2o_file(path)
    X = chordal_initialization(num_poses, mm)                       # same X on both ranks
    # pack with the LIBRARY's host packing (dpgo_host_pack_sent: the lay-out dpgo_group_pack_sent produces on the
    # device), all-gather with gloo, unpack with the library (dpgo_host_unpack_recv)
    allkeys = [None] * world
    dist.all_gather_object(allkeys, (sn.tolist(), sp_.tolist()))
    stride = max(max(len(k[0]) for k in allkeys), 1)
    send = torch.zeros(stride * RS, dtype=torch.float64)
    packed = G.host_pack_sent(mine, X)
    send[:len(packed)] = torch.from_numpy(packed)
    gathered = torch.zeros(world * stride * RS, dtype=torch.float64)
    dist.all_gather_into_tensor(gathered, send)
    # the oracle's in-process communicate() gives the neighbour rows every node must end up with
    orc = ODistPGO(path, nn, X0=X, mm=mm, num_poses=num_poses)
    ok, imported = True, 0
    for a in mine:
        n0, n1, _, _ = G.node_sizes(a)
        Z = np.zeros(((d + 1) * (n0 + n1), d), order="F")
        imported += G.host_unpack_recv(mine, a, stride, allkeys, gathered.numpy(), Z)
        want = orc.nodes[a].results.Xk
        nb_node, _ = G.node_neighbours(a)
        for k in range(n1):
            if nb_node[k] in mine:
                continue                                            # hosted by this rank: dpgo_group_communicate_local
            ok = ok and np.array_equal(Z[(d + 1) * n0 + k], want[(d + 1) * n0 + k])
            r0 = (d + 1) * n0 + n1 + k * d
            ok = ok and np.array_equal(Z[r0:r0 + d], want[r0:r0 + d])
    # every imported key must be exported by exactly the rank that hosts its node
    for a, p in zip(rn.tolist(), rp.tolist()):
        owner = a // per
        ok = ok and ((a, p) in set(zip(*allkeys[owner])))
    out[rank] = (bool(ok), len(rn), len(sn), imported)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name,nn", [("smallGrid3D", 4), ("M3500", 4)])
def test_exchange_protocol_world2(fixtures_dir, name, nn):
    path = os.path.join(fixtures_dir, name + ".g2o")
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, path, nn, out), nprocs=2, join=True)
    assert out[0][0] and out[1][0]
    assert out[0][1] > 0 and out[1][1] > 0      # both ranks import something
    # symmetric-free sanity: what rank 0 imports, rank 1 exports (2 ranks only)
    assert out[0][1] == out[1][2] and out[1][1] == out[0][2]
    assert out[0][3] >= out[0][1] and out[1][3] >= out[1][1]     # every imported key fills at least one neighbour row


def test_plan_matches_oracle_recv_sets(fixtures_dir):
    import dpgo_amd
    from oracle import g2o as og
    path = os.path.join(fixtures_dir, "smallGrid3D.g2o")
    G = dpgo_amd.read_g2o(path, 4)
    num_poses, mm = og.read_g2o_file(path)
    _, meas, _ = og.partition_measurements(num_poses, mm, 4)
    (sn, sp_), (rn, rp) = G.exchange_plan([0, 1])
    want_recv, want_sent = set(), set()
    for a in (0, 1):
        info = og.generate_data_info(a, meas[a])
        for b, poses in info.recv.items():
            if b not in (0, 1):
                want_recv |= {(b, p) for p in poses}
        for b, poses in info.sent.items():
            if b not in (0, 1):
                want_sent |= {(a, p) for p in poses}
    assert set(zip(rn.tolist(), rp.tolist())) == want_recv
    assert set(zip(sn.tolist(), sp_.tolist())) == want_sent


def _p2p_worker(rank, world, port, path, nn, out):
    """Every rank sends each real neighbour exactly the poses the plan says (gloo isend / irecv of host buffers laid out
    by the plan) and must end up with its oracle neighbour rows."""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import dpgo_amd
    from oracle import g2o as og
    G = dpgo_amd.read_g2o(path, nn)
    per = nn // world
    plans = [G.exchange_plan(list(range(r * per, (r + 1) * per))) for r in range(world)]
    exported = [list(zip(p[0][0].tolist(), p[0][1].tolist())) for p in plans]
    needed = [list(zip(p[1][0].tolist(), p[1][1].tolist())) for p in plans]
    peers, skeys, rkeys = dpgo_amd.p2p_plan(rank, exported, needed)
    # a "pose record" that identifies itself: (node, pose, 1000 node + pose)
    rec = lambda k: [float(k[0]), float(k[1]), 1000.0 * k[0] + k[1]]
    send = torch.tensor([rec(k) for k in skeys] or [[0.0, 0.0, 0.0]], dtype=torch.float64)
    recv = torch.full((max(len(rkeys), 1), 3), -1.0, dtype=torch.float64)
    reqs = []
    for (q, so, sc, ro, rc) in peers:
        if sc:
            reqs.append(dist.isend(send[so:so + sc].contiguous(), dst=q))
        if rc:
            buf = torch.empty((rc, 3), dtype=torch.float64)
            reqs.append((dist.irecv(buf, src=q), buf, ro, rc))
    for r in reqs:
        if isinstance(r, tuple):
            r[0].wait()
            recv[r[2]:r[2] + r[3]] = r[1]
        else:
            r.wait()
    got = {(int(v[0]), int(v[1])): float(v[2]) for v in recv.tolist() if v[0] >= 0}
    ok = set(got) == set(needed[rank]) and all(abs(got[k] - (1000.0 * k[0] + k[1])) == 0 for k in got)
    ok = ok and [tuple(k) for k in rkeys] == sorted(set(needed[rank]) & set().union(*[set(exported[q]) for q in range(world) if q != rank]))
    ok = ok and all(k in set(exported[rank]) for k in skeys)
    out[rank] = (bool(ok), len(peers), len(skeys), len(rkeys), sum(len(e) for e in exported))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name,nn,world", [("smallGrid3D", 4, 4), ("torus3D", 8, 4), ("M3500", 4, 2)])
def test_neighbour_to_neighbour_plan_delivers_every_needed_pose(fixtures_dir, name, nn, world):
    """dpgo_comm_exchange with several ranks talks to the real neighbours only (grouped ncclSend / ncclRecv laid out by
    comm.cpp::p2p_plan).  RCCL cannot run several ranks here, so the plan is exercised with gloo messages on the CPU:
    every rank receives exactly the (node, pose) records it needs, each from the rank that exports it, and nothing else;
    the traffic is what the neighbours need, not world x the largest export."""
    path = os.path.join(fixtures_dir, name + ".g2o")
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_p2p_worker, args=(world, _free_port(), path, nn, out), nprocs=world, join=True)
    assert all(out[r][0] for r in range(world)), dict(out)
    assert sum(out[r][2] for r in range(world)) == sum(out[r][3] for r in range(world))   # everything sent is received once
    for r in range(world):
        assert out[r][1] <= world - 1
