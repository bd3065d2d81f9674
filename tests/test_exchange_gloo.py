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
