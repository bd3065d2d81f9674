"""The N > 1 boundary exchange on CPU: two gloo processes, each hosting half of the nodes, run the
same plan -> pack -> all_gather -> unpack protocol bench.py uses with RCCL, and must reproduce the
neighbour rows the oracle's in-process communicate() (DPGOHash.h:28-86) produces."""
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
    from oracle.star import chordal_initialization
    G = dpgo_amd.read_g2o(path, nn)
    d, N = G.d, G.num_poses
    RS = (d + 1) * d
    per = nn // world
    mine = list(range(rank * per, (rank + 1) * per))
    (sn, sp_), (rn, rp) = G.exchange_plan(mine)
    num_poses, mm = og.read_g2o_file(path)
    X = chordal_initialization(num_poses, mm) + 0.01 * rank * 0     # same X on both ranks
    # pack: records of the exported poses, in key order (what dpgo_group_pack_sent does on the device)
    def record(node, pose):
        gid = G.node_offset(node) + pose
        return np.concatenate([X[gid], X[N + gid * d: N + gid * d + d].ravel()])
    allkeys = [None] * world
    dist.all_gather_object(allkeys, (sn.tolist(), sp_.tolist()))
    stride = max(max(len(k[0]) for k in allkeys), 1)
    send = torch.zeros(stride * RS, dtype=torch.float64)
    for i, (a, p) in enumerate(zip(sn, sp_)):
        send[i * RS:(i + 1) * RS] = torch.from_numpy(record(a, p))
    gathered = torch.zeros(world * stride * RS, dtype=torch.float64)
    dist.all_gather_into_tensor(gathered, send)
    # unpack: slot of key k of rank r = r * stride + k (dpgo_group_set_recv_layout)
    slot = {}
    for r, (ns, ps) in enumerate(allkeys):
        for k, key in enumerate(zip(ns, ps)):
            slot[key] = r * stride + k
    ok = True
    for a, p in zip(rn.tolist(), rp.tolist()):
        got = gathered[slot[(a, p)] * RS:(slot[(a, p)] + 1) * RS].numpy()
        ok = ok and np.array_equal(got, record(a, p))
    # every imported key must be exported by exactly the rank that hosts its node
    for a, p in zip(rn.tolist(), rp.tolist()):
        owner = a // per
        ok = ok and ((a, p) in set(zip(*allkeys[owner])))
    out[rank] = (bool(ok), len(rn), len(sn))
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
