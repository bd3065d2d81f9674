"""The distributed chordal initialisation (dist_pgo --dist_init true, C++/examples/dist_pgo.cpp:144-416) with the nodes of
the graph spread over two processes -- both on the one GPU of the test box, collectives lent from torch.distributed (gloo) --
against the single-group run: every rank runs the four stages for its own nodes, the stage halos (boundary rows of the sparse
stages per iteration, the reduced stages' per-node blocks, own-pose blocks between the stages) travel through the
collectives, and all ranks must end with the single group's initial guess and sampled stage objectives.  (The groups cut
their solve tiles differently -- the tile classes depend on how many fronts a group holds -- so 1e-9, not bit-equality.)"""
import json
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _options():
    import dpgo_amd
    return dpgo_amd.DChordalOptions(local_iters=5)


def _worker(rank, world, port, path, nn, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import dpgo_amd
    G = dpgo_amd.read_g2o(path, nn)
    per = nn // world
    grp = dpgo_amd.NodeGroup(G, list(range(rank * per, (rank + 1) * per)), dpgo_amd.Options.driver(dpgo_amd.LOSS_NONE, True), device=0)
    grp.connect_torch()
    X, obj = grp.dist_chordal_initialization(_options())
    np.save(out + ".%d.npy" % rank, X)
    if rank == 0:
        with open(out, "w") as fh:
            json.dump(obj.tolist(), fh)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name,nn", [("smallGrid3D", 4), ("M3500", 4)])
def test_dist_init_two_processes_match_one_group(fixtures_dir, tmp_path, name, nn):
    import torch.multiprocessing as mp
    import dpgo_amd
    path = os.path.join(fixtures_dir, name + ".g2o")
    out = str(tmp_path / "obj.json")
    mp.spawn(_worker, args=(2, _free_port(), path, nn, out), nprocs=2, join=True)
    obj2 = np.asarray(json.load(open(out)))
    X2 = [np.load(out + ".%d.npy" % r) for r in range(2)]
    np.testing.assert_array_equal(X2[0], X2[1])                      # every rank holds the same complete guess
    G = dpgo_amd.read_g2o(path, nn)
    grp = dpgo_amd.NodeGroup(G, range(nn), dpgo_amd.Options.driver(dpgo_amd.LOSS_NONE, True))
    X1, obj1 = grp.dist_chordal_initialization(_options())
    assert obj1.shape == obj2.shape
    np.testing.assert_allclose(obj2, obj1, rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(X2[0], X1, atol=1e-8)
