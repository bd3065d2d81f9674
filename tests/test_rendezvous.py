"""The rank rendezvous of the dist_pgo driver (dpgo_amd/csrc/rdv.h): the RCCL id reaches every rank through a private
directory with a nonce handshake; files left behind by an earlier, dead run are never taken for this run's id
(a stale id would make ncclCommInitRank wait forever), and a missing rank is an error after the time-out, not a hang."""
import os
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HARNESS = r'''
#include "rdv.h"
int main(int argc, char **argv) {
  const int rank = atoi(argv[2]), world = atoi(argv[3]);
  unsigned char id[128];
  for (int i = 0; i < 128; i++) id[i] = rank == 0 ? (unsigned char)(atoi(argv[4]) + i) : 0;
  if (dpgo_rdv::rendezvous(argv[1], rank, world, id) != 0) return 3;
  for (int i = 0; i < 128; i++) if (id[i] != (unsigned char)(atoi(argv[4]) + i)) return 4;
  return 0;
}
'''


@pytest.fixture(scope="module")
def harness():
    d = tempfile.mkdtemp(prefix="dpgo_rdv_test_")
    src = os.path.join(d, "h.cpp")
    open(src, "w").write(HARNESS)
    exe = os.path.join(d, "h")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "dpgo_amd", "csrc"), src, "-o", exe, "-pthread"])
    return exe


def _run(exe, d, world, tag, env=None, ranks=None):
    ps = [subprocess.Popen([exe, d, str(r), str(world), str(tag)], env=env) for r in (ranks if ranks is not None else range(world))]
    return [p.wait(timeout=60) for p in ps]


def test_rendezvous_delivers_the_id(harness, tmp_path):
    d = str(tmp_path / "rdv")
    assert _run(harness, d, 4, 17) == [0, 0, 0, 0]
    assert not os.path.exists(d) or os.listdir(d) == []      # nothing is left behind


def test_stale_files_of_a_dead_run_are_ignored(harness, tmp_path):
    d = str(tmp_path / "rdv")
    os.mkdir(d, 0o700)
    # what a killed earlier run may have left: an answer with an old id for rank 1, a hello of its rank 2
    open(os.path.join(d, "id.1"), "wb").write(b"\x01" * 8 + b"\xee" * 128)
    open(os.path.join(d, "hello.2"), "wb").write(b"\x02" * 8)
    assert _run(harness, d, 3, 42) == [0, 0, 0]              # every rank got THIS run's id (tag 42), not 0xee...


def test_missing_rank_times_out(harness, tmp_path):
    d = str(tmp_path / "rdv")
    env = dict(os.environ, DPGO_RDV_TIMEOUT="1.5")
    assert _run(harness, d, 3, 5, env=env, ranks=[0, 1]) == [3, 0]          # rank 1 is answered, rank 0 misses rank 2
    rc = _run(harness, d, 2, 5, env=env, ranks=[1])
    assert rc == [3]                                          # no rank 0: error, not a hang
    assert not os.path.exists(os.path.join(d, "hello.1"))    # and its hello is removed on the way out


def test_directory_must_be_private(harness, tmp_path):
    d = str(tmp_path / "rdv")
    os.mkdir(d, 0o755)
    os.chmod(d, 0o755)
    assert _run(harness, d, 1, 1, ranks=[0]) == [3]
