"""dpgo_amd -- MI355X-native DPGO hot path (ctypes binding of libdpgo_amd.so).

The host-side mirror of the reference's C++ interface for the per-node MM / AMM
inner step of ``dist_pgo``:

  reference (C++/DPGO/include/DPGO)              here
  ---------------------------------------------  --------------------------------
  DPGO::read_g2o            DPGO_utils.h:49-51    read_g2o(filename, num_nodes) -> Graph
  DPGO::Options             DPGO_types.h:78-201   Options (same field names/defaults)
  DPGOHash(node, meas, opt) DPGOHash.h:13-107     NodeGroup(graph, node_ids, opt)[k] -> DPGOHash view
    initialize/update/iterate/communicate           same names, return 0 / -1
    results()               DPGO_types.h:204-322  .results() (scalars), .Xk(), .Xak()
  dist_pgo driver loop      dist_pgo.cpp:446-531  DistPGO

All compute runs in hand-written HIP kernels behind the C ABI of
include/dpgo_amd.h.  There is NO CPU fallback: creating a NodeGroup without a
HIP device raises.  numpy is used only to hold host matrices.
"""
from __future__ import annotations

import ctypes as C
import os

import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (DPGO_AMD_LIB: A/B builds of the same library from tools/build_variant.sh; never a different implementation)
LIB_PATH = os.environ.get("DPGO_AMD_LIB") or os.path.join(_HERE, "libdpgo_amd.so")

LOSS_NONE, LOSS_HUBER, LOSS_GM, LOSS_WELSCH = 0, 1, 2, 3
LOSS_NAMES = {"trivial": 0, "none": 0, "huber": 1, "gm": 2, "welsch": 3}
SCHEME_MM, SCHEME_AMM = 0, 1
RESCALE_STATIC, RESCALE_DYNAMIC = 0, 1
PRECON_NONE, PRECON_JACOBI, PRECON_ICHOL, PRECON_REG_CHOLESKY = 0, 1, 2, 3


class Options(C.Structure):
    """DPGO::Options (C++/DPGO/include/DPGO/DPGO_types.h:78-201)."""
    _fields_ = [
        ("scheme", C.c_int), ("regularizer", C.c_double), ("accepted_delta", C.c_double),
        ("eta", C.c_double * 2), ("psi", C.c_double), ("phi", C.c_double),
        ("max_soft_restart_hits", C.c_int * 2), ("oscillation_cnt_period", C.c_int),
        ("max_oscillations", C.c_int), ("loss", C.c_int), ("loss_reg", C.c_double),
        ("rescale", C.c_int), ("max_rescale_count", C.c_int), ("grad_norm_tol", C.c_double), ("rel_func_decrease_tol", C.c_double), ("stepsize_tol", C.c_double),
        ("max_iterations", C.c_int), ("max_iterations_accepted", C.c_int),
        ("reg_Cholesky_precon_max_condition_number", C.c_double),
        ("preconditioned_grad_norm_tol", C.c_double), ("max_tCG_iterations", C.c_int),
        ("STPCG_kappa", C.c_double), ("STPCG_theta", C.c_double), ("preconditioner", C.c_int),
        ("verbose", C.c_int),
    ]

    def __init__(self, **kw):
        super().__init__()
        lib().dpgo_options_default(C.byref(self))
        for k, v in kw.items():
            setattr(self, k, v)

    @staticmethod
    def driver(loss=LOSS_NONE, accelerated=True, **kw):
        """The hard-coded options of C++/examples/dist_pgo.cpp:103-120."""
        o = Options()
        lib().dpgo_options_driver(C.byref(o), int(loss), int(bool(accelerated)))
        for k, v in kw.items():
            setattr(o, k, v)
        return o


def p2p_plan(rank, exported, needed):
    """The neighbour-to-neighbour exchange plan of `rank` (comm.cpp::p2p_plan through dpgo_debug_p2p_plan).
    exported[r] / needed[r]: lists of (node, pose) keys of every rank.  Returns (peers, send_keys, recv_keys) with
    peers = [(rank, send_off, send_cnt, recv_off, recv_cnt), ...]."""
    n = len(exported)
    def flat(lists):
        cnt = np.asarray([len(l) for l in lists], np.int32)
        nodes = np.asarray([k[0] for l in lists for k in l] or [0], np.int32)
        poses = np.asarray([k[1] for l in lists for k in l] or [0], np.int32)
        return cnt, nodes, poses
    ec, en, ep = flat(exported)
    nc, nn_, npz = flat(needed)
    sizes = np.zeros(3, np.int32)
    if lib().dpgo_debug_p2p_plan(rank, n, _ip(ec), _ip(en), _ip(ep), _ip(nc), _ip(nn_), _ip(npz), None, None, None, _ip(sizes)) != 0:
        raise ValueError("p2p_plan")
    peers = np.zeros(max(5 * int(sizes[0]), 1), np.int32)
    sk = np.zeros(max(2 * int(sizes[1]), 1), np.int32)
    rk = np.zeros(max(2 * int(sizes[2]), 1), np.int32)
    lib().dpgo_debug_p2p_plan(rank, n, _ip(ec), _ip(en), _ip(ep), _ip(nc), _ip(nn_), _ip(npz), _ip(peers), _ip(sk), _ip(rk), _ip(sizes))
    P = [tuple(int(v) for v in peers[5 * i:5 * i + 5]) for i in range(int(sizes[0]))]
    S = [(int(sk[2 * i]), int(sk[2 * i + 1])) for i in range(int(sizes[1]))]
    R = [(int(rk[2 * i]), int(rk[2 * i + 1])) for i in range(int(sizes[2]))]
    return P, S, R


class DChordalOptions(C.Structure):
    """DChordal::Options::reg_G + the driver's stage schedule (dist_pgo.cpp:205,274,344,383) + stage-0 length."""
    _fields_ = [("iters", C.c_int * 4), ("local_iters", C.c_int), ("reg_G", C.c_double)]

    def __init__(self, **kw):
        super().__init__()
        lib().dpgo_dchordal_options_default(C.byref(self))
        for k, v in kw.items():
            setattr(self, k, v)


class Results(C.Structure):
    """Scalar part of DPGOResult (C++/DPGO/include/DPGO/DPGO_types.h:204-322)."""
    _fields_ = [
        ("updated", C.c_int), ("iters", C.c_int), ("gradFnorm", C.c_double), ("fobjE", C.c_double),
        ("Fk", C.c_double * 2), ("Gk", C.c_double), ("Gkh", C.c_double), ("fobj", C.c_double),
        ("f", C.c_double), ("gamma", C.c_double), ("s", C.c_double * 2),
        ("soft_restart_hits", C.c_int * 2), ("num_oscillations", C.c_int), ("refined", C.c_int),
        ("tnt_status", C.c_int), ("tnt_inner_iterations", C.c_int), ("restarts", C.c_int),
    ]


_lib = None
_DP = C.POINTER(C.c_double)
_IP = C.POINTER(C.c_int)

# every symbol of include/dpgo_amd.h: (restype, argtypes)
SYMBOLS = {
    "dpgo_options_default": (None, [C.POINTER(Options)]),
    "dpgo_options_driver": (None, [C.POINTER(Options), C.c_int, C.c_int]),
    "dpgo_read_g2o": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]),
    "dpgo_graph_from_edges": (C.c_int, [C.c_int, C.c_int, C.c_int, _IP, _IP, _DP, _DP, _DP, _DP, C.c_int,
                                        C.POINTER(C.c_void_p)]),
    "dpgo_graph_free": (None, [C.c_void_p]),
    "dpgo_graph_info": (C.c_int, [C.c_void_p, _IP, _IP, _IP, _IP]),
    "dpgo_graph_edges": (C.c_int, [C.c_void_p, _IP, _IP, _DP, _DP, _DP, _DP]),
    "dpgo_graph_node_sizes": (C.c_int, [C.c_void_p, C.c_int, _IP, _IP, _IP, _IP]),
    "dpgo_graph_node_neighbours": (C.c_int, [C.c_void_p, C.c_int, _IP, _IP]),
    "dpgo_graph_node_offset": (C.c_int, [C.c_void_p, C.c_int]),
    "dpgo_graph_exchange_plan": (C.c_int, [C.c_void_p, _IP, C.c_int, _IP, _IP, _IP, _IP, _IP]),
    "dpgo_chordal_initialization": (C.c_int, [C.c_void_p, _DP, C.c_int]),
    "dpgo_graph_node_maps": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _IP, _IP, _IP, _IP, _IP]),
    "dpgo_write_g2o": (C.c_int, [C.c_void_p, _DP, C.c_int, C.c_char_p]),
    "dpgo_group_evaluate": (C.c_int, [C.c_void_p, _DP, C.c_int, _DP, _DP, _DP, C.c_int]),
    "dpgo_group_set_options": (C.c_int, [C.c_void_p, C.POINTER(Options)]),
    "dpgo_group_get_options": (C.c_int, [C.c_void_p, C.POINTER(Options)]),
    "dpgo_comm_unique_id": (C.c_int, [C.c_void_p]),
    "dpgo_comm_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "dpgo_comm_free": (None, [C.c_void_p]),
    "dpgo_comm_exchange": (C.c_int, [C.c_void_p]),
    "dpgo_comm_allreduce_sum": (C.c_int, [C.c_void_p, _DP, C.c_long]),
    "dpgo_comm_barrier": (C.c_int, [C.c_void_p]),
    "dpgo_comm_exchange_kind": (C.c_int, [C.c_void_p]),
    "dpgo_comm_bytes_sent": (C.c_long, [C.c_void_p]),
    "dpgo_comm_self_exchange": (C.c_int, [C.c_void_p]),
    "dpgo_comm_create_self": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "dpgo_comm_enable_timing": (C.c_int, [C.c_void_p]),
    "dpgo_comm_exchange_time": (C.c_int, [C.c_void_p, _DP, C.POINTER(C.c_long)]),
    "dpgo_debug_comm_p2p_self": (C.c_int, [C.c_void_p]),
    "dpgo_host_pack_sent": (C.c_int, [C.c_void_p, _IP, C.c_int, _DP, C.c_int, _DP]),
    "dpgo_host_unpack_recv": (C.c_int, [C.c_void_p, _IP, C.c_int, C.c_int, C.c_int, C.c_int, _IP, _IP, _IP, _DP, _DP,
                                        C.c_int]),
    "dpgo_dchordal_options_default": (None, [C.c_void_p]),
    "dpgo_group_dist_chordal_initialization": (C.c_int, [C.c_void_p, C.c_void_p, _DP, C.c_int, _DP, C.c_int, _DP, _IP]),
    "dpgo_group_create": (C.c_int, [C.c_void_p, _IP, C.c_int, C.POINTER(Options), C.c_int, C.POINTER(C.c_void_p)]),
    "dpgo_group_free": (None, [C.c_void_p]),
    "dpgo_group_initialize": (C.c_int, [C.c_void_p, C.c_int, _DP, C.c_int]),
    "dpgo_group_initialize_global": (C.c_int, [C.c_void_p, _DP, C.c_int]),
    "dpgo_group_update": (C.c_int, [C.c_void_p, _IP, C.c_int]),
    "dpgo_group_iterate": (C.c_int, [C.c_void_p, _IP, C.c_int]),
    "dpgo_group_communicate_local": (C.c_int, [C.c_void_p]),
    "dpgo_group_step": (C.c_int, [C.c_void_p, C.c_void_p]),
    "dpgo_group_message_sizes": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _IP, _IP]),
    "dpgo_group_send": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _DP, C.c_int]),
    "dpgo_group_receive": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _DP, C.c_int]),
    "dpgo_group_set_collectives": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dpgo_group_star_initialize": (C.c_int, [C.c_void_p, _DP, C.c_int]),
    "dpgo_group_star_update": (C.c_int, [C.c_void_p]),
    "dpgo_group_star_iterate": (C.c_int, [C.c_void_p]),
    "dpgo_group_star_state": (C.c_int, [C.c_void_p, _DP, _DP, _DP, _IP]),
    "dpgo_group_num_sent": (C.c_int, [C.c_void_p]),
    "dpgo_group_sent_keys": (C.c_int, [C.c_void_p, _IP, _IP]),
    "dpgo_group_set_recv_layout": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _IP, _IP, _IP]),
    "dpgo_group_pack_sent": (C.c_int, [C.c_void_p, C.c_void_p]),
    "dpgo_group_unpack_recv": (C.c_int, [C.c_void_p, C.c_void_p]),
    "dpgo_group_get_Xk": (C.c_int, [C.c_void_p, C.c_int, _DP, C.c_int]),
    "dpgo_group_get_Xak": (C.c_int, [C.c_void_p, C.c_int, _DP, C.c_int]),
    "dpgo_group_scatter_global": (C.c_int, [C.c_void_p, _DP, C.c_int]),
    "dpgo_group_results": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(Results)]),
    "dpgo_group_node_id": (C.c_int, [C.c_void_p, C.c_int]),
    "dpgo_group_sync": (C.c_int, [C.c_void_p]),
    "dpgo_group_stream": (C.c_void_p, [C.c_void_p]),
    "dpgo_prof_enable": (C.c_int, [C.c_int]),
    "dpgo_prof_num_kinds": (C.c_int, []),
    "dpgo_prof_kind_name": (C.c_char_p, [C.c_int]),
    "dpgo_prof_collect": (C.c_int, [_DP, _DP, C.POINTER(C.c_long)]),
    "dpgo_prof_collect_operands": (C.c_int, [_DP]),
    "dpgo_group_solver_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_long), C.POINTER(C.c_long), _IP, _IP]),
    "dpgo_group_graph_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_long), C.POINTER(C.c_long), C.POINTER(C.c_long)]),
    "dpgo_debug_node_matrix": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(Options), C.c_char_p, _IP, _IP, _DP]),
    "dpgo_debug_node_proximal": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(Options), _DP, _DP, _DP]),
    "dpgo_debug_spd_solve": (C.c_int, [C.c_int, _IP, _IP, _DP, _DP, C.c_int, C.c_int]),
    "dpgo_debug_spd_stats": (C.c_int, [C.c_int, _IP, _IP, _DP, C.c_int, C.POINTER(C.c_long), _IP, _IP]),
    "dpgo_debug_p2p_plan": (C.c_int, [C.c_int, C.c_int, _IP, _IP, _IP, _IP, _IP, _IP, _IP, _IP, _IP, _IP]),
    "dpgo_group_debug_apply": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, _DP, C.c_int, _DP, C.c_int]),
}


def lib():
    """Load libdpgo_amd.so (built by dpgo_amd/csrc/Makefile).  Fails loudly if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("dpgo_amd: %s not found -- build it with __graft_entry__.build() "
                               "(make -C dpgo_amd/csrc); there is no fallback path" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(_DP)


def _ip(a):
    return a.ctypes.data_as(_IP)


def _fcol(X):
    """Column-major float64 copy/view and its leading dimension."""
    X = np.asfortranarray(X, dtype=np.float64)
    return X, X.shape[0]


class Graph:
    """Result of DPGO::read_g2o: measurements partitioned over num_nodes."""

    def __init__(self, handle):
        self._h = handle
        d, n, k, m = (C.c_int() for _ in range(4))
        lib().dpgo_graph_info(self._h, C.byref(d), C.byref(n), C.byref(k), C.byref(m))
        self.d, self.num_poses, self.num_nodes, self.num_edges = d.value, n.value, k.value, m.value

    def __del__(self):
        if getattr(self, "_h", None):
            lib().dpgo_graph_free(self._h)
            self._h = None

    def edges(self):
        d, m = self.d, self.num_edges
        I, J = np.empty(m, np.int32), np.empty(m, np.int32)
        R, t = np.empty((m, d, d)), np.empty((m, d))
        kap, tau = np.empty(m), np.empty(m)
        lib().dpgo_graph_edges(self._h, _ip(I), _ip(J), _dp(R), _dp(t), _dp(kap), _dp(tau))
        return I, J, R, t, kap, tau

    def node_sizes(self, node):
        v = [C.c_int() for _ in range(4)]
        if lib().dpgo_graph_node_sizes(self._h, node, *[C.byref(x) for x in v]) != 0:
            raise ValueError("node %d" % node)
        return tuple(x.value for x in v)   # n0, n1, m0, m1

    def node_neighbours(self, node):
        n1 = self.node_sizes(node)[1]
        a, b = np.empty(n1, np.int32), np.empty(n1, np.int32)
        lib().dpgo_graph_node_neighbours(self._h, node, _ip(a), _ip(b))
        return a, b

    def node_offset(self, node):
        return lib().dpgo_graph_node_offset(self._h, node)

    def exchange_plan(self, node_ids):
        """(sent (node, pose) keys, recv (node, pose) keys) of a group hosting node_ids (host only)."""
        ids = np.asarray(list(node_ids), np.int32)
        cnt = np.zeros(2, np.int32)
        if lib().dpgo_graph_exchange_plan(self._h, _ip(ids), len(ids), None, None, None, None, _ip(cnt)) != 0:
            raise ValueError("exchange_plan")
        sn, sp_, rn, rp = (np.empty(c, np.int32) for c in (cnt[0], cnt[0], cnt[1], cnt[1]))
        lib().dpgo_graph_exchange_plan(self._h, _ip(ids), len(ids), _ip(sn), _ip(sp_), _ip(rn), _ip(rp), _ip(cnt))
        return (sn, sp_), (rn, rp)

    def node_maps(self, node, which):
        """DPGOProblem::index() / sent() / recv() (which = "index" | "sent" | "recv"):
        list of ((node, pose), (block, k)) in map order."""
        w = {"index": 0, "sent": 1, "recv": 2}[which]
        cnt = C.c_int()
        if lib().dpgo_graph_node_maps(self._h, node, w, None, None, None, None, C.byref(cnt)) != 0:
            raise ValueError("node_maps")
        a, b, c, d = (np.empty(cnt.value, np.int32) for _ in range(4))
        lib().dpgo_graph_node_maps(self._h, node, w, _ip(a), _ip(b), _ip(c), _ip(d), C.byref(cnt))
        return [((int(a[i]), int(b[i])), (int(c[i]), int(d[i]))) for i in range(cnt.value)]

    def write_g2o(self, filename, X=None):
        """VERTEX_* lines from X (optional) + EDGE_* lines; returns 0 / -1."""
        if X is None:
            return lib().dpgo_write_g2o(self._h, None, 0, os.fsencode(filename))
        X, ld = _fcol(X)
        return lib().dpgo_write_g2o(self._h, _dp(X), ld, os.fsencode(filename))

    def host_pack_sent(self, node_ids, X):
        """Records of the poses a group hosting node_ids exports (key order), from a global X: what
        dpgo_group_pack_sent puts into the device buffer (host version, no GPU)."""
        ids = np.asarray(list(node_ids), np.int32)
        (sn, _), _ = self.exchange_plan(ids)
        X, ld = _fcol(X)
        buf = np.zeros(max(len(sn), 1) * (self.d + 1) * self.d)
        n = lib().dpgo_host_pack_sent(self._h, _ip(ids), len(ids), _dp(X), ld, _dp(buf))
        if n < 0:
            raise RuntimeError("dpgo_host_pack_sent failed")
        return buf[:n * (self.d + 1) * self.d]

    def host_unpack_recv(self, node_ids, node, stride, keys_per_rank, gathered, Z):
        """Fill the neighbour rows of node's Z ((d+1)(n0+n1) x d, F order) from the gathered buffers (host version
        of dpgo_group_unpack_recv); returns the number of poses written."""
        ids = np.asarray(list(node_ids), np.int32)
        counts = np.asarray([len(k[0]) for k in keys_per_rank], np.int32)
        nodes = np.ascontiguousarray(np.concatenate([np.asarray(k[0], np.int32) for k in keys_per_rank]), np.int32)
        poses = np.ascontiguousarray(np.concatenate([np.asarray(k[1], np.int32) for k in keys_per_rank]), np.int32)
        gathered = np.ascontiguousarray(gathered, np.float64)
        assert Z.flags.f_contiguous
        n = lib().dpgo_host_unpack_recv(self._h, _ip(ids), len(ids), int(node), len(counts), int(stride), _ip(counts),
                                        _ip(nodes), _ip(poses), _dp(gathered), _dp(Z), Z.shape[0])
        if n < 0:
            raise RuntimeError("dpgo_host_unpack_recv failed")
        return n

    def chordal_initialization(self):
        """Centralised chordal init (dist_pgo.cpp:416-444): X, (d+1)N x d, reference layout."""
        X = np.zeros(((self.d + 1) * self.num_poses, self.d), order="F")
        if lib().dpgo_chordal_initialization(self._h, _dp(X), X.shape[0]) != 0:
            raise RuntimeError("chordal initialisation failed")
        return X

    def node_matrix(self, node, opt, name):
        """Assembled operator in the reference's row order, as a scipy COO matrix (test hook)."""
        import scipy.sparse as sp
        cnt = lib().dpgo_debug_node_matrix(self._h, node, C.byref(opt), name.encode(), None, None, None)
        if cnt < 0:
            raise ValueError(name)
        r, c, v = np.empty(cnt, np.int32), np.empty(cnt, np.int32), np.empty(cnt)
        lib().dpgo_debug_node_matrix(self._h, node, C.byref(opt), name.encode(), _ip(r), _ip(c), _dp(v))
        n0, n1, _, _ = self.node_sizes(node)
        D1 = self.d + 1
        shape = {"G": (D1 * n0, D1 * n0), "D": (D1 * n0, D1 * n0), "S": (D1 * n0, D1 * (n0 + n1))}.get(
            name, (D1 * (n0 + n1), D1 * (n0 + n1)))
        return sp.coo_matrix((v, (r, c)), shape=shape).tocsr()

    def node_proximal(self, node, opt):
        n0 = self.node_sizes(node)[0]
        d = self.d
        T, N, V = np.empty(n0), np.empty((n0, d)), np.empty((n0, d, d))
        lib().dpgo_debug_node_proximal(self._h, node, C.byref(opt), _dp(T), _dp(N), _dp(V))
        return T, N, V


def read_g2o(filename, num_nodes):
    """DPGO::read_g2o (C++/DPGO/src/DPGO_utils.cpp:140-202)."""
    h = C.c_void_p()
    if lib().dpgo_read_g2o(os.fsencode(filename), int(num_nodes), C.byref(h)) != 0:
        raise IOError("read_g2o failed for %s" % filename)
    return Graph(h)


def graph_from_edges(d, num_poses, I, J, R, t, kappa, tau, num_nodes):
    I, J = np.ascontiguousarray(I, np.int32), np.ascontiguousarray(J, np.int32)
    R, t = np.ascontiguousarray(R, np.float64), np.ascontiguousarray(t, np.float64)
    kappa, tau = np.ascontiguousarray(kappa, np.float64), np.ascontiguousarray(tau, np.float64)
    h = C.c_void_p()
    if lib().dpgo_graph_from_edges(d, num_poses, len(I), _ip(I), _ip(J), _dp(R), _dp(t), _dp(kappa), _dp(tau),
                                   int(num_nodes), C.byref(h)) != 0:
        raise ValueError("graph_from_edges failed")
    return Graph(h)


def spd_solve_host(A_csr, B, leaf=32):
    """Host multifrontal factor + solve (test hook for the solver set-up path)."""
    A = A_csr.tocsr()
    A.sort_indices()
    ptr, col = A.indptr.astype(np.int32), A.indices.astype(np.int32)
    val = A.data.astype(np.float64)
    X = np.ascontiguousarray(B, np.float64).copy()
    if X.ndim == 1:
        X = X[:, None]
    if lib().dpgo_debug_spd_solve(A.shape[0], _ip(ptr), _ip(col), _dp(val), _dp(X), X.shape[1], leaf) != 0:
        raise RuntimeError("spd solve failed")
    return X


def prof_enable(on):
    lib().dpgo_prof_enable(int(bool(on)))


def prof_collect():
    """{kernel family: (total_ms, algorithmic_bytes, launches)} since prof_enable(True)."""
    n = lib().dpgo_prof_num_kinds()
    ms, by, cnt = np.zeros(n), np.zeros(n), np.zeros(n, np.int64)
    lib().dpgo_prof_collect(_dp(ms), _dp(by), cnt.ctypes.data_as(C.POINTER(C.c_long)))
    return {lib().dpgo_prof_kind_name(k).decode(): (float(ms[k]), float(by[k]), int(cnt[k])) for k in range(n)}


def prof_collect_operands():
    """{kernel family: bytes of every operand its launches moved, counted one by one} -- non-zero for the fused passes
    (k_inter, k_proximal), whose duties SURVEY 8(d)'s per-unit formula does not price."""
    n = lib().dpgo_prof_num_kinds()
    ob = np.zeros(n)
    lib().dpgo_prof_collect_operands(_dp(ob))
    return {lib().dpgo_prof_kind_name(k).decode(): float(ob[k]) for k in range(n)}


def spd_stats(A_csr, leaf):
    A = A_csr.tocsr()
    A.sort_indices()
    ptr, col, val = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)
    nnz, lv, mf = C.c_long(), C.c_int(), C.c_int()
    if lib().dpgo_debug_spd_stats(A.shape[0], _ip(ptr), _ip(col), _dp(val), leaf, C.byref(nnz), C.byref(lv), C.byref(mf)) != 0:
        raise RuntimeError("spd_stats failed")
    return nnz.value, lv.value, mf.value


class NodeGroup:
    """The DPGOHash objects of the nodes hosted by one GPU (one process)."""

    def __init__(self, graph, node_ids, options, device=0):
        self.graph, self.options = graph, options
        self.node_ids = [int(a) for a in node_ids]
        ids = np.asarray(self.node_ids, np.int32)
        h = C.c_void_p()
        if lib().dpgo_group_create(graph._h, _ip(ids), len(ids), C.byref(options), int(device), C.byref(h)) != 0:
            raise RuntimeError("dpgo_group_create failed (no HIP device, or inconsistent input); "
                               "the DPGO hot path has no CPU fallback")
        self._h = h
        self.d = graph.d
        self.sizes = [graph.node_sizes(a) for a in self.node_ids]

    def __del__(self):
        if getattr(self, "_h", None):
            lib().dpgo_group_free(self._h)
            self._h = None

    def __len__(self):
        return len(self.node_ids)

    def __getitem__(self, k):
        return DPGOHash(self, k)

    def _sel(self, locals_):
        if locals_ is None:
            return None, 0
        a = np.asarray(list(locals_), np.int32)
        return _ip(a), len(a)

    def initialize_global(self, X):
        X, ld = _fcol(X)
        return lib().dpgo_group_initialize_global(self._h, _dp(X), ld)

    def update(self, locals_=None):
        p, n = self._sel(locals_)
        return lib().dpgo_group_update(self._h, p, n)

    def iterate(self, locals_=None):
        p, n = self._sel(locals_)
        return lib().dpgo_group_iterate(self._h, p, n)

    def communicate_local(self):
        return lib().dpgo_group_communicate_local(self._h)

    def p2p_self_check(self):
        """Test hook (dpgo_debug_comm_p2p_self): the grouped ncclSend / ncclRecv path of the boundary exchange on a one-rank
        communicator that is its own peer, for the rows this group exports."""
        return lib().dpgo_debug_comm_p2p_self(self._h)

    def step(self, comm=None):
        """iterate() of every node, the boundary exchange of `comm` (a Comm of this group) if any, communicate(), update():
        the body of the driver's loop (dist_pgo.cpp:496-521) in one native call."""
        return lib().dpgo_group_step(self._h, comm._h if comm is not None else None)

    def sync(self):
        return lib().dpgo_group_sync(self._h)

    def stream(self):
        return lib().dpgo_group_stream(self._h)

    # boundary exchange across groups
    def sent_keys(self):
        n = lib().dpgo_group_num_sent(self._h)
        a, b = np.empty(n, np.int32), np.empty(n, np.int32)
        lib().dpgo_group_sent_keys(self._h, _ip(a), _ip(b))
        return a, b

    def set_recv_layout(self, stride, keys_per_rank):
        counts = np.asarray([len(k[0]) for k in keys_per_rank], np.int32)
        nodes = np.ascontiguousarray(np.concatenate([k[0] for k in keys_per_rank]) if len(counts) else [], np.int32)
        poses = np.ascontiguousarray(np.concatenate([k[1] for k in keys_per_rank]) if len(counts) else [], np.int32)
        return lib().dpgo_group_set_recv_layout(self._h, len(counts), int(stride), _ip(counts), _ip(nodes), _ip(poses))

    def set_collectives(self, send_ptr, gathered_ptr, allgather, allreduce):
        """Lend the group an all-gather of its boundary buffer and a sum over the groups (AMM-PGO* with the
        nodes spread over several processes).  allgather() -> 0; allreduce(numpy array) -> 0, in place."""
        def _ag(_user):
            try:
                return int(allgather() or 0)
            except Exception as e:      # an exception must not unwind through the C frames
                sys.stderr.write("allgather callback: %r\n" % (e,))
                return -1

        def _ar(_user, vals, n):
            try:
                a = np.ctypeslib.as_array(vals, shape=(n,))
                return int(allreduce(a) or 0)
            except Exception as e:
                sys.stderr.write("allreduce callback: %r\n" % (e,))
                return -1
        self._cb = (C.CFUNCTYPE(C.c_int, C.c_void_p)(_ag), C.CFUNCTYPE(C.c_int, C.c_void_p, _DP, C.c_int)(_ar))
        return lib().dpgo_group_set_collectives(self._h, C.c_void_p(send_ptr), C.c_void_p(gathered_ptr),
                                                C.cast(self._cb[0], C.c_void_p), C.cast(self._cb[1], C.c_void_p), None)

    def pack_sent(self, dev_ptr):
        return lib().dpgo_group_pack_sent(self._h, C.c_void_p(dev_ptr))

    def unpack_recv(self, dev_ptr):
        return lib().dpgo_group_unpack_recv(self._h, C.c_void_p(dev_ptr))

    def scatter_global(self, X):
        assert X.flags.f_contiguous
        return lib().dpgo_group_scatter_global(self._h, _dp(X), X.shape[0])

    def connect_torch(self):
        """Connect this group to the groups of the other processes of an initialised torch.distributed process group
        (one process per GPU: backend nccl = RCCL; gloo: staged through the host, lets several processes share a GPU
        in tests): the recv lay-out of the boundary poses and the two collectives the library needs when the nodes
        of the graph are spread over several groups (AMM-PGO*, the distributed chordal initialisation, global
        evaluations).  Returns (dist, torch, send, gathered, stream, allgather)."""
        import torch
        import torch.distributed as dist
        world, RS = dist.get_world_size(), (self.d + 1) * self.d
        host = dist.get_backend() == "gloo"          # gloo: staged through pinned host tensors
        keys = self.sent_keys()
        allkeys = [None] * world
        dist.all_gather_object(allkeys, (keys[0].tolist(), keys[1].tolist()))
        stride = max(max(len(k[0]) for k in allkeys), 1)
        self.set_recv_layout(stride, [(np.asarray(k[0], np.int32), np.asarray(k[1], np.int32)) for k in allkeys])
        send = torch.zeros(stride * RS, dtype=torch.float64, device="cuda")
        gathered = torch.zeros(world * stride * RS, dtype=torch.float64, device="cuda")
        send_h = torch.zeros(stride * RS, dtype=torch.float64) if host else None
        gathered_h = torch.zeros(world * stride * RS, dtype=torch.float64) if host else None
        ext = torch.cuda.ExternalStream(self.stream())
        torch.cuda.synchronize()

        def allgather():
            with torch.cuda.stream(ext):
                if host:
                    send_h.copy_(send)
                    dist.all_gather_into_tensor(gathered_h, send_h)
                    gathered.copy_(gathered_h)
                else:
                    dist.all_gather_into_tensor(gathered, send)
            return 0

        def allreduce(vals):
            t = torch.from_numpy(vals.copy())
            if not host:
                t = t.cuda()
            dist.all_reduce(t)                       # same reduction order on every rank: identical branches
            vals[:] = t.cpu().numpy()
            return 0
        if self.set_collectives(send.data_ptr(), gathered.data_ptr(), allgather, allreduce) != 0:
            raise RuntimeError("dpgo_group_set_collectives failed")
        self._torch_link = (dist, torch, send, gathered, ext, allgather)
        return self._torch_link

    def dist_chordal_initialization(self, options=None, X_local=None):
        """The --dist_init true branch of dist_pgo (dist_pgo.cpp:144-416): returns (X, objectives) -- the initial
        guess ((d+1)N x d) and the stage objectives sampled every 20 iterations."""
        o = options or DChordalOptions()
        N, d = self.graph.num_poses, self.d
        X = np.zeros(((d + 1) * N, d), order="F")
        cap = sum((o.iters[k] + 19) // 20 for k in range(4))
        obj, cnt = np.zeros(max(cap, 1)), C.c_int(cap)
        xl, ldl = (None, 0)
        if X_local is not None:
            Xl, ldl = _fcol(X_local)
            xl = _dp(Xl)
        if lib().dpgo_group_dist_chordal_initialization(self._h, C.byref(o), xl, ldl, _dp(X), X.shape[0], _dp(obj),
                                                        C.byref(cnt)) != 0:
            raise RuntimeError("distributed chordal initialisation failed")
        return X, obj[:cnt.value]

    def evaluate(self, X, want_grad=False):
        """DPGOStar::evaluate_f / evaluate_grad at an arbitrary global X: (F, |grad F|^2[, grad]) summed over
        this group's nodes (over all groups when collectives are attached)."""
        X, ld = _fcol(X)
        F, g2 = C.c_double(), C.c_double()
        G = np.zeros_like(X, order="F") if want_grad else None
        if lib().dpgo_group_evaluate(self._h, _dp(X), ld, C.byref(F), C.byref(g2), _dp(G) if want_grad else None,
                                     ld if want_grad else 0) != 0:
            raise RuntimeError("dpgo_group_evaluate failed")
        return (F.value, g2.value, G) if want_grad else (F.value, g2.value)

    def set_options(self, options):
        rc = lib().dpgo_group_set_options(self._h, C.byref(options))
        if rc == 0:
            self.options = options
        return rc

    def get_options(self):
        o = Options()
        lib().dpgo_group_get_options(self._h, C.byref(o))
        return o

    def results(self, k):
        r = Results()
        lib().dpgo_group_results(self._h, k, C.byref(r))
        return r

    def solver_stats(self):
        a, b, c, d = C.c_long(), C.c_long(), C.c_int(), C.c_int()
        lib().dpgo_group_solver_stats(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d))
        return dict(nnz_tt=a.value, nnz_rr=b.value, levels_tt=c.value, levels_rr=d.value)

    def graph_stats(self):
        """Segments of the iteration replayed from captured graphs, graphs captured, segments launched eagerly."""
        a, b, c = C.c_long(), C.c_long(), C.c_long()
        lib().dpgo_group_graph_stats(self._h, C.byref(a), C.byref(b), C.byref(c))
        return dict(replays=a.value, captures=b.value, eager=c.value)

    def debug_apply(self, k, op, X, out_rows):
        X, ld = _fcol(X)
        out = np.zeros((out_rows, self.d), order="F")
        if lib().dpgo_group_debug_apply(self._h, k, op.encode(), _dp(X), ld, _dp(out), out.shape[0]) != 0:
            raise RuntimeError("debug_apply(%s) failed" % op)
        return out


class Comm:
    """The RCCL communicator of one group (one process per GPU): dpgo_comm_* of include/dpgo_amd.h.  `bcast`
    carries the 128-byte unique id from rank 0 to the others: a function bytes -> bytes (e.g. through
    torch.distributed's gloo store, MPI, a file); with one rank it is not needed."""

    def __init__(self, group, rank, nranks, bcast=None):
        self.group, self.rank, self.nranks = group, rank, nranks
        if nranks > 1 and bcast is None:
            raise ValueError("nranks > 1 needs a broadcast function for the unique id")
        idb = C.create_string_buffer(128)
        raw = b""
        if rank == 0 and lib().dpgo_comm_unique_id(idb) == 0:
            raw = bytes(idb.raw)
        if nranks > 1:
            raw = bcast(raw)          # (an empty id tells every rank that rank 0 could not get one: nobody is left waiting)
        if len(raw) != 128:
            raise RuntimeError("dpgo_comm_unique_id failed (RCCL not available)")
        idb = C.create_string_buffer(raw, 128)
        h = C.c_void_p()
        if lib().dpgo_comm_create(group._h, rank, nranks, idb, C.byref(h)) != 0:
            raise RuntimeError("dpgo_comm_create failed")
        self._h = h

    @classmethod
    def self_exchange_only(cls, group):
        """A one-rank communicator that serves exchange() alone, with the rank as its own peer (dpgo_comm_create_self): for a
        group whose neighbours no rank hosts -- one rank of an N-GPU run emulated on one GPU."""
        self = cls.__new__(cls)
        self.group, self.rank, self.nranks = group, 0, 1
        h = C.c_void_p()
        if lib().dpgo_comm_create_self(group._h, C.byref(h)) != 0:
            raise RuntimeError("dpgo_comm_create_self failed")
        self._h = h
        return self

    def __del__(self):
        if getattr(self, "_h", None):
            lib().dpgo_comm_free(self._h)
            self._h = None

    def close(self):
        self.__del__()

    def exchange(self):
        return lib().dpgo_comm_exchange(self._h)

    def exchange_kind(self):
        """"p2p" (grouped ncclSend / ncclRecv to the real neighbours) or "allgather"."""
        return "p2p" if lib().dpgo_comm_exchange_kind(self._h) == 1 else "allgather"

    def bytes_sent(self):
        """Bytes this rank hands to RCCL per exchange."""
        return int(lib().dpgo_comm_bytes_sent(self._h))

    def self_exchange(self):
        """One rank only: exchange() runs the neighbour-to-neighbour path with this rank as its own peer from now on
        (a measurement mode, dpgo_comm_self_exchange)."""
        if lib().dpgo_comm_self_exchange(self._h) != 0:
            raise RuntimeError("dpgo_comm_self_exchange failed")

    def enable_timing(self):
        if lib().dpgo_comm_enable_timing(self._h) != 0:
            raise RuntimeError("dpgo_comm_enable_timing failed")

    def exchange_time(self):
        """(mean microseconds from 'iterate final' to 'neighbour rows in place', exchanges counted)."""
        us, n = C.c_double(), C.c_long()
        if lib().dpgo_comm_exchange_time(self._h, C.byref(us), C.byref(n)) != 0:
            raise RuntimeError("dpgo_comm_exchange_time failed")
        return us.value, n.value

    def allreduce_sum(self, vals):
        a = np.ascontiguousarray(vals, np.float64).ravel().copy()
        if lib().dpgo_comm_allreduce_sum(self._h, _dp(a), len(a)) != 0:
            raise RuntimeError("dpgo_comm_allreduce_sum failed")
        return a

    def barrier(self):
        return lib().dpgo_comm_barrier(self._h)


class DPGOHash:
    """View of one node of a NodeGroup with the reference's DPGOHash method names
    (C++/DPGO/include/DPGO/DPGOHash.h:13-107)."""

    def __init__(self, group, local):
        self.group, self.local = group, local
        self.node = group.node_ids[local]
        self.n = group.sizes[local][:2]
        self.m = group.sizes[local][2:]
        self.d = group.d

    def initialize(self, X):
        X, ld = _fcol(X)
        if X.shape != ((self.d + 1) * (self.n[0] + self.n[1]), self.d):
            return -1
        return lib().dpgo_group_initialize(self.group._h, self.local, _dp(X), ld)

    def update(self):
        return self.group.update([self.local])

    def iterate(self):
        return self.group.iterate([self.local])

    def results(self):
        return self.group.results(self.local)

    def message_for(self, beta):
        """The message this node owes node beta: ((d+1) |sent[beta]|) x d, [t rows ; R rows]."""
        ns = C.c_int()
        if lib().dpgo_group_message_sizes(self.group._h, self.local, int(beta), C.byref(ns), None) != 0 or ns.value == 0:
            return None
        M = np.zeros(((self.d + 1) * ns.value, self.d), order="F")
        lib().dpgo_group_send(self.group._h, self.local, int(beta), _dp(M), M.shape[0])
        return M

    def receive(self, msg):
        """DPGOHash::receive(const std::map<int, Matrix>&)."""
        rc = 0
        for beta, M in msg.items():
            M, ld = _fcol(M)
            rc |= lib().dpgo_group_receive(self.group._h, self.local, int(beta), _dp(M), ld)
        return rc

    def Xk(self):
        X = np.zeros(((self.d + 1) * (self.n[0] + self.n[1]), self.d), order="F")
        lib().dpgo_group_get_Xk(self.group._h, self.local, _dp(X), X.shape[0])
        return X

    def Xak(self):
        X = np.zeros(((self.d + 1) * self.n[0], self.d), order="F")
        lib().dpgo_group_get_Xak(self.group._h, self.local, _dp(X), X.shape[0])
        return X


class DistPGO:
    """Single-process dist_pgo driver loop (C++/examples/dist_pgo.cpp:446-531): every node of the graph
    hosted by one GPU.  bench.py holds the multi-process (one rank per GPU) variant."""

    def __init__(self, graph, options, X0=None, device=0):
        self.graph, self.options = graph, options
        self.group = NodeGroup(graph, range(graph.num_nodes), options, device)
        self.X0 = graph.chordal_initialization() if X0 is None else np.asfortranarray(X0)
        if self.group.initialize_global(self.X0) != 0:
            raise RuntimeError("initialize failed")
        self.group.update()

    def step(self):
        return self.group.step()

    def X(self):
        X = np.zeros(((self.graph.d + 1) * self.graph.num_poses, self.graph.d), order="F")
        self.group.scatter_global(X)
        return X

    def sum_fobj(self):
        """sum_a fobj^a == F(X_k) (invariant 1)."""
        return sum(self.group.results(k).fobj for k in range(len(self.group)))

    def evaluate(self):
        """(2F, 2|grad F|) as printed by the reference driver (dist_pgo.cpp:477-481, 523-527), from the
        per-node device reductions: F = sum_a fobj^a and |grad F|^2 = sum_a gradFnorm_a^2 (the rows of
        the global Riemannian gradient that belong to node a are exactly Proj(Dfobj^a), SURVEY Appendix
        B-1/B-4; DPGOStar::evaluate_f / evaluate_grad, DPGOStar.cpp:713-829)."""
        r = [self.group.results(k) for k in range(len(self.group))]
        return 2.0 * sum(x.fobj for x in r), 2.0 * float(np.sqrt(sum(x.gradFnorm ** 2 for x in r)))


class DPGOStar:
    """AMM-PGO* with the method names of the reference's DPGOStar
    (C++/DPGO/include/DPGO/DPGOStar.h:13-61): initialize / update / iterate / communicate.
    The master's global objective is the sum of per-node device reductions.  By default all nodes are hosted
    by one GPU; with `nodes` (this process's share) and an initialised torch.distributed process group the
    nodes are spread over one process per GPU: boundary poses travel by all-gather (after every iterate, and
    for every trial point the master evaluates), the global scalars by all-reduce."""

    def __init__(self, graph, options, device=0, nodes=None):
        self.graph, self.options = graph, options
        self.group = NodeGroup(graph, range(graph.num_nodes) if nodes is None else nodes, options, device)
        self._dist = None
        if nodes is not None and len(list(nodes)) != graph.num_nodes:
            self._connect()

    def _connect(self):
        self._dist = self.group.connect_torch()

    def initialize(self, X):
        X, ld = _fcol(X)
        return lib().dpgo_group_star_initialize(self.group._h, _dp(X), ld)

    def update(self):
        return lib().dpgo_group_star_update(self.group._h)

    def iterate(self):
        return lib().dpgo_group_star_iterate(self.group._h)

    def communicate(self):
        rc = self.group.communicate_local()
        if self._dist is not None:                   # DPGOHash::communicate across processes
            _, torch, send, gathered, ext, allgather = self._dist
            with torch.cuda.stream(ext):
                self.group.pack_sent(send.data_ptr())
            allgather()
            with torch.cuda.stream(ext):
                self.group.unpack_recv(gathered.data_ptr())
        return rc

    def state(self):
        F, f, fh, b = C.c_double(), C.c_double(), C.c_double(), C.c_int()
        lib().dpgo_group_star_state(self.group._h, C.byref(F), C.byref(f), C.byref(fh), C.byref(b))
        return dict(F=F.value, fobj=f.value, fobjh=fh.value, branches=b.value)

    def step(self):
        return self.update() | self.iterate() | self.communicate()

    def X(self):
        X = np.zeros(((self.graph.d + 1) * self.graph.num_poses, self.graph.d), order="F")
        self.group.scatter_global(X)
        return X
