/* dpgo_amd -- C ABI of the MI355X-native DPGO hot path.
 *
 * Drop-in boundary for the per-node MM / AMM inner step of the reference's
 * `dist_pgo` (MurpheyLab/DPGO).  Plain pointers and sizes only; the caller owns
 * host buffers, the library owns device memory behind opaque handles.  Every
 * function returns 0 on success and -1 on error (with a line on stderr), which
 * is the reference's own convention (`return -1` + LOG(ERROR), e.g.
 * C++/DPGO/include/DPGO/DPGOHash.h:43-47, C++/DPGO/include/DPGO/DPGO_utils.h:402-406).
 *
 * Matrices X are column-major with an explicit leading dimension, in the
 * reference layout: rows [0,n) translations t_i^T, rows [n + d i, n + d i + d)
 * the block R_i^T (C++/DPGO/include/DPGO/DPGOProblem.h:167-171).  The
 * conversion to the pose-contiguous device layout happens inside the library.
 *
 * A "group" is the set of nodes hosted by one GPU / one process; it replaces a
 * std::vector<std::shared_ptr<DPGOHash>> restricted to those nodes, and its
 * batched calls replace the driver's `for (alpha...) dpgo_hash[alpha]->...()`
 * loops (C++/examples/dist_pgo.cpp:455-462, 496-521).
 */
#ifndef DPGO_AMD_H
#define DPGO_AMD_H

#ifdef __cplusplus
extern "C" {
#endif

/* DPGO::Options -- C++/DPGO/include/DPGO/DPGO_types.h:78-201: same names, same defaults, enums as their integer
 * values.  verbose prints one summary line per truncated-Newton refinement.  Not carried: user_function, log_iterates
 * (no effect on the iterates).
 * dpgo_group_create fails (-1) for what is not implemented: preconditioner IncompleteCholesky. */
typedef struct dpgo_options {
  int scheme;                 /* 0 = Scheme::MM, 1 = Scheme::AMM */
  double regularizer;
  double accepted_delta;
  double eta[2];
  double psi;
  double phi;
  int max_soft_restart_hits[2];
  int oscillation_cnt_period;
  int max_oscillations;
  int loss;                   /* 0 None, 1 Huber, 2 GemanMcClure, 3 Welsch */
  double loss_reg;
  int rescale;                /* Rescale (DPGO_types.h:43-46): 0 Static, 1 Dynamic (the reference default, :128) */
  int max_rescale_count;      /* DPGO_types.h:131 */
  double grad_norm_tol;
  double rel_func_decrease_tol;
  double stepsize_tol;
  int max_iterations;
  int max_iterations_accepted;
  double reg_Cholesky_precon_max_condition_number;
  double preconditioned_grad_norm_tol;
  int max_tCG_iterations;
  double STPCG_kappa;
  double STPCG_theta;
  int preconditioner;         /* Preconditioner (DPGO_types.h:35-40): 0 None, 1 Jacobi, 2 IncompleteCholesky,
                                 3 RegularizedCholesky (default) */
  int verbose;                /* DPGO_types.h:87: 0 (default) quiet; 1: a line per node and refinement on stdout */
} dpgo_options_t;

/* The scalar part of DPGOResult -- C++/DPGO/include/DPGO/DPGO_types.h:204-322. */
typedef struct dpgo_results {
  int updated;
  int iters;
  double gradFnorm;
  double fobjE;
  double Fk[2];
  double Gk;
  double Gkh;
  double fobj;                /* fobj[iters] */
  double f;                   /* f[iters] */
  double gamma;
  double s[2];                /* s[iters], s[iters+1] */
  int soft_restart_hits[2];
  int num_oscillations;
  int refined;                /* whether the last iterate() ran the TNT refinement */
  int tnt_status;             /* TNTStatus of that run (TNT.h:134-164), -1 if none */
  int tnt_inner_iterations;
  int restarts;
} dpgo_results_t;


typedef struct dpgo_graph dpgo_graph_t;
typedef struct dpgo_group dpgo_group_t;

/* Options() defaults, and the overrides of C++/examples/dist_pgo.cpp:103-120. */
void dpgo_options_default(dpgo_options_t *opt);
void dpgo_options_driver(dpgo_options_t *opt, int loss, int accelerated);

/* DPGO::read_g2o -- C++/DPGO/include/DPGO/DPGO_utils.h:49-51, C++/DPGO/src/DPGO_utils.cpp:140-202. */
int dpgo_read_g2o(const char *filename, int num_nodes, dpgo_graph_t **out);
/* The same partition applied to measurements already in memory (tail I, head J global pose ids,
 * R row-major d x d, t d). */
int dpgo_graph_from_edges(int d, int num_poses, int m, const int *I, const int *J, const double *R,
                          const double *t, const double *kappa, const double *tau, int num_nodes,
                          dpgo_graph_t **out);
void dpgo_graph_free(dpgo_graph_t *g);
int dpgo_graph_info(const dpgo_graph_t *g, int *d, int *num_poses, int *num_nodes, int *num_edges);
/* copies the parsed measurements (file order); any output pointer may be NULL */
int dpgo_graph_edges(const dpgo_graph_t *g, int *I, int *J, double *R, double *t, double *kappa, double *tau);
/* DPGOProblem::n() / m() -- C++/DPGO/include/DPGO/DPGOProblem.h:241-248 (host only) */
int dpgo_graph_node_sizes(const dpgo_graph_t *g, int node, int *n0, int *n1, int *m0, int *m1);
/* neighbour ordering of generate_data_info (C++/DPGO/src/DPGO_utils.cpp:400-418): n1 (node, pose) pairs */
int dpgo_graph_node_neighbours(const dpgo_graph_t *g, int node, int *nbr_node, int *nbr_pose);
/* first global pose id of a node (g_index[node].begin()->second, dist_pgo.cpp:470) */
int dpgo_graph_node_offset(const dpgo_graph_t *g, int node);
/* Boundary-exchange plan of a group of nodes (host only): the (node, pose) keys it exports to and
 * imports from nodes outside the group, both sorted.  counts[0] = exported, counts[1] = imported; the
 * key arrays may be NULL to query the counts.  (sent_ / recv_ of C++/DPGO/src/DPGO_utils.cpp:428-435.) */
int dpgo_graph_exchange_plan(const dpgo_graph_t *g, const int *node_ids, int num_local, int *sent_nodes,
                             int *sent_poses, int *recv_nodes, int *recv_poses, int *counts);
/* DPGOProblem::index() / sent() / recv() -- C++/DPGO/include/DPGO/DPGOProblem.h:212-225 (host only): the entries
 * {(node, pose) -> (block, k)} of the map `which` (0 index, 1 sent, 2 recv), in map order.  Arrays may be NULL
 * to query *count. */
int dpgo_graph_node_maps(const dpgo_graph_t *g, int node, int which, int *nodes, int *poses, int *block, int *local,
                         int *count);
/* Result format: VERTEX_SE2 / VERTEX_SE3:QUAT lines from X ((d+1)N x d; NULL: edges only) followed
 * by the graph's EDGE_* lines with the isotropic information the loader's formulas invert. */
int dpgo_write_g2o(const dpgo_graph_t *g, const double *X, int ld, const char *filename);
/* Centralised chordal initialisation -- C++/examples/dist_pgo.cpp:416-444
 * (C++/SESync/src/SESync_utils.cpp:573-652).  Host, set-up only.  X: (d+1)N x d. */
int dpgo_chordal_initialization(const dpgo_graph_t *g, double *X, int ld);

/* Distributed chordal initialisation -- the `--dist_init true` branch of C++/examples/dist_pgo.cpp:144-416 on
 * C++/DChordal (DChordalReduced_R, DChordal_R, DChordalReduced_t, DChordal_t; DChordal.cpp:79-152,
 * DChordalProblem.h:128-246, DChordal_utils.cpp:67-1204).  The two sparse stages run on the group's GPU, the two
 * reduced ones (one d x d block / one translation per node) on the host.  Every node of the graph must be in the
 * group, in order.  opts: DChordal::Options::reg_G, the driver's stage schedule, and the length of the stage-0
 * stand-in (the reference's stage 0 is a per-node SE-Sync solve, out of scope: here chordal initialisation of the
 * node's own subgraph + local_iters refined MM-PGO iterations).  X_local (optional): stage-0 poses of every node in
 * the global layout.  X: (d+1)N x d result.  objectives (optional, capacity *num_objectives on entry): 0.5 sum |B X +
 * b|^2 sampled every 20 iterations through the four stages (what the driver prints, :206-210). */
typedef struct dpgo_dchordal_options {
  int iters[4];      /* 100, 400, 150, 250 */
  int local_iters;   /* 30 */
  double reg_G;      /* 1e-12 (DChordal_types.h:50) */
} dpgo_dchordal_options_t;
void dpgo_dchordal_options_default(dpgo_dchordal_options_t *opt);
int dpgo_group_dist_chordal_initialization(dpgo_group_t *grp, const dpgo_dchordal_options_t *opt, const double *X_local,
                                           int ld_local, double *X, int ld, double *objectives, int *num_objectives);

/* DPGOHash(node, measurements, options) for every node in node_ids -- C++/DPGO/src/DPGOHash.cpp:11-18,
 * C++/DPGO/src/DPGOProblem.cpp:11-125.  Fails (-1) when no HIP device is present: there is no CPU path. */
int dpgo_group_create(const dpgo_graph_t *g, const int *node_ids, int num_local, const dpgo_options_t *opt,
                      int device, dpgo_group_t **out);
void dpgo_group_free(dpgo_group_t *grp);

/* DPGOHash::initialize -- C++/DPGO/src/DPGOHash.cpp:20-43.  X: (d+1)(n0+n1) x d. */
int dpgo_group_initialize(dpgo_group_t *grp, int local, const double *X, int ld);
/* Split a global X over the nodes and fill neighbour rows -- dist_pgo.cpp:435-446 + DPGO::communicate. */
int dpgo_group_initialize_global(dpgo_group_t *grp, const double *X, int ld);
/* DPGOHash::update / iterate -- C++/DPGO/src/DPGOHash.cpp:84-228, 583-628.  locals == NULL: every node. */
int dpgo_group_update(dpgo_group_t *grp, const int *locals, int n);
int dpgo_group_iterate(dpgo_group_t *grp, const int *locals, int n);
/* DPGOHash::communicate -- C++/DPGO/include/DPGO/DPGOHash.h:28-86 -- for neighbours hosted by this group. */
int dpgo_group_communicate_local(dpgo_group_t *grp);
/* One pass of the driver's loop body -- C++/examples/dist_pgo.cpp:496-521: iterate() of every node of the group, the
 * boundary exchange (`comm`: a communicator made by dpgo_comm_create for this group, or NULL when every neighbour is hosted
 * here), communicate(), update() of every node -- in one call, so that a host in another language does not put its own
 * call overhead between the launches. */
struct dpgo_comm;
int dpgo_group_step(dpgo_group_t *grp, struct dpgo_comm *comm);
/* DPGOHash::receive -- C++/DPGO/src/DPGOHash.cpp:45-82: one message per neighbour node beta, a
 * ((d+1) |recv[beta]|) x d matrix [translation rows ; rotation rows] with the poses in the order of
 * recv[beta].  dpgo_group_send builds the message node `local` owes node beta (the poses of sent[beta],
 * C++/DPGO/src/DPGO_utils.cpp:428-435) from its current Xk.  Host matrices, column-major. */
int dpgo_group_message_sizes(const dpgo_group_t *grp, int local, int beta, int *num_send, int *num_recv);
int dpgo_group_send(const dpgo_group_t *grp, int local, int beta, double *msg, int ld);
int dpgo_group_receive(dpgo_group_t *grp, int local, int beta, const double *msg, int ld);
/* AMM-PGO* -- DPGOStar::{initialize, update, iterate} (C++/DPGO/src/DPGOStar.cpp:107-213; per-node
 * helpers :215-711); communicate() is dpgo_group_communicate_local (+ the boundary exchange between groups).
 * Either every node of the graph is in the group, or the groups are connected by dpgo_group_set_collectives
 * (the master's global objective is the sum of the per-node device reductions over all groups).  Loop:
 * star_initialize(X); repeat { star_update; star_iterate; communicate_local }.
 * star_state: F (running average, :210), fobj = F(X_k+1), fobjh = F(X_k+1/2), branches bit 0 = plain
 * proximal redo (:149-155), bit 1 = MM redo (:159-169), bit 2 = proximal-rotation fallback (:171-192). */
/* AMM-PGO* with the nodes spread over several groups (one process per GPU): the caller lends the library two
 * collectives.  allgather(user): all-gather the registered device buffer `send` (stride * (d+1)*d doubles, the
 * layout of dpgo_group_pack_sent) into `gathered` (the layout given to dpgo_group_set_recv_layout), ordered on
 * dpgo_group_stream().  allreduce(user, vals, n): in-place sum of n host doubles over all groups; every group
 * must get bit-identical results (they steer the same branches: DPGOStar.cpp:147-192).  Both return 0 on success. */
typedef int (*dpgo_allgather_fn)(void *user);
typedef int (*dpgo_allreduce_fn)(void *user, double *vals, int n);
int dpgo_group_set_collectives(dpgo_group_t *grp, void *send, void *gathered, dpgo_allgather_fn allgather,
                               dpgo_allreduce_fn allreduce, void *user);
int dpgo_group_star_initialize(dpgo_group_t *grp, const double *X, int ld);
int dpgo_group_star_update(dpgo_group_t *grp);
int dpgo_group_star_iterate(dpgo_group_t *grp);
int dpgo_group_star_state(const dpgo_group_t *grp, double *F, double *fobj, double *fobjh, int *branches);
/* Boundary exchange with other groups (the message of DPGOHash::receive, DPGOHash.cpp:45-82):
 * pack this group's exported poses into a device buffer of num_sent * (d+1)*d doubles, all-gather
 * the buffers of all groups (RCCL), then unpack.  Keys are (node, pose).
 * dpgo_group_unpack_recv is LAZY for the robust losses (round 6): the neighbour rows stay in `device_gathered` until the
 * group next looks at them -- the next dpgo_group_update reads them from there inside its inter-edge pass and stores them
 * into Xk on the way (no unpack kernel); any other reader (dpgo_group_get_Xk, ...) gets the plain indexed copy first.  The
 * buffer must therefore stay untouched until that update (or any call that reads the group's state) has been made;
 * everything is ordered on dpgo_group_stream(). */
int dpgo_group_num_sent(const dpgo_group_t *grp);
int dpgo_group_sent_keys(const dpgo_group_t *grp, int *nodes, int *poses);
int dpgo_group_set_recv_layout(dpgo_group_t *grp, int nranks, int stride, const int *counts, const int *nodes,
                               const int *poses);
int dpgo_group_pack_sent(dpgo_group_t *grp, void *device_buffer);
int dpgo_group_unpack_recv(dpgo_group_t *grp, const void *device_gathered);

/* DPGOStar::evaluate_f / evaluate_grad -- C++/DPGO/src/DPGOStar.cpp:713-829 -- at an ARBITRARY global X
 * ((d+1)N x d); the optimizer state is not touched.  *F and *grad_sqnorm (= |grad F|^2, Riemannian) are sums over
 * the nodes of this group -- over all groups when collectives are set (dpgo_group_set_collectives / dpgo_comm_create);
 * the driver prints 2 F and 2 sqrt(grad_sqnorm) (dist_pgo.cpp:477-481).  grad (optional, (d+1)N x d, leading
 * dimension ldg): the rows of this group's own poses are written.  Any output pointer may be NULL. */
int dpgo_group_evaluate(dpgo_group_t *grp, const double *X, int ld, double *F, double *grad_sqnorm, double *grad,
                        int ldg);
/* DPGOHash::set_options / options -- C++/DPGO/include/DPGO/DPGOHash.h:91-96.  Fields that are part of the problem
 * built at construction (loss, loss_reg, regularizer, rescale, preconditioner) must keep their values (-1). */
int dpgo_group_set_options(dpgo_group_t *grp, const dpgo_options_t *opt);
int dpgo_group_get_options(const dpgo_group_t *grp, dpgo_options_t *opt);

/* ---- the exchange between GPUs: RCCL behind the C ABI (one process per GPU; replaces the in-process copies of
 * DPGOHash::communicate, C++/DPGO/include/DPGO/DPGOHash.h:28-86, and the master's sums of DPGOStar.cpp:147-192).
 * dpgo_comm_unique_id: ncclGetUniqueId on one rank; the caller carries the 128 bytes to the others (file, socket,
 *   MPI, torch.distributed -- any channel).
 * dpgo_comm_create: ncclCommInitRank + the exchange lay-out (the ranks all-gather their exported (node, pose) keys);
 *   also connects the group's AMM-PGO* / global-evaluation collectives (dpgo_group_set_collectives) to RCCL.
 * dpgo_comm_exchange: communicate() for neighbours hosted by other ranks; returns at once.  Neighbour to neighbour (the
 *   default with several ranks): ncclGroupStart, one ncclSend + ncclRecv per real neighbour, ncclGroupEnd on the GROUP's own
 *   stream -- the pack has ridden on the tail of dpgo_group_iterate, the unpack is the next dpgo_group_update's inter-edge
 *   pass reading the receive buffer (round 6; on a stream of its own the exchange paid two event hand-overs to hide one
 *   9 us kernel: profiles/r06_exchange_streams.txt).  All-gather (the fallback): pack, ncclAllGather, unpack on the
 *   communicator's own stream; the group's next update() joins it after queueing the part of the surrogate build that
 *   needs no neighbour data.  Call after dpgo_group_iterate, every rank, every iteration.
 * dpgo_comm_allreduce_sum: in-place sum of n host doubles over the ranks (bit-identical on every rank).
 * The library binds RCCL at run time (librccl.so.1); these calls return -1 when it cannot be loaded. */
typedef struct dpgo_comm dpgo_comm_t;
int dpgo_comm_unique_id(void *id128);
int dpgo_comm_create(dpgo_group_t *grp, int rank, int nranks, const void *id128, dpgo_comm_t **out);
void dpgo_comm_free(dpgo_comm_t *comm);
int dpgo_comm_exchange(dpgo_comm_t *comm);
int dpgo_comm_allreduce_sum(dpgo_comm_t *comm, double *vals, long n);
int dpgo_comm_barrier(dpgo_comm_t *comm);
/* How dpgo_comm_exchange moves the boundary poses: 1 = neighbour to neighbour (grouped ncclSend / ncclRecv, the default with
 * several ranks once its self-check passed on every rank), 0 = all-gather of fixed-size buffers; -1 on error. */
int dpgo_comm_exchange_kind(const dpgo_comm_t *comm);
/* Bytes this rank hands to RCCL per dpgo_comm_exchange (neighbour to neighbour: the records its peers need; all-gather: one
 * padded block); -1 on error.  Reported per rank by bench.py's N > 1 line. */
long dpgo_comm_bytes_sent(const dpgo_comm_t *comm);
/* Measurement hooks (bench.py --emulate-world N --force-exchange; no counterpart in the reference, whose "exchange" is a
 * memory copy, C++/DPGO/include/DPGO/DPGOHash.h:28-86):
 * dpgo_comm_self_exchange: on a communicator of ONE rank, dpgo_comm_exchange from now on runs the neighbour-to-neighbour path
 *   in its steady state with the rank as its own peer (the pack on the tail of iterate(), ncclGroupStart, ncclSend + ncclRecv
 *   of every exported record, ncclGroupEnd on the group's stream; what arrives -- the rank's own rows -- is not unpacked, the
 *   trajectory stays that of the run without an exchange): what that path costs an iteration, short of the wire;
 * dpgo_comm_enable_timing / dpgo_comm_exchange_time: mean time in microseconds between an event recorded in front of the
 *   exchange ("the iterate is final") and one behind it ("the neighbour rows have arrived"), over the exchanges since timing
 *   was enabled. */
int dpgo_comm_self_exchange(dpgo_comm_t *comm);
/* ... the same for a group whose neighbours NO rank hosts (one rank of an N-GPU run emulated on one GPU): a communicator of
 * one rank without the exchange lay-out, serving dpgo_comm_exchange in the self-exchange mode only */
int dpgo_comm_create_self(dpgo_group_t *grp, dpgo_comm_t **out);
int dpgo_comm_enable_timing(dpgo_comm_t *comm);
int dpgo_comm_exchange_time(dpgo_comm_t *comm, double *mean_us, long *count);
/* Test hook: the grouped ncclSend / ncclRecv path of dpgo_comm_exchange on a communicator of ONE rank that is its own peer
 * (legal inside ncclGroupStart / End): every row `grp` exports travels pack kernel -> group call -> unpack kernel on the
 * communicator's stream (the form of the creation-time check), on records that carry their own keys.  `grp` may host any subset of the nodes.  0 = every record
 * arrived in its row and no other row was touched; -1 otherwise (also when the group exports nothing). */
int dpgo_debug_comm_p2p_self(dpgo_group_t *grp);
/* The same pack / unpack on host matrices (no GPU needed; what a host-staged transport or a test uses): records
 * of the poses a group exports, in key order, from a global X ((d+1)N x d) into buf (count x (d+1)d doubles,
 * [t | rows of R^T] per pose); and the neighbour rows of node `node` ((d+1)(n0+n1) x d matrix Z, DPGOHash::initialize
 * lay-out) from the gathered buffers of all ranks (slot of key k of rank r = r * stride + k, as in
 * dpgo_group_set_recv_layout).  Return the number of poses written, -1 on error. */
int dpgo_host_pack_sent(const dpgo_graph_t *g, const int *node_ids, int num_local, const double *X, int ld, double *buf);
int dpgo_host_unpack_recv(const dpgo_graph_t *g, const int *node_ids, int num_local, int node, int nranks, int stride,
                          const int *counts, const int *nodes, const int *poses, const double *gathered, double *Z,
                          int ldz);

/* results().Xk / results().Xak -- DPGO_types.h:207-214 */
int dpgo_group_get_Xk(const dpgo_group_t *grp, int local, double *X, int ld);
int dpgo_group_get_Xak(const dpgo_group_t *grp, int local, double *X, int ld);
/* gather X^alpha into the global X -- dist_pgo.cpp:502-511 */
int dpgo_group_scatter_global(const dpgo_group_t *grp, double *X, int ld);
int dpgo_group_results(const dpgo_group_t *grp, int local, dpgo_results_t *out);
int dpgo_group_node_id(const dpgo_group_t *grp, int local);
int dpgo_group_sync(const dpgo_group_t *grp);
void *dpgo_group_stream(const dpgo_group_t *grp);   /* hipStream_t all work is enqueued on */

/* ---- measurement hooks (bench.py) --------------------------------------------------------
 * Per-launch timing of every kernel family with HIP events on the launch stream.  Off by default. */
int dpgo_prof_enable(int on);
int dpgo_prof_num_kinds(void);
const char *dpgo_prof_kind_name(int kind);
int dpgo_prof_collect(double *ms, double *bytes, long *count);   /* arrays of dpgo_prof_num_kinds() */
/* for the fused passes (k_inter, k_proximal): the bytes of every operand they move, counted one by one (0 for the others) */
int dpgo_prof_collect_operands(double *operand_bytes);
/* size of the two multifrontal factors: dense front entries and number of tree levels */
int dpgo_group_solver_stats(const dpgo_group_t *grp, long *nnz_tt, long *nnz_rr, int *levels_tt, int *levels_rr);
/* The branch-free segments of an iteration (C++/examples/dist_pgo.cpp:496-521: iterate, update) go to the GPU as replays of
 * captured HIP graphs where the host's launch rate would bound the group (DPGO_ITER_GRAPH=0 / 1 forces it off / on):
 * segments replayed, graphs captured, segments launched eagerly since the group was created. */
int dpgo_group_graph_stats(const dpgo_group_t *grp, long *replays, long *captures, long *eager);

/* ---- test hooks ------------------------------------------------------------------------- */
/* Host: the assembled operator `name` in {"G","S","P","P0","Q","D"} of a node as COO triplets in
 * the REFERENCE row/column order.  Call with rows == NULL to get the count. */
int dpgo_debug_node_matrix(const dpgo_graph_t *g, int node, const dpgo_options_t *opt, const char *name,
                           int *rows, int *cols, double *vals);
/* Host: the proximal coefficients T (n0), N (n0 x d), V (n0 x d x d). */
int dpgo_debug_node_proximal(const dpgo_graph_t *g, int node, const dpgo_options_t *opt, double *T, double *N,
                             double *V);
/* Host: multifrontal factor + solve of a CSR SPD matrix, X (n x ncols row-major) <- A^-1 X. */
int dpgo_debug_spd_solve(int n, const int *ptr, const int *col, const double *val, double *X, int ncols,
                         int leaf);
/* Host: size of the multifrontal factor of a CSR SPD matrix for a given nested-dissection leaf size. */
int dpgo_debug_spd_stats(int n, const int *ptr, const int *col, const double *val, int leaf, long *nnz, int *levels,
                         int *max_front);
/* Host: the neighbour-to-neighbour exchange plan of `rank` (what dpgo_comm_exchange uses with more than one rank) from
 * every rank's exported and needed (node, pose) keys (dpgo_graph_exchange_plan of its nodes): per peer 5 ints (rank,
 * send_off, send_cnt, recv_off, recv_cnt), the send and receive keys (node, pose interleaved, concatenated over the peers
 * in the order both ends agree on: ascending (node, pose)); sizes[0..2] = peers, send keys, receive keys.  Output
 * pointers may be NULL to query the sizes. */
int dpgo_debug_p2p_plan(int rank, int nranks, const int *exp_counts, const int *exp_nodes, const int *exp_poses,
                        const int *need_counts, const int *need_nodes, const int *need_poses, int *peers, int *send_keys,
                        int *recv_keys, int *sizes);
/* Device: single operators of one node on reference-layout inputs:
 *  "project" (d n0 x d -> nearest rotations), "solve_tt" ((d+1) n0 x d, translation rows),
 *  "solve_rr" (rotation rows), "G" ((d+1) n0 x d -> G X), "proximal" (in = [Z ; Df] stacked). */
int dpgo_group_debug_apply(dpgo_group_t *grp, int local, const char *op, const double *in, int ld_in,
                           double *out, int ld_out);

#ifdef __cplusplus
}
#endif
#endif /* DPGO_AMD_H */
