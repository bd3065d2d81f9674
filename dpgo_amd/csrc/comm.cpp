// Inter-GPU exchange of the boundary poses: RCCL behind the C ABI (one process per GPU, xGMI).
//
// Replaces the in-process copies of the reference
//   DPGOHash::communicate      C++/DPGO/include/DPGO/DPGOHash.h:28-86   (neighbour rows of Xk)
//   DPGOStar::communicate_n    C++/DPGO/src/DPGOStar.cpp:276-313
// and supplies the sums over nodes that the reference takes in one address space (the master's global objective of
// AMM-PGO*, DPGOStar.cpp:147-192; the logged F and |grad F|, dist_pgo.cpp:477-481).
//
// One exchange = pack the exported poses (k_copy_indexed) -> ncclAllGather of fixed-size buffers -> unpack into the
// neighbour rows, all on the communicator's own HIP stream: it starts when the iterate is final (an event on the
// group's stream) and the group's next update() waits for it only after it has queued the part of the surrogate
// build that needs no neighbour data (Group::update).  The payload is tiny (<= ~0.5 MB per GPU at the headline
// size), so the exchange is latency bound: one fused collective per iteration, no host round trip.
//
// RCCL is bound at run time (dlopen of librccl.so.1): the library loads -- and its host-side entry points work --
// on machines without RCCL or without a GPU, and inside a process that already carries an RCCL (PyTorch) the same
// library instance is used.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <iterator>
#include <thread>
#include <map>
#include <cstdio>
#include <cstring>
#include <string>

#include "comm.h"

namespace dpgo {

namespace {
struct Rccl {
  void *h = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommAbort) CommAbort = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  bool ok = false;
};
Rccl &rccl() {
  static Rccl r;
  static bool tried = false;
  if (tried) return r;
  tried = true;
  for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    r.h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (r.h) break;
  }
  if (!r.h) {
    fprintf(stderr, "[dpgo_amd] ERROR: cannot load RCCL (librccl.so.1): %s\n", dlerror());
    return r;
  }
#define BIND(f) r.f = reinterpret_cast<decltype(r.f)>(dlsym(r.h, "nccl" #f))
  BIND(GetUniqueId); BIND(CommInitRank); BIND(CommDestroy); BIND(AllGather); BIND(AllReduce); BIND(GetErrorString);
  BIND(Send); BIND(Recv); BIND(GroupStart); BIND(GroupEnd);   // (optional: without them the all-gather stays)
  BIND(CommAbort);                                             // (optional: lets a timed-out communicator go)
#undef BIND
  r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.AllReduce && r.GetErrorString;
  if (!r.ok) fprintf(stderr, "[dpgo_amd] ERROR: librccl.so.1 lacks an expected symbol.\n");
  return r;
}
#define NCCL_OK(x)                                                                                  \
  do {                                                                                              \
    ncclResult_t r_ = (x);                                                                          \
    if (r_ != ncclSuccess) {                                                                        \
      fprintf(stderr, "[dpgo_amd] ERROR: RCCL: %s at %s:%d\n", rccl().GetErrorString(r_), __FILE__, __LINE__); \
      throw DeviceError("RCCL call failed");                                                        \
    }                                                                                               \
  } while (0)
#define HIP_OK(x)                                                                                   \
  do {                                                                                              \
    hipError_t e_ = (x);                                                                            \
    if (e_ != hipSuccess) {                                                                         \
      fprintf(stderr, "[dpgo_amd] ERROR: HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      throw DeviceError(hipGetErrorString(e_));                                                     \
    }                                                                                               \
  } while (0)
}  // namespace

P2PPlan p2p_plan(int rank, const std::vector<std::vector<PoseKey>> &exported, const std::vector<std::vector<PoseKey>> &needed) {
  P2PPlan P;
  const int n = (int)exported.size();
  auto common = [](const std::vector<PoseKey> &a, const std::vector<PoseKey> &b) {
    std::vector<PoseKey> x(a), y(b), out;
    std::sort(x.begin(), x.end());
    x.erase(std::unique(x.begin(), x.end()), x.end());
    std::sort(y.begin(), y.end());
    y.erase(std::unique(y.begin(), y.end()), y.end());
    std::set_intersection(x.begin(), x.end(), y.begin(), y.end(), std::back_inserter(out));
    return out;   // ascending (node, pose)
  };
  for (int q = 0; q < n; q++) {
    if (q == rank) continue;
    const std::vector<PoseKey> snd = common(exported[rank], needed[q]), rcv = common(needed[rank], exported[q]);
    if (snd.empty() && rcv.empty()) continue;
    P.peers.push_back({q, (int)P.send_keys.size(), (int)snd.size(), (int)P.recv_keys.size(), (int)rcv.size()});
    P.send_keys.insert(P.send_keys.end(), snd.begin(), snd.end());
    P.recv_keys.insert(P.recv_keys.end(), rcv.begin(), rcv.end());
  }
  return P;
}

int Comm::unique_id(void *id128) {
  static_assert(sizeof(ncclUniqueId) == 128, "the C ABI hands the id over as 128 bytes");
  if (!rccl().ok) return -1;
  ncclUniqueId id;
  if (rccl().GetUniqueId(&id) != ncclSuccess) return -1;
  std::memcpy(id128, &id, sizeof(id));
  return 0;
}

Comm::Comm(Group *grp, int rank, int nranks, const void *id128, bool layout) : grp_(grp), rank_(rank), nranks_(nranks) {
  if (!rccl().ok || !grp || rank < 0 || rank >= nranks) return;
  // a constructor that throws never runs its destructor: release what was created, then pass the error on
  try {
    init(id128, layout);
  } catch (...) {
    release();
    throw;
  }
}

void Comm::init(const void *id128, bool layout) {
  Group *grp = grp_;
  const int rank = rank_, nranks = nranks_;
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  HIP_OK(hipSetDevice(grp->device()));
  ncclComm_t c = nullptr;
  NCCL_OK(rccl().CommInitRank(&c, nranks, id, rank));
  comm_ = c;
  HIP_OK(hipStreamCreateWithFlags(&cs_, hipStreamNonBlocking));
  HIP_OK(hipEventCreateWithFlags(&ev_ready_, hipEventDisableTiming));
  HIP_OK(hipEventCreateWithFlags(&ev_done_, hipEventDisableTiming));
  const int RS = (grp->d() + 1) * grp->d();
  red_.alloc(64);
  HIP_OK(hipHostMalloc((void **)&h_red_, sizeof(double) * 64, hipHostMallocDefault));
  if (!layout) return;   // (ok_ stays false: no exchange, no collectives lent to the group)
  // ---- who exports what: all-gather of the key counts, then of the (node, pose) keys, through RCCL itself
  const auto &keys = grp->sent_keys();
  DevBuf<int> cnt_d, cnts_d;
  cnt_d.upload(std::vector<int>{(int)keys.size()});
  cnts_d.alloc(nranks);
  NCCL_OK(rccl().AllGather(cnt_d.p, cnts_d.p, 1, ncclInt32, (ncclComm_t)comm_, cs_));
  sync_comm_stream();
  std::vector<int> counts;
  cnts_d.download(counts);
  stride_ = std::max(1, *std::max_element(counts.begin(), counts.end()));
  std::vector<int> mine(2 * (size_t)stride_, -1), all;
  for (size_t k = 0; k < keys.size(); k++) { mine[2 * k] = keys[k].first; mine[2 * k + 1] = keys[k].second; }
  DevBuf<int> mine_d, all_d;
  mine_d.upload(mine);
  all_d.alloc(2 * (size_t)stride_ * nranks);
  NCCL_OK(rccl().AllGather(mine_d.p, all_d.p, 2 * (size_t)stride_, ncclInt32, (ncclComm_t)comm_, cs_));
  sync_comm_stream();
  all_d.download(all);
  std::vector<int> nodes, poses;
  for (int r = 0; r < nranks; r++)
    for (int k = 0; k < counts[r]; k++) {
      nodes.push_back(all[2 * ((size_t)r * stride_ + k)]);
      poses.push_back(all[2 * ((size_t)r * stride_ + k) + 1]);
    }
  if (grp->set_recv_layout(nranks, stride_, counts.data(), nodes.data(), poses.data()) != 0) return;
  send_.alloc((size_t)stride_ * RS);
  gathered_.alloc((size_t)stride_ * RS * nranks);
  {
    std::vector<std::vector<PoseKey>> exported(nranks);
    size_t at = 0;
    for (int r = 0; r < nranks; r++)
      for (int k = 0; k < counts[r]; k++, at++) exported[r].push_back({nodes[at], poses[at]});
    // Whether the neighbour-to-neighbour path is even tried is decided per PROCESS (environment, the symbols this
    // process's RCCL exports), so the ranks agree on it first: one dissenting rank and everybody keeps the all-gather --
    // nobody may enter setup_p2p()'s collectives alone.
    const char *kind = getenv("DPGO_EXCHANGE");
    const bool want = !(kind && std::string(kind) == "allgather") && rccl().Send && rccl().Recv && rccl().GroupStart && rccl().GroupEnd;
    if (nranks > 1) {
      double against = want ? 0.0 : 1.0;
      if (allreduce_impl(&against, 1) != 0) return;
      if (against == 0.0 && setup_p2p(exported) != 0) p2p_ = false;
    }
  }
  // AMM-PGO* and the global evaluations borrow the same buffers on the group's own stream
  if (grp->set_collectives(send_.p, gathered_.p, &Comm::cb_allgather, &Comm::cb_allreduce, this) != 0) return;
  lent_ = true;
  grp->set_device_allreduce(&Comm::cb_allreduce_dev);   // (sums that are born on the device stay there: Group::star_sums)
  ok_ = true;
}

Comm::~Comm() { release(); }

void Comm::release() {
  if (grp_) {
    // an exchange that no update() has joined yet: let it finish, and take its event away from the group before the
    // event is destroyed
    if (ev_done_ && !broken_) (void)hipEventSynchronize(ev_done_);
    // (only what THIS communicator lent the group is taken back: a lay-out-free one, p2p_self_check's, lent nothing, and
    // the group may be working with another communicator's collectives)
    if (attached_) {
      grp_->flush_pending_recv();   // (a lazy unpack reads this communicator's receive buffer)
      grp_->set_exchange_pack(nullptr, 0, nullptr);
      grp_->set_stuck_handler(nullptr, nullptr);
      attached_ = false;
    }
    if (lent_) {
      grp_->set_pending_exchange(nullptr);
      grp_->set_collectives(nullptr, nullptr, nullptr, nullptr, nullptr);
      grp_->set_device_allreduce(nullptr);
      lent_ = false;
    }
  }
  // (a communicator whose stream ran into a deadline holds a kernel that will never end: it was aborted at the timeout
  // site -- abandon() -- not waited for)
  if (broken_ && comm_ && rccl().CommAbort) { (void)rccl().CommAbort((ncclComm_t)comm_); comm_ = nullptr; }
  if (cs_ && !broken_) (void)hipStreamSynchronize(cs_);
  if (comm_ && rccl().ok && !broken_) (void)rccl().CommDestroy((ncclComm_t)comm_);
  if (ev_ready_) (void)hipEventDestroy(ev_ready_);
  if (ev_done_) (void)hipEventDestroy(ev_done_);
  if (cs_ && !broken_) (void)hipStreamDestroy(cs_);
  if (h_red_) (void)hipHostFree(h_red_);
  comm_ = nullptr; ev_ready_ = ev_done_ = nullptr; cs_ = nullptr; h_red_ = nullptr;
  ok_ = false;
}

// The neighbour-to-neighbour exchange: all-gather the keys every rank NEEDS (set-up only), derive the plan (p2p_plan),
// upload the pack / unpack lists, and CHECK the whole path once with a message whose records carry their own keys: only
// if every rank received exactly what it expected (a vote through ncclAllReduce) does exchange() use it; otherwise the
// all-gather stays and a line says why.  RCCL with more than one rank cannot run on the boxes this was written on, so
// the first 8-GPU run is also this path's first execution (one rank talking to itself: p2p_self_check, run on the
// MI355X by tests/test_gpu_comm.py).  Hence the rules of this function: every rank runs the SAME sequence of
// collectives whatever happens to it locally -- a local failure (a key it cannot place, an exception from a kernel
// launch or an upload) becomes a "no" in the vote, never an early return that leaves the peers inside a collective --
// and every wait on the communicator's stream has a deadline (sync_comm_stream), after which the communicator counts
// as broken and its creation fails on this rank (bench.py and dist_pgo vote on THAT through their control plane).
int Comm::setup_p2p(const std::vector<std::vector<PoseKey>> &exported) {
  Group *grp = grp_;
  const int RS = (grp->d() + 1) * grp->d();
  P2P &x = p2p_state_;
  std::vector<PoseKey> need;
  std::vector<int> need_rows;
  bool ok = true;
  try {
    grp->needed_keys(need, need_rows);
  } catch (...) {
    ok = false; need.clear(); need_rows.clear();
  }
  // ---- all-gather of the needed keys (counts, then padded key lists).  Nothing between here and the final vote may
  // leave this rank's peers inside a collective it never joins: the small all-gathers run out of buffers that exist
  // already (small_allgather: no allocation, so no throw site but the enqueue itself), and a rank whose local work fails
  // -- the list of needed keys, an allocation, an upload -- SAYS so in the next of them (a count of -1, a "not ready"),
  // upon which every rank leaves together, before the collective the failed rank could not have posted.
  std::vector<int> counts;
  if (small_allgather(ok ? (int)need.size() : -1, counts) != 0) return -1;
  if (*std::min_element(counts.begin(), counts.end()) < 0) return p2p_refused("a rank could not list the poses it needs");
  const int stride = std::max(1, *std::max_element(counts.begin(), counts.end()));
  std::vector<int> mine(2 * (size_t)stride, -1), all;
  for (size_t k = 0; k < need.size(); k++) { mine[2 * k] = need[k].first; mine[2 * k + 1] = need[k].second; }
  DevBuf<int> mine_d, all_d;
  int ready = 1;
  try {
    mine_d.upload(mine);
    all_d.alloc(2 * (size_t)stride * nranks_);
  } catch (...) {
    ready = 0;
  }
  std::vector<int> readies;
  if (small_allgather(ready, readies) != 0) return -1;
  if (*std::min_element(readies.begin(), readies.end()) == 0) return p2p_refused("a rank could not allocate its key lists");
  NCCL_OK(rccl().AllGather(mine_d.p, all_d.p, 2 * (size_t)stride, ncclInt32, (ncclComm_t)comm_, cs_));
  sync_comm_stream();
  int got = 1;
  try {
    all_d.download(all);
  } catch (...) {
    got = 0;
  }
  if (small_allgather(got, readies) != 0) return -1;   // (without the keys a rank has no plan: it could not post its messages)
  if (*std::min_element(readies.begin(), readies.end()) == 0) return p2p_refused("a rank could not read the gathered keys");
  std::vector<std::vector<PoseKey>> needed(nranks_);
  for (int r = 0; r < nranks_; r++)
    for (int k = 0; k < counts[r]; k++) needed[r].push_back({all[2 * ((size_t)r * stride + k)], all[2 * ((size_t)r * stride + k) + 1]});
  x.plan = p2p_plan(rank_, exported, needed);   // (the same pure function of the same inputs on every rank)
  // ---- local part: lists, buffers, and the probe records; whatever fails here only changes this rank's vote
  DevBuf<double> probe_d;
  std::vector<double> back;
  const int nrec = grp->num_records();
  try {
    // pack list: the own row of every key sent; unpack lists: every neighbour row that wants a received key
    std::map<PoseKey, int> own_row;
    {
      const auto &keys = grp->sent_keys();
      const auto &rows = grp->sent_rows();
      for (size_t k = 0; k < keys.size(); k++) own_row[keys[k]] = rows[k];
    }
    // (the lists keep the sizes of the plan even when a key cannot be placed: this rank must be able to run the checking
    // exchange, or its peers would wait for it)
    std::vector<int> srows;
    for (const PoseKey &k : x.plan.send_keys) {
      auto it = own_row.find(k);
      if (it == own_row.end()) ok = false;
      srows.push_back(it == own_row.end() ? 0 : it->second);
    }
    std::map<PoseKey, int> slot;
    for (size_t i = 0; i < x.plan.recv_keys.size(); i++) slot[x.plan.recv_keys[i]] = (int)i;
    std::vector<int> rdst, rsrc;
    for (size_t i = 0; i < need.size(); i++) {   // (a key may be wanted by several local nodes: one row each)
      auto it = slot.find(need[i]);
      if (it == slot.end()) { ok = false; continue; }
      rdst.push_back(need_rows[i]);
      rsrc.push_back(it->second);
    }
    x.send_rows.upload(srows);
    x.recv_dst.upload(rdst);
    x.recv_src.upload(rsrc);
    x.send.alloc(std::max<size_t>(x.plan.send_keys.size(), 1) * RS);
    x.recv.alloc(std::max<size_t>(x.plan.recv_keys.size(), 1) * RS);
    // self-check records: (node, pose) of their key in the first two entries
    std::vector<double> probe((size_t)nrec * RS, -1.0);
    const auto &keys = grp->sent_keys();
    const auto &rows = grp->sent_rows();
    for (size_t k = 0; k < keys.size(); k++) { probe[(size_t)rows[k] * RS] = keys[k].first; probe[(size_t)rows[k] * RS + 1] = keys[k].second; }
    probe_d.upload(probe);
  } catch (...) {
    ok = false;
  }
  // ---- the checking exchange.  A rank whose local part failed still posts its sends and receives (from / into whatever
  // buffers it has: the sizes are the plan's), so that its peers' GroupEnd returns.
  if (!x.send.p) x.send.alloc(std::max<size_t>(x.plan.send_keys.size(), 1) * RS);
  if (!x.recv.p) x.recv.alloc(std::max<size_t>(x.plan.recv_keys.size(), 1) * RS);
  if (ok) {
    if (run_p2p(x, probe_d.p, probe_d.p) != 0) ok = false;
  } else {
    P2P bare;   // no pack / unpack lists: the messages alone
    bare.plan = x.plan;
    bare.send.swap(x.send);
    bare.recv.swap(x.recv);
    (void)run_p2p(bare, nullptr, nullptr);
    bare.send.swap(x.send);
    bare.recv.swap(x.recv);
  }
  sync_comm_stream();
  if (ok) {
    try {
      probe_d.download(back);
      for (size_t i = 0; i < need.size() && ok; i++)
        ok = back[(size_t)need_rows[i] * RS] == (double)need[i].first && back[(size_t)need_rows[i] * RS + 1] == (double)need[i].second;
    } catch (...) {
      ok = false;
    }
  }
  double vote = ok ? 0.0 : 1.0;
  if (allreduce_impl(&vote, 1) != 0) return -1;
  if (vote != 0.0) {
    if (rank_ == 0) fprintf(stderr, "[dpgo_amd] WARNING: the neighbour-to-neighbour exchange failed its self-check on %d rank(s); using the all-gather.\n", (int)vote);
    return -1;
  }
  p2p_ = true;
  attach_p2p();
  return 0;
}

// One int per rank through buffers that exist since init() (red_: 64 doubles = 128 ints; ranks beyond that keep the
// all-gather exchange, refused by the caller's vote).  No allocation: the only thing that can fail locally is the enqueue.
int Comm::small_allgather(int mine, std::vector<int> &all) {
  if (2 * nranks_ + 2 > 128) return -1;
  int *dev = reinterpret_cast<int *>(red_.p);
  HIP_OK(hipMemcpyAsync(dev, &mine, sizeof(int), hipMemcpyHostToDevice, cs_));   // (pageable source: returns when staged)
  NCCL_OK(rccl().AllGather(dev, dev + 2, 1, ncclInt32, (ncclComm_t)comm_, cs_));
  sync_comm_stream();
  all.assign(nranks_, 0);
  HIP_OK(hipMemcpy(all.data(), dev + 2, sizeof(int) * nranks_, hipMemcpyDeviceToHost));
  return 0;
}

int Comm::p2p_refused(const char *why) {
  if (rank_ == 0) fprintf(stderr, "[dpgo_amd] WARNING: the neighbour-to-neighbour exchange is not used (%s); using the all-gather.\n", why);
  return -1;
}

// pack the send keys' rows of src_records -> grouped send / recv with the real neighbours -> unpack into dst_records
// (null record arrays: the messages alone).  Between GroupStart and GroupEnd nothing returns early: a Send or Recv that
// fails is remembered and GroupEnd is still called, so the group is never left open.
int Comm::run_p2p(P2P &x, const double *src_records, double *dst_records, hipStream_t st) {
  const int RS = (grp_->d() + 1) * grp_->d();
  hipStream_t cs_ = st ? st : this->cs_;   // (the stream of this exchange: the communicator's own, or the caller's)
  if (src_records && x.send_rows.n > 0) grp_->copy_records(cs_, (int)x.send_rows.n, nullptr, x.send_rows.p, src_records, x.send.p);
  ncclResult_t bad = ncclSuccess;
  NCCL_OK(rccl().GroupStart());
  for (const auto &pr : x.plan.peers) {
    ncclResult_t r = ncclSuccess;
    if (pr.send_cnt) r = rccl().Send(x.send.p + (size_t)pr.send_off * RS, (size_t)pr.send_cnt * RS, ncclFloat64, pr.rank, (ncclComm_t)comm_, cs_);
    if (r != ncclSuccess) bad = r;
    r = ncclSuccess;
    if (pr.recv_cnt) r = rccl().Recv(x.recv.p + (size_t)pr.recv_off * RS, (size_t)pr.recv_cnt * RS, ncclFloat64, pr.rank, (ncclComm_t)comm_, cs_);
    if (r != ncclSuccess) bad = r;
  }
  const ncclResult_t fin = rccl().GroupEnd();
  if (bad != ncclSuccess) NCCL_OK(bad);
  NCCL_OK(fin);
  if (dst_records && x.recv_dst.n > 0) grp_->copy_records(cs_, (int)x.recv_dst.n, x.recv_dst.p, x.recv_src.p, x.recv.p, dst_records);
  return 0;
}

// Wait for the communicator's stream, but not for ever: a peer that died or never posted its half of a message leaves an
// RCCL kernel spinning on the device.  After the deadline the communicator is marked broken (release() then aborts it
// instead of waiting for its stream) and the caller gets an exception, i.e. dpgo_comm_create / the entry point returns -1.
void Comm::sync_comm_stream() { sync_stream(cs_); }

void Comm::sync_stream(hipStream_t st) {
  double limit = 120.0;
  if (const char *e = getenv("DPGO_COMM_TIMEOUT")) limit = std::max(1.0, atof(e));
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    const hipError_t q = hipStreamQuery(st);
    if (q == hipSuccess) return;
    if (q != hipErrorNotReady) HIP_OK(q);
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) {
      fprintf(stderr, "[dpgo_amd] ERROR: rank %d: an RCCL collective did not finish within %.0f s (a peer is gone or "
                      "never joined); the communicator is abandoned.\n", rank_, limit);
      abandon(st);
      throw DeviceError("RCCL collective timed out");
    }
    std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
}

// A wait ran into its deadline: an RCCL kernel is spinning on `st` and will never end by itself.  The unwinding that
// follows frees device buffers (the temporaries of init() / setup_p2p() / allreduce_large(), later the members), and
// hipFree waits for every stream of the device -- so the stuck kernel has to go FIRST: ncclCommAbort here, at the
// timeout site, not in release().  Without ncclCommAbort (an RCCL that does not export it) nothing can end that kernel:
// from then on no device buffer of this process is freed any more (DevBuf leaks, dev_leak_buffers) -- a leak in a process
// that is about to report failure, instead of a hang.  The group whose stream the collective ran on cannot be used
// again either way: it is marked failed (its update() / iterate() return -1, its destructor does not wait).
void Comm::abandon(hipStream_t st) {
  broken_ = true;
  ok_ = false;
  if (comm_ && rccl().CommAbort) {
    (void)rccl().CommAbort((ncclComm_t)comm_);
    comm_ = nullptr;
  } else {
    dev_leak_buffers(true);
  }
  if (grp_ && (st == grp_->stream() || lent_)) grp_->mark_failed();
}

// One rank talking to itself through the grouped send / recv path (tests: the first execution of this code on real
// hardware must not be the first multi-GPU run).  Every exported row is sent to this rank and unpacked into the same row
// of a scratch record array; the records carry their (node, pose) keys.
int Comm::p2p_self_check() {
  if (!comm_ || !cs_ || !(rccl().Send && rccl().Recv && rccl().GroupStart && rccl().GroupEnd)) return -1;
  Group *grp = grp_;
  const int RS = (grp->d() + 1) * grp->d();
  const auto &keys = grp->sent_keys();
  const auto &rows = grp->sent_rows();
  const int n = (int)keys.size();
  if (n == 0) return -1;   // (a group that exports nothing has nothing to check: the caller picks a group with neighbours elsewhere)
  P2P x;
  x.plan.peers.push_back({rank_, 0, n, 0, n});
  x.plan.send_keys = keys;
  x.plan.recv_keys = keys;
  std::vector<int> ident(n);
  for (int i = 0; i < n; i++) ident[i] = i;
  x.send_rows.upload(rows);
  x.recv_dst.upload(rows);
  x.recv_src.upload(ident);
  x.send.alloc((size_t)n * RS);
  x.recv.alloc((size_t)n * RS);
  const int nrec = grp->num_records();
  std::vector<double> probe((size_t)nrec * RS, -1.0), back;
  for (int k = 0; k < n; k++)
    for (int c = 0; c < RS; c++) probe[(size_t)rows[k] * RS + c] = c == 0 ? keys[k].first : (c == 1 ? keys[k].second : 1000.0 * k + c);
  DevBuf<double> src_d, dst_d;
  src_d.upload(probe);
  dst_d.upload(std::vector<double>((size_t)nrec * RS, -7.0));
  HIP_OK(hipDeviceSynchronize());   // (the uploads ran on the null stream; cs_ does not wait for it)
  if (run_p2p(x, src_d.p, dst_d.p) != 0) return -1;
  sync_comm_stream();
  dst_d.download(back);
  std::vector<char> sent(nrec, 0);
  for (int k = 0; k < n; k++) sent[rows[k]] = 1;
  for (int r = 0; r < nrec; r++)
    for (int c = 0; c < RS; c++)
      if (back[(size_t)r * RS + c] != (sent[r] ? probe[(size_t)r * RS + c] : -7.0)) {
        fprintf(stderr, "[dpgo_amd] ERROR: p2p self check: record %d entry %d is %.17g\n", r, c, back[(size_t)r * RS + c]);
        return -1;
      }
  return 0;
}

int Comm::enable_self_exchange() {
  // (also on a communicator made without the exchange lay-out -- a group whose neighbours no rank hosts, the emulated rank of
  // bench.py: exchange() is then all this communicator serves)
  if (nranks_ != 1 || !comm_ || !cs_ || !(rccl().Send && rccl().Recv && rccl().GroupStart && rccl().GroupEnd)) return -1;
  Group *grp = grp_;
  const int RS = (grp->d() + 1) * grp->d();
  const auto &keys = grp->sent_keys();
  const auto &rows = grp->sent_rows();
  const int n = (int)keys.size();
  if (n == 0) return -1;
  P2P &x = p2p_state_;
  x.plan = P2PPlan();
  x.plan.peers.push_back({rank_, 0, n, 0, n});
  x.plan.send_keys = keys;
  x.plan.recv_keys = keys;
  std::vector<int> ident(n);
  for (int i = 0; i < n; i++) ident[i] = i;
  x.send_rows.upload(rows);
  x.recv_dst.upload(rows);      // (into the same rows of the SCRATCH array)
  x.recv_src.upload(ident);
  x.send.alloc((size_t)n * RS);
  x.recv.alloc((size_t)n * RS);
  HIP_OK(hipDeviceSynchronize());
  p2p_ = true;
  self_ = true;
  attach_p2p();
  ok_ = true;
  return 0;
}

int Comm::enable_timing() {
  if (!cs_) return -1;
  sync_comm_stream();
  if (ev_ready_) (void)hipEventDestroy(ev_ready_);
  if (ev_done_) (void)hipEventDestroy(ev_done_);
  ev_ready_ = ev_done_ = nullptr;
  grp_->set_pending_exchange(nullptr);
  HIP_OK(hipEventCreate(&ev_ready_));
  HIP_OK(hipEventCreate(&ev_done_));
  timing_ = true; timed_pending_ = false; time_sum_us_ = 0; time_n_ = 0;
  return 0;
}

// the last exchange's two events, once both have happened (looked at before they are recorded again)
void Comm::take_time() {
  if (!timing_ || !timed_pending_) return;
  if (hipEventQuery(ev_done_) != hipSuccess) return;
  float ms = 0.f;
  if (hipEventElapsedTime(&ms, ev_ready_, ev_done_) == hipSuccess) { time_sum_us_ += 1e3 * ms; time_n_++; }
  timed_pending_ = false;
}

int Comm::exchange_time(double *mean_us, long *count) {
  if (!timing_) return -1;
  if (timed_pending_) {
    if (p2p_) (void)hipEventSynchronize(ev_done_);   // (recorded on the group's stream)
    else sync_comm_stream();
    take_time();
  }
  if (mean_us) *mean_us = time_n_ ? time_sum_us_ / time_n_ : 0.0;
  if (count) *count = time_n_;
  return 0;
}

size_t Comm::bytes_sent_per_exchange() const {
  const size_t RS = (size_t)(grp_->d() + 1) * grp_->d();
  if (p2p_) return p2p_state_.plan.send_keys.size() * RS * sizeof(double);
  return (size_t)stride_ * RS * sizeof(double);
}

// What the neighbour-to-neighbour exchange asks of the group it serves: the pack rides on the tail of iterate(), and a
// collective of ours that never ends on the group's stream (a peer is gone) is aborted when the group's wait for that
// stream runs into its deadline (Group::wait_flag), as sync_stream() does for the communicator's own stream.
void Comm::attach_p2p() {
  grp_->set_exchange_pack(p2p_state_.send_rows.p, (int)p2p_state_.send_rows.n, p2p_state_.send.p);
  grp_->set_stuck_handler(&Comm::cb_stuck, this);
  attached_ = true;
}
void Comm::cb_stuck(void *user) {
  Comm *c = static_cast<Comm *>(user);
  fprintf(stderr, "[dpgo_amd] ERROR: rank %d: the boundary exchange did not finish (a peer is gone or never joined); the communicator "
                  "is abandoned.\n", c->rank_);
  c->abandon(c->grp_->stream());
}

// DPGOHash::communicate for the neighbours hosted by other ranks (DPGOHash.h:64-82), asynchronous: the group's
// stream is not held up; Group::update() joins (Group::pending_exchange).
int Comm::exchange() {
  if (!ok_) return -1;
  {   // DPGO_DEBUG_FAIL_EXCHANGE=n (test hook): the n-th exchange of this process fails the way an RCCL error would
    static const int fail_at = getenv("DPGO_DEBUG_FAIL_EXCHANGE") ? atoi(getenv("DPGO_DEBUG_FAIL_EXCHANGE")) : 0;
    static int calls = 0;
    if (fail_at > 0 && ++calls == fail_at) {
      fprintf(stderr, "[dpgo_amd] ERROR: rank %d: exchange %d failed (DPGO_DEBUG_FAIL_EXCHANGE)\n", rank_, calls);
      return -1;
    }
  }
  const int RS = (grp_->d() + 1) * grp_->d();
  if (timing_) {
    // (the previous exchange was joined by an update() long ago; its events are about to be recorded again)
    if (timed_pending_ && hipEventQuery(ev_done_) != hipSuccess) sync_comm_stream();
    take_time();
  }
  if (p2p_) {
    // Neighbour to neighbour: the whole exchange on the GROUP's own stream (round 6).  On a stream of its own it paid two event
    // hand-overs (31-40 us from "iterate final" to "rows in place" against 24 in line, measured with this rank as its own
    // peer; profiles/r06_exchange_streams.txt) to hide one 9 us kernel.  The pack has ridden on the tail of iterate()
    // (Group::set_exchange_pack) unless that ran before this communicator existed; the unpack is the next update()'s
    // inter-edge pass reading the receive buffer (Group::set_pending_recv), or -- trivial loss -- one indexed copy.
    hipStream_t gs = grp_->stream();
    if (timing_) HIP_OK(hipEventRecord(ev_ready_, gs));
    const bool packed = grp_->take_packed();
    P2P &x = p2p_state_;
    if (run_p2p(x, packed ? nullptr : grp_->Xk_records(), nullptr, gs) != 0) return -1;
    if (self_) {
      // (measurement mode: what arrived are this rank's own rows -- nothing for the inter-edge pass to read; the trajectory stays
      // that of the run without an exchange)
    } else if (x.recv_dst.n > 0 && grp_->set_pending_recv(x.recv.p, (int)x.recv_dst.n, x.recv_dst.p, x.recv_src.p) != 0) {
      grp_->copy_records(gs, (int)x.recv_dst.n, x.recv_dst.p, x.recv_src.p, x.recv.p, grp_->Xk_records());
    }
    if (timing_) HIP_OK(hipEventRecord(ev_done_, gs));
    timed_pending_ = timing_;
    return 0;
  }
  HIP_OK(hipEventRecord(ev_ready_, grp_->stream()));      // Xk of this iteration is final
  HIP_OK(hipStreamWaitEvent(cs_, ev_ready_, 0));
  {
    grp_->pack_sent(send_.p, cs_);
    NCCL_OK(rccl().AllGather(send_.p, gathered_.p, (size_t)stride_ * RS, ncclFloat64, (ncclComm_t)comm_, cs_));
    grp_->unpack_recv(gathered_.p, cs_);
  }
  HIP_OK(hipEventRecord(ev_done_, cs_));
  timed_pending_ = timing_;
  grp_->set_pending_exchange(ev_done_);
  return 0;
}

// sum of n host doubles over all ranks (every rank gets the same bits: one ring, same order)
int Comm::allreduce(double *vals, int n) {
  if (!ok_ || n < 0) return -1;
  return allreduce_impl(vals, n);
}

int Comm::allreduce_impl(double *vals, int n) {
  if (broken_ || !comm_) return -1;
  hipStream_t st = grp_->stream();
  for (int off = 0; off < n; off += 64) {
    const int m = std::min(64, n - off);
    std::memcpy(h_red_, vals + off, sizeof(double) * m);
    HIP_OK(hipMemcpyAsync(red_.p, h_red_, sizeof(double) * m, hipMemcpyHostToDevice, st));
    NCCL_OK(rccl().AllReduce(red_.p, red_.p, m, ncclFloat64, ncclSum, (ncclComm_t)comm_, st));
    HIP_OK(hipMemcpyAsync(h_red_, red_.p, sizeof(double) * m, hipMemcpyDeviceToHost, st));
    sync_stream(st);
    std::memcpy(vals + off, h_red_, sizeof(double) * m);
  }
  return 0;
}

// large host arrays (the global X for the result files): staged through a device buffer of their own
int Comm::allreduce_large(double *vals, size_t n) {
  if (!ok_ || broken_ || !comm_) return -1;
  DevBuf<double> buf;
  buf.alloc(n, false);
  hipStream_t st = grp_->stream();
  HIP_OK(hipMemcpyAsync(buf.p, vals, sizeof(double) * n, hipMemcpyHostToDevice, st));
  NCCL_OK(rccl().AllReduce(buf.p, buf.p, n, ncclFloat64, ncclSum, (ncclComm_t)comm_, st));
  HIP_OK(hipMemcpyAsync(vals, buf.p, sizeof(double) * n, hipMemcpyDeviceToHost, st));
  sync_stream(st);
  return 0;
}

int Comm::barrier() {
  double one = 1.0;
  return allreduce(&one, 1);
}

int Comm::cb_allgather(void *user) {
  Comm *c = static_cast<Comm *>(user);
  if (c->broken_ || !c->comm_) return -1;
  const int RS = (c->grp_->d() + 1) * c->grp_->d();
  try {
    NCCL_OK(rccl().AllGather(c->send_.p, c->gathered_.p, (size_t)c->stride_ * RS, ncclFloat64, (ncclComm_t)c->comm_,
                             c->grp_->stream()));
  } catch (const std::exception &) {
    return -1;
  }
  return 0;
}

// in-place sum of n device doubles over the ranks, enqueued on the group's stream: no host in between
int Comm::cb_allreduce_dev(void *user, double *dev_vals, int n) {
  Comm *c = static_cast<Comm *>(user);
  if (c->broken_ || !c->comm_) return -1;
  try {
    NCCL_OK(rccl().AllReduce(dev_vals, dev_vals, (size_t)n, ncclFloat64, ncclSum, (ncclComm_t)c->comm_, c->grp_->stream()));
  } catch (const std::exception &) {
    return -1;
  }
  return 0;
}

int Comm::cb_allreduce(void *user, double *vals, int n) {
  try {
    if (n > 4096) return static_cast<Comm *>(user)->allreduce_large(vals, (size_t)n);   // (global arrays of the distributed initialisation)
    return static_cast<Comm *>(user)->allreduce(vals, n);
  } catch (const std::exception &) {
    return -1;
  }
}

}  // namespace dpgo
