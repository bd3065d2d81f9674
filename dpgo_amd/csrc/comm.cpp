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
#include <cstdio>
#include <cstring>

#include "comm.h"

namespace dpgo {

namespace {
struct Rccl {
  void *h = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
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

int Comm::unique_id(void *id128) {
  static_assert(sizeof(ncclUniqueId) == 128, "the C ABI hands the id over as 128 bytes");
  if (!rccl().ok) return -1;
  ncclUniqueId id;
  if (rccl().GetUniqueId(&id) != ncclSuccess) return -1;
  std::memcpy(id128, &id, sizeof(id));
  return 0;
}

Comm::Comm(Group *grp, int rank, int nranks, const void *id128) : grp_(grp), rank_(rank), nranks_(nranks) {
  if (!rccl().ok || !grp || rank < 0 || rank >= nranks) return;
  // a constructor that throws never runs its destructor: release what was created, then pass the error on
  try {
    init(id128);
  } catch (...) {
    release();
    throw;
  }
}

void Comm::init(const void *id128) {
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
  // ---- who exports what: all-gather of the key counts, then of the (node, pose) keys, through RCCL itself
  const auto &keys = grp->sent_keys();
  red_.alloc(64);
  HIP_OK(hipHostMalloc((void **)&h_red_, sizeof(double) * 64, hipHostMallocDefault));
  DevBuf<int> cnt_d, cnts_d;
  cnt_d.upload(std::vector<int>{(int)keys.size()});
  cnts_d.alloc(nranks);
  NCCL_OK(rccl().AllGather(cnt_d.p, cnts_d.p, 1, ncclInt32, (ncclComm_t)comm_, cs_));
  HIP_OK(hipStreamSynchronize(cs_));
  std::vector<int> counts;
  cnts_d.download(counts);
  stride_ = std::max(1, *std::max_element(counts.begin(), counts.end()));
  std::vector<int> mine(2 * (size_t)stride_, -1), all;
  for (size_t k = 0; k < keys.size(); k++) { mine[2 * k] = keys[k].first; mine[2 * k + 1] = keys[k].second; }
  DevBuf<int> mine_d, all_d;
  mine_d.upload(mine);
  all_d.alloc(2 * (size_t)stride_ * nranks);
  NCCL_OK(rccl().AllGather(mine_d.p, all_d.p, 2 * (size_t)stride_, ncclInt32, (ncclComm_t)comm_, cs_));
  HIP_OK(hipStreamSynchronize(cs_));
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
  // AMM-PGO* and the global evaluations borrow the same buffers on the group's own stream
  if (grp->set_collectives(send_.p, gathered_.p, &Comm::cb_allgather, &Comm::cb_allreduce, this) != 0) return;
  ok_ = true;
}

Comm::~Comm() { release(); }

void Comm::release() {
  if (grp_) {
    // an exchange that no update() has joined yet: let it finish, and take its event away from the group before the
    // event is destroyed
    if (ev_done_) (void)hipEventSynchronize(ev_done_);
    grp_->set_pending_exchange(nullptr);
    grp_->set_collectives(nullptr, nullptr, nullptr, nullptr, nullptr);
  }
  if (cs_) (void)hipStreamSynchronize(cs_);
  if (comm_ && rccl().ok) (void)rccl().CommDestroy((ncclComm_t)comm_);
  if (ev_ready_) (void)hipEventDestroy(ev_ready_);
  if (ev_done_) (void)hipEventDestroy(ev_done_);
  if (cs_) (void)hipStreamDestroy(cs_);
  if (h_red_) (void)hipHostFree(h_red_);
  comm_ = nullptr; ev_ready_ = ev_done_ = nullptr; cs_ = nullptr; h_red_ = nullptr;
  ok_ = false;
}

// DPGOHash::communicate for the neighbours hosted by other ranks (DPGOHash.h:64-82), asynchronous: the group's
// stream is not held up; Group::update() joins (Group::pending_exchange).
int Comm::exchange() {
  if (!ok_) return -1;
  const int RS = (grp_->d() + 1) * grp_->d();
  HIP_OK(hipEventRecord(ev_ready_, grp_->stream()));      // Xk of this iteration is final
  HIP_OK(hipStreamWaitEvent(cs_, ev_ready_, 0));
  grp_->pack_sent(send_.p, cs_);
  NCCL_OK(rccl().AllGather(send_.p, gathered_.p, (size_t)stride_ * RS, ncclFloat64, (ncclComm_t)comm_, cs_));
  grp_->unpack_recv(gathered_.p, cs_);
  HIP_OK(hipEventRecord(ev_done_, cs_));
  grp_->set_pending_exchange(ev_done_);
  return 0;
}

// sum of n host doubles over all ranks (every rank gets the same bits: one ring, same order)
int Comm::allreduce(double *vals, int n) {
  if (!ok_ || n < 0) return -1;
  hipStream_t st = grp_->stream();
  for (int off = 0; off < n; off += 64) {
    const int m = std::min(64, n - off);
    std::memcpy(h_red_, vals + off, sizeof(double) * m);
    HIP_OK(hipMemcpyAsync(red_.p, h_red_, sizeof(double) * m, hipMemcpyHostToDevice, st));
    NCCL_OK(rccl().AllReduce(red_.p, red_.p, m, ncclFloat64, ncclSum, (ncclComm_t)comm_, st));
    HIP_OK(hipMemcpyAsync(h_red_, red_.p, sizeof(double) * m, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    std::memcpy(vals + off, h_red_, sizeof(double) * m);
  }
  return 0;
}

// large host arrays (the global X for the result files): staged through a device buffer of their own
int Comm::allreduce_large(double *vals, size_t n) {
  if (!ok_) return -1;
  DevBuf<double> buf;
  buf.alloc(n, false);
  hipStream_t st = grp_->stream();
  HIP_OK(hipMemcpyAsync(buf.p, vals, sizeof(double) * n, hipMemcpyHostToDevice, st));
  NCCL_OK(rccl().AllReduce(buf.p, buf.p, n, ncclFloat64, ncclSum, (ncclComm_t)comm_, st));
  HIP_OK(hipMemcpyAsync(vals, buf.p, sizeof(double) * n, hipMemcpyDeviceToHost, st));
  HIP_OK(hipStreamSynchronize(st));
  return 0;
}

int Comm::barrier() {
  double one = 1.0;
  return allreduce(&one, 1);
}

int Comm::cb_allgather(void *user) {
  Comm *c = static_cast<Comm *>(user);
  const int RS = (c->grp_->d() + 1) * c->grp_->d();
  try {
    NCCL_OK(rccl().AllGather(c->send_.p, c->gathered_.p, (size_t)c->stride_ * RS, ncclFloat64, (ncclComm_t)c->comm_,
                             c->grp_->stream()));
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
