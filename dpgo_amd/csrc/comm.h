// RCCL communicator of one group (one process per GPU): see comm.cpp.
#pragma once
#include <cstddef>
#include <utility>
#include <vector>

#include "group.h"

namespace dpgo {

// Who sends what to whom when every rank talks to its real neighbours only.  exported[r] / needed[r]: the (node, pose)
// keys rank r's group exports / needs from other groups.  Rank `rank` sends peer q the keys q needs among its exports
// and receives from q the keys it needs among q's exports, both in ascending (node, pose) order -- the same rule on
// both ends, so a message's lay-out needs no negotiation.  Pure host logic (tests/test_exchange_gloo.py drives it
// through dpgo_debug_p2p_plan with gloo messages on the CPU).
typedef std::pair<int, int> PoseKey;
struct P2PPlan {
  struct Peer { int rank, send_off, send_cnt, recv_off, recv_cnt; };
  std::vector<Peer> peers;                 // ascending rank, only peers with traffic
  std::vector<PoseKey> send_keys, recv_keys;   // concatenated over the peers
};
P2PPlan p2p_plan(int rank, const std::vector<std::vector<PoseKey>> &exported, const std::vector<std::vector<PoseKey>> &needed);

class Comm {
 public:
  static int unique_id(void *id128);                                  // ncclGetUniqueId, 128 bytes
  // ncclCommInitRank + exchange lay-out (layout = false: the communicator and its stream only, for p2p_self_check on a group
  // whose neighbours no rank hosts)
  Comm(Group *grp, int rank, int nranks, const void *id128, bool layout = true);
  ~Comm();
  Comm(const Comm &) = delete;
  Comm &operator=(const Comm &) = delete;
  bool ok() const { return ok_; }
  int rank() const { return rank_; }
  int nranks() const { return nranks_; }
  int exchange();                          // pack -> (grouped send / recv | ncclAllGather) -> unpack on the communicator's stream
  const char *exchange_kind() const { return p2p_ ? "p2p" : "allgather"; }
  int allreduce(double *vals, int n);      // in-place sum of host doubles over the ranks
  int allreduce_impl(double *vals, int n);
  int allreduce_large(double *vals, size_t n);
  int barrier();
  // test hook (one rank is enough): the grouped ncclSend / ncclRecv path with THIS rank as its own peer -- pack, GroupStart,
  // Send + Recv to self, GroupEnd, unpack, on records that carry their own keys; 0 = every record arrived where it should
  int p2p_self_check();
  // Measurement mode for a communicator of ONE rank (bench.py --emulate-world N --force-exchange): from now on exchange()
  // runs the neighbour-to-neighbour path in its steady state with this rank as its own peer -- the pack on the tail of
  // iterate(), ncclGroupStart, one ncclSend + ncclRecv of every exported record, ncclGroupEnd, on the group's stream; what
  // arrives (this rank's own rows) is not unpacked, so that the iterate's trajectory is that of the run without it.  What
  // the p2p path costs an iteration, short of the wire and of the inter-edge pass reading its neighbour rows through the
  // receive buffer.
  int enable_self_exchange();
  // time from "Xk is final" (ev_ready_, the group's stream) to "the neighbour rows are in place" (ev_done_, the
  // communicator's stream), mean over the exchanges since enable_timing(); events carry time stamps only in this mode
  int enable_timing();
  int exchange_time(double *mean_us, long *count);
  size_t bytes_sent_per_exchange() const;   // what this rank hands to RCCL per exchange (p2p: its peers' records; all-gather: one padded block)

 private:
  void init(const void *id128, bool layout);
  void release();
  static int cb_allgather(void *user);
  static int cb_allreduce(void *user, double *vals, int n);
  static int cb_allreduce_dev(void *user, double *dev_vals, int n);
  Group *grp_ = nullptr;
  int rank_ = 0, nranks_ = 1, stride_ = 1;
  bool ok_ = false;
  void *comm_ = nullptr;                   // ncclComm_t
  hipStream_t cs_ = nullptr;
  hipEvent_t ev_ready_ = nullptr, ev_done_ = nullptr;
  DevBuf<double> send_, gathered_, red_;
  double *h_red_ = nullptr;
  // neighbour-to-neighbour exchange (grouped ncclSend / ncclRecv): the default with more than one rank once its
  // self-check has passed on every rank; DPGO_EXCHANGE=allgather keeps the all-gather
  bool p2p_ = false;
  bool broken_ = false;   // a wait on the communicator's stream ran into its deadline: the stream is never waited for again
  bool lent_ = false;     // the group works with this communicator's collectives (taken back by release())
  bool self_ = false;     // enable_self_exchange()
  bool timing_ = false, timed_pending_ = false;
  double time_sum_us_ = 0;
  long time_n_ = 0;
  void take_time();
  bool attached_ = false;   // attach_p2p(): the group packs for us and knows whom to call when its stream is stuck
  void attach_p2p();
  static void cb_stuck(void *user);
  struct P2P {            // one neighbour-to-neighbour exchange: the plan, its message buffers, its pack / unpack lists
    P2PPlan plan;
    DevBuf<double> send, recv;
    DevBuf<int> send_rows, recv_dst, recv_src;
  };
  P2P p2p_state_;
  int setup_p2p(const std::vector<std::vector<PoseKey>> &exported);
  int run_p2p(P2P &x, const double *src_records, double *dst_records, hipStream_t st = nullptr);   // pack from / unpack into record arrays, on cs_ (or st)
  void sync_comm_stream();   // hipStreamSynchronize(cs_) with a deadline (DPGO_COMM_TIMEOUT seconds, default 120)
  void sync_stream(hipStream_t st);
  void abandon(hipStream_t st);                             // a wait timed out: abort the communicator NOW (comm.cpp)
  int small_allgather(int mine, std::vector<int> &all);     // one int per rank, no allocation
  int p2p_refused(const char *why);
};

}  // namespace dpgo
