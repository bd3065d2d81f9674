// RCCL communicator of one group (one process per GPU): see comm.cpp.
#pragma once
#include <cstddef>

#include "group.h"

namespace dpgo {

class Comm {
 public:
  static int unique_id(void *id128);                                  // ncclGetUniqueId, 128 bytes
  Comm(Group *grp, int rank, int nranks, const void *id128);          // ncclCommInitRank + exchange lay-out
  ~Comm();
  Comm(const Comm &) = delete;
  Comm &operator=(const Comm &) = delete;
  bool ok() const { return ok_; }
  int rank() const { return rank_; }
  int nranks() const { return nranks_; }
  int exchange();                          // pack -> ncclAllGather -> unpack on the communicator's stream
  int allreduce(double *vals, int n);      // in-place sum of host doubles over the ranks
  int allreduce_large(double *vals, size_t n);
  int barrier();

 private:
  void init(const void *id128);
  void release();
  static int cb_allgather(void *user);
  static int cb_allreduce(void *user, double *vals, int n);
  Group *grp_ = nullptr;
  int rank_ = 0, nranks_ = 1, stride_ = 1;
  bool ok_ = false;
  void *comm_ = nullptr;                   // ncclComm_t
  hipStream_t cs_ = nullptr;
  hipEvent_t ev_ready_ = nullptr, ev_done_ = nullptr;
  DevBuf<double> send_, gathered_, red_;
  double *h_red_ = nullptr;
};

}  // namespace dpgo
