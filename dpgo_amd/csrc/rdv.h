// Rendezvous of the ranks of one dist_pgo launch (header-only; tests/test_rendezvous.py builds it into a harness).
#pragma once
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

// ---- rendezvous of the ranks: the 128-byte RCCL id travels from rank 0 to the others through a PRIVATE directory
// (mode 0700, owned by this user) with a nonce handshake, so that a file left behind by an earlier run can never be
// taken for this run's id (a stale id makes ncclCommInitRank wait forever):
//   rank r > 0 : writes hello.<r> = its fresh nonce; waits for id.<r> that carries the SAME nonce; removes both
//   rank 0     : answers every hello.<r> it sees with id.<r> = nonce + id (again when the nonce changes: a stale hello
//                gets an answer nobody accepts); rank r counts as arrived when its hello.<r> is gone
// Every file is created under a temporary name with O_EXCL | O_NOFOLLOW and renamed; both sides give up after
// DPGO_RDV_TIMEOUT seconds (default 120) and remove what they wrote on every way out.
namespace dpgo_rdv {
struct RdvFiles {
  std::vector<std::string> mine;
  ~RdvFiles() { for (const auto &f : mine) unlink(f.c_str()); }
};
inline bool write_atomic(const std::string &path, const void *buf, size_t n) {
  const std::string tmp = path + ".tmp." + std::to_string((long)getpid());
  unlink(tmp.c_str());
  const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW, 0600);
  if (fd < 0) return false;
  const bool ok = write(fd, buf, n) == (ssize_t)n;
  close(fd);
  if (!ok || rename(tmp.c_str(), path.c_str()) != 0) { unlink(tmp.c_str()); return false; }
  return true;
}
inline bool read_exact(const std::string &path, void *buf, size_t n) {
  const int fd = open(path.c_str(), O_RDONLY | O_NOFOLLOW);
  if (fd < 0) return false;
  const bool ok = read(fd, buf, n) == (ssize_t)n;
  close(fd);
  return ok;
}
inline int rendezvous(const std::string &dir, int rank, int world, unsigned char *id) {
  using clk = std::chrono::steady_clock;
  const double limit = getenv("DPGO_RDV_TIMEOUT") ? atof(getenv("DPGO_RDV_TIMEOUT")) : 120.0;
  const auto t0 = clk::now();
  auto late = [&] { return std::chrono::duration<double>(clk::now() - t0).count() > limit; };
  if (mkdir(dir.c_str(), 0700) != 0 && errno != EEXIST) { fprintf(stderr, "Cannot create the rendezvous directory %s.\n", dir.c_str()); return -1; }
  struct stat sb;
  if (lstat(dir.c_str(), &sb) != 0 || !S_ISDIR(sb.st_mode) || sb.st_uid != geteuid() || (sb.st_mode & 077) != 0) {
    fprintf(stderr, "The rendezvous directory %s must be a directory of this user with mode 0700.\n", dir.c_str());
    return -1;
  }
  RdvFiles guard;
  struct Msg { unsigned long long nonce; unsigned char id[128]; } msg;
  if (rank > 0) {
    std::random_device rd;
    const unsigned long long nonce = ((unsigned long long)rd() << 32) ^ rd() ^ ((unsigned long long)getpid() << 17);
    const std::string hello = dir + "/hello." + std::to_string(rank), idf = dir + "/id." + std::to_string(rank);
    guard.mine.push_back(hello);
    if (!write_atomic(hello, &nonce, sizeof nonce)) { fprintf(stderr, "Cannot write %s.\n", hello.c_str()); return -1; }
    for (;;) {
      if (read_exact(idf, &msg, sizeof msg) && msg.nonce == nonce) break;
      if (late()) { fprintf(stderr, "Rendezvous: no answer from rank 0 in %s within %g s.\n", dir.c_str(), limit); return -1; }
      std::this_thread::sleep_for(std::chrono::milliseconds(10));
    }
    memcpy(id, msg.id, 128);
    unlink(idf.c_str());
    return 0;   // (the guard removes hello.<r>: that is the acknowledgement rank 0 waits for)
  }
  memcpy(msg.id, id, 128);
  std::vector<unsigned long long> answered(world, 0);
  std::vector<char> seen(world, 0), done(world, 0);
  for (int r = 1; r < world; r++) {
    unlink((dir + "/id." + std::to_string(r)).c_str());   // nothing of an earlier run survives
    guard.mine.push_back(dir + "/id." + std::to_string(r));
  }
  for (int left = world - 1; left > 0;) {
    for (int r = 1; r < world; r++) {
      if (done[r]) continue;
      unsigned long long nonce;
      const std::string hello = dir + "/hello." + std::to_string(r);
      if (read_exact(hello, &nonce, sizeof nonce)) {
        if (!seen[r] || nonce != answered[r]) {
          msg.nonce = nonce;
          if (!write_atomic(dir + "/id." + std::to_string(r), &msg, sizeof msg)) { fprintf(stderr, "Cannot write to %s.\n", dir.c_str()); return -1; }
          answered[r] = nonce;
          seen[r] = 1;
        }
      } else if (seen[r] && access(hello.c_str(), F_OK) != 0) {
        done[r] = 1;
        left--;
      }
    }
    if (left > 0 && late()) { fprintf(stderr, "Rendezvous: %d rank(s) missing in %s after %g s.\n", left, dir.c_str(), limit); return -1; }
    if (left > 0) std::this_thread::sleep_for(std::chrono::milliseconds(10));
  }
  guard.mine.clear();
  rmdir(dir.c_str());   // empty by now (fails harmlessly otherwise)
  return 0;
}
}  // namespace dpgo_rdv

