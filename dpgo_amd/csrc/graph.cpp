#include "graph.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <set>
#include <sstream>
#include <thread>

#include <sched.h>

namespace dpgo {

int host_threads() {
  static int cached = 0;
  if (cached) return cached;
  if (const char *e = getenv("DPGO_HOST_THREADS")) {
    const int v = atoi(e);
    if (v > 0) return cached = v;
  }
  int n = (int)std::thread::hardware_concurrency();
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
  if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char q[64];
    long period = 0;
    if (fscanf(f, "%63s %ld", q, &period) == 2 && q[0] != 'm' && period > 0) {
      long quota = atol(q);
      if (quota > 0) n = std::min<long>(n, std::max<long>(1, quota / period));
    }
    fclose(f);
  } else if (FILE *f1 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
    long quota = -1, period = 100000;
    if (fscanf(f1, "%ld", &quota) != 1) quota = -1;
    fclose(f1);
    if (FILE *f2 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
      if (fscanf(f2, "%ld", &period) != 1) period = 100000;
      fclose(f2);
    }
    if (quota > 0 && period > 0) n = std::min<long>(n, std::max<long>(1, quota / period));
  }
  cached = std::max(1, std::min(n, 64));
  return cached;
}

static double trace_inv3(const double a[9]) {
  // trace of the inverse of a 3x3 matrix = (sum of principal 2x2 minors) / det
  double c00 = a[4] * a[8] - a[5] * a[7];
  double c11 = a[0] * a[8] - a[2] * a[6];
  double c22 = a[0] * a[4] - a[1] * a[3];
  double det = a[0] * c00 - a[1] * (a[3] * a[8] - a[5] * a[6]) + a[2] * (a[3] * a[7] - a[4] * a[6]);
  return (c00 + c11 + c22) / det;
}

int read_g2o_file(const std::string &filename, int &num_poses, int &d, measurements_t &out) {
  std::ifstream in(filename);
  if (!in.is_open()) {
    fprintf(stderr, "[dpgo_amd] ERROR: cannot open %s\n", filename.c_str());
    return -1;
  }
  out.clear();
  num_poses = 0;
  d = 0;
  std::string line, token;
  while (std::getline(in, line)) {
    std::stringstream ss(line);
    token.clear();
    ss >> token;
    if (token.empty()) continue;
    Measurement m{};
    m.inode = m.jnode = 0;
    if (token == "EDGE_SE2") {
      double dx, dy, dth, I11, I12, I13, I22, I23, I33;
      ss >> m.ipose >> m.jpose >> dx >> dy >> dth >> I11 >> I12 >> I13 >> I22 >> I23 >> I33;
      double c = std::cos(dth), s = std::sin(dth);
      m.R[0] = c; m.R[1] = -s; m.R[2] = s; m.R[3] = c;
      m.t[0] = dx; m.t[1] = dy;
      // tau = 2 / tr(TranCov^-1), kappa = I33   (DPGO_utils.cpp:63-67)
      double det = I11 * I22 - I12 * I12;
      m.tau = 2.0 / ((I11 + I22) / det);
      m.kappa = I33;
      d = 2;
    } else if (token == "EDGE_SE3:QUAT") {
      double dx, dy, dz, qx, qy, qz, qw, I[21];
      ss >> m.ipose >> m.jpose >> dx >> dy >> dz >> qx >> qy >> qz >> qw;
      for (int k = 0; k < 21; k++) ss >> I[k];
      // Quaternion -> rotation without normalisation (Eigen semantics, DPGO_utils.cpp:100-101)
      double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
      double twx = tx * qw, twy = ty * qw, twz = tz * qw;
      double txx = tx * qx, txy = ty * qx, txz = tz * qx;
      double tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
      m.R[0] = 1 - (tyy + tzz); m.R[1] = txy - twz;       m.R[2] = txz + twy;
      m.R[3] = txy + twz;       m.R[4] = 1 - (txx + tzz); m.R[5] = tyz - twx;
      m.R[6] = txz - twy;       m.R[7] = tyz + twx;       m.R[8] = 1 - (txx + tyy);
      m.t[0] = dx; m.t[1] = dy; m.t[2] = dz;
      // I11 I12 I13 I14 I15 I16 | I22 I23 I24 I25 I26 | I33 I34 I35 I36 | I44 I45 I46 | I55 I56 | I66
      double tc[9] = {I[0], I[1], I[2], I[1], I[6], I[7], I[2], I[7], I[11]};
      double rc[9] = {I[15], I[16], I[17], I[16], I[18], I[19], I[17], I[19], I[20]};
      m.tau = 3.0 / trace_inv3(tc);            // :107-109
      m.kappa = 3.0 / (2.0 * trace_inv3(rc));  // :114-116
      d = 3;
    } else if (token == "VERTEX_SE2" || token == "VERTEX_SE3:QUAT") {
      continue;
    } else {
      fprintf(stderr, "[dpgo_amd] ERROR: unrecognized type: %s!\n", token.c_str());
      return -1;
    }
    num_poses = std::max(num_poses, std::max(m.ipose, m.jpose));
    out.push_back(m);
  }
  num_poses++;
  return out.empty() ? -1 : 0;
}

int partition(Graph &g, int num_nodes) {
  if (num_nodes < 1 || g.all.empty()) return -1;
  g.num_nodes = num_nodes;
  const int q = g.num_poses / num_nodes;
  const int inc_n = g.num_poses - num_nodes * q;
  const int inc = inc_n * (q + 1);
  auto index = [&](int i, int &node, int &pose) {
    if (i < inc) {
      node = i / (q + 1);
      pose = i % (q + 1);
    } else {
      i -= inc;
      node = i / q + inc_n;
      pose = i % q;
    }
  };
  g.measurements.assign(num_nodes, measurements_t());
  g.g_index.assign(num_nodes, std::map<int, int>());
  for (const auto &mm : g.all) {
    if (mm.ipose < 0 || mm.ipose >= g.num_poses || mm.jpose < 0 || mm.jpose >= g.num_poses) {
      fprintf(stderr, "[dpgo_amd] ERROR: pose id out of range in edge (%d, %d): num_poses = %d.\n", mm.ipose, mm.jpose, g.num_poses);
      return -1;
    }
    Measurement m = mm;
    index(mm.ipose, m.inode, m.ipose);
    index(mm.jpose, m.jnode, m.jpose);
    if (m.inode >= num_nodes || m.jnode >= num_nodes) return -1;
    g.g_index[m.inode].emplace(m.ipose, mm.ipose);
    g.g_index[m.jnode].emplace(m.jpose, mm.jpose);
    g.measurements[m.inode].push_back(m);
    if (m.inode != m.jnode) g.measurements[m.jnode].push_back(m);
  }
  return 0;
}

int read_g2o(const std::string &filename, int num_nodes, Graph &g) {
  if (read_g2o_file(filename, g.num_poses, g.d, g.all) != 0) return -1;
  return partition(g, num_nodes);
}

int DataInfo::tail(const Measurement &mm) const {
  int k = index.at({mm.inode, mm.ipose});
  return mm.inode == node ? k : n[0] + k;
}
int DataInfo::head(const Measurement &mm) const {
  int k = index.at({mm.jnode, mm.jpose});
  return mm.jnode == node ? k : n[0] + k;
}

int generate_data_info(int a, int d, const measurements_t &meas, DataInfo &info) {
  info = DataInfo();
  info.node = a;
  if (meas.empty()) {
    fprintf(stderr, "[dpgo_amd] WARNING: No measurements are specified for node %d.\n", a);
    return -1;
  }
  std::set<int> own;
  std::set<std::pair<int, int>> nbr;
  std::map<int, std::set<int>> sent;
  for (const auto &m : meas) {
    if (m.inode != a && m.jnode != a) {
      fprintf(stderr, "[dpgo_amd] ERROR: The measurement is not associated with node %d.\n", a);
      continue;
    }
    if (m.inode == a && m.jnode == a) info.intra.push_back(m); else info.inter.push_back(m);
    if (m.inode == a) own.insert(m.ipose); else nbr.insert({m.inode, m.ipose});
    if (m.jnode == a) own.insert(m.jpose); else nbr.insert({m.jnode, m.jpose});
    if (m.inode != a) sent[m.inode].insert(m.jpose);
    if (m.jnode != a) sent[m.jnode].insert(m.ipose);
  }
  info.d = d;
  info.m[0] = (int)info.intra.size();
  info.m[1] = (int)info.inter.size();
  // ordering: own poses by id, then neighbours by (node, id)   (DPGO_utils.cpp:400-418)
  int k = 0;
  for (int p : own) { info.index[{a, p}] = k++; info.own_pose.push_back(p); }
  info.n[0] = k;
  k = 0;
  for (const auto &np : nbr) {
    info.index[np] = k++;
    info.nbr_key.push_back(np);
    info.recv[np.first].push_back({np.second, k - 1});
  }
  info.n[1] = k;
  for (const auto &s : sent)
    for (int p : s.second) info.sent[s.first].push_back(info.index.at({a, p}));
  return 0;
}

}  // namespace dpgo
