// Pose-graph data: .g2o loader, node partition and per-node index maps.
//
// Host side of the drop-in boundary.  Mirrors the behaviour of
//   read_g2o_file / read_g2o     C++/DPGO/src/DPGO_utils.cpp:8-202
//   generate_data_info           C++/DPGO/src/DPGO_utils.cpp:326-438
// of the reference (same precision formulas, same contiguous-range partition,
// same own-then-neighbour local ordering) without Eigen.
#pragma once
#include <map>
#include <string>
#include <utility>
#include <vector>

namespace dpgo {

// RelativePoseMeasurement (C++/DPGO/include/DPGO/RelativePoseMeasurement.h:11-36).
// R is row-major d x d in the first d*d entries, t in the first d.
struct Measurement {
  int inode, ipose, jnode, jpose;
  double R[9];
  double t[3];
  double kappa, tau;
};
using measurements_t = std::vector<Measurement>;

// Output of read_g2o (DPGO_utils.cpp:140-202).
struct Graph {
  int d = 0;
  int num_poses = 0;
  int num_nodes = 0;
  measurements_t all;                          // file order, global pose ids in ipose/jpose, node = 0
  std::vector<measurements_t> measurements;    // per node, inter-node edges duplicated on both endpoints
  std::vector<std::map<int, int>> g_index;     // per node: local pose id -> global pose id
};

// Parse a .g2o file (EDGE_SE2 / EDGE_SE3:QUAT; VERTEX_* ignored).  Returns 0 / -1.
int read_g2o_file(const std::string &filename, int &num_poses, int &d, measurements_t &out);
// Partition `all` into num_nodes contiguous ranges (DPGO_utils.cpp:147-158).
int partition(Graph &g, int num_nodes);
int read_g2o(const std::string &filename, int num_nodes, Graph &g);

// generate_data_info (DPGO_utils.cpp:326-438).
struct DataInfo {
  int node = 0;
  int d = 0;
  measurements_t intra, inter;
  int n[2] = {0, 0};   // own / neighbour pose counts
  int m[2] = {0, 0};   // intra / inter measurement counts
  // (node, pose) -> local index; own poses are [0, n0), neighbours [0, n1)
  std::map<std::pair<int, int>, int> index;
  std::vector<int> own_pose;                            // own local index -> pose id
  std::vector<std::pair<int, int>> nbr_key;            // neighbour local index -> (node, pose)
  std::map<int, std::vector<int>> sent;                // nbr node -> own local indices it needs (sorted)
  std::map<int, std::vector<std::pair<int, int>>> recv;  // nbr node -> (its pose id, our nbr local index)
  // unified pose id inside the node: own k -> k, neighbour k -> n0 + k
  int tail(const Measurement &m) const;
  int head(const Measurement &m) const;
};
int generate_data_info(int a, int d, const measurements_t &meas, DataInfo &info);

// Host threads this process may really use: min(cgroup CPU quota, affinity mask), at least 1.
// (The GPU boxes expose 256 logical CPUs behind a 16-core quota; spinning 256 OpenMP threads there
// is an order of magnitude slower than 16.)
int host_threads();

}  // namespace dpgo
