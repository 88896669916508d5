// Single-process communicator with the interface subset of the reference's Comm (include/sctl/comm.hpp:35-435) that
// ParticleFMM's direct path touches: Self/World, Rank, Size.  It behaves like the reference built WITHOUT
// SCTL_HAVE_MPI (comm.txx:198-212: Rank() = 0, Size() = 1).  On an MI355X node the "ranks" of
// ParticleFMM::EvalDirect are the GPUs in DeviceSet (common.hpp), driven from this one process; multi-process runs
// (one process per GPU, RCCL all-gather over xGMI) go through sctl_amd.distributed on the Python side.
#ifndef SCTL_AMD_COMM_HPP_
#define SCTL_AMD_COMM_HPP_

#include "common.hpp"

namespace sctl_amd {

class Comm {
 public:
  Comm() {}
  static Comm Self() { return Comm(); }
  static Comm World() { return Comm(); }
  Integer Rank() const { return 0; }
  Integer Size() const { return 1; }
  void Barrier() const {}
};

}  // namespace sctl_amd
#endif  // SCTL_AMD_COMM_HPP_
