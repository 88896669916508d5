// Comm: the slice of the reference's communicator (include/sctl/comm.hpp:35-435) that ParticleFMM's direct path uses — Self,
// World, Rank, Size, Barrier — for rank-parallel runs with ONE PROCESS PER GPU.  The reference's World() is MPI_COMM_WORLD
// (comm.txx:117-140); there is no MPI here: World() reads the launcher's environment,
//     rank        SCTL_AMD_RANK | RANK | OMPI_COMM_WORLD_RANK | PMI_RANK | SLURM_PROCID
//     size        SCTL_AMD_WORLD_SIZE | WORLD_SIZE | OMPI_COMM_WORLD_SIZE | PMI_SIZE | SLURM_NTASKS
//     rendezvous  MASTER_ADDR (default 127.0.0.1), MASTER_PORT (default 29411)
//     GPU         SCTL_AMD_LOCAL_RANK | LOCAL_RANK | OMPI_COMM_WORLD_LOCAL_RANK | SLURM_LOCALID (default: rank) modulo the device count
// (what torchrun, mpirun and srun export), and builds a sctl_amd_comm: a TCP rendezvous plus, when every rank has its own GPU,
// an RCCL communicator whose all-gathers run GPU to GPU over xGMI (sctl_amd/csrc/comm.hip).  MASTER_ADDR may be a dotted
// address or a host name (`localhost`, a node name).
// WHEN World() goes rank-parallel: a world size above 1 alone is not enough — independent single-rank tasks started by srun or
// mpirun also see SLURM_NTASKS / PMI_SIZE > 1 and must not sit in a rendezvous nobody else joins.  It takes
//     SCTL_AMD_WORLD_SIZE > 1                                   (explicit opt-in; the rendezvous defaults apply), or
//     a launcher's size variable > 1 AND both MASTER_ADDR and MASTER_PORT in the environment (torchrun exports them; under
//     mpirun / srun the job script does: that is the opt-in),
// and SCTL_AMD_COMM=0 always keeps World() == Self().  With no such environment World() is Self(): Rank() = 0, Size() = 1,
// like the reference built without SCTL_HAVE_MPI (comm.txx:198-212).
// ParticleFMM::EvalDirect then follows the reference's rank-parallel contract (fmm-wrapper.txx:504-561): every rank passes the
// sources and targets IT owns and gets the potential at ITS targets from the sources of ALL ranks.
// Inside one process several GPUs are still driven through DeviceSet (common.hpp); the two do not combine.
#ifndef SCTL_AMD_COMM_HPP_
#define SCTL_AMD_COMM_HPP_

#include <cstdio>
#include <cstdlib>
#include <initializer_list>
#include <memory>

#include "common.hpp"

namespace sctl_amd {

class Comm {
 public:
  Comm() {}
  static Comm Self() { return Comm(); }
  static Comm World() {
    static Comm* world = new Comm(FromEnvironment());   // never destroyed: its RCCL communicator must not outlive the HIP runtime's own teardown
    return *world;
  }
  Integer Rank() const { return h_ ? h_->rank : 0; }
  Integer Size() const { return h_ ? h_->size : 1; }
  void Barrier() const {
    if (h_ && h_->size > 1) CheckStatus(sctl_amd_comm_barrier(h_->c), "sctl_amd_comm_barrier");
  }
  // the GPU this rank evaluates on (rank-parallel runs: one per rank), and the library handle for the collective entries
  int Device() const { return h_ ? h_->device : 0; }
  bool UsesRCCL() const { return h_ && h_->rccl; }
  sctl_amd_comm* Handle() const { return h_ ? h_->c : nullptr; }

  // explicit construction (a launcher that exports none of the variables above)
  static Comm Connect(int rank, int size, const char* master_addr, int master_port, int device, bool sockets_only = false) {
    Comm comm;
    if (size <= 1) return comm;
    std::shared_ptr<State> st(new State);
    CheckStatus(sctl_amd_comm_create(rank, size, master_addr, master_port, device, sockets_only ? SCTL_AMD_COMM_SOCKETS_ONLY : 0, &st->c), "sctl_amd_comm_create");
    int transport = SCTL_AMD_COMM_SOCKETS;
    CheckStatus(sctl_amd_comm_info(st->c, &st->rank, &st->size, &st->device, &transport), "sctl_amd_comm_info");
    st->rccl = (transport == SCTL_AMD_COMM_RCCL);
    comm.h_ = st;
    return comm;
  }

 private:
  struct State {
    sctl_amd_comm* c = nullptr;
    int rank = 0, size = 1, device = 0;
    bool rccl = false;
    ~State() { if (c) sctl_amd_comm_destroy(c); }
  };
  static long Env(std::initializer_list<const char*> names, long fallback) {
    for (const char* n : names)
      if (const char* v = std::getenv(n)) { char* end = nullptr; const long x = std::strtol(v, &end, 10); if (end != v) return x; }
    return fallback;
  }
  static Comm FromEnvironment() {
    if (Env({"SCTL_AMD_COMM"}, 1) == 0) return Comm();
    const long explicit_size = Env({"SCTL_AMD_WORLD_SIZE"}, 0);
    const long size = explicit_size > 0 ? explicit_size : Env({"WORLD_SIZE", "OMPI_COMM_WORLD_SIZE", "PMI_SIZE", "SLURM_NTASKS"}, 1);
    if (size <= 1) return Comm();
    if (explicit_size <= 1 && !(std::getenv("MASTER_ADDR") && std::getenv("MASTER_PORT"))) {
      // A launcher reports several tasks but names no rendezvous: they run INDEPENDENTLY, each on its own sources — under the reference's contract
      // (the potential from ALL ranks' sources, fmm-wrapper.txx:504-561) that is another answer, so it is said once, loudly.
      if (Env({"SCTL_AMD_COMM_QUIET"}, 0) == 0)
        std::fprintf(stderr, "sctl_amd: %ld tasks detected (launcher environment) but no rendezvous is named: every task evaluates on its own, NOT rank-parallel. "
                             "Set MASTER_ADDR and MASTER_PORT (or SCTL_AMD_WORLD_SIZE / SCTL_AMD_RANK) for one rank-parallel evaluation, SCTL_AMD_COMM=0 to "
                             "run independent tasks without this message.\n", size);
      return Comm();
    }
    const long rank = Env({"SCTL_AMD_RANK", "RANK", "OMPI_COMM_WORLD_RANK", "PMI_RANK", "SLURM_PROCID"}, 0);
    const long local = Env({"SCTL_AMD_LOCAL_RANK", "LOCAL_RANK", "OMPI_COMM_WORLD_LOCAL_RANK", "SLURM_LOCALID"}, rank);
    const int ndev = sctl_amd_device_count();
    const char* addr = std::getenv("MASTER_ADDR");
    return Connect((int)rank, (int)size, addr ? addr : "127.0.0.1", (int)Env({"MASTER_PORT"}, 29411), ndev > 0 ? (int)(local % ndev) : -1,
                   Env({"SCTL_AMD_COMM_SOCKETS_ONLY"}, 0) != 0);
  }
  std::shared_ptr<State> h_;
};

}  // namespace sctl_amd
#endif  // SCTL_AMD_COMM_HPP_
