// Rank-parallel ParticleFMM::EvalDirect from C++ (include/sctl_amd/comm.hpp + fmm-wrapper.hpp): one process per rank, started by
// the test with RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment, the way mpirun or torchrun would.
// Every rank draws the SAME global problem (srand48(0)), keeps an UNEVEN slice of the targets and of the sources — rank r owns the
// points [N w_r, N w_{r+1}) with weights 1:2:3:... (the last rank the most, rank 0 possibly none of the second source type) — and
// calls Eval; its slice of the potential goes to <out>.r<rank>.  Two source types (a double layer and a single layer) act on one
// target type, as in ParticleFMM::test (fmm-wrapper.txx:35-92).
//   fmm_dist <N> <out prefix> [hostonly]
// "hostonly": no evaluation (no GPU needed) — only the communicator: an uneven all-gather of doubles and a barrier.
#include <sctl_amd.hpp>

#include <cstdio>
#include <cstdlib>
#include <string>

using namespace sctl_amd;

static Long cut(Long N, int r, int np) {   // first point of rank r: weights 1, 2, ..., np
  const Long tot = (Long)np * (np + 1) / 2, upto = (Long)r * (r + 1) / 2;
  return N * upto / tot;
}

int main(int argc, char** argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: fmm_dist <N> <out prefix> [hostonly]\n"); return 2; }
  const Long N = std::atol(argv[1]);
  const std::string out = argv[2];
  const bool hostonly = argc > 3 && std::string(argv[3]) == "hostonly";
  const Comm comm = Comm::World();
  const int rank = (int)comm.Rank(), np = (int)comm.Size();
  std::printf("rank %d of %d, device %d, transport %s\n", rank, np, comm.Device(), comm.UsesRCCL() ? "rccl" : "sockets");
  if (hostonly && !comm.Handle()) return 0;   // World() == Self(): nothing to exchange
  if (hostonly) {
    std::vector<double> mine((size_t)(rank + 1) * 3, 100.0 * rank), all((size_t)np * (np + 1) / 2 * 3);
    std::vector<int64_t> bytes((size_t)np);
    for (size_t i = 0; i < mine.size(); i++) mine[i] += (double)i;
    CheckStatus(sctl_amd_comm_allgatherv_host(comm.Handle(), mine.data(), (int64_t)mine.size() * 8, all.data(), (int64_t)all.size() * 8, bytes.data()), "allgatherv_host");
    size_t at = 0;
    for (int r = 0; r < np; r++) {
      SCTL_AMD_ASSERT(bytes[(size_t)r] == (int64_t)(r + 1) * 24);
      for (int i = 0; i < (r + 1) * 3; i++) SCTL_AMD_ASSERT(all[at++] == 100.0 * r + i);
    }
    comm.Barrier();
    std::printf("rank %d host collectives ok\n", rank);
    return 0;
  }
  if (comm.UsesRCCL()) CheckStatus(sctl_amd_comm_selftest(comm.Handle(), 1 << 16), "sctl_amd_comm_selftest");   // ring of sends over xGMI
  srand48(0);
  Vector<double> Xt(N * 3), Xd(N * 3), Nd(N * 3), Fd(N * 3), Xs(N * 3), Fs(N * 3);
  for (auto& a : Xt) a = drand48() - 0.5;
  for (auto& a : Xd) a = drand48() - 0.5;
  for (auto& a : Nd) a = drand48() - 0.5;
  for (auto& a : Fd) a = drand48() - 0.5;
  for (auto& a : Xs) a = drand48() - 0.5;
  for (auto& a : Fs) a = drand48() - 0.5;
  const Long t0 = cut(N, rank, np), t1 = cut(N, rank + 1, np);
  const Long s0 = cut(N, np - 1 - rank, np) , s1 = cut(N, np - rank, np);              // sources the other way round: ranks own different shares
  const Long q0 = (rank == 0 ? 0 : cut(N, rank, np)), q1 = (rank == 0 ? 0 : cut(N, rank + 1, np)), q_first = cut(N, 1, np);   // rank 0 owns no single-layer sources...
  auto part = [](const Vector<double>& v, Long a, Long b, Long dof) { return Vector<double>((b - a) * dof, (Iterator<double>)v.begin() + a * dof, false); };
  Stokes3D_DxU ker_dl;
  Stokes3D_FxU ker_sl;
  Stokes3D_FSxU ker_m2l;
  ParticleFMM<double, 3> fmm(comm);
  fmm.SetAccuracy(16);
  fmm.SetKernels(ker_m2l, ker_m2l, ker_sl);
  fmm.AddTrg("Velocity", ker_m2l, ker_sl);
  fmm.AddSrc("DoubleLayer", ker_dl, ker_dl);
  fmm.AddSrc("SingleLayer", ker_sl, ker_sl);
  fmm.SetKernelS2T("DoubleLayer", "Velocity", ker_dl);
  fmm.SetKernelS2T("SingleLayer", "Velocity", ker_sl);
  fmm.SetTrgCoord("Velocity", part(Xt, t0, t1, 3));
  fmm.SetSrcCoord("DoubleLayer", part(Xd, s0, s1, 3), part(Nd, s0, s1, 3));
  fmm.SetSrcDensity("DoubleLayer", part(Fd, s0, s1, 3));
  // ...and the LAST rank owns rank 0's share of them as well (so that all N are owned by somebody)
  Vector<double> xs_mine = part(Xs, q0, q1, 3), fs_mine = part(Fs, q0, q1, 3);
  if (rank == np - 1) {
    Vector<double> x((q1 - q0 + q_first) * 3), f((q1 - q0 + q_first) * 3);
    for (Long i = 0; i < (q1 - q0) * 3; i++) { x[i] = xs_mine[i]; f[i] = fs_mine[i]; }
    for (Long i = 0; i < q_first * 3; i++) { x[(q1 - q0) * 3 + i] = Xs[i]; f[(q1 - q0) * 3 + i] = Fs[i]; }
    xs_mine = x; fs_mine = f;
  }
  fmm.SetSrcCoord("SingleLayer", xs_mine);
  fmm.SetSrcDensity("SingleLayer", fs_mine);
  Vector<double> U;
  fmm.Eval(U, "Velocity");
  SCTL_AMD_ASSERT(U.Dim() == (t1 - t0) * 3);
  Vector<double> U2;
  fmm.Eval(U2, "Velocity");                    // a second collective evaluation on the same object: same bits
  for (Long i = 0; i < U.Dim(); i++) SCTL_AMD_ASSERT(U[i] == U2[i]);
  U.Write((out + ".r" + std::to_string(rank)).c_str());
  // ONE rank moves its double-layer sources (here: the last rank negates its normals); every rank's next Eval must see it
  if (rank == np - 1) {
    Vector<double> nd = part(Nd, s0, s1, 3);
    Vector<double> flipped(nd.Dim());
    for (Long i = 0; i < nd.Dim(); i++) flipped[i] = -nd[i];
    fmm.SetSrcCoord("DoubleLayer", part(Xd, s0, s1, 3), flipped);
  }
  Vector<double> U3;
  fmm.Eval(U3, "Velocity");
  U3.Write((out + ".moved.r" + std::to_string(rank)).c_str());
  comm.Barrier();
  return 0;
}
