// Rank-parallel runs from C++ (sctl_amd_comm_*): what the reference gets from MPI in ParticleFMM::EvalDirect
// (include/sctl/fmm-wrapper.txx:504-561: every rank owns some targets and some sources; source blocks travel round a ring),
// re-designed for one process per GPU of an MI355X node: the sources of all ranks are ALL-GATHERED into every rank's
// device-resident operator (they are O(N) data for O(N^2/P) work) — over RCCL/xGMI, device buffer to device buffer, when
// every rank has its own GPU — and each rank then evaluates its own targets.  No MPI in this image: ranks find each other
// through a TCP rendezvous on MASTER_ADDR:MASTER_PORT (rank 0 listens), which also carries the RCCL unique id, the small
// host-side collectives (counts, barrier) and — when several ranks SHARE one GPU, which RCCL refuses (a one-GPU rehearsal) —
// the data itself.  librccl is dlopen()ed on first use, so single-rank users of libsctl_amd.so do not load it.
#include "internal.hpp"
#include "workspace.hpp"

#include <arpa/inet.h>
#include <dlfcn.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <sys/socket.h>
#include <unistd.h>

#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace sctl_amd {
namespace {

bool send_all(int fd, const void* p, size_t n) {
  const char* c = (const char*)p;
  while (n) {
    const ssize_t k = ::send(fd, c, n, MSG_NOSIGNAL);
    if (k <= 0) return false;
    c += k; n -= (size_t)k;
  }
  return true;
}
bool recv_all(int fd, void* p, size_t n) {
  char* c = (char*)p;
  while (n) {
    const ssize_t k = ::recv(fd, c, n, 0);
    if (k <= 0) return false;
    c += k; n -= (size_t)k;
  }
  return true;
}

// the few RCCL entry points used, resolved at run time (types from rccl.h restated: the library is not a link dependency)
struct RcclUniqueId { char internal[128]; };
struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(RcclUniqueId*) = nullptr;
  int (*CommInitRank)(void**, int, RcclUniqueId, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool load() {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (lib) break;
    }
    if (!lib) return false;
    GetUniqueId = (int (*)(RcclUniqueId*))dlsym(lib, "ncclGetUniqueId");
    CommInitRank = (int (*)(void**, int, RcclUniqueId, int))dlsym(lib, "ncclCommInitRank");
    CommDestroy = (int (*)(void*))dlsym(lib, "ncclCommDestroy");
    GroupStart = (int (*)())dlsym(lib, "ncclGroupStart");
    GroupEnd = (int (*)())dlsym(lib, "ncclGroupEnd");
    Send = (int (*)(const void*, size_t, int, int, void*, hipStream_t))dlsym(lib, "ncclSend");
    Recv = (int (*)(void*, size_t, int, int, void*, hipStream_t))dlsym(lib, "ncclRecv");
    GetErrorString = (const char* (*)(int))dlsym(lib, "ncclGetErrorString");
    return GetUniqueId && CommInitRank && CommDestroy && GroupStart && GroupEnd && Send && Recv;
  }
};
constexpr int kRcclChar = 0;   // ncclInt8 / ncclChar

}  // namespace
}  // namespace sctl_amd

struct sctl_amd_comm {
  int rank = 0, size = 1, device = -1;
  std::vector<int> fd;          // rank 0: fd[r] = socket to rank r; other ranks: fd[0] = socket to rank 0
  int listen_fd = -1;
  sctl_amd::Rccl rccl;
  void* nccl = nullptr;         // ncclComm_t when the data path is RCCL
  ~sctl_amd_comm() {
    if (nccl && rccl.CommDestroy) (void)rccl.CommDestroy(nccl);
    for (int f : fd) if (f >= 0) ::close(f);
    if (listen_fd >= 0) ::close(listen_fd);
  }
};

namespace sctl_amd {

// ---- host-side collectives over the rendezvous sockets (a star through rank 0: meant for counts, ids, test-sized data) ---------
// The largest contribution rank 0 accepts from one peer: the star carries counts, ids and the data of one-GPU rehearsals; a length
// above this is a corrupt or foreign message, not a request to allocate it.
constexpr int64_t kMaxStarBytes = int64_t(1) << 34;
// every rank contributes n bytes; all receives the concatenation in rank order, bytes_of_rank the sizes.  false = a peer went away.
static bool star_allgatherv(sctl_amd_comm* c, const void* send, int64_t n, std::vector<char>& all, std::vector<int64_t>& bytes_of_rank) {
  bytes_of_rank.assign((size_t)c->size, 0);
  if (c->size == 1) {
    bytes_of_rank[0] = n;
    all.assign((const char*)send, (const char*)send + n);
    return true;
  }
  if (c->rank == 0) {
    std::vector<std::vector<char>> part((size_t)c->size);
    part[0].assign((const char*)send, (const char*)send + n);
    bytes_of_rank[0] = n;
    for (int r = 1; r < c->size; r++) {
      int64_t m = 0;
      if (!recv_all(c->fd[(size_t)r], &m, 8) || m < 0 || m > kMaxStarBytes) return false;
      part[(size_t)r].resize((size_t)m);
      if (m && !recv_all(c->fd[(size_t)r], part[(size_t)r].data(), (size_t)m)) return false;
      bytes_of_rank[(size_t)r] = m;
    }
    all.clear();
    for (auto& p : part) all.insert(all.end(), p.begin(), p.end());
    for (int r = 1; r < c->size; r++) {
      if (!send_all(c->fd[(size_t)r], bytes_of_rank.data(), 8 * (size_t)c->size)) return false;
      if (!all.empty() && !send_all(c->fd[(size_t)r], all.data(), all.size())) return false;
    }
    return true;
  }
  if (!send_all(c->fd[0], &n, 8) || (n && !send_all(c->fd[0], send, (size_t)n))) return false;
  if (!recv_all(c->fd[0], bytes_of_rank.data(), 8 * (size_t)c->size)) return false;
  int64_t tot = 0;
  for (int64_t b : bytes_of_rank) tot += b;
  all.resize((size_t)tot);
  return tot == 0 || recv_all(c->fd[0], all.data(), (size_t)tot);
}

// Every rank reports the status of what it did LOCALLY before a collective; all ranks return together: the caller's own failure as it is,
// SCTL_AMD_ERR_PEER when only another rank failed — so no rank enqueues sends or receives towards a rank that has already left the call
// (which would hang the stream instead of failing).
int comm_agree(sctl_amd_comm* c, int local_rc, const char* what) {
  if (!c || c->size == 1) return local_rc;
  const std::string own = local_rc ? std::string(sctl_amd_last_error()) : std::string();
  std::vector<char> all;
  std::vector<int64_t> sizes;
  const int32_t mine = local_rc;
  if (!star_allgatherv(c, &mine, 4, all, sizes)) return set_error(SCTL_AMD_ERR_PEER, std::string(what) + ": rank exchange failed, a peer closed its connection");
  if (local_rc) return set_error(local_rc, own);
  for (int r = 0; r < c->size; r++) {
    int32_t v = 0;
    std::memcpy(&v, all.data() + 4 * (size_t)r, 4);
    if (v) return set_error(SCTL_AMD_ERR_PEER, std::string(what) + ": rank " + std::to_string(r) + " failed with status " + std::to_string(v) + " before the collective; no rank entered it");
  }
  return SCTL_AMD_OK;
}

// All ranks' host arrays, concatenated in rank order, into a device buffer of the calling rank (grown on demand):
// RCCL send/recv between the ranks' device buffers when every rank has its own GPU, the rendezvous sockets otherwise.
// bytes_of_rank receives the sizes.  Enqueued on st; the staging slices must stay untouched until st is synchronised.
int comm_gather_to_device(sctl_amd_comm* c, const void* local, int64_t nbytes, void** dbuf, size_t* dcap, std::vector<int64_t>* bytes_of_rank,
                          char* (*stage_take)(void*, size_t), void* stage, hipStream_t st) {
  std::vector<char> tmp;
  std::vector<int64_t>& sizes = *bytes_of_rank;
  auto grow = [&](size_t bytes) -> hipError_t {
    if (bytes <= *dcap) return hipSuccess;
    if (*dbuf) { hipError_t e = hipFree(*dbuf); *dbuf = nullptr; *dcap = 0; if (e != hipSuccess) return e; }
    hipError_t e = hipMalloc(dbuf, bytes);
    if (e == hipSuccess) *dcap = bytes;
    return e;
  };
  if (!c->nccl) {   // sockets: the concatenation arrives in host memory and is uploaded whole
    if (!star_allgatherv(c, local, nbytes, tmp, sizes)) return set_error(SCTL_AMD_ERR_PEER, "rank exchange failed: a peer closed its connection");
    auto upload_all = [&]() -> int {
      if (tmp.empty()) return SCTL_AMD_OK;
      hipError_t e = grow(tmp.size());
      if (e != hipSuccess) return set_error(SCTL_AMD_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
      char* q = stage_take(stage, tmp.size());
      if (!q) return set_error(SCTL_AMD_ERR_HIP, "cannot allocate pinned staging memory");
      std::memcpy(q, tmp.data(), tmp.size());
      e = hipMemcpyAsync(*dbuf, q, tmp.size(), hipMemcpyHostToDevice, st);
      return e == hipSuccess ? SCTL_AMD_OK : set_error(SCTL_AMD_ERR_HIP, std::string("hipMemcpyAsync: ") + hipGetErrorString(e));
    };
    return comm_agree(c, upload_all(), "host all-gather");   // a rank that could not take the data tells the others before they move on
  }
  // RCCL: sizes over the sockets, payload GPU to GPU over xGMI
  if (!star_allgatherv(c, &nbytes, 8, tmp, sizes)) return set_error(SCTL_AMD_ERR_PEER, "rank exchange failed: a peer closed its connection");
  std::vector<int64_t> n((size_t)c->size), off((size_t)c->size + 1, 0);
  for (int r = 0; r < c->size; r++) { std::memcpy(&n[(size_t)r], tmp.data() + 8 * (size_t)r, 8); off[(size_t)r + 1] = off[(size_t)r] + n[(size_t)r]; }
  sizes = n;
  const int64_t total = off[(size_t)c->size];
  if (total == 0) return SCTL_AMD_OK;
  // local preparation (buffer, staging, own block up); its status is agreed on BEFORE any rank posts a send or a receive
  char* base = nullptr;
  auto prepare = [&]() -> int {
    hipError_t e = grow((size_t)total);
    if (e != hipSuccess) return set_error(SCTL_AMD_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
    base = (char*)*dbuf;
    if (nbytes) {
      char* q = stage_take(stage, (size_t)nbytes);
      if (!q) return set_error(SCTL_AMD_ERR_HIP, "cannot allocate pinned staging memory");
      std::memcpy(q, local, (size_t)nbytes);
      e = hipMemcpyAsync(base + off[(size_t)c->rank], q, (size_t)nbytes, hipMemcpyHostToDevice, st);
      if (e != hipSuccess) return set_error(SCTL_AMD_ERR_HIP, std::string("hipMemcpyAsync: ") + hipGetErrorString(e));
    }
    return SCTL_AMD_OK;
  };
  int rc = comm_agree(c, prepare(), "device all-gather");
  if (rc) return rc;
  rc = c->rccl.GroupStart();
  for (int r = 0; r < c->size && rc == 0; r++) {
    if (r == c->rank) continue;
    if (nbytes) rc = c->rccl.Send(base + off[(size_t)c->rank], (size_t)nbytes, kRcclChar, r, c->nccl, st);
    if (rc == 0 && n[(size_t)r]) rc = c->rccl.Recv(base + off[(size_t)r], (size_t)n[(size_t)r], kRcclChar, r, c->nccl, st);
  }
  const int rc2 = c->rccl.GroupEnd();
  if (rc == 0) rc = rc2;
  if (rc != 0) return set_error(SCTL_AMD_ERR_HIP, std::string("RCCL all-gather failed: ") + (c->rccl.GetErrorString ? c->rccl.GetErrorString(rc) : "?"));
  return SCTL_AMD_OK;
}

bool comm_uses_rccl(const sctl_amd_comm* c) { return c && c->nccl != nullptr; }

}  // namespace sctl_amd

using namespace sctl_amd;

extern "C" {

int sctl_amd_comm_create(int rank, int size, const char* master_addr, int master_port, int device, int flags, sctl_amd_comm** out) {
  if (!out || size < 1 || rank < 0 || rank >= size) return set_error(SCTL_AMD_ERR_BAD_ARGUMENT, "bad rank / size / handle pointer");
  *out = nullptr;
  std::unique_ptr<sctl_amd_comm> c(new sctl_amd_comm);
  c->rank = rank; c->size = size; c->device = device;
  if (size == 1) {
    if (flags & SCTL_AMD_COMM_FORCE_RCCL) {   // a one-rank RCCL communicator: lets a one-GPU box exercise the RCCL binding (sctl_amd_comm_selftest)
      if (device < 0 || device >= device_count_quiet()) return set_error(SCTL_AMD_ERR_NO_DEVICE, "no HIP device for the RCCL communicator");
      if (!c->rccl.load()) return set_error(SCTL_AMD_ERR_HIP, "librccl could not be loaded");
      RcclUniqueId id{};
      DeviceScope scope(device);
      if (scope.err != hipSuccess) return set_error(SCTL_AMD_ERR_HIP, "hipSetDevice failed for the rank's device");
      int rc = c->rccl.GetUniqueId(&id);
      if (rc == 0) rc = c->rccl.CommInitRank(&c->nccl, 1, id, 0);
      if (rc != 0) { c->nccl = nullptr; return set_error(SCTL_AMD_ERR_HIP, std::string("RCCL initialisation failed: ") + (c->rccl.GetErrorString ? c->rccl.GetErrorString(rc) : "?")); }
    }
    *out = c.release();
    return SCTL_AMD_OK;
  }
  if (!master_addr || !master_addr[0]) master_addr = "127.0.0.1";
  if (master_port <= 0 || master_port > 65535) return set_error(SCTL_AMD_ERR_BAD_ARGUMENT, "bad rendezvous port");
  // MASTER_ADDR as launchers export it: dotted IPv4 (fast path) or a host name (`localhost`, a node name from scontrol) -> getaddrinfo
  sockaddr_in sa{};
  sa.sin_family = AF_INET;
  sa.sin_port = htons((uint16_t)master_port);
  if (inet_pton(AF_INET, master_addr, &sa.sin_addr) != 1) {
    addrinfo hints{};
    hints.ai_family = AF_INET;
    hints.ai_socktype = SOCK_STREAM;
    addrinfo* res = nullptr;
    const int gai = getaddrinfo(master_addr, nullptr, &hints, &res);
    if (gai != 0 || !res) return set_error(SCTL_AMD_ERR_BAD_ARGUMENT, std::string("cannot resolve the rendezvous address '") + master_addr + "': " + (gai ? gai_strerror(gai) : "no IPv4 address"));
    sa.sin_addr = ((const sockaddr_in*)res->ai_addr)->sin_addr;
    freeaddrinfo(res);
  }
  // The hello every rank sends: a magic word, its rank, the world size it believes in and a job token (SCTL_AMD_JOB_TOKEN, else the
  // launcher's run / job id; 0 when there is none) — rank 0 drops connections that do not belong to this job and keeps listening.
  // The token is a guard against two jobs MIXING UP their rendezvous ports, not authentication: a job id can be guessed; keep the port off untrusted networks.
  struct Hello { uint32_t magic; int32_t rank; int32_t size; uint32_t pad; uint64_t token; };
  constexpr uint32_t kMagic = 0x5343544cu;   // "SCTL"
  uint64_t token = 0;
  for (const char* v : {"SCTL_AMD_JOB_TOKEN", "TORCHELASTIC_RUN_ID", "SLURM_JOB_ID", "PBS_JOBID", "OMPI_MCA_ess_base_jobid"})
    if (const char* t = std::getenv(v)) {
      token = 1469598103934665603ull;        // FNV-1a of the text
      for (const char* q = t; *q; q++) token = (token ^ (unsigned char)*q) * 1099511628211ull;
      break;
    }
  const int one = 1;
  if (rank == 0) {
    c->listen_fd = ::socket(AF_INET, SOCK_STREAM, 0);
    if (c->listen_fd < 0) return set_error(SCTL_AMD_ERR_HIP, "socket() failed");
    (void)setsockopt(c->listen_fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
    if (::bind(c->listen_fd, (sockaddr*)&sa, sizeof sa) != 0) {   // the name may resolve to an address that is not an interface of this host (NAT): any interface
      sockaddr_in any = sa;
      any.sin_addr.s_addr = htonl(INADDR_ANY);
      if (::bind(c->listen_fd, (sockaddr*)&any, sizeof any) != 0) return set_error(SCTL_AMD_ERR_HIP, std::string("cannot bind ") + master_addr + ":" + std::to_string(master_port));
    }
    if (::listen(c->listen_fd, size) != 0) return set_error(SCTL_AMD_ERR_HIP, std::string("cannot listen on ") + master_addr + ":" + std::to_string(master_port));
    c->fd.assign((size_t)size, -1);
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(300);
    for (int joined = 1; joined < size;) {
      const long long left_ms = std::chrono::duration_cast<std::chrono::milliseconds>(deadline - std::chrono::steady_clock::now()).count();
      pollfd pf{c->listen_fd, POLLIN, 0};
      if (left_ms <= 0 || ::poll(&pf, 1, (int)left_ms) <= 0)
        return set_error(SCTL_AMD_ERR_PEER, "rendezvous: " + std::to_string(size - joined) + " rank(s) did not connect within 5 minutes");
      const int f = ::accept(c->listen_fd, nullptr, nullptr);
      if (f < 0) {                             // transient (the peer went away between poll and accept): try again; anything else (EMFILE, ...) will not
        if (errno == EINTR || errno == EAGAIN || errno == EWOULDBLOCK || errno == ECONNABORTED) continue;   // mend itself by spinning until the deadline
        return set_error(SCTL_AMD_ERR_HIP, std::string("rendezvous: accept() failed: ") + std::strerror(errno));
      }
      timeval tv{2, 0};                        // a connection that says nothing within 2 s is not one of ours (a member sends its hello at once)
      (void)setsockopt(f, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
      Hello h{};
      if (!recv_all(f, &h, sizeof h) || h.magic != kMagic || h.size != size || h.token != token || h.rank < 1 || h.rank >= size || c->fd[(size_t)h.rank] >= 0) {
        ::close(f);                            // stray, foreign or duplicate: drop it and keep accepting until the deadline
        continue;
      }
      tv = timeval{0, 0};                      // members may legitimately stay silent for long (they compute between collectives)
      (void)setsockopt(f, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
      (void)setsockopt(f, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
      c->fd[(size_t)h.rank] = f;
      joined++;
    }
  } else {
    int f = -1;
    for (int attempt = 0; attempt < 1200; attempt++) {   // rank 0 may start later: retry for two minutes
      f = ::socket(AF_INET, SOCK_STREAM, 0);
      if (f >= 0 && ::connect(f, (sockaddr*)&sa, sizeof sa) == 0) break;
      if (f >= 0) ::close(f);
      f = -1;
      std::this_thread::sleep_for(std::chrono::milliseconds(100));
    }
    const Hello h{kMagic, rank, size, 0, token};
    if (f < 0 || !send_all(f, &h, sizeof h)) { if (f >= 0) ::close(f); return set_error(SCTL_AMD_ERR_PEER, std::string("cannot reach rank 0 at ") + master_addr + ":" + std::to_string(master_port)); }
    (void)setsockopt(f, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
    c->fd.assign(1, f);
  }
  // Which data path?  RCCL needs one GPU per rank: every rank reports "<host name>|<PCI bus id of its device>" (the same bus id on two
  // nodes is two GPUs); all distinct -> RCCL.
  char bus[160] = {0};
  if (device >= 0 && device < device_count_quiet() && !(flags & SCTL_AMD_COMM_SOCKETS_ONLY)) {
    char pci[64] = {0}, host[64] = {0};
    if (hipDeviceGetPCIBusId(pci, (int)sizeof pci - 1, device) != hipSuccess) { (void)hipGetLastError(); pci[0] = 0; }
    if (gethostname(host, sizeof host - 1) != 0) host[0] = 0;
    if (pci[0]) std::snprintf(bus, sizeof bus, "%s|%s", host, pci);
  }
  std::vector<char> all;
  std::vector<int64_t> sizes;
  if (!star_allgatherv(c.get(), bus, (int64_t)sizeof bus, all, sizes)) return set_error(SCTL_AMD_ERR_PEER, "rendezvous: exchange failed");
  bool distinct = true;
  for (int a = 0; a < size && distinct; a++) {
    const char* ba = all.data() + (size_t)a * sizeof bus;
    if (!ba[0]) distinct = false;
    for (int b = 0; b < a && distinct; b++) distinct = std::strcmp(ba, all.data() + (size_t)b * sizeof bus) != 0;
  }
  char ok = (distinct && c->rccl.load()) ? 1 : 0;
  if (!star_allgatherv(c.get(), &ok, 1, all, sizes)) return set_error(SCTL_AMD_ERR_PEER, "rendezvous: exchange failed");
  for (char v : all) ok = ok && v;
  if (ok) {
    RcclUniqueId id{};
    if (rank == 0 && c->rccl.GetUniqueId(&id) != 0) std::memset(&id, 0, sizeof id);
    // rank 0's id to everybody (all ranks contribute 128 bytes; block 0 is the one that counts)
    if (!star_allgatherv(c.get(), &id, (int64_t)sizeof id, all, sizes)) return set_error(SCTL_AMD_ERR_PEER, "rendezvous: exchange failed");
    std::memcpy(&id, all.data(), sizeof id);
    DeviceScope scope(device);
    if (scope.err != hipSuccess) return set_error(SCTL_AMD_ERR_HIP, "hipSetDevice failed for the rank's device");
    const int rc = c->rccl.CommInitRank(&c->nccl, size, id, rank);
    if (rc != 0) { c->nccl = nullptr; return set_error(SCTL_AMD_ERR_HIP, std::string("ncclCommInitRank failed: ") + (c->rccl.GetErrorString ? c->rccl.GetErrorString(rc) : "?")); }
  }
  *out = c.release();
  return SCTL_AMD_OK;
}

void sctl_amd_comm_destroy(sctl_amd_comm* c) { delete c; }

int sctl_amd_comm_info(const sctl_amd_comm* c, int* rank, int* size, int* device, int* transport) {
  if (!c) return set_error(SCTL_AMD_ERR_BAD_ARGUMENT, "null communicator");
  if (rank) *rank = c->rank;
  if (size) *size = c->size;
  if (device) *device = c->device;
  if (transport) *transport = c->nccl ? SCTL_AMD_COMM_RCCL : SCTL_AMD_COMM_SOCKETS;
  return SCTL_AMD_OK;
}

int sctl_amd_comm_allgatherv_host(sctl_amd_comm* c, const void* send, int64_t send_bytes, void* recv, int64_t recv_capacity, int64_t* bytes_of_rank) {
  if (!c || send_bytes < 0 || (send_bytes > 0 && !send)) return set_error(SCTL_AMD_ERR_BAD_ARGUMENT, "null communicator or bad send buffer");
  std::vector<char> all;
  std::vector<int64_t> sizes;
  if (!star_allgatherv(c, send, send_bytes, all, sizes)) return set_error(SCTL_AMD_ERR_PEER, "rank exchange failed: a peer closed its connection");
  if (bytes_of_rank) std::memcpy(bytes_of_rank, sizes.data(), 8 * (size_t)c->size);
  if ((int64_t)all.size() > recv_capacity) return set_error(SCTL_AMD_ERR_BAD_ARGUMENT, "receive buffer too small: " + std::to_string(all.size()) + " bytes arrive");
  if (!all.empty()) std::memcpy(recv, all.data(), all.size());
  return SCTL_AMD_OK;
}

// RCCL data path check on the communicator's own device: every rank sends `bytes` bytes to its right neighbour and receives from its
// left one (itself, with one rank) with the grouped ncclSend / ncclRecv the gathers use, and compares what arrived with what the
// sender holds (a pattern of rank and index).  SCTL_AMD_ERR_BAD_ARGUMENT when the communicator's data path is not RCCL.
int sctl_amd_comm_selftest(sctl_amd_comm* c, int64_t bytes) {
  if (!c || bytes <= 0) return set_error(SCTL_AMD_ERR_BAD_ARGUMENT, "null communicator or empty message");
  if (!c->nccl) return set_error(SCTL_AMD_ERR_BAD_ARGUMENT, "this communicator's data path is not RCCL");
  DeviceScope scope(c->device);
  if (scope.err != hipSuccess) return set_error(SCTL_AMD_ERR_HIP, "hipSetDevice failed");
  const int right = (c->rank + 1) % c->size, left = (c->rank + c->size - 1) % c->size;
  std::vector<unsigned char> h((size_t)bytes), back((size_t)bytes);
  for (int64_t i = 0; i < bytes; i++) h[(size_t)i] = (unsigned char)(c->rank * 31 + i * 7);
  void *a = nullptr, *b = nullptr;
  hipStream_t st = nullptr;
  auto done = [&](int rc) { if (a) (void)hipFree(a); if (b) (void)hipFree(b); if (st) (void)hipStreamDestroy(st); return rc; };
  if (hipMalloc(&a, (size_t)bytes) != hipSuccess || hipMalloc(&b, (size_t)bytes) != hipSuccess || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess)
    return done(set_error(SCTL_AMD_ERR_HIP, "allocation failed"));
  if (hipMemcpy(a, h.data(), (size_t)bytes, hipMemcpyHostToDevice) != hipSuccess || hipMemset(b, 0, (size_t)bytes) != hipSuccess) return done(set_error(SCTL_AMD_ERR_HIP, "copy failed"));
  int rc = c->rccl.GroupStart();
  if (rc == 0) rc = c->rccl.Send(a, (size_t)bytes, kRcclChar, right, c->nccl, st);
  if (rc == 0) rc = c->rccl.Recv(b, (size_t)bytes, kRcclChar, left, c->nccl, st);
  const int rc2 = c->rccl.GroupEnd();
  if (rc == 0) rc = rc2;
  if (rc != 0) return done(set_error(SCTL_AMD_ERR_HIP, std::string("RCCL send/recv failed: ") + (c->rccl.GetErrorString ? c->rccl.GetErrorString(rc) : "?")));
  if (hipStreamSynchronize(st) != hipSuccess || hipMemcpy(back.data(), b, (size_t)bytes, hipMemcpyDeviceToHost) != hipSuccess) return done(set_error(SCTL_AMD_ERR_HIP, "synchronisation failed"));
  for (int64_t i = 0; i < bytes; i++)
    if (back[(size_t)i] != (unsigned char)(left * 31 + i * 7)) return done(set_error(SCTL_AMD_ERR_HIP, "RCCL delivered wrong bytes at offset " + std::to_string(i)));
  return done(SCTL_AMD_OK);
}

int sctl_amd_comm_barrier(sctl_amd_comm* c) {
  if (!c) return set_error(SCTL_AMD_ERR_BAD_ARGUMENT, "null communicator");
  std::vector<char> all;
  std::vector<int64_t> sizes;
  const char z = 0;
  return star_allgatherv(c, &z, 1, all, sizes) ? SCTL_AMD_OK : set_error(SCTL_AMD_ERR_PEER, "rank exchange failed: a peer closed its connection");
}

}  // extern "C"
