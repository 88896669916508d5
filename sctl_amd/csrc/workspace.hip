#include "workspace.hpp"

#include <atomic>
#include <map>
#include <mutex>
#include <utility>

namespace sctl_amd {
namespace {
struct Block { void* p = nullptr; size_t cap = 0; };
std::mutex g_mu;
std::atomic<bool> g_poison{false};
std::map<std::pair<int, hipStream_t>, Block>& blocks() {   // leaked on purpose: no HIP calls from static destructors
  static auto* m = new std::map<std::pair<int, hipStream_t>, Block>;
  return *m;
}
}  // namespace

hipError_t workspace_acquire(hipStream_t st, size_t bytes, void** base) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lock(g_mu);
  Block& b = blocks()[std::make_pair(dev, st)];
  if (bytes > b.cap) {
    if (b.p) { e = hipFree(b.p); b.p = nullptr; b.cap = 0; if (e != hipSuccess) return e; }   // hipFree waits for pending work
    const size_t want = bytes + bytes / 8 + 4096;                                               // slack: sizes creep between calls
    e = hipMalloc(&b.p, want);
    if (e != hipSuccess) { b.p = nullptr; return e; }
    b.cap = want;
  }
  *base = b.p;
  if (g_poison.load() && bytes) return hipMemsetAsync(b.p, 0xFF, bytes, st);
  return hipSuccess;
}

void workspace_poison(bool on) { g_poison.store(on); }

void workspace_forget(hipStream_t st) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return; }
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = blocks().find(std::make_pair(dev, st));
  if (it == blocks().end()) return;
  if (it->second.p) (void)hipFree(it->second.p);
  blocks().erase(it);
}

void workspace_release_all() {
  int cur = 0;
  const bool have = hipGetDevice(&cur) == hipSuccess;
  std::lock_guard<std::mutex> lock(g_mu);
  for (auto& kv : blocks()) {
    if (!kv.second.p) continue;
    if (hipSetDevice(kv.first.first) == hipSuccess) (void)hipFree(kv.second.p);
  }
  blocks().clear();
  if (have) (void)hipSetDevice(cur);
  (void)hipGetLastError();
}

}  // namespace sctl_amd
