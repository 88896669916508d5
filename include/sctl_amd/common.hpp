// Host-side (header-only) surface of the MI355X direct-summation path: scalar types, error convention, device set.
//
// Mirrors the pieces of the reference's include/sctl/common.hpp the hot path touches:
//   Integer / Long                     common.hpp:44-45
//   SCTL_ASSERT -> fprintf + abort()   common.hpp:59-70   (no return codes, no exceptions at this level)
//   Iterator<T> = T*                   common.hpp:75-83   (the non-SCTL_MEMDEBUG definition)
// Everything lives in namespace sctl_amd; define SCTL_AMD_AS_SCTL before including to also get `namespace sctl`.
#ifndef SCTL_AMD_COMMON_HPP_
#define SCTL_AMD_COMMON_HPP_

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../sctl_amd.h"

namespace sctl_amd {

typedef long Integer;   // bounded numbers < 32k
typedef int64_t Long;   // problem size

template <typename ValueType> using Iterator = ValueType*;
template <typename ValueType> using ConstIterator = const ValueType*;
template <typename ValueType> inline Iterator<ValueType> NullIterator() { return nullptr; }
template <typename ValueType> inline Iterator<ValueType> Ptr2Itr(void* ptr, Long) { return (Iterator<ValueType>)ptr; }
template <typename ValueType> inline ConstIterator<ValueType> Ptr2ConstItr(const void* ptr, Long) { return (ConstIterator<ValueType>)ptr; }

}  // namespace sctl_amd

// Error convention of this surface, the reference's (no return codes, no exceptions: report on stderr, then abort): one
// noreturn function does the reporting; the macros only capture the call site.
namespace sctl_amd {
namespace detail {
[[noreturn]] inline void fatal(const char* file, int line, const char* func, const char* what, const char* detail_text) {
  std::fprintf(stderr, "\nsctl_amd fatal error: %s%s%s\n    at %s:%d (%s)\n", what, detail_text ? ": " : "", detail_text ? detail_text : "", file, line, func);
  std::fflush(stderr);
  std::abort();
}
}  // namespace detail
}  // namespace sctl_amd

#define SCTL_AMD_ERROR(msg) ::sctl_amd::detail::fatal(__FILE__, __LINE__, __func__, (msg), nullptr)
#define SCTL_AMD_ASSERT(cond) ((cond) ? (void)0 : ::sctl_amd::detail::fatal(__FILE__, __LINE__, __func__, "requirement not met", #cond))
#define SCTL_AMD_ASSERT_MSG(cond, msg) ((cond) ? (void)0 : ::sctl_amd::detail::fatal(__FILE__, __LINE__, __func__, (msg), #cond))

namespace sctl_amd {

// Status of a C-ABI call -> the reference's convention: print and abort (common.hpp:48-70).
inline void CheckStatus(int rc, const char* what) {
  if (rc != SCTL_AMD_OK) {
    char text[64];
    std::snprintf(text, sizeof text, "%s failed with status %d", what, rc);
    detail::fatal(__FILE__, __LINE__, __func__, text, sctl_amd_last_error());
  }
}

// The GPUs of this node that evaluations are spread over (targets block-partitioned).
// Default: device 0, or the comma-separated list in $SCTL_AMD_DEVICES (e.g. "0,1,2,3,4,5,6,7").
class DeviceSet {
 public:
  static std::vector<int>& Get() {
    static std::vector<int> devs = FromEnv();
    return devs;
  }
  static void Set(const std::vector<int>& devs) {
    SCTL_AMD_ASSERT(!devs.empty());
    Get() = devs;
  }
  static void UseAll() {
    const int n = sctl_amd_device_count();
    SCTL_AMD_ASSERT_MSG(n > 0, "no HIP device visible: sctl_amd has no CPU fallback");
    std::vector<int> d(n);
    for (int i = 0; i < n; i++) d[i] = i;
    Get() = d;
  }

 private:
  static std::vector<int> FromEnv() {
    std::vector<int> d;
    if (const char* e = std::getenv("SCTL_AMD_DEVICES")) {
      const char* p = e;
      while (*p) {
        char* end = nullptr;
        const long v = std::strtol(p, &end, 10);
        if (end == p) break;
        d.push_back((int)v);
        p = (*end == ',') ? end + 1 : end;
      }
    }
    if (d.empty()) d.push_back(0);
    return d;
  }
};

template <class Real> struct RealTag;
template <> struct RealTag<double> { static constexpr int value = SCTL_AMD_F64; };
template <> struct RealTag<float> { static constexpr int value = SCTL_AMD_F32; };

}  // namespace sctl_amd

#ifdef SCTL_AMD_AS_SCTL
namespace sctl = sctl_amd;
#endif

#endif  // SCTL_AMD_COMMON_HPP_
