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

#define SCTL_AMD_ERROR(msg)                                     \
  do {                                                          \
    fprintf(stderr, "\n\033[1;31mError:\033[0m %s\n", (msg));   \
    abort();                                                    \
  } while (0)

#define SCTL_AMD_ASSERT(cond)                                                                                    \
  do {                                                                                                           \
    if (!(cond)) {                                                                                               \
      fprintf(stderr, "\n%s:%d: %s: Assertion `%s' failed.\n", __FILE__, __LINE__, __PRETTY_FUNCTION__, #cond);  \
      abort();                                                                                                   \
    }                                                                                                            \
  } while (0)

#define SCTL_AMD_ASSERT_MSG(cond, msg) \
  do {                                 \
    if (!(cond)) SCTL_AMD_ERROR(msg);  \
  } while (0)

namespace sctl_amd {

// Status of a C-ABI call -> the reference's convention: print and abort (common.hpp:48-70).
inline void CheckStatus(int rc, const char* what) {
  if (rc != SCTL_AMD_OK) {
    fprintf(stderr, "\n\033[1;31mError:\033[0m %s failed (status %d): %s\n", what, rc, sctl_amd_last_error());
    abort();
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
