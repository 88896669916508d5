// Minimal Vector<T> with the observable API subset of the reference's include/sctl/vector.hpp:36-184 that the
// direct-summation path and its callers use: owning storage or a non-owning view over caller memory
// (vector.hpp:45, vector.txx:26-41), Dim/begin/end/operator[], ReInit, SetZero, PushBack, element-wise
// arithmetic, and the raw binary Write/Read format (vector.txx:107-140: two uint64 {dim, 1} then the data).
// Written from the interface documentation, not from the reference's implementation (no memory manager,
// 64-byte aligned allocations through posix_memalign; common.hpp:36-38 asks for 64-byte alignment).
#ifndef SCTL_AMD_VECTOR_HPP_
#define SCTL_AMD_VECTOR_HPP_

#include <cstring>
#include <initializer_list>
#include <new>
#include <type_traits>

#include "common.hpp"

namespace sctl_amd {

template <class ValueType> class Vector {
 public:
  typedef ValueType value_type;
  typedef Long size_type;

  Vector() : dim_(0), cap_(0), data_(nullptr), own_(true) {}
  explicit Vector(Long dim, Iterator<ValueType> data = nullptr, bool own_data = true) : dim_(0), cap_(0), data_(nullptr), own_(true) {
    Init(dim, data, own_data);
  }
  Vector(const Vector& v) : dim_(0), cap_(0), data_(nullptr), own_(true) {
    Init(v.dim_, nullptr, true);
    Copy(v.data_, v.dim_);
  }
  explicit Vector(const std::vector<ValueType>& v) : dim_(0), cap_(0), data_(nullptr), own_(true) {
    Init((Long)v.size(), nullptr, true);
    Copy(v.data(), (Long)v.size());
  }
  Vector(std::initializer_list<ValueType> v) : dim_(0), cap_(0), data_(nullptr), own_(true) {
    Init((Long)v.size(), nullptr, true);
    Long i = 0;
    for (const auto& x : v) data_[i++] = x;
  }
  ~Vector() { Release(); }

  void Swap(Vector& o) {
    std::swap(dim_, o.dim_); std::swap(cap_, o.cap_); std::swap(data_, o.data_); std::swap(own_, o.own_);
  }

  // own_data = true with a pointer copies from it; own_data = false makes a view (vector.txx:26-41)
  void ReInit(Long dim, Iterator<ValueType> data = nullptr, bool own_data = true) {
    if (own_data && own_ && dim <= cap_) {
      dim_ = dim;
      if (data) Copy(data, dim);
      return;
    }
    Release();
    Init(dim, data, own_data);
  }

  Long Dim() const { return dim_; }
  Iterator<ValueType> begin() { return data_; }
  ConstIterator<ValueType> begin() const { return data_; }
  Iterator<ValueType> end() { return data_ + dim_; }
  ConstIterator<ValueType> end() const { return data_ + dim_; }
  ValueType& operator[](Long j) { return data_[j]; }
  const ValueType& operator[](Long j) const { return data_[j]; }

  void SetZero() {
    for (Long i = 0; i < dim_; i++) data_[i] = ValueType();
  }
  void PushBack(const ValueType& x) {
    if (dim_ == cap_ || !own_) {
      Vector grown(cap_ > 0 ? std::max<Long>(2 * cap_, dim_ + 1) : 8);
      for (Long i = 0; i < dim_; i++) grown.data_[i] = data_[i];
      grown.dim_ = dim_;
      Swap(grown);
    }
    data_[dim_++] = x;
  }

  Vector& operator=(const Vector& v) {
    if (this != &v) {
      if (dim_ != v.dim_) ReInit(v.dim_);
      Copy(v.data_, v.dim_);
    }
    return *this;
  }
  Vector& operator=(const std::vector<ValueType>& v) {
    if (dim_ != (Long)v.size()) ReInit((Long)v.size());
    Copy(v.data(), (Long)v.size());
    return *this;
  }
  Vector& operator=(ValueType s) { for (Long i = 0; i < dim_; i++) data_[i] = s; return *this; }
  Vector& operator+=(const Vector& v) { SCTL_AMD_ASSERT(v.dim_ == dim_); for (Long i = 0; i < dim_; i++) data_[i] += v.data_[i]; return *this; }
  Vector& operator-=(const Vector& v) { SCTL_AMD_ASSERT(v.dim_ == dim_); for (Long i = 0; i < dim_; i++) data_[i] -= v.data_[i]; return *this; }
  Vector& operator*=(ValueType s) { for (Long i = 0; i < dim_; i++) data_[i] *= s; return *this; }
  Vector operator+(const Vector& v) const { Vector r(*this); r += v; return r; }
  Vector operator-(const Vector& v) const { Vector r(*this); r -= v; return r; }
  Vector operator*(ValueType s) const { Vector r(*this); r *= s; return r; }

  // vector.txx:107-140: uint64 dim, uint64 1, then dim values
  void Write(const char* fname) const {
    FILE* f = fopen(fname, "wb");
    SCTL_AMD_ASSERT_MSG(f != nullptr, "Unable to open file for writing");
    const uint64_t hdr[2] = {(uint64_t)dim_, 1};
    fwrite(hdr, sizeof(uint64_t), 2, f);
    if (dim_) fwrite(data_, sizeof(ValueType), (size_t)dim_, f);
    fclose(f);
  }
  void Read(const char* fname) {
    FILE* f = fopen(fname, "rb");
    SCTL_AMD_ASSERT_MSG(f != nullptr, "Unable to open file for reading");
    uint64_t hdr[2] = {0, 0};
    SCTL_AMD_ASSERT(fread(hdr, sizeof(uint64_t), 2, f) == 2);
    const Long n = (Long)(hdr[0] * hdr[1]);
    if (n != dim_) ReInit(n);
    if (n) SCTL_AMD_ASSERT(fread(data_, sizeof(ValueType), (size_t)n, f) == (size_t)n);
    fclose(f);
  }

 private:
  void Init(Long dim, Iterator<ValueType> data, bool own_data) {
    dim_ = dim; own_ = own_data;
    if (own_data) {
      cap_ = dim;
      data_ = nullptr;
      if (dim > 0) {
        void* p = nullptr;
        if (posix_memalign(&p, 64, sizeof(ValueType) * (size_t)dim) != 0) SCTL_AMD_ERROR("memory allocation failed");
        data_ = (ValueType*)p;
        if (!std::is_trivially_default_constructible<ValueType>::value)
          for (Long i = 0; i < dim; i++) new (data_ + i) ValueType();
        if (data) Copy(data, dim);
      }
    } else {
      cap_ = dim;
      data_ = data;
    }
  }
  void Release() {
    if (own_ && data_) {
      if (!std::is_trivially_destructible<ValueType>::value)
        for (Long i = 0; i < cap_; i++) data_[i].~ValueType();
      free(data_);
    }
    data_ = nullptr; dim_ = 0; cap_ = 0; own_ = true;
  }
  void Copy(const ValueType* src, Long n) {
    if (std::is_trivially_copyable<ValueType>::value) {
      if (n) std::memcpy((void*)data_, (const void*)src, sizeof(ValueType) * (size_t)n);
    } else {
      for (Long i = 0; i < n; i++) data_[i] = src[i];
    }
  }

  Long dim_, cap_;
  ValueType* data_;
  bool own_;
};

}  // namespace sctl_amd
#endif  // SCTL_AMD_VECTOR_HPP_
