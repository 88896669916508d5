// Minimal row-major Matrix<T> with the API subset of the reference's include/sctl/matrix.hpp:35-388 that
// GenericKernel::KernelMatrix and its callers use: owning storage or a view (matrix.hpp ctor with own_data),
// Dim(i), ReInit, SetZero, operator[] returning a row, begin().  BLAS/LAPACK operations of the reference's Matrix
// (GEMM, pinv, SVD) are outside the direct-summation path (SURVEY.md §2 "Containers") and are not provided.
#ifndef SCTL_AMD_MATRIX_HPP_
#define SCTL_AMD_MATRIX_HPP_

#include "vector.hpp"

namespace sctl_amd {

template <class ValueType> class Matrix {
 public:
  Matrix() : d0_(0), d1_(0) {}
  Matrix(Long dim0, Long dim1, Iterator<ValueType> data = nullptr, bool own_data = true) : d0_(dim0), d1_(dim1), v_(dim0 * dim1, data, own_data) {}
  Matrix(const Matrix& m) : d0_(m.d0_), d1_(m.d1_), v_(m.v_) {}
  Matrix& operator=(const Matrix& m) {
    if (this != &m) { d0_ = m.d0_; d1_ = m.d1_; v_ = m.v_; }
    return *this;
  }
  void ReInit(Long dim0, Long dim1, Iterator<ValueType> data = nullptr, bool own_data = true) {
    d0_ = dim0; d1_ = dim1;
    v_.ReInit(dim0 * dim1, data, own_data);
  }
  Long Dim(Long i) const { return i == 0 ? d0_ : d1_; }
  void SetZero() { v_.SetZero(); }
  Iterator<ValueType> begin() { return v_.begin(); }
  ConstIterator<ValueType> begin() const { return v_.begin(); }
  Iterator<ValueType> operator[](Long i) { return v_.begin() + i * d1_; }
  ConstIterator<ValueType> operator[](Long i) const { return v_.begin() + i * d1_; }
  ValueType& operator()(Long i, Long j) { return v_[i * d1_ + j]; }
  const ValueType& operator()(Long i, Long j) const { return v_[i * d1_ + j]; }

 private:
  Long d0_, d1_;
  Vector<ValueType> v_;
};

}  // namespace sctl_amd
#endif  // SCTL_AMD_MATRIX_HPP_
