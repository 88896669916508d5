// BoundaryIntegralOp<Real,Kernel>: the FAR-FIELD leg of the reference's boundary-integral operator
// (include/sctl/boundary_integral.hpp:223-410), which is the caller of the direct-summation hot path
// (SURVEY.md §8 a14): U = K_far[F] evaluated as
//     F_far = density at the far-field quadrature nodes x quadrature weights      boundary_integral.txx:1021-1053
//     fmm.SetSrcDensity("Src", F_far); fmm.Eval(U, "Trg")                          boundary_integral.txx:1054-1073
//     optional dot product of the K1/3 x 3 output with the target normals          boundary_integral.txx:1060-1071
// with the wiring of the constructor (:500-509), SetupBasic (:690-766) and SetupFar (:744-782).
//
// Same names and argument meaning as the reference for everything on that leg: ElementListBase (Size, GetNodeCoord,
// GetFarFieldNodes, GetFarFieldDensity, MatrixFree), BoundaryIntegralOp (SetAccuracy, AddElemList, GetElemList,
// DeleteElemList, SetTargetCoord, SetTargetNormal, Dim, Setup, ClearSetup, ComputeFarField, ComputePotential).
//
// OUT OF SCOPE here: the near/self corrections — SetupSelf, SetupNear (Morton-sorted near
// lists, KernelMatrix subtraction, boundary_integral.txx:46-468, 786-1009) and ComputeNearInterac (:1079-1142).
// ComputeFarField is therefore public here (it is private in the reference), and ComputePotential is provided only
// for operators whose element lists declare no near zone (every far-field distance is 0), where the reference's
// ComputePotential reduces to ComputeFarField; otherwise it aborts with a message instead of silently returning an
// uncorrected potential.
#ifndef SCTL_AMD_BOUNDARY_INTEGRAL_HPP_
#define SCTL_AMD_BOUNDARY_INTEGRAL_HPP_

#include <cmath>
#include <map>
#include <string>
#include <typeinfo>

#include "fmm-wrapper.hpp"

namespace sctl_amd {

// The part of the reference's ElementListBase (boundary_integral.hpp:64-213) the far field needs.
template <class Real> class ElementListBase {
 public:
  virtual ~ElementListBase() {}
  virtual Long Size() const = 0;   // number of elements
  // surface discretisation nodes (X, normals Xn) and the node count of each element
  virtual void GetNodeCoord(Vector<Real>* X, Vector<Real>* Xn, Vector<Long>* element_wise_node_cnt) const = 0;
  // far-field quadrature: nodes, normals, weights, distance beyond which it is accurate to `tol`, node count per element
  virtual void GetFarFieldNodes(Vector<Real>& X, Vector<Real>& Xn, Vector<Real>& wts, Vector<Real>& dist_far,
                                Vector<Long>& element_wise_node_cnt, const Real tol) const = 0;
  // density at the far-field nodes from the density at the surface nodes; leaving Fout empty means "same nodes"
  virtual void GetFarFieldDensity(Vector<Real>& Fout, const Vector<Real>& Fin) const {
    if (Fout.Dim() != 0) Fout.ReInit(0);
  }
  virtual bool MatrixFree() const { return false; }
};

template <class Real, class Kernel> class BoundaryIntegralOp {
  static constexpr Integer KDIM0 = Kernel::SrcDim();
  static constexpr Integer KDIM1 = Kernel::TrgDim();
  static constexpr Integer COORD_DIM = Kernel::CoordDim();

 public:
  BoundaryIntegralOp() = delete;
  BoundaryIntegralOp(const BoundaryIntegralOp&) = delete;
  BoundaryIntegralOp& operator=(const BoundaryIntegralOp&) = delete;

  // boundary_integral.txx:500-509
  explicit BoundaryIntegralOp(const Kernel& ker, bool trg_normal_dot_prod = false, const Comm& comm = Comm::Self())
      : tol_(1e-10), ker_(ker), trg_normal_dot_prod_(trg_normal_dot_prod), comm_(comm), fmm(comm) {
    SCTL_AMD_ASSERT(!trg_normal_dot_prod_ || (KDIM1 % COORD_DIM == 0));
    ClearSetup();
    fmm.SetKernels(ker, ker, ker);
    fmm.AddSrc("Src", ker, ker);
    fmm.AddTrg("Trg", ker, ker);
    fmm.SetKernelS2T("Src", "Trg", ker);
    fmm.SetAccuracy((Integer)(std::log(tol_) / std::log(0.1)) + 1);
  }
  ~BoundaryIntegralOp() {
    for (auto& it : elem_lst_map) delete it.second;
  }

  // boundary_integral.txx:516-522: tolerance -> digits for the kernel evaluation
  void SetAccuracy(Real tol) {
    setup_far_flag = false;
    tol_ = tol;
    fmm.SetAccuracy((Integer)(std::log(tol_) / std::log(0.1)) + 1);
  }

  template <class ElemLstType> void AddElemList(const ElemLstType& elem_lst, const std::string& name = std::to_string(typeid(ElemLstType).hash_code())) {
    if (elem_lst_map.find(name) != elem_lst_map.end()) DeleteElemList(name);
    elem_lst_map[name] = static_cast<ElementListBase<Real>*>(new ElemLstType(elem_lst));
    ClearSetup();
  }
  template <class ElemLstType> const ElemLstType& GetElemList(const std::string& name = std::to_string(typeid(ElemLstType).hash_code())) const {
    SCTL_AMD_ASSERT_MSG(elem_lst_map.find(name) != elem_lst_map.end(), "Element list does not exist.");
    return *dynamic_cast<const ElemLstType*>(elem_lst_map.at(name));
  }
  void DeleteElemList(const std::string& name) {
    if (elem_lst_map.find(name) == elem_lst_map.end()) return;
    delete elem_lst_map[name];
    elem_lst_map.erase(name);
    ClearSetup();
  }

  void SetTargetCoord(const Vector<Real>& Xtrg_) {
    Xt = Xtrg_;
    setup_flag = false;
    setup_far_flag = false;
  }
  void SetTargetNormal(const Vector<Real>& Xn_trg_) {
    Xnt = Xn_trg_;
    setup_flag = false;
  }

  // boundary_integral.txx:572-586: k = 0 input (density) dimension, k = 1 output (potential) dimension
  Long Dim(Integer k) const {
    SetupBasic();
    if (k == 0) {
      const Long Nelem = elem_nds_cnt.Dim();
      return (Nelem ? (elem_nds_dsp[Nelem - 1] + elem_nds_cnt[Nelem - 1]) * KDIM0 : 0);
    }
    if (k == 1) return (Xtrg.Dim() / COORD_DIM) * (trg_normal_dot_prod_ ? KDIM1 / COORD_DIM : KDIM1);
    SCTL_AMD_ASSERT(false);
    return -1;
  }

  void Setup() const {
    SetupBasic();
    SetupFar();
  }
  void ClearSetup() const {
    setup_flag = false;
    setup_far_flag = false;
  }

  // boundary_integral.txx:1016-1077
  void ComputeFarField(Vector<Real>& U, const Vector<Real>& F) const {
    Setup();
    const Long Nsrc = X_far.Dim() / COORD_DIM;
    const Long Ntrg = Xtrg.Dim() / COORD_DIM;
    SCTL_AMD_ASSERT(F.Dim() == Dim(0));
    {  // F_far = (density at the far-field nodes) * wts_far
      if (F_far.Dim() != Nsrc * KDIM0) F_far.ReInit(Nsrc * KDIM0);
      const Long Nlst = (Long)elem_lst_name.size();
      for (Long i = 0; i < Nlst; i++) {
        const Long elem_idx0 = elem_lst_dsp[i], elem_idx1 = elem_lst_dsp[i] + elem_lst_cnt[i];
        const Long offset0 = (!elem_lst_cnt[i] ? 0 : elem_nds_dsp[elem_idx0]);
        const Long offset1 = (!elem_lst_cnt[i] ? 0 : elem_nds_dsp[elem_idx1 - 1] + elem_nds_cnt[elem_idx1 - 1]);
        const Vector<Real> F_((offset1 - offset0) * KDIM0, (Iterator<Real>)F.begin() + offset0 * KDIM0, false);
        const Long offset0_far = (!elem_lst_cnt[i] ? 0 : elem_nds_dsp_far[elem_idx0]);
        const Long offset1_far = (!elem_lst_cnt[i] ? 0 : elem_nds_dsp_far[elem_idx1 - 1] + elem_nds_cnt_far[elem_idx1 - 1]);
        Vector<Real> F_far_((offset1_far - offset0_far) * KDIM0, F_far.begin() + offset0_far * KDIM0, false);
        elem_lst_map.at(elem_lst_name[i])->GetFarFieldDensity(F_far_, F_);
        if (F_far_.Dim()) {
          SCTL_AMD_ASSERT(F_far_.begin() == F_far.begin() + offset0_far * KDIM0);   // filled in place, not reallocated
          for (Long j = offset0_far; j < offset1_far; j++)
            for (Long k = 0; k < KDIM0; k++) F_far[j * KDIM0 + k] *= wts_far[j];
        } else {
          SCTL_AMD_ASSERT(offset1_far - offset0_far == offset1 - offset0);
          for (Long j = offset0_far; j < offset1_far; j++)
            for (Long k = 0; k < KDIM0; k++) F_far[j * KDIM0 + k] = F_[(j - offset0_far) * KDIM0 + k] * wts_far[j];
        }
      }
    }
    fmm.SetSrcDensity("Src", F_far);

    const Integer KDIM1_ = (trg_normal_dot_prod_ ? KDIM1 / COORD_DIM : KDIM1);
    if (U.Dim() != Ntrg * KDIM1_) U.ReInit(Ntrg * KDIM1_);
    U.SetZero();
    if (trg_normal_dot_prod_) {
      Vector<Real> U_(Ntrg * KDIM1);
      U_.SetZero();
      fmm.Eval(U_, "Trg");
      for (Long i = 0; i < Ntrg; i++)
        for (Long k = 0; k < KDIM1_; k++)
          for (Long l = 0; l < COORD_DIM; l++) U[i * KDIM1_ + k] += U_[(i * KDIM1_ + k) * COORD_DIM + l] * Xn_trg[i * COORD_DIM + l];
    } else {
      fmm.Eval(U, "Trg");
    }
  }

  // boundary_integral.txx:608-614 is ComputeFarField + ComputeNearInterac; only the first is implemented here.
  void ComputePotential(Vector<Real>& U, const Vector<Real>& F) const {
    Setup();
    for (Long i = 0; i < dist_far.Dim(); i++) {
      if (dist_far[i] > 0)
        SCTL_AMD_ERROR("BoundaryIntegralOp::ComputePotential: this element list has a near zone (far-field distance > 0); the "
                       "near/self corrections of the reference (boundary_integral.txx:786-1142) are outside sctl_amd — call "
                       "ComputeFarField and apply your own corrections");
    }
    ComputeFarField(U, F);
  }

 private:
  static void concat(Vector<Real>& out, const std::vector<Vector<Real>>& parts) {
    Long n = 0;
    for (const auto& p : parts) n += p.Dim();
    out.ReInit(n);
    Long off = 0;
    for (const auto& p : parts) {
      for (Long i = 0; i < p.Dim(); i++) out[off + i] = p[i];
      off += p.Dim();
    }
  }
  static void concat(Vector<Long>& out, const std::vector<Vector<Long>>& parts) {
    Long n = 0;
    for (const auto& p : parts) n += p.Dim();
    out.ReInit(n);
    Long off = 0;
    for (const auto& p : parts) {
      for (Long i = 0; i < p.Dim(); i++) out[off + i] = p[i];
      off += p.Dim();
    }
  }
  static void scan(const Vector<Long>& cnt, Vector<Long>& dsp) {   // exclusive prefix sum (omp_par::scan in the reference)
    dsp.ReInit(cnt.Dim());
    Long s = 0;
    for (Long i = 0; i < cnt.Dim(); i++) { dsp[i] = s; s += cnt[i]; }
  }

  // boundary_integral.txx:690-766
  void SetupBasic() const {
    if (setup_flag) return;
    elem_lst_name.clear();
    for (const auto& x : elem_lst_map) elem_lst_name.push_back(x.first);
    const Long Nlst = (Long)elem_lst_name.size();
    std::vector<Vector<Real>> Xsurf_(Nlst), Xn_surf_(Nlst);
    std::vector<Vector<Long>> cnt_(Nlst);
    elem_lst_cnt.ReInit(Nlst);
    for (Long i = 0; i < Nlst; i++) {
      elem_lst_map.at(elem_lst_name[i])->GetNodeCoord(&Xsurf_[i], &Xn_surf_[i], &cnt_[i]);
      elem_lst_cnt[i] = cnt_[i].Dim();
    }
    concat(Xsurf, Xsurf_);
    concat(Xn_surf, Xn_surf_);
    concat(elem_nds_cnt, cnt_);
    scan(elem_lst_cnt, elem_lst_dsp);
    scan(elem_nds_cnt, elem_nds_dsp);
    Xtrg = (Xt.Dim() ? Xt : Xsurf);   // no explicit targets: evaluate on the surface nodes (:748-756)
    if (trg_normal_dot_prod_) {
      if (Xnt.Dim()) {
        Xn_trg = Xnt;
        SCTL_AMD_ASSERT_MSG(Xn_trg.Dim() == Xtrg.Dim(), "Invalid normal vector at targets.");
      } else {
        Xn_trg = Xn_surf;
      }
    }
    setup_flag = true;
  }

  // boundary_integral.txx:744-782
  void SetupFar() const {
    if (setup_far_flag) return;
    SetupBasic();
    const Long Nlst = (Long)elem_lst_name.size();
    std::vector<Vector<Real>> X_far_(Nlst), Xn_far_(Nlst), wts_far_(Nlst), dist_far_(Nlst);
    std::vector<Vector<Long>> cnt_far_(Nlst);
    for (Long i = 0; i < Nlst; i++)
      elem_lst_map.at(elem_lst_name[i])->GetFarFieldNodes(X_far_[i], Xn_far_[i], wts_far_[i], dist_far_[i], cnt_far_[i], tol_);
    concat(X_far, X_far_);
    concat(Xn_far, Xn_far_);
    concat(wts_far, wts_far_);
    concat(dist_far, dist_far_);
    concat(elem_nds_cnt_far, cnt_far_);
    SCTL_AMD_ASSERT(elem_nds_cnt_far.Dim() == elem_nds_cnt.Dim());
    scan(elem_nds_cnt_far, elem_nds_dsp_far);
    fmm.SetSrcCoord("Src", X_far, Xn_far);
    fmm.SetTrgCoord("Trg", Xtrg);
    setup_far_flag = true;
  }

  std::map<std::string, ElementListBase<Real>*> elem_lst_map;
  Vector<Real> Xt, Xnt;   // user-specified targets and target normals
  Real tol_;
  Kernel ker_;
  bool trg_normal_dot_prod_;
  Comm comm_;

  mutable bool setup_flag, setup_far_flag;
  mutable std::vector<std::string> elem_lst_name;
  mutable Vector<Long> elem_lst_cnt, elem_lst_dsp;          // elements per element list
  mutable Vector<Long> elem_nds_cnt, elem_nds_dsp;          // surface nodes per element
  mutable Vector<Real> Xsurf, Xn_surf, Xtrg, Xn_trg;
  mutable ParticleFMM<Real, COORD_DIM> fmm;
  mutable Vector<Long> elem_nds_cnt_far, elem_nds_dsp_far;  // far-field nodes per element
  mutable Vector<Real> X_far, Xn_far, wts_far, dist_far, F_far;
};

}  // namespace sctl_amd
#endif  // SCTL_AMD_BOUNDARY_INTEGRAL_HPP_
