// BoundaryIntegralOp<Real,Kernel>: the reference's boundary-integral operator, far field and near field
// (include/sctl/boundary_integral.hpp:223-410), which is the caller of the direct-summation hot path
// (SURVEY.md §8 a14): U = K_far[F] evaluated as
//     F_far = density at the far-field quadrature nodes x quadrature weights      boundary_integral.txx:1021-1053
//     fmm.SetSrcDensity("Src", F_far); fmm.Eval(U, "Trg")                          boundary_integral.txx:1054-1073
//     optional dot product of the K1/3 x 3 output with the target normals          boundary_integral.txx:1060-1071
// (here one sctl_amd_op_eval: the weights multiply and the normal contraction run on the device)
// with the wiring of the constructor (:500-509), SetupBasic (:690-766) and SetupFar (:744-782).
//
// Same names and argument meaning as the reference for everything on that leg: ElementListBase (Size, GetNodeCoord,
// GetFarFieldNodes, GetFarFieldDensity, MatrixFree), BoundaryIntegralOp (SetAccuracy, AddElemList, GetElemList,
// DeleteElemList, SetTargetCoord, SetTargetNormal, Dim, Setup, ClearSetup, ComputeFarField, ComputePotential).
//
// NEAR FIELD.  For element lists with a near zone (far-field distance > 0) the reference corrects
// the far-field quadrature near each element with precomputed operator matrices:
//     SetupSelf   K_self[e] from ElemLstType::SelfInterac<Kernel>                    boundary_integral.txx:784-814
//     SetupNear   near lists (BuildNearList, :46-468), K_near from K_self / NearInterac (:860-942), minus the direct
//                 far-field quadrature through KernelMatrix and FarFieldDensityOperatorTranspose (:944-1009)
//     ComputeNearInterac   U_ = F_ K_near_ per element, scatter, accumulate          boundary_integral.txx:1079-1142
// Here the setup runs on the host (it calls the user's element-list code; the direct part comes from ONE batched device
// KernelMatrix launch per element list) and the application runs on the device: the assembled arrays go to sctl_amd_near_create once and every
// ComputeNearInterac is one sctl_amd_near_apply_host.  The near list is built for ONE rank with a uniform cell grid
// instead of the reference's distributed Morton tree; it yields the same lists (element-major, targets ascending).
// Matrix-free element lists (EvalNearInterac) are evaluated on the host, as in the reference.
#ifndef SCTL_AMD_BOUNDARY_INTEGRAL_HPP_
#define SCTL_AMD_BOUNDARY_INTEGRAL_HPP_

#include <algorithm>
#include <cmath>
#include <map>
#include <string>
#include <typeinfo>
#include <unordered_map>
#include <vector>

#include "comm.hpp"
#include "generic-kernel.hpp"
#include "kernel_functions.hpp"

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
  // transpose of the GetFarFieldDensity operator of ONE element applied to the columns of Min; an empty Mout means identity
  virtual void FarFieldDensityOperatorTranspose(Matrix<Real>& Mout, const Matrix<Real>& Min, const Long elem_idx) const {
    if (Mout.Dim(0) != 0 && Mout.Dim(1) != 0) Mout.ReInit(0, 0);
  }
  // Singular / near-singular quadratures of a concrete element list (boundary_integral.hpp:146-206).  An element list
  // with a near zone redefines the ones it needs as static member templates of the same signature; these defaults abort.
  template <class Kernel> static void SelfInterac(std::vector<Matrix<Real>>& M_lst, const Kernel& ker, Real tol, bool trg_dot_prod, const ElementListBase<Real>* self) {
    SCTL_AMD_ERROR("ElementListBase::SelfInterac: this element list has a near zone but defines no SelfInterac<Kernel>");
  }
  template <class Kernel> static void NearInterac(Matrix<Real>& M, const Vector<Real>& Xt, const Vector<Real>& normal_trg, const Kernel& ker, Real tol, const Long elem_idx, const ElementListBase<Real>* self) {
    SCTL_AMD_ERROR("ElementListBase::NearInterac: this element list has a near zone but defines no NearInterac<Kernel>");
  }
  template <class Kernel> static void EvalNearInterac(Vector<Real>& u, const Vector<Real>& f, const Vector<Real>& Xt, const Vector<Real>& normal_trg, const Kernel& ker, Real tol, const Long elem_idx, const ElementListBase<Real>* self) {
    SCTL_AMD_ERROR("ElementListBase::EvalNearInterac: this matrix-free element list has a near zone but defines no EvalNearInterac<Kernel>");
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
      : tol_(1e-10), ker_(ker), trg_normal_dot_prod_(trg_normal_dot_prod), comm_(comm) {
    SCTL_AMD_ASSERT(!trg_normal_dot_prod_ || (KDIM1 % COORD_DIM == 0));
    ClearSetup();
  }
  ~BoundaryIntegralOp() {
    ReleaseNearOp();
    ReleaseFarOp();
    for (auto& it : elem_lst_map) delete it.second;
  }

  // boundary_integral.txx:516-522: tolerance -> digits for the kernel evaluation
  void SetAccuracy(Real tol) {
    setup_far_flag = false;
    setup_self_flag = false;
    setup_near_flag = false;
    tol_ = tol;
  }

  template <class ElemLstType> void AddElemList(const ElemLstType& elem_lst, const std::string& name = std::to_string(typeid(ElemLstType).hash_code())) {
    if (elem_lst_map.find(name) != elem_lst_map.end()) DeleteElemList(name);
    elem_lst_map[name] = static_cast<ElementListBase<Real>*>(new ElemLstType(elem_lst));
    ElemData& fn = elem_data_map[name];      // boundary_integral.txx:541-543
    fn.SelfInterac = &ElemLstType::template SelfInterac<Kernel>;
    fn.NearInterac = &ElemLstType::template NearInterac<Kernel>;
    fn.EvalNearInterac = &ElemLstType::template EvalNearInterac<Kernel>;
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
    elem_data_map.erase(name);
    ClearSetup();
  }

  void SetTargetCoord(const Vector<Real>& Xtrg_) {
    Xt = Xtrg_;
    setup_flag = false;
    setup_far_flag = false;
    setup_near_flag = false;
  }
  void SetTargetNormal(const Vector<Real>& Xn_trg_) {
    Xnt = Xn_trg_;
    setup_flag = false;
    setup_near_flag = false;
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

  void Setup() const {   // boundary_integral.txx:591-599
    if (setup_flag && setup_far_flag && setup_self_flag && setup_near_flag) return;
    SetupBasic();
    SetupFar();
    SetupSelf();
    SetupNear();
  }
  void ClearSetup() const {
    setup_flag = false;
    setup_far_flag = false;
    setup_self_flag = false;
    setup_near_flag = false;
  }

  // boundary_integral.txx:1016-1077
  void ComputeFarField(Vector<Real>& U, const Vector<Real>& F) const {
    SetupBasic();
    SetupFar();
    const Long Nsrc = X_far.Dim() / COORD_DIM;
    const Long Ntrg = Xtrg.Dim() / COORD_DIM;
    SCTL_AMD_ASSERT(F.Dim() == Dim(0));
    GatherFarFieldDensity(F);
    const Integer KDIM1_ = (trg_normal_dot_prod_ ? KDIM1 / COORD_DIM : KDIM1);
    if (U.Dim() != Ntrg * KDIM1_) U.ReInit(Ntrg * KDIM1_);
    U.SetZero();
    if (!Ntrg || !Nsrc) return;
    // The weights and the target normals live on the devices with the coordinates (sctl_amd_op_set_source_weights /
    // _target_normals, uploaded by SetupFar): one evaluation moves the unweighted density down and the contracted potential up.
    const int rc = sctl_amd_op_eval(far_op, F_far.begin(), U.begin(), /*accumulate*/ 0, fmm_digits(), ker_.GetCtxPtr(), (int)Kernel::CTX_BYTES);
    CheckStatus(rc, "sctl_amd_op_eval");
  }

  // boundary_integral.txx:608-614: far field, then the near-zone correction added to it.  Here both run in ONE pass over the devices of
  // the far-field operator (sctl_amd_op_eval_potential): the densities go down once, the near field is accumulated into the far-field
  // result where it lies (each device holds the columns of K_near that belong to its target slab), the potential comes up once.
  void ComputePotential(Vector<Real>& U, const Vector<Real>& F) const {
    Setup();
    const Long Nelem = near_elem_cnt.Dim();
    const Long N_near = (Nelem ? near_elem_dsp[Nelem - 1] + near_elem_cnt[Nelem - 1] : 0);
    if (!N_near || !far_op) {   // no near zone (or nothing to evaluate on the devices): the two legs as they are
      ComputeFarField(U, F);
      ComputeNearInterac(U, F);
      return;
    }
    const Integer KDIM1_ = (trg_normal_dot_prod_ ? KDIM1 / COORD_DIM : KDIM1);
    const Long Ntrg = Xtrg.Dim() / COORD_DIM;
    GatherFarFieldDensity(F);
    if (U.Dim() != Ntrg * KDIM1_) U.ReInit(Ntrg * KDIM1_);
    if (!near_attached) {
      CheckStatus(sctl_amd_op_set_near(far_op, (int)KDIM0, (int)KDIM1_, Nelem, PtrOf(elem_nds_cnt), PtrOf(near_elem_cnt), PtrOf(K_near_cnt),
                                       K_near.Dim() ? (const void*)K_near.begin() : nullptr, PtrOf(near_scatter_index), PtrOf(near_trg_cnt), PtrOf(near_trg_dsp)),
                  "sctl_amd_op_set_near");
      near_attached = true;
    }
    CheckStatus(sctl_amd_op_eval_potential(far_op, F_far.begin(), F.begin(), U.begin(), /*accumulate*/ 0, fmm_digits(), ker_.GetCtxPtr(), (int)Kernel::CTX_BYTES),
                "sctl_amd_op_eval_potential");
    AddMatrixFreeNearField(U, F);
  }

  // boundary_integral.txx:1079-1142 (private in the reference).  U is ACCUMULATED into when it has the right size.
  void ComputeNearInterac(Vector<Real>& U, const Vector<Real>& F) const {
    Setup();
    const Integer KDIM1_ = (trg_normal_dot_prod_ ? KDIM1 / COORD_DIM : KDIM1);
    const Long Ntrg = Xtrg.Dim() / COORD_DIM;
    const Long Nelem = near_elem_cnt.Dim();
    SCTL_AMD_ASSERT(F.Dim() == Dim(0));
    if (U.Dim() != Ntrg * KDIM1_) {
      U.ReInit(Ntrg * KDIM1_);
      U.SetZero();
    }
    const Long N_near = (Nelem ? near_elem_dsp[Nelem - 1] + near_elem_cnt[Nelem - 1] : 0);
    if (!N_near) return;
    // precomputed operator matrices: on the device (:1092-1102, 1129-1140)
    if (!near_op) {
      const int rc = sctl_amd_near_create(RealTag<Real>::value, DeviceSet::Get()[0], Nelem, (int)KDIM0, (int)KDIM1_, PtrOf(elem_nds_cnt), PtrOf(near_elem_cnt),
                                          PtrOf(K_near_cnt), K_near.Dim() ? (const void*)K_near.begin() : nullptr, Ntrg, PtrOf(near_scatter_index),
                                          PtrOf(near_trg_cnt), PtrOf(near_trg_dsp), &near_op);
      CheckStatus(rc, "sctl_amd_near_create");
    }
    CheckStatus(sctl_amd_near_apply_host(near_op, F.begin(), U.begin()), "sctl_amd_near_apply_host");
    AddMatrixFreeNearField(U, F);
  }

 private:
  // Density at the far-field quadrature nodes, one element list at a time (each list owns a contiguous run of surface nodes and
  // of far-field nodes); the quadrature weights are NOT applied here: they live on the device (boundary_integral.txx:1040-1052).
  void GatherFarFieldDensity(const Vector<Real>& F) const {
    const Long Nsrc = X_far.Dim() / COORD_DIM;
    SCTL_AMD_ASSERT(F.Dim() == Dim(0));
    if (F_far.Dim() != Nsrc * KDIM0) F_far.ReInit(Nsrc * KDIM0);
    for (size_t lst = 0; lst < elem_lst_name.size(); lst++) {
      const NodeRange surf = ListNodes(lst, elem_nds_cnt, elem_nds_dsp), far = ListNodes(lst, elem_nds_cnt_far, elem_nds_dsp_far);
      const Vector<Real> f_in((surf.end - surf.begin) * KDIM0, (Iterator<Real>)F.begin() + surf.begin * KDIM0, false);
      Real* const dst = F_far.begin() + far.begin * KDIM0;
      Vector<Real> f_out((far.end - far.begin) * KDIM0, dst, false);
      elem_lst_map.at(elem_lst_name[lst])->GetFarFieldDensity(f_out, f_in);
      if (f_out.Dim() == 0) {   // the list's answer for "far-field nodes are the surface nodes": the density passes through
        SCTL_AMD_ASSERT(far.end - far.begin == surf.end - surf.begin);
        std::copy(f_in.begin(), f_in.begin() + f_in.Dim(), dst);
      } else {
        SCTL_AMD_ASSERT(f_out.begin() == dst);   // written into the view, not into a reallocated vector
      }
    }
  }

  // Matrix-free element lists: their near zone is evaluated by the user's code on the host (boundary_integral.txx:1104-1125) and
  // scattered to the targets like the device part (:1129-1140).
  void AddMatrixFreeNearField(Vector<Real>& U, const Vector<Real>& F) const {
    bool any_matrix_free = false;
    for (const auto& name : elem_lst_name) any_matrix_free = any_matrix_free || elem_lst_map.at(name)->MatrixFree();
    if (!any_matrix_free) return;
    const Integer KDIM1_ = (trg_normal_dot_prod_ ? KDIM1 / COORD_DIM : KDIM1);
    const Long Ntrg = Xtrg.Dim() / COORD_DIM, Nelem = near_elem_cnt.Dim();
    const Long N_near = (Nelem ? near_elem_dsp[Nelem - 1] + near_elem_cnt[Nelem - 1] : 0);
    Vector<Real> U_near(N_near * KDIM1_);
    U_near.SetZero();
    for (Long i = 0; i < (Long)elem_lst_name.size(); i++) {
      const ElementListBase<Real>* elem_lst = elem_lst_map.at(elem_lst_name[i]);
      if (!elem_lst->MatrixFree()) continue;
      for (Long j = 0; j < elem_lst_cnt[i]; j++) {
        const Long e = elem_lst_dsp[i] + j, nt = near_elem_cnt[e], src_dof = elem_nds_cnt[e] * KDIM0;
        if (!nt || !src_dof) continue;
        const Vector<Real> Xt_(nt * COORD_DIM, Xtrg_near.begin() + near_elem_dsp[e] * COORD_DIM, false);
        const Vector<Real> Xn_(trg_normal_dot_prod_ ? nt * COORD_DIM : 0, trg_normal_dot_prod_ ? Xn_trg_near.begin() + near_elem_dsp[e] * COORD_DIM : nullptr, false);
        const Vector<Real> F_(src_dof, (Iterator<Real>)F.begin() + elem_nds_dsp[e] * KDIM0, false);
        Vector<Real> U_(nt * KDIM1_, U_near.begin() + near_elem_dsp[e] * KDIM1_, false);
        // the reference hands EvalNearInterac the GLOBAL element index (boundary_integral.txx:1122), unlike NearInterac, which gets
        // the list-local one (:925): kept, so that an element list written against the reference behaves the same here
        elem_data_map.at(elem_lst_name[i]).EvalNearInterac(U_, F_, Xt_, Xn_, ker_, tol_, e, elem_lst);
      }
    }
    for (Long i = 0; i < Ntrg; i++)
      for (Long p = near_trg_dsp[i]; p < near_trg_dsp[i] + near_trg_cnt[i]; p++)
        for (Long k = 0; k < KDIM1_; k++) U[i * KDIM1_ + k] += U_near[near_scatter_index[p] * KDIM1_ + k];
  }

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
  // nodes [begin, end) owned by element list `lst` in a per-element (count, displacement) layout; empty lists own nothing
  struct NodeRange { Long begin, end; };
  NodeRange ListNodes(size_t lst, const Vector<Long>& cnt, const Vector<Long>& dsp) const {
    const Long n = elem_lst_cnt[(Long)lst];
    if (n == 0) return NodeRange{0, 0};
    const Long first = elem_lst_dsp[(Long)lst], last = first + n - 1;
    return NodeRange{dsp[first], dsp[last] + cnt[last]};
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
    // The reference hands these to its ParticleFMM member (:776-779); here they go straight to a device-resident operator,
    // together with the quadrature weights and the target normals, which ComputeFarField then never touches on the host.
    ReleaseFarOp();
    const Long Nsrc = X_far.Dim() / COORD_DIM, Ntrg = Xtrg.Dim() / COORD_DIM;
    if (Nsrc && Ntrg) {
      const std::vector<int>& devs = DeviceSet::Get();
      CheckStatus(sctl_amd_op_create(ker_.DeviceKernelId(), RealTag<Real>::value, devs.data(), (int)devs.size(), &far_op), "sctl_amd_op_create");
      CheckStatus(sctl_amd_op_set_targets(far_op, Ntrg, Xtrg.begin()), "sctl_amd_op_set_targets");
      CheckStatus(sctl_amd_op_set_sources(far_op, Nsrc, X_far.begin(), Kernel::NormalDim() ? Xn_far.begin() : nullptr), "sctl_amd_op_set_sources");
      CheckStatus(sctl_amd_op_set_source_weights(far_op, wts_far.begin()), "sctl_amd_op_set_source_weights");
      if (trg_normal_dot_prod_) CheckStatus(sctl_amd_op_set_target_normals(far_op, Xn_trg.begin()), "sctl_amd_op_set_target_normals");
    }
    setup_far_flag = true;
  }

  // boundary_integral.txx:784-814
  void SetupSelf() const {
    if (setup_self_flag) return;
    SetupBasic();
    SetupFar();
    K_self.assign((size_t)elem_nds_cnt.Dim(), Matrix<Real>());
    bool near_zone = false;
    for (Long i = 0; i < dist_far.Dim(); i++) near_zone = near_zone || dist_far[i] > 0;
    for (Long i = 0; near_zone && i < (Long)elem_lst_name.size(); i++) {   // no near zone: nothing will read K_self
      const ElementListBase<Real>* elem_lst = elem_lst_map.at(elem_lst_name[i]);
      if (elem_lst->MatrixFree()) continue;
      std::vector<Matrix<Real>> K_(elem_lst_cnt[i]);
      elem_data_map.at(elem_lst_name[i]).SelfInterac(K_, ker_, tol_, trg_normal_dot_prod_, elem_lst);
      SCTL_AMD_ASSERT((Long)K_.size() == elem_lst_cnt[i]);
      for (Long j = 0; j < elem_lst_cnt[i]; j++) K_self[elem_lst_dsp[i] + j] = K_[j];
    }
    setup_self_flag = true;
  }

  // Single-rank equivalent of BuildNearList (boundary_integral.txx:46-468): target t is near element e when it lies
  // strictly inside the sphere of radius dist_far of one of e's far-field nodes (:374).  Lists are element-major with
  // targets ascending (:383-392); near_scatter_index groups the entries by target, ties in list order (:436-443).
  void BuildNearList() const {
    const Long Ntrg = Xtrg.Dim() / COORD_DIM, Nelem = elem_nds_cnt_far.Dim();
    Real rmax = 0;
    for (Long i = 0; i < dist_far.Dim(); i++) rmax = std::max(rmax, dist_far[i]);
    std::vector<std::vector<Long>> near(Nelem);
    if (rmax > 0 && Ntrg > 0) {
      Real lo[3] = {Xtrg[0], Xtrg[1], Xtrg[2]};
      for (Long t = 0; t < Ntrg; t++)
        for (Integer k = 0; k < COORD_DIM; k++) lo[k] = std::min(lo[k], Xtrg[t * COORD_DIM + k]);
      auto cell_of = [&](const Real* x, long long (&c)[3]) { for (Integer k = 0; k < COORD_DIM; k++) c[k] = (long long)std::floor((x[k] - lo[k]) / rmax); };
      auto key_of = [](const long long (&c)[3]) { return (unsigned long long)((c[0] * 73856093LL) ^ (c[1] * 19349663LL) ^ (c[2] * 83492791LL)); };
      std::unordered_map<unsigned long long, std::vector<Long>> grid;
      for (Long t = 0; t < Ntrg; t++) {
        long long c[3];
        cell_of(&Xtrg[t * COORD_DIM], c);
        grid[key_of(c)].push_back(t);
      }
      std::vector<Long> cand;
      for (Long e = 0; e < Nelem; e++) {
        cand.clear();
        for (Long s = elem_nds_dsp_far[e]; s < elem_nds_dsp_far[e] + elem_nds_cnt_far[e]; s++) {
          const Real rad = dist_far[s];
          if (!(rad > 0)) continue;
          long long c[3], d[3];
          cell_of(&X_far[s * COORD_DIM], c);
          for (d[0] = c[0] - 1; d[0] <= c[0] + 1; d[0]++)
            for (d[1] = c[1] - 1; d[1] <= c[1] + 1; d[1]++)
              for (d[2] = c[2] - 1; d[2] <= c[2] + 1; d[2]++) {
                const auto it = grid.find(key_of(d));
                if (it == grid.end()) continue;
                for (const Long t : it->second) {   // hash collisions only add candidates: the distance test decides
                  Real r2 = 0;
                  for (Integer k = 0; k < COORD_DIM; k++) r2 += (X_far[s * COORD_DIM + k] - Xtrg[t * COORD_DIM + k]) * (X_far[s * COORD_DIM + k] - Xtrg[t * COORD_DIM + k]);
                  if (r2 < rad * rad) cand.push_back(t);
                }
              }
        }
        std::sort(cand.begin(), cand.end());
        cand.erase(std::unique(cand.begin(), cand.end()), cand.end());
        near[e] = cand;
      }
    }
    near_elem_cnt.ReInit(Nelem);
    for (Long e = 0; e < Nelem; e++) near_elem_cnt[e] = (Long)near[e].size();
    scan(near_elem_cnt, near_elem_dsp);
    const Long N_near = (Nelem ? near_elem_dsp[Nelem - 1] + near_elem_cnt[Nelem - 1] : 0);
    Xtrg_near.ReInit(N_near * COORD_DIM);
    Xn_trg_near.ReInit(trg_normal_dot_prod_ ? N_near * COORD_DIM : 0);
    std::vector<Long> trg_idx(N_near);
    for (Long e = 0, i = 0; e < Nelem; e++)
      for (const Long t : near[e]) {
        for (Integer k = 0; k < COORD_DIM; k++) Xtrg_near[i * COORD_DIM + k] = Xtrg[t * COORD_DIM + k];
        if (trg_normal_dot_prod_) for (Integer k = 0; k < COORD_DIM; k++) Xn_trg_near[i * COORD_DIM + k] = Xn_trg[t * COORD_DIM + k];
        trg_idx[i++] = t;
      }
    std::vector<Long> order(N_near);
    for (Long i = 0; i < N_near; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](Long a, Long b) { return trg_idx[a] < trg_idx[b]; });
    near_scatter_index.ReInit(N_near);
    near_trg_cnt.ReInit(Ntrg);
    near_trg_dsp.ReInit(Ntrg);
    for (Long t = 0; t < Ntrg; t++) near_trg_cnt[t] = near_trg_dsp[t] = 0;
    for (Long p = 0; p < N_near; p++) {
      near_scatter_index[p] = order[p];
      const Long t = trg_idx[order[p]];
      if (near_trg_cnt[t] == 0) near_trg_dsp[t] = p;
      near_trg_cnt[t]++;
    }
  }

  // boundary_integral.txx:816-1012
  void SetupNear() const {
    if (setup_near_flag) return;
    SetupBasic();
    SetupFar();
    SetupSelf();
    ReleaseNearOp();
    BuildNearList();
    const Integer KDIM1_ = (trg_normal_dot_prod_ ? KDIM1 / COORD_DIM : KDIM1);
    const Long Nelem = near_elem_cnt.Dim(), Nlst = (Long)elem_lst_name.size();
    SCTL_AMD_ASSERT(Nelem == elem_nds_cnt.Dim());
    K_near_cnt.ReInit(Nelem);
    for (Long i = 0; i < Nlst; i++) {
      const bool matrix_free = elem_lst_map.at(elem_lst_name[i])->MatrixFree();
      for (Long j = 0; j < elem_lst_cnt[i]; j++) {
        const Long e = elem_lst_dsp[i] + j;
        K_near_cnt[e] = matrix_free ? 0 : elem_nds_cnt[e] * near_elem_cnt[e];
      }
    }
    scan(K_near_cnt, K_near_dsp);
    K_near.ReInit(Nelem ? (K_near_dsp[Nelem - 1] + K_near_cnt[Nelem - 1]) * KDIM0 * KDIM1_ : 0);
    for (Long i = 0; i < Nlst; i++) {
      const ElementListBase<Real>* elem_lst = elem_lst_map.at(elem_lst_name[i]);
      if (elem_lst->MatrixFree()) continue;
      const ElemData& fn = elem_data_map.at(elem_lst_name[i]);
      // The direct far-field quadrature of every element of this list at its own near targets, in ONE device launch: block j
      // is what the reference gets from KernelMatrix(Mker, Xtrg_near_, X, Xn) for element j (:971,986).
      Vector<Real> Mfull_all;
      Vector<Long> blk_nt(elem_lst_cnt[i]), blk_ns(elem_lst_cnt[i]), blk_dsp(elem_lst_cnt[i]);
      if (elem_lst_cnt[i]) {
        const Long e0 = elem_lst_dsp[i], e1 = e0 + elem_lst_cnt[i];
        Long off = 0;
        for (Long e = e0; e < e1; e++) {
          blk_nt[e - e0] = near_elem_cnt[e];
          blk_ns[e - e0] = elem_nds_cnt_far[e];
          blk_dsp[e - e0] = off;
          off += elem_nds_cnt_far[e] * KDIM0 * near_elem_cnt[e] * KDIM1;
        }
        const Long t0 = near_elem_dsp[e0], t1 = near_elem_dsp[e1 - 1] + near_elem_cnt[e1 - 1];
        const Long s0 = elem_nds_dsp_far[e0], s1 = elem_nds_dsp_far[e1 - 1] + elem_nds_cnt_far[e1 - 1];
        const Vector<Real> Xt_((t1 - t0) * COORD_DIM, Xtrg_near.begin() + t0 * COORD_DIM, false);
        const Vector<Real> Xs_((s1 - s0) * COORD_DIM, X_far.begin() + s0 * COORD_DIM, false), Xn_((s1 - s0) * COORD_DIM, Xn_far.begin() + s0 * COORD_DIM, false);
        ker_.template KernelMatrixBatch<Real, true>(Mfull_all, blk_nt, blk_ns, Xt_, Xs_, Xn_);
      }
      for (Long j = 0; j < elem_lst_cnt[i]; j++) {
        const Long e = elem_lst_dsp[i] + j, nt = near_elem_cnt[e], nds = elem_nds_cnt[e], N0 = nds * KDIM0;
        if (!nt || !nds) continue;
        Matrix<Real> K_near_(N0, nt * KDIM1_, K_near.begin() + K_near_dsp[e] * KDIM0 * KDIM1_, false);
        for (Long k = 0; k < nt; k++) {   // singular / near-singular quadrature, one target at a time (:860-942)
          const Vector<Real> Xt_(COORD_DIM, Xtrg_near.begin() + (near_elem_dsp[e] + k) * COORD_DIM, false);
          const Vector<Real> Xn_(trg_normal_dot_prod_ ? COORD_DIM : 0, trg_normal_dot_prod_ ? Xn_trg_near.begin() + (near_elem_dsp[e] + k) * COORD_DIM : nullptr, false);
          Long min_node = -1;
          Real min_r2 = -1;
          for (Long n = 0; n < nds; n++) {
            Real r2 = 0;
            for (Integer l = 0; l < COORD_DIM; l++) r2 += (Xt_[l] - Xsurf[(elem_nds_dsp[e] + n) * COORD_DIM + l]) * (Xt_[l] - Xsurf[(elem_nds_dsp[e] + n) * COORD_DIM + l]);
            if (min_r2 < 0 || r2 < min_r2) { min_r2 = r2; min_node = n; }
          }
          if (min_r2 == 0) {   // the target is a node of this element: its column block of the self-interaction matrix
            const Matrix<Real>& K0 = K_self[e];
            const bool have = K0.Dim(0) && K0.Dim(1);
            SCTL_AMD_ASSERT(!have || K0.Dim(0) == N0);
            for (Long l = 0; l < N0; l++)
              for (Long k1 = 0; k1 < KDIM1_; k1++) K_near_[l][k * KDIM1_ + k1] = have ? K0[l][min_node * KDIM1_ + k1] : 0;
          } else {
            Matrix<Real> K0;
            fn.NearInterac(K0, Xt_, Xn_, ker_, tol_, j, elem_lst);
            const bool have = K0.Dim(0) && K0.Dim(1);
            SCTL_AMD_ASSERT(!have || (K0.Dim(0) == N0 && K0.Dim(1) == KDIM1_));
            for (Long l = 0; l < N0; l++)
              for (Long k1 = 0; k1 < KDIM1_; k1++) K_near_[l][k * KDIM1_ + k1] = have ? K0[l][k1] : 0;
          }
        }
        {  // minus what the far-field quadrature of this element already contributes at these targets (:944-1009)
          const Long ns = elem_nds_cnt_far[e], s0 = elem_nds_dsp_far[e];
          Matrix<Real> Mker(ns * KDIM0, nt * KDIM1_);
          const Matrix<Real> Mfull(ns * KDIM0, nt * KDIM1, Mfull_all.begin() + blk_dsp[j], false);   // computed on the device, full precision
          for (Long sidx = 0; sidx < ns; sidx++)
            for (Long k0 = 0; k0 < KDIM0; k0++)
              for (Long t = 0; t < nt; t++)
                for (Long k1 = 0; k1 < KDIM1_; k1++) {
                  Real v = 0;
                  if (trg_normal_dot_prod_) {
                    for (Long l = 0; l < COORD_DIM; l++)
                      v += Mfull[sidx * KDIM0 + k0][(t * KDIM1_ + k1) * COORD_DIM + l] * wts_far[s0 + sidx] * Xn_trg_near[(near_elem_dsp[e] + t) * COORD_DIM + l];
                  } else {
                    v = Mfull[sidx * KDIM0 + k0][t * KDIM1_ + k1] * wts_far[s0 + sidx];
                  }
                  Mker[sidx * KDIM0 + k0][t * KDIM1_ + k1] = v;
                }
          Matrix<Real> K_direct;
          elem_lst->FarFieldDensityOperatorTranspose(K_direct, Mker, j);
          const Matrix<Real>& D = (K_direct.Dim(0) && K_direct.Dim(1)) ? K_direct : Mker;
          SCTL_AMD_ASSERT(D.Dim(0) == K_near_.Dim(0) && D.Dim(1) == K_near_.Dim(1));
          for (Long l = 0; l < N0; l++)
            for (Long c = 0; c < nt * KDIM1_; c++) K_near_[l][c] -= D[l][c];
        }
      }
    }
    setup_near_flag = true;
  }

  int fmm_digits() const { return (int)(std::log(tol_) / std::log(0.1)) + 1; }   // tolerance -> digits, as :520
  void ReleaseFarOp() const {
    if (far_op) sctl_amd_op_destroy(far_op);
    far_op = nullptr;
    near_attached = false;
  }
  void ReleaseNearOp() const {
    if (near_op) sctl_amd_near_destroy(near_op);
    near_op = nullptr;
    if (near_attached && far_op) CheckStatus(sctl_amd_op_set_near(far_op, (int)KDIM0, 1, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr), "sctl_amd_op_set_near");
    near_attached = false;
  }
  static const int64_t* PtrOf(const Vector<Long>& v) {
    static_assert(sizeof(Long) == sizeof(int64_t), "Long must be 64 bits wide");
    return v.Dim() ? reinterpret_cast<const int64_t*>(&v[0]) : nullptr;
  }

  struct ElemData {   // boundary_integral.hpp:398-402
    void (*SelfInterac)(std::vector<Matrix<Real>>&, const Kernel&, Real, bool, const ElementListBase<Real>*);
    void (*NearInterac)(Matrix<Real>&, const Vector<Real>&, const Vector<Real>&, const Kernel&, Real, const Long, const ElementListBase<Real>*);
    void (*EvalNearInterac)(Vector<Real>&, const Vector<Real>&, const Vector<Real>&, const Vector<Real>&, const Kernel&, Real, const Long, const ElementListBase<Real>*);
  };

  std::map<std::string, ElementListBase<Real>*> elem_lst_map;
  std::map<std::string, ElemData> elem_data_map;
  Vector<Real> Xt, Xnt;   // user-specified targets and target normals
  Real tol_;
  Kernel ker_;
  bool trg_normal_dot_prod_;
  Comm comm_;

  mutable bool setup_flag, setup_far_flag, setup_self_flag = false, setup_near_flag = false;
  mutable std::vector<std::string> elem_lst_name;
  mutable Vector<Long> elem_lst_cnt, elem_lst_dsp;          // elements per element list
  mutable Vector<Long> elem_nds_cnt, elem_nds_dsp;          // surface nodes per element
  mutable Vector<Real> Xsurf, Xn_surf, Xtrg, Xn_trg;
  mutable sctl_amd_op* far_op = nullptr;                    // far field: coordinates, weights and target normals resident on the GPUs
  mutable Vector<Long> elem_nds_cnt_far, elem_nds_dsp_far;  // far-field nodes per element
  mutable Vector<Real> X_far, Xn_far, wts_far, dist_far, F_far;
  // near field: the arrays of boundary_integral.hpp:381-396, and their device-resident form
  mutable std::vector<Matrix<Real>> K_self;
  mutable Vector<Real> Xtrg_near, Xn_trg_near, K_near;
  mutable Vector<Long> near_elem_cnt, near_elem_dsp, near_scatter_index, near_trg_cnt, near_trg_dsp, K_near_cnt, K_near_dsp;
  mutable sctl_amd_near* near_op = nullptr;                 // ComputeNearInterac on its own (device 0)
  mutable bool near_attached = false;                       // K_near partitioned over far_op's devices (ComputePotential)
};

}  // namespace sctl_amd
#endif  // SCTL_AMD_BOUNDARY_INTEGRAL_HPP_
