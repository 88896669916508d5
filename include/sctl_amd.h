/* sctl_amd.h — C ABI of the MI355X (gfx950) direct kernel-summation library, libsctl_amd.so.
 *
 * Drop-in boundary for ONE hot path of SCTL (reference snapshot iostanin1/SCTL @ 2024-11-08; file:line
 * citations are relative to that tree): the all-pairs source->target kernel summation
 *
 *     v_trg[t,k1] += scale * sum_s sum_k0 U(x_t - x_s, n_s)[k0][k1] * v_src[s,k0]
 *
 * that the reference performs in GenericKernel<uKernel>::Eval (include/sctl/generic-kernel.txx:76-189),
 * reached from ParticleFMM::EvalDirect (include/sctl/fmm-wrapper.txx:557) and
 * BoundaryIntegralOp::ComputeFarField (include/sctl/boundary_integral.txx:1063,1073), plus the dense
 * operator build GenericKernel::KernelMatrix (generic-kernel.txx:191-307).
 *
 * Only PODs cross this boundary: plain pointers, sizes, enums.  No C++ types, no torch types.
 * All arrays are contiguous AoS exactly as SCTL's Vector<Real> holds them (generic-kernel.txx:127-129):
 *     r_trg[Nt*3], r_src[Ns*3], n_src[Ns*NormalDim] (NULL when NormalDim == 0), v_src[Ns*SrcDim],
 *     v_trg[Nt*TrgDim].
 * Every function returns SCTL_AMD_OK (0) or a negative error code; sctl_amd_last_error() gives the text
 * for the calling thread.  The reference's convention (SCTL_ASSERT -> abort, common.hpp:59-70) is restored
 * by the header-only C++ wrapper include/sctl_amd/generic-kernel.hpp, not here.
 * Thread safety: every entry point is re-entrant (KernelMatrix is called from inside an OpenMP parallel
 * region by boundary_integral.txx:949-986); there is no global mutable state beyond per-thread error text.
 * There is NO CPU fallback: without a HIP device every compute entry returns SCTL_AMD_ERR_NO_DEVICE.
 */
#ifndef SCTL_AMD_H_
#define SCTL_AMD_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCTL_AMD_VERSION 200 /* 0.2.0 */

/* Precision of every array in a call (the reference's template parameter Real). */
enum sctl_amd_real { SCTL_AMD_F64 = 0, SCTL_AMD_F32 = 1 };

/* Kernel identities.  0-7 replace the functors of include/sctl/kernel_functions.hpp:15-198 (same Name()
 * strings, scale factors and FLOPS()); 8 and 9 are functors the reference does not have. */
enum sctl_amd_kernel {
  SCTL_AMD_LAPLACE3D_FXU = 0,    /* "Laplace3D-FxU"    1x1, 1/(4 pi) / r                        kernel_functions.hpp:15-31   */
  SCTL_AMD_LAPLACE3D_DXU = 1,    /* "Laplace3D-DxU"    1x1, normal, (r.n) / r^3                 kernel_functions.hpp:33-51   */
  SCTL_AMD_LAPLACE3D_FXDU = 2,   /* "Laplace3D-FxdU"   1x3, -1/(4 pi) r_j / r^3                 kernel_functions.hpp:53-72   */
  SCTL_AMD_STOKES3D_FXU = 3,     /* "Stokes3D-FxU"     3x3, Stokeslet, 1/(8 pi)                 kernel_functions.hpp:74-95   */
  SCTL_AMD_STOKES3D_DXU = 4,     /* "Stokes3D-DxU"     3x3, normal, stresslet, 3/(4 pi)         kernel_functions.hpp:97-120  */
  SCTL_AMD_STOKES3D_FXT = 5,     /* "Stokes3D-FxT"     3x9, traction tensor, -3/(4 pi)          kernel_functions.hpp:122-146 */
  SCTL_AMD_STOKES3D_FSXU = 6,    /* "Stokes3D-FSxU"    4x3, Stokeslet + source/sink             kernel_functions.hpp:148-172 */
  SCTL_AMD_STOKES3D_FXUP = 7,    /* "Stokes3D-FxUP"    3x4, velocity + pressure                 kernel_functions.hpp:174-198 */
  SCTL_AMD_LAPLACE3D_FDXUDU = 8, /* "Laplace3D-FDxUdU" 2x4, normal, {q, mu} -> {u, grad u}      (new, BASELINE config 2)     */
  SCTL_AMD_HELMHOLTZ3D_FXU = 9,  /* "Helmholtz3D-FxU"  2x2, exp(ikr)/(4 pi r), ctx = {Re k, Im k} as 2 doubles (new, config 5) */
  SCTL_AMD_NUM_KERNELS = 10      /* built-in kernels; ids from here on belong to kernels registered by plugins (below) */
};

enum sctl_amd_status {
  SCTL_AMD_OK = 0,
  SCTL_AMD_ERR_UNKNOWN_KERNEL = -1, /* functor not implemented on the device: caller keeps its own CPU path          */
  SCTL_AMD_ERR_BAD_ARGUMENT = -2,   /* the size checks of generic-kernel.txx:94-97                                     */
  SCTL_AMD_ERR_NO_DEVICE = -3,      /* no HIP device / bad device index: there is no CPU fallback behind this ABI      */
  SCTL_AMD_ERR_HIP = -4,            /* a HIP runtime call failed; text in sctl_amd_last_error()                        */
  SCTL_AMD_ERR_BAD_CONTEXT = -5,    /* kernel needs a context blob (Helmholtz wavenumber) of another size              */
  SCTL_AMD_ERR_PEER = -6            /* rank-parallel call: another rank failed, left or never joined; no rank is left waiting */
};

/* ---- library / registry -------------------------------------------------------------------------------- */
int sctl_amd_version(void);
const char* sctl_amd_last_error(void);          /* message of the last failure on the calling thread ("" if none) */
int sctl_amd_device_count(void);                /* number of HIP devices visible, 0 if none (never an error)       */
/* Optional (everything initialises lazily).  init: touch every visible GPU now, so that the first evaluation does not pay the runtime's
 * first-use cost; returns the number of GPUs (0 without any: not an error) or a negative status.  finalize: give back what the library
 * keeps between calls (sctl_amd_trim + the calling thread's cached streams, device buffers and pinned staging); the library stays
 * usable afterwards.  Neither exists in the reference, whose CPU path has no such state; a patched SCTL calls them from Comm::MPI_Init /
 * MPI_Finalize (comm.txx:117-140) if at all. */
int sctl_amd_init(void);
void sctl_amd_finalize(void);

/* Kernel id for a functor's Name() string (kernel_functions.hpp:16-19), or SCTL_AMD_ERR_UNKNOWN_KERNEL:
 * the "is this kernel supported on the device" query of the header wrapper. */
int sctl_amd_kernel_id(const char* name);
const char* sctl_amd_kernel_name(int kernel);   /* NULL for an unknown id */
/* Shape table: SrcDim, TrgDim, NormalDim (generic-kernel.hpp:59-84), FLOPS() (kernel_functions.hpp:20-22),
 * uKerScaleFactor<double>() (:23-25) and the size in bytes of the context blob the kernel needs (0 = none).
 * Any output pointer may be NULL. */
int sctl_amd_kernel_info(int kernel, int* src_dim, int* trg_dim, int* normal_dim, int* flops, double* scale, int* ctx_bytes);
/* Algorithmic flops per pair interaction by SURVEY.md §8(d): 3 + FLOPS() + 2*SrcDim*TrgDim. */
int sctl_amd_flops_per_pair(int kernel);
int sctl_amd_num_kernels(void);                 /* built-in + registered: valid ids are 0 .. sctl_amd_num_kernels()-1 */

/* ---- user-defined kernel functors (the device side of doc/tutorial/kernels.rst:11-84) ------------------------ */
/* SCTL lets a user write a functor (Name, FLOPS, uKerScaleFactor, uKerMatrix) and get Eval / KernelMatrix for it from
 * GenericKernel<uKernel> (generic-kernel.hpp:31-152).  The device counterpart: the user writes the functor's device form (a struct
 * with pack()/pair(), see include/sctl_amd/device/kernel_plugin.hpp and the built-in ones in device/ukernels.hpp), compiles it
 * with hipcc for gfx950 into a shared object, and registers it; every entry of this ABI then accepts its id, and the header
 * wrappers find it by Name().  desc->launch_table points at the sctl_amd::KernelEntry that device/launch.hpp's
 * make_entry<Ker>() builds (function pointers into the plugin's own code object); abi_version and desc_bytes guard against a
 * plugin compiled with other device headers.  Returns the new kernel id (>= SCTL_AMD_NUM_KERNELS) or a negative error code
 * (a name that is already registered is refused).  Registered kernels live until the process ends. */
#define SCTL_AMD_DEVICE_ABI 3
typedef struct sctl_amd_kernel_desc {
  int abi_version;          /* SCTL_AMD_DEVICE_ABI of the headers the plugin was compiled with */
  int desc_bytes;           /* sizeof(sctl_amd_kernel_desc) */
  int entry_bytes;          /* sizeof(sctl_amd::KernelEntry) */
  int src_dim, trg_dim, normal_dim, flops, ctx_bytes;
  double scale;             /* uKerScaleFactor<double>() */
  const char* name;         /* Name() */
  const void* launch_table; /* const sctl_amd::KernelEntry* */
} sctl_amd_kernel_desc;
int sctl_amd_register_kernel(const sctl_amd_kernel_desc* desc);
/* dlopen()s a plugin; its static initialisers (SCTL_AMD_REGISTER_KERNEL in device/kernel_plugin.hpp) register its kernels.
 * Returns the number of kernels the plugin added, 0 when this very object had been loaded before, or a negative error code —
 * also when the object loads but every registration in it was refused (ABI mismatch, duplicate name, incomplete table:
 * sctl_amd_last_error() carries the reason) or it registers nothing.  A C++ program may instead simply link the plugin's object file. */
int sctl_amd_load_plugin(const char* path);

/* ---- the hot path: GenericKernel::Eval ------------------------------------------------------------------- */
/* Device-resident form.  All five arrays are DEVICE pointers on the current HIP device; `stream` is a
 * hipStream_t (NULL = default stream); the call only enqueues work on that stream and returns.
 * v_trg must already hold Nt*TrgDim values and is ACCUMULATED into (generic-kernel.txx:182-186);
 * resizing-and-zeroing (generic-kernel.txx:98-101) is the host wrapper's job.
 * digits: requested decimal digits, -1 = full precision of `real` (generic-kernel.txx:46-74,77).
 * ctx/ctx_bytes: HOST pointer to the kernel's context blob, copied at launch (replaces the unsized
 * ctx_ptr of generic-kernel.hpp:90,150); NULL/0 for kernels without one. */
int sctl_amd_eval_device(int kernel, int real, int64_t Nt, int64_t Ns, const void* r_trg, const void* r_src, const void* n_src,
                         const void* v_src, void* v_trg, int digits, const void* ctx, int ctx_bytes, void* stream);

/* The same for a SPATIALLY COMPACT slab of a larger target set: the Nt targets are a contiguous run of the Nt_whole
 * targets in space-filling-curve order, as a rank of a multi-GPU job holds them (fmm-wrapper.txx:504-512 partitions the
 * targets; here the partition follows a Morton curve so that a slab keeps the point density of the whole set).  The
 * result is the same as sctl_amd_eval_device's; Nt_whole informs the choice between the exact kernel and the tile-centred
 * one, which depends on the target density and not on the count, and a proper slab (Nt_whole > Nt) is taken to be in curve order
 * already, so the tile-centred path skips its own sort.  Nt_whole >= Nt. */
int sctl_amd_eval_device_slab(int kernel, int real, int64_t Nt, int64_t Ns, int64_t Nt_whole, const void* r_trg, const void* r_src,
                              const void* n_src, const void* v_src, void* v_trg, int digits, const void* ctx, int ctx_bytes,
                              void* stream);

/* Host-buffer form: the drop-in for GenericKernel<uKer>::Eval<Real,enable_openmp,digits>
 * (generic-kernel.hpp:123) and for the type-erased static entry ParticleFMM stores
 * (generic-kernel.hpp:110, fmm-wrapper.txx:152-153).  All arrays are HOST pointers; the call uploads,
 * evaluates on `device`, downloads and accumulates into v_trg, and returns when v_trg is final. */
int sctl_amd_eval_host(int kernel, int real, int64_t Nt, int64_t Ns, const void* r_trg, const void* r_src, const void* n_src,
                       const void* v_src, void* v_trg, int digits, const void* ctx, int ctx_bytes, int device);

/* Host-buffer form over several GPUs of one node from ONE process: targets are block-partitioned,
 * GPU g of G gets [Nt*g/G, Nt*(g+1)/G) — the rank partition formula of fmm-wrapper.txx:507 — and sources are
 * replicated.  devices == NULL means devices 0..n_devices-1. */
int sctl_amd_eval_host_multi(int kernel, int real, int64_t Nt, int64_t Ns, const void* r_trg, const void* r_src, const void* n_src,
                             const void* v_src, void* v_trg, int digits, const void* ctx, int ctx_bytes, const int* devices,
                             int n_devices);

/* ---- GenericKernel::KernelMatrix (generic-kernel.txx:191-307) ---------------------------------------------- */
/* M is (Ns*SrcDim) x (Nt*TrgDim), row-major, OVERWRITTEN, scale factor included. */
int sctl_amd_kernel_matrix_device(int kernel, int real, int64_t Nt, int64_t Ns, const void* r_trg, const void* r_src,
                                  const void* n_src, void* M, int digits, const void* ctx, int ctx_bytes, void* stream);
int sctl_amd_kernel_matrix_host(int kernel, int real, int64_t Nt, int64_t Ns, const void* r_trg, const void* r_src,
                                const void* n_src, void* M, int digits, const void* ctx, int ctx_bytes, int device);

/* Many operator blocks in one launch: block b = KernelMatrix of targets [sum(Nt[:b]), +Nt[b]) of r_trg against sources
 * [sum(Ns[:b]), +Ns[b]) of r_src / n_src, stored (Ns[b]*SrcDim) x (Nt[b]*TrgDim) row-major; the blocks are concatenated in
 * M in batch order.  This is the shape of BoundaryIntegralOp::SetupNear's direct-part subtraction, which calls KernelMatrix
 * once per element on that element's near targets and far-field nodes (boundary_integral.txx:946-1009, Xtrg_near by
 * near_elem_dsp, X_far by elem_nds_dsp_far): one call here replaces that loop.  HOST arrays. */
int sctl_amd_kernel_matrix_batch_host(int kernel, int real, int64_t nbatch, const int64_t* Nt, const int64_t* Ns, const void* r_trg,
                                      const void* r_src, const void* n_src, void* M, int digits, const void* ctx, int ctx_bytes,
                                      int device);

/* ---- device-resident operator: coordinates stay on the GPUs between evaluations ------------------------------- */
/* The MI355X-first form of what ParticleFMM keeps between SetSrcCoord/SetTrgCoord and repeated Eval calls
 * (fmm-wrapper.txx:444-479: the object owns copies of X, Xn, F; boundary_integral.txx:1054,1063: an iterative solver
 * changes only the density between evaluations).  Coordinates are uploaded once — targets block-partitioned over the
 * device list with the formula of fmm-wrapper.txx:507 (with more than one device the blocks are cut from the Morton order
 * of the targets; results always come back in the caller's order), sources replicated — and every sctl_amd_op_eval moves
 * only the density down and the potential up.  A handle may be used from one thread at a time. */
typedef struct sctl_amd_op sctl_amd_op;
int sctl_amd_op_create(int kernel, int real, const int* devices, int n_devices, sctl_amd_op** op);
/* HOST arrays, copied to the devices before returning; either may be called again at any time (new coordinates). */
int sctl_amd_op_set_targets(sctl_amd_op* op, int64_t Nt, const void* r_trg);
int sctl_amd_op_set_sources(sctl_amd_op* op, int64_t Ns, const void* r_src, const void* n_src);
/* The far-field pre/post steps of BoundaryIntegralOp::ComputeFarField, kept on the device (both optional, HOST arrays, NULL clears):
 * weights[Ns]: every evaluation multiplies the uploaded density by the quadrature weights, f[s][k] *= w[s] (boundary_integral.txx:
 * 1040-1052); call after sctl_amd_op_set_sources (new sources drop the weights).  n_trg[Nt*3]: the TrgDim = 3*m output of the kernel is
 * contracted with the target normal, u[t][k] = sum_l v[t][k][l] n_trg[t][l] (:1060-1071), and sctl_amd_op_eval's v_trg then holds
 * Nt*TrgDim/3 values; call after sctl_amd_op_set_targets (new targets drop the normals). */
int sctl_amd_op_set_source_weights(sctl_amd_op* op, const void* weights);
int sctl_amd_op_set_target_normals(sctl_amd_op* op, const void* n_trg);
/* v_src[Ns*SrcDim] and v_trg[Nt*TrgDim] are HOST arrays.  accumulate != 0: v_trg += result (GenericKernel::Eval,
 * generic-kernel.txx:184); accumulate == 0: v_trg = result (ParticleFMM::EvalDirect, fmm-wrapper.txx:501-502). */
int sctl_amd_op_eval(sctl_amd_op* op, const void* v_src, void* v_trg, int accumulate, int digits, const void* ctx, int ctx_bytes);
void sctl_amd_op_destroy(sctl_amd_op* op);

/* ---- rank-parallel evaluation: one process per GPU (ParticleFMM::EvalDirect under MPI, fmm-wrapper.txx:504-561) --------------- */
/* The reference partitions targets and sources over MPI ranks and rotates the source blocks round a ring (:537-558).  Here every
 * rank keeps ITS targets, and the sources (and, per evaluation, the densities) of ALL ranks are all-gathered into each rank's
 * device-resident operator: over RCCL / xGMI, GPU buffer to GPU buffer, when every rank drives its own GPU.  There is no MPI
 * behind this: ranks meet through a TCP rendezvous (rank 0 listens on master_addr:master_port; dotted IPv4), which carries the
 * RCCL unique id, the small host-side collectives below, and — when ranks SHARE a GPU, which RCCL refuses: a one-GPU rehearsal —
 * the data itself.  create() is collective (every rank calls it; blocks until all have connected).  device: the HIP device of
 * this rank, or -1 for a host-only communicator (the host collectives work without any GPU).  flags: SCTL_AMD_COMM_SOCKETS_ONLY
 * keeps RCCL out even when it could be used. */
typedef struct sctl_amd_comm sctl_amd_comm;
enum { SCTL_AMD_COMM_SOCKETS_ONLY = 1, SCTL_AMD_COMM_FORCE_RCCL = 2 /* size == 1 only: build a one-rank RCCL communicator (for sctl_amd_comm_selftest) */ };
enum { SCTL_AMD_COMM_SOCKETS = 0, SCTL_AMD_COMM_RCCL = 1 };   /* sctl_amd_comm_info: transport of the data path */
int sctl_amd_comm_create(int rank, int size, const char* master_addr, int master_port, int device, int flags, sctl_amd_comm** comm);
int sctl_amd_comm_info(const sctl_amd_comm* comm, int* rank, int* size, int* device, int* transport);
/* Host-side collectives over the rendezvous sockets (counts, small arrays): every rank contributes send_bytes bytes, recv gets
 * the concatenation in rank order (at most recv_capacity bytes) and bytes_of_rank[size] the contributions. */
int sctl_amd_comm_allgatherv_host(sctl_amd_comm* comm, const void* send, int64_t send_bytes, void* recv, int64_t recv_capacity, int64_t* bytes_of_rank);
int sctl_amd_comm_barrier(sctl_amd_comm* comm);
/* Checks the RCCL data path: every rank sends `bytes` bytes to its right neighbour (itself, with one rank) with the grouped
 * ncclSend / ncclRecv the gathers use, and verifies what arrives.  Collective.  BAD_ARGUMENT if the data path is the sockets. */
int sctl_amd_comm_selftest(sctl_amd_comm* comm, int64_t bytes);
void sctl_amd_comm_destroy(sctl_amd_comm* comm);
/* Collective forms of sctl_amd_op_set_sources / sctl_amd_op_eval for an operator with ONE device (this rank's): the operator's
 * sources become the concatenation, in rank order, of all ranks' Ns_local sources (HOST arrays); eval_dist gathers the ranks'
 * densities the same way and evaluates THIS rank's targets (set with sctl_amd_op_set_targets), v_trg holding their Nt*TrgDim
 * values.  Every rank must call them, in the same order; a rank may own no sources or no targets. */
int sctl_amd_op_set_sources_dist(sctl_amd_op* op, sctl_amd_comm* comm, int64_t Ns_local, const void* r_src, const void* n_src);
int sctl_amd_op_eval_dist(sctl_amd_op* op, sctl_amd_comm* comm, int64_t Ns_local, const void* v_src_local, void* v_trg, int accumulate, int digits,
                          const void* ctx, int ctx_bytes);

/* ---- BoundaryIntegralOp near field: ComputeNearInterac (boundary_integral.txx:1079-1142) -------------------------- */
/* The step that follows the far field in ComputePotential (:608-614): for every element the precomputed operator block
 * K_near_ ((elem_nds_cnt[e]*src_dim) x (near_elem_cnt[e]*trg_dim), row-major) is applied to the element's density,
 * U_ = F_ . K_near_ (:1092-1102); the results are permuted by near_scatter_index (ScatterForward, :1129: out[i] =
 * in[index[i]]) and each target adds its near_trg_cnt[i] entries starting at near_trg_dsp[i] (:1131-1140).
 * create() takes exactly the HOST arrays BoundaryIntegralOp::SetupNear leaves behind (:816-1012) and keeps them on the
 * device; K_near is the concatenation of the blocks in element order (the reference's K_near with K_near_dsp, :854-857).
 * K_near_cnt may be NULL (every block present); K_near_cnt[e] == 0 marks an element without a matrix (MatrixFree, :849),
 * whose near targets receive nothing from this routine.  trg_dim is the number of potential components per target
 * (KDIM1, or KDIM1/3 when the operator was set up with trg_normal_dot_prod, :1080).
 * apply: F holds sum(elem_nds_cnt)*src_dim densities in element order; U (Ntrg*trg_dim) is ACCUMULATED into.
 * A handle may be used from one thread at a time. */
typedef struct sctl_amd_near sctl_amd_near;
int sctl_amd_near_create(int real, int device, int64_t Nelem, int src_dim, int trg_dim, const int64_t* elem_nds_cnt,
                         const int64_t* near_elem_cnt, const int64_t* K_near_cnt, const void* K_near, int64_t Ntrg,
                         const int64_t* near_scatter_index, const int64_t* near_trg_cnt, const int64_t* near_trg_dsp,
                         sctl_amd_near** op);
int sctl_amd_near_apply_host(sctl_amd_near* op, const void* F, void* U);                   /* HOST arrays            */
int sctl_amd_near_apply_device(sctl_amd_near* op, const void* F, void* U, void* stream);   /* DEVICE arrays, enqueue */
/* Sizes of an operator: density and potential lengths, near-list entries, bytes of K_near resident in HBM (the
 * algorithmic traffic of one application), workgroups of the GEMV launch.  Any output pointer may be NULL. */
/* The whole of BoundaryIntegralOp::ComputePotential (boundary_integral.txx:608-614: far field, then the near-zone correction added
 * to it) on the devices of a direct-sum operator handle.  set_near attaches the near-field operator of the same BoundaryIntegralOp —
 * the arrays of sctl_amd_near_create, for the operator's CURRENT targets (Ntrg = the Nt of sctl_amd_op_set_targets; call again
 * after new targets) — block-partitioned like the targets: device g keeps, of every element block, the columns whose targets lie
 * in its slab.  eval_potential then uploads both densities once (v_src_far at the far-field nodes, f_near at the element nodes:
 * sum(elem_nds_cnt)*SrcDim values), runs far field (+ weights, + target-normal contraction) and near field back to back on each
 * device's stream, the near field ACCUMULATING into the far-field result where it lies, and downloads the potential once.
 * trg_dim: TrgDim, or TrgDim/3 when target normals are set.  Results equal sctl_amd_op_eval followed by sctl_amd_near_apply_host up
 * to the order in which the far field and a target's near entries are added (and do not depend on the number of devices).
 * Nelem = 0 with null arrays detaches. */
int sctl_amd_op_set_near(sctl_amd_op* op, int src_dim, int trg_dim, int64_t Nelem, const int64_t* elem_nds_cnt, const int64_t* near_elem_cnt,
                         const int64_t* K_near_cnt, const void* K_near, const int64_t* near_scatter_index, const int64_t* near_trg_cnt,
                         const int64_t* near_trg_dsp);
int sctl_amd_op_eval_potential(sctl_amd_op* op, const void* v_src_far, const void* f_near, void* v_trg, int accumulate, int digits, const void* ctx,
                               int ctx_bytes);

int sctl_amd_near_info(const sctl_amd_near* op, int64_t* density_len, int64_t* potential_len, int64_t* near_entries,
                       int64_t* operator_bytes, int64_t* workgroups);
void sctl_amd_near_destroy(sctl_amd_near* op);

/* ---- batched list evaluation: many (target range x source range) direct sums in ONE launch ------------------------ */
/* The near-field (P2P, U-list) shape of a tree code — SURVEY.md §8f row 4, second half: PVFMM calls the kernel once per
 * (target box, source box) pair through pvfmm::GenericKernel<PVFMMKernelFn_<Ker>> on sub-ranges of the particle arrays
 * (fmm-wrapper.txx:756-786); boxes hold 1-500 points, so on a GPU the pairs must share a launch.  List l adds to the targets
 * [trg_off[l], trg_off[l] + trg_cnt[l]) the potential of the sources [src_off[l], src_off[l] + src_cnt[l]):
 *     v_trg[t] += scale * sum_{s in list l} U(x_t - x_s, n_s) v_src[s]         (ACCUMULATED into, like GenericKernel::Eval)
 * Offsets and counts are in POINTS.  The target ranges of any two lists must be IDENTICAL or DISJOINT (the leaf boxes of a
 * tree): every target is then owned by one wave, sums run in list order in registers and are written once — deterministic, no
 * atomics; anything else is SCTL_AMD_ERR_BAD_ARGUMENT.  Source ranges may overlap freely.  The four index arrays are HOST
 * arrays (nlists entries each), read by create() only.
 * A plan keeps the grouped, cost-ordered work list on `device`; evaluating it is one kernel launch.  eval_device: DEVICE
 * arrays on the plan's device, which must be the current device; enqueues on `stream` and returns.  eval_host: HOST arrays.
 * A handle may be used from one thread at a time. */
typedef struct sctl_amd_lists sctl_amd_lists;
int sctl_amd_lists_create(int kernel, int real, int device, int64_t nlists, const int64_t* trg_off, const int64_t* trg_cnt, const int64_t* src_off,
                          const int64_t* src_cnt, int64_t Nt, int64_t Ns, sctl_amd_lists** plan);
int sctl_amd_lists_eval_device(sctl_amd_lists* plan, const void* r_trg, const void* r_src, const void* n_src, const void* v_src, void* v_trg,
                               int digits, const void* ctx, int ctx_bytes, void* stream);
int sctl_amd_lists_eval_host(sctl_amd_lists* plan, const void* r_trg, const void* r_src, const void* n_src, const void* v_src, void* v_trg,
                             int digits, const void* ctx, int ctx_bytes);
/* pair interactions of one evaluation, work items (waves) and source ranges of the launch.  NULL = skip. */
int sctl_amd_lists_info(const sctl_amd_lists* plan, int64_t* pairs, int64_t* work_items, int64_t* source_ranges);
void sctl_amd_lists_destroy(sctl_amd_lists* plan);
/* One-shot forms (plan, evaluate, release).  _device: arrays on the current device; returns after the stream has finished. */
int sctl_amd_eval_lists_device(int kernel, int real, int64_t nlists, const int64_t* trg_off, const int64_t* trg_cnt, const int64_t* src_off,
                               const int64_t* src_cnt, int64_t Nt, int64_t Ns, const void* r_trg, const void* r_src, const void* n_src,
                               const void* v_src, void* v_trg, int digits, const void* ctx, int ctx_bytes, void* stream);
int sctl_amd_eval_lists_host(int kernel, int real, int64_t nlists, const int64_t* trg_off, const int64_t* trg_cnt, const int64_t* src_off,
                             const int64_t* src_cnt, int64_t Nt, int64_t Ns, const void* r_trg, const void* r_src, const void* n_src,
                             const void* v_src, void* v_trg, int digits, const void* ctx, int ctx_bytes, int device);

/* ---- accounting (the reference's Profile::IncrementCounter(FLOP, Ns*Nt*FLOPS()), generic-kernel.txx:188) ---- */
/* Process-wide counters, updated atomically by every eval / kernel_matrix call. */
void sctl_amd_counters(int64_t* pair_interactions, int64_t* sctl_flops);
void sctl_amd_reset_counters(void);
/* Frees the device scratch memory the library keeps per (device, stream) between calls (partial sums, the sort buffers
 * of the tile-centred path: up to ~1 GB per stream at 2^20 points) and the operators sctl_amd_eval_host_multi keeps between
 * calls (streams, device copies of the last coordinates, pinned staging; at most 8).  Waits for the devices.  Optional. */
void sctl_amd_trim(void);

/* Debugging switches (a bit mask; returns the previous mask).  SCTL_AMD_DEBUG_POISON_SCRATCH: every evaluation first fills the
 * device scratch it is about to use (partial sums, sort buffers) with NaN bit patterns, so that a read of scratch the call did not
 * write shows up as NaN in the result instead of as the previous call's numbers.  Costs one memset per call; off by default. */
#define SCTL_AMD_DEBUG_POISON_SCRATCH 1
int sctl_amd_set_debug(int flags);

/* Launch geometry chosen for a problem (for benchmarks and DESIGN.md; no side effects):
 * targets per lane, source splits, workgroups, and bytes of the partial-sum workspace.
 * Nt_whole: as in sctl_amd_eval_device_slab; 0 (or Nt) for a whole target set. */
int sctl_amd_eval_plan(int kernel, int real, int64_t Nt, int64_t Ns, int64_t Nt_whole, int digits, int* trg_per_lane,
                       int* src_splits, int64_t* workgroups, int64_t* workspace_bytes);

/* Which device algorithm sctl_amd_eval_device/_host will use for a problem at the DEFAULT accuracy: 0 = the exact all-pairs kernel
 * (d = x_t - x_s per pair, as generic-kernel.txx:83), 1 = the tile-centred path (targets Morton-sorted on the device, far sources through
 * r2 = |x_t'|^2 + |x_s'|^2 - 2 x_t'.x_s', near sources exact): Laplace3D-FxU/-DxU (f64, f32), Laplace3D-FxdU and
 * Stokes3D-FxUP (f64), Stokes3D-FxU/-FSxU/-FxUP/-DxU/-FxT and Laplace3D-FxdU/-FDxUdU (f32, at the seed's accuracy only: sctl_amd_eval_pipe answers for a given `digits`).
 * Negative = error code.  Setting SCTL_AMD_CENTERED=0 in the environment forces 0. */
int sctl_amd_eval_path(int kernel, int real, int64_t Nt, int64_t Ns, int64_t Nt_whole);

/* Which execution units the far pairs of that problem run on at `digits` (for roofline labels; no side effects): 0 = the vector pipe, exact kernel;
 * 1 = the vector pipe, tile-centred path; 2 = tile-centred path with r2 (and the double layer's / the Stokeslet's dot product) as split-bf16 contractions on
 * the MATRIX cores (v_mfma_f32_32x32x16_bf16) and v_rsq_f32 + the accumulation on the vector pipe — fp32 Laplace3D-FxU/-DxU/-FxdU/-FDxUdU and five fp32 Stokes kernels at
 * the seed's accuracy (digits < 8); SCTL_AMD_MFMA_F32=0 in the environment keeps Laplace3D-FxU/-DxU on 1 and the others on 0.  Negative = error
 * code.  (The reference has one pipe, the host's SIMD units: vec.hpp.) */
int sctl_amd_eval_pipe(int kernel, int real, int64_t Nt, int64_t Ns, int64_t Nt_whole, int digits);

#ifdef __cplusplus
}
#endif
#endif /* SCTL_AMD_H_ */
