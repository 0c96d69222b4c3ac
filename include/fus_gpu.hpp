/*
 * fus_gpu.hpp -- header-only C++ functors over the C ABI of libfusgpu.so, shaped like the reference's
 *
 *     template <typename T, int P> class MassSpectral3D       cpp/common/spectral_op.hpp:29-107
 *     template <typename T, int P> class StiffnessSpectral3D  cpp/common/spectral_op.hpp:132-284
 *         void operator()(const la::Vector<T>& x, std::span<T> coeffs, la::Vector<T>& y)      (y += A(coeffs) x)
 *
 * The reference's constructors take a dolfinx FunctionSpace and do three things with it: reorder the dofmap to the
 * tensor-product order (cpp/common/permute.hpp), tabulate the GLL rule / 1-D derivative table (basix) and precompute the
 * geometry factors (cpp/common/precompute.hpp).  dolfinx / basix are not part of this library, so the constructors here
 * take what those steps produce or consume, as DEVICE arrays: the tensor-product dofmap, the P1 geometry (x_dofs, x_g)
 * with the tabulated P1 gradients + weights (the geometry factors are then computed on the device, the twin of
 * precompute.hpp:101-213) or the factors themselves, and the 1-D derivative table.  operator() takes device pointers
 * where the reference takes la::Vector / std::span (which live on the host there).
 *
 * Every call is asynchronous on ``stream``; errors are exceptions carrying fus_error_string().  The functors own the
 * batch-plan workspace (and the geometry factors they computed); everything else is caller-owned, as in the C ABI.
 */
#ifndef FUS_GPU_HPP
#define FUS_GPU_HPP

#include <hip/hip_runtime.h>

#include <cstdint>
#include <stdexcept>
#include <string>
#include <type_traits>

#include "fus_gpu.h"

namespace fus_gpu {

inline void check(int rc, const char* what) {
  if (rc != FUS_OK) throw std::runtime_error(std::string(what) + ": " + fus_error_string(rc));
}
inline void check_hip(hipError_t e, const char* what) {
  if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
inline void check_abi() {
  if (fus_abi_version() != FUS_ABI_VERSION)
    throw std::runtime_error("libfusgpu.so ABI version " + std::to_string(fus_abi_version()) + ", header " +
                             std::to_string(FUS_ABI_VERSION));
}

template <typename T>
struct Geometry {  // what compute_scaled_geometrical_factor / _jacobian_determinant read (device arrays)
  const T* x_g;           // [nvert][3]
  const int32_t* x_dofs;  // [ncell][8], vertex v = vx + 2 vy + 4 vz
  const T* dphi_p1;       // [3][nq][8] gradients of the P1 shape functions at the quadrature points
  const T* weights;       // [nq] tensor GLL weights
};

namespace detail {
template <typename T>
struct Abi;
template <>
struct Abi<double> {
  static int geometry(const double* xg, const int32_t* xd, const double* dphi, const double* w, int nq, int64_t nc, double* G,
                      double* dJ, void* s) {
    return fus_geometry_factors_f64(xg, xd, dphi, w, nq, nc, G, dJ, s);
  }
  static int stiffness(const double* x, const double* c, double* y, const double* G, const void* ws, const double* dphi, int P,
                       int64_t nc, void* s) {
    return fus_stiffness_apply_planned_f64(x, c, y, G, ws, dphi, P, nc, s);
  }
  static int mass(const double* x, const double* c, double* y, const double* dJ, const void* ws, int N, int epb, int64_t ne,
                  void* s) {
    return fus_mass_apply_planned_f64(x, c, y, dJ, ws, N, epb, ne, s);
  }
  static int mass_gather(const double* x, const double* c, double* y, const double* dJ, const void* ws, int N, int64_t ne, void* s) {
    return fus_mass_apply_gather_f64(x, c, y, dJ, ws, N, ne, s);
  }
  static int static_build(const void* ws, const double* dJ, void* sws, int64_t bytes, void* s) {
    return fus_mass_gather_static_build_f64(ws, dJ, sws, bytes, s);
  }
  static int mass_gather_static(const double* x, const double* c, double* y, const void* ws, const void* sws, int N, int64_t ne, void* s) {
    return fus_mass_apply_gather_static_f64(x, c, y, ws, sws, N, ne, s);
  }
};
template <>
struct Abi<float> {
  static int geometry(const float* xg, const int32_t* xd, const float* dphi, const float* w, int nq, int64_t nc, float* G, float* dJ,
                      void* s) {
    return fus_geometry_factors_f32(xg, xd, dphi, w, nq, nc, G, dJ, s);
  }
  static int stiffness(const float* x, const float* c, float* y, const float* G, const void* ws, const float* dphi, int P, int64_t nc,
                       void* s) {
    return fus_stiffness_apply_planned_f32(x, c, y, G, ws, dphi, P, nc, s);
  }
  static int mass(const float* x, const float* c, float* y, const float* dJ, const void* ws, int N, int epb, int64_t ne, void* s) {
    return fus_mass_apply_planned_f32(x, c, y, dJ, ws, N, epb, ne, s);
  }
  static int mass_gather(const float* x, const float* c, float* y, const float* dJ, const void* ws, int N, int64_t ne, void* s) {
    return fus_mass_apply_gather_f32(x, c, y, dJ, ws, N, ne, s);
  }
  static int static_build(const void* ws, const float* dJ, void* sws, int64_t bytes, void* s) {
    return fus_mass_gather_static_build_f32(ws, dJ, sws, bytes, s);
  }
  static int mass_gather_static(const float* x, const float* c, float* y, const void* ws, const void* sws, int N, int64_t ne, void* s) {
    return fus_mass_apply_gather_static_f32(x, c, y, ws, sws, N, ne, s);
  }
};

// a device allocation owned by a functor: released when the functor's constructor throws after it, too (a member that is
// fully constructed is destroyed then; a raw pointer freed by the class's own destructor is not).  The stream the geometry
// kernel was enqueued on is drained first: the kernel may still be writing the buffer.
template <typename T>
struct DeviceBuffer {
  T* p = nullptr;
  hipStream_t stream = nullptr;
  DeviceBuffer() = default;
  DeviceBuffer(const DeviceBuffer&) = delete;
  DeviceBuffer& operator=(const DeviceBuffer&) = delete;
  void allocate(size_t count, hipStream_t s, const char* what) {
    stream = s;
    check_hip(hipMalloc(&p, sizeof(T) * count), what);
  }
  ~DeviceBuffer() {
    if (p) {
      (void)hipStreamSynchronize(stream);
      (void)hipFree(p);
    }
  }
};

// the batch plan of a dofmap (device workspace owned by the functor; registered with the library at this address)
struct Plan {
  void* ws = nullptr;
  int N = 0, epb = 0;
  int64_t nent = 0;
  Plan() = default;
  Plan(const Plan&) = delete;
  Plan& operator=(const Plan&) = delete;
  void build(const int32_t* dofmap, int ndof_per_entity, int64_t n_entities, hipStream_t stream) {
    N = ndof_per_entity;
    nent = n_entities;
    epb = fus_plan_entities_per_batch(N);
    check(epb < 0 ? epb : FUS_OK, "fus_plan_entities_per_batch");
    const int64_t bytes = fus_plan_bytes(N, epb, nent);
    check(bytes < 0 ? (int)bytes : FUS_OK, "fus_plan_bytes");
    check_hip(hipMalloc(&ws, (size_t)bytes), "hipMalloc(plan workspace)");
    check(fus_plan_build(dofmap, N, epb, nent, ws, bytes, stream), "fus_plan_build");
  }
  ~Plan() {
    if (ws) {
      (void)fus_plan_release(ws);
      (void)hipFree(ws);
    }
  }
};
// the transposed dofmap of the atomic-free mass apply (fus_mass_gather_plan_build); ``ok`` stays false when the library
// declines the dofmap (a dof in more than 255 entities): the functor then keeps to the atomic kernel
struct GatherPlan {
  void* ws = nullptr;
  bool ok = false;
  GatherPlan() = default;
  GatherPlan(const GatherPlan&) = delete;
  GatherPlan& operator=(const GatherPlan&) = delete;
  void release() {
    if (ws) {
      (void)fus_plan_release(ws);
      (void)hipFree(ws);
    }
    ws = nullptr;
    ok = false;
  }
  void build(const int32_t* dofmap, int ndof_per_entity, int64_t n_entities, int64_t ndofs, hipStream_t stream) {
    release();  // a second enable_gather() replaces the first plan instead of leaking it
    const int64_t bytes = fus_mass_gather_plan_bytes(ndof_per_entity, n_entities, ndofs);
    if (bytes < 0) return;  // 2^31 entries or more: atomic kernel
    check_hip(hipMalloc(&ws, (size_t)bytes), "hipMalloc(gather plan workspace)");
    const int rc = fus_mass_gather_plan_build(dofmap, ndof_per_entity, n_entities, ndofs, ws, bytes, stream);
    if (rc == FUS_ERR_UNSUPPORTED_ENTITY) return;
    check(rc, "fus_mass_gather_plan_build");
    ok = true;
  }
  ~GatherPlan() { release(); }
};
// the static companion of a transposed dofmap (fus_mass_gather_static_build): detJ in row order; ``ok`` stays false when the
// library declines (a block of 256 dofs spanning more than 65 535 entities): the functor then gathers detJ as before
struct GatherStaticPlan {
  void* ws = nullptr;
  bool ok = false;
  GatherStaticPlan() = default;
  GatherStaticPlan(const GatherStaticPlan&) = delete;
  GatherStaticPlan& operator=(const GatherStaticPlan&) = delete;
  void release() {
    if (ws) {
      (void)fus_plan_release(ws);
      (void)hipFree(ws);
    }
    ws = nullptr;
    ok = false;
  }
  template <typename T>
  void build(const void* gather_ws, const T* detJ, int ndof_per_entity, int64_t n_entities, hipStream_t stream) {
    release();
    const int64_t bytes = fus_mass_gather_static_bytes(ndof_per_entity, n_entities, (int)sizeof(T));
    if (bytes < 0) return;
    check_hip(hipMalloc(&ws, (size_t)bytes), "hipMalloc(static companion of the gather plan)");
    const int rc = Abi<T>::static_build(gather_ws, detJ, ws, bytes, stream);
    if (rc == FUS_ERR_UNSUPPORTED_ENTITY) return;
    check(rc, "fus_mass_gather_static_build");
    ok = true;
  }
  ~GatherStaticPlan() { release(); }
};
}  // namespace detail

/// y += M(coeffs) x, M the collocated (diagonal) GLL mass operator; cpp/common/spectral_op.hpp:29-107
template <typename T, int P>
class MassSpectral3D {
  static_assert(std::is_same<T, double>::value || std::is_same<T, float>::value, "T: float or double");
  static_assert(P >= FUS_MIN_DEGREE && P <= FUS_MAX_DEGREE, "degree out of range");

public:
  static constexpr int Nd = (P + 1) * (P + 1) * (P + 1);
  /// dofmap: device int32[ncells][(P+1)^3], tensor-product local order; the scaled Jacobian determinant is computed here
  MassSpectral3D(const int32_t* dofmap, int64_t ncells, const Geometry<T>& geo, hipStream_t stream = nullptr) : Nc(ncells) {
    check_abi();  // before anything is allocated
    detJ_own_.allocate((size_t)Nc * Nd, stream, "hipMalloc(detJ)");
    check(detail::Abi<T>::geometry(geo.x_g, geo.x_dofs, geo.dphi_p1, geo.weights, Nd, Nc, nullptr, detJ_own_.p, stream),
          "fus_geometry_factors (detJ)");
    detJ_ = detJ_own_.p;
    plan_.build(dofmap, Nd, Nc, stream);
  }
  /// with the factors the caller already has (device T[ncells][(P+1)^3])
  MassSpectral3D(const int32_t* dofmap, int64_t ncells, const T* detJ, hipStream_t stream = nullptr) : Nc(ncells), detJ_(detJ) {
    check_abi();
    plan_.build(dofmap, Nd, Nc, stream);
  }
  MassSpectral3D(const MassSpectral3D&) = delete;
  MassSpectral3D& operator=(const MassSpectral3D&) = delete;
  /// Opt in to the atomic-free kernel (one thread per dof over the transposed dofmap: no float atomics, bitwise
  /// reproducible, 0.091 against 0.133 ms at P = 4 / 10 M dofs): ``ndofs`` = length of the vectors the operator is applied to
  /// (every dofmap value < ndofs).  A launch then assumes that nothing else adds into y while it runs (other launches of the
  /// same stream are fine); ``apply_atomic`` stays safe next to concurrent writers (a halo receive, another stream).
  /// ``static_detJ``: detJ is ALSO kept in ROW order and the kernel streams that copy instead of gathering detJ through the
  /// transposed dofmap (0.080 against 0.091 ms; bitwise the same result) -- a snapshot that is never re-read.  kStaticAuto
  /// (default): only where the functor OWNS detJ (the constructor that computes it once, as the reference's does,
  /// cpp/common/spectral_op.hpp:60-66: nobody else can change it); a caller-owned detJ is gathered live unless the caller
  /// promises with kStaticAlways (or ``true``) that it stays constant; kStaticNever (or ``false``): never.  Calling
  /// enable_gather() again rebuilds both plans (and re-takes the snapshot).
  static constexpr int kStaticAuto = -1, kStaticNever = 0, kStaticAlways = 1;
  void enable_gather(const int32_t* dofmap, int64_t ndofs, hipStream_t stream = nullptr, int static_detJ = kStaticAuto) {
    static_.release();  // (belongs to the gather plan that is about to be replaced)
    gather_.build(dofmap, Nd, Nc, ndofs, stream);
    const bool snapshot = static_detJ == kStaticAuto ? detJ_own_.p != nullptr : static_detJ != kStaticNever;
    if (gather_.ok && snapshot) static_.template build<T>(gather_.ws, detJ_, Nd, Nc, stream);
  }
  bool gather_enabled() const { return gather_.ok; }
  bool static_detJ_enabled() const { return static_.ok; }
  /// y += M x   (x, y: device vectors of nlocal + nghost entries; coeffs: device T[ncells])
  void operator()(const T* x, const T* coeffs, T* y, hipStream_t stream = nullptr) const {
    if (static_.ok)
      check(detail::Abi<T>::mass_gather_static(x, coeffs, y, gather_.ws, static_.ws, Nd, Nc, stream), "fus_mass_apply_gather_static");
    else if (gather_.ok)
      check(detail::Abi<T>::mass_gather(x, coeffs, y, detJ_, gather_.ws, Nd, Nc, stream), "fus_mass_apply_gather");
    else
      apply_atomic(x, coeffs, y, stream);
  }
  void apply_atomic(const T* x, const T* coeffs, T* y, hipStream_t stream = nullptr) const {
    check(detail::Abi<T>::mass(x, coeffs, y, detJ_, plan_.ws, Nd, plan_.epb, Nc, stream), "fus_mass_apply_planned");
  }
  const T* detJ() const { return detJ_; }

private:
  int64_t Nc;
  const T* detJ_ = nullptr;
  detail::DeviceBuffer<T> detJ_own_;  // members are released in reverse order, also when the constructor throws
  detail::Plan plan_;
  detail::GatherPlan gather_;
  detail::GatherStaticPlan static_;  // (declared after gather_: released before the plan it belongs to)
};

/// y += K(coeffs) x, the sum-factorised stiffness operator; cpp/common/spectral_op.hpp:132-284
template <typename T, int P>
class StiffnessSpectral3D {
  static_assert(std::is_same<T, double>::value || std::is_same<T, float>::value, "T: float or double");
  static_assert(P >= FUS_MIN_DEGREE && P <= FUS_MAX_DEGREE, "degree out of range");

public:
  static constexpr int Nd = (P + 1) * (P + 1) * (P + 1);
  /// dphi: device T[(P+1)][(P+1)] 1-D GLL derivative table [q][i]; G is computed here (device twin of precompute.hpp:101-213)
  StiffnessSpectral3D(const int32_t* dofmap, int64_t ncells, const Geometry<T>& geo, const T* dphi, hipStream_t stream = nullptr)
      : Nc(ncells), dphi_(dphi) {
    check_abi();  // before anything is allocated
    G_own_.allocate((size_t)Nc * Nd * 6, stream, "hipMalloc(G)");
    check(detail::Abi<T>::geometry(geo.x_g, geo.x_dofs, geo.dphi_p1, geo.weights, Nd, Nc, G_own_.p, nullptr, stream),
          "fus_geometry_factors (G)");
    G_ = G_own_.p;
    plan_.build(dofmap, Nd, Nc, stream);
  }
  /// with the factors the caller already has (device T[ncells][(P+1)^3][6])
  StiffnessSpectral3D(const int32_t* dofmap, int64_t ncells, const T* G, const T* dphi, hipStream_t stream = nullptr)
      : Nc(ncells), G_(G), dphi_(dphi) {
    check_abi();
    plan_.build(dofmap, Nd, Nc, stream);
  }
  StiffnessSpectral3D(const StiffnessSpectral3D&) = delete;
  StiffnessSpectral3D& operator=(const StiffnessSpectral3D&) = delete;
  /// y += K x
  void operator()(const T* x, const T* coeffs, T* y, hipStream_t stream = nullptr) const {
    check(detail::Abi<T>::stiffness(x, coeffs, y, G_, plan_.ws, dphi_, P, Nc, stream), "fus_stiffness_apply_planned");
  }
  const T* G() const { return G_; }

private:
  int64_t Nc;
  const T* G_ = nullptr;
  detail::DeviceBuffer<T> G_own_;
  const T* dphi_ = nullptr;
  detail::Plan plan_;
};

}  // namespace fus_gpu

#endif /* FUS_GPU_HPP */
