// Plain C++ host using ONLY the C ABI (include/fus_gpu.h + libfusgpu.so + the HIP runtime): no
// Python, no torch.  This is what the reference's C++ flavour would do inside
// StiffnessSpectral3D<T,P>::operator() (cpp/common/spectral_op.hpp:132-284) -- see INTEGRATION.md.
//
// It assembles a small affine P = 2 box on the host (2 x 2 x 2 cells, GLL nodes and weights in closed
// form), uploads the reference-layout arrays, applies the stiffness operator through the plan-free
// and the planned entry points and the mass operator, and checks operator identities that need no
// second implementation:   K 1 = 0,   v.Ku = u.Kv,   u.Ku = |a|^2 vol for u = a.x,   sum M 1 = vol,
// planned == plan-free.  Exit code 0 on success.
//
//   hipcc --offload-arch=gfx950 -O2 -Iinclude examples/c_abi_demo.cpp -o examples/c_abi_demo \
//         -Lfenicsx-fus-gpu_amd/csrc -lfusgpu -Wl,-rpath,$PWD/fenicsx-fus-gpu_amd/csrc
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fus_gpu.h"

#define HIP_OK(e)                                                              \
  do {                                                                         \
    hipError_t err_ = (e);                                                     \
    if (err_ != hipSuccess) {                                                  \
      std::fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(err_), __LINE__); \
      return 2;                                                                \
    }                                                                          \
  } while (0)
#define FUS_CHECK(e)                                                           \
  do {                                                                         \
    int rc_ = (e);                                                             \
    if (rc_ != FUS_OK) {                                                       \
      std::fprintf(stderr, "fus error %d (%s) at line %d\n", rc_, fus_error_string(rc_), __LINE__); \
      return 3;                                                                \
    }                                                                          \
  } while (0)

template <typename T>
static T* upload(const std::vector<T>& h) {
  T* d = nullptr;
  if (hipMalloc(&d, h.size() * sizeof(T)) != hipSuccess) return nullptr;
  if (hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  return d;
}

int main() {
  if (fus_abi_version() != FUS_ABI_VERSION) {
    std::fprintf(stderr, "libfusgpu.so has ABI version %d, this host was built against %d\n", fus_abi_version(), FUS_ABI_VERSION);
    return 2;
  }
  constexpr int P = 2, n = 3, Nd = 27, N = 2;  // degree, nodes per direction, dofs per cell, cells per direction
  const double pts[n] = {0.0, 0.5, 1.0}, wts[n] = {1.0 / 6, 4.0 / 6, 1.0 / 6};
  const double D[n * n] = {-3, 4, -1, -1, 0, 1, 1, -4, 3};  // D[q][i] = l_i'(pts[q]) on [0, 1]
  const double h[3] = {0.5, 0.25, 0.2};                      // anisotropic affine cells
  const int M = P * N + 1, ncell = N * N * N, ndofs = M * M * M;

  std::vector<int32_t> dofmap(ncell * Nd);
  std::vector<double> G(ncell * Nd * 6, 0.0), detJ(ncell * Nd), cc(ncell, 1.0), coords(ndofs * 3);
  const double vol_cell = h[0] * h[1] * h[2];
  for (int cx = 0; cx < N; ++cx)
    for (int cy = 0; cy < N; ++cy)
      for (int cz = 0; cz < N; ++cz) {
        const int c = (cx * N + cy) * N + cz;
        for (int i = 0; i < n; ++i)
          for (int j = 0; j < n; ++j)
            for (int k = 0; k < n; ++k) {
              const int l = (i * n + j) * n + k;
              const int dof = ((cx * P + i) * M + (cy * P + j)) * M + (cz * P + k);
              dofmap[c * Nd + l] = dof;
              coords[3 * dof + 0] = (cx + pts[i]) * h[0];
              coords[3 * dof + 1] = (cy + pts[j]) * h[1];
              coords[3 * dof + 2] = (cz + pts[k]) * h[2];
              const double w = wts[i] * wts[j] * wts[k];
              detJ[c * Nd + l] = vol_cell * w;  // |det J| w_q
              double* g = &G[(c * Nd + l) * 6];  // w |det J| diag(1/h_a^2): (G00,G01,G02,G11,G12,G22)
              g[0] = vol_cell * w / (h[0] * h[0]);
              g[3] = vol_cell * w / (h[1] * h[1]);
              g[5] = vol_cell * w / (h[2] * h[2]);
            }
      }
  const double a[3] = {1.5, -2.0, 0.5};
  std::vector<double> ones(ndofs, 1.0), lin(ndofs), u(ndofs), v(ndofs), Dv(D, D + n * n);
  for (int d = 0; d < ndofs; ++d) {
    lin[d] = a[0] * coords[3 * d] + a[1] * coords[3 * d + 1] + a[2] * coords[3 * d + 2];
    u[d] = std::sin(1.0 + 0.37 * d);
    v[d] = std::cos(0.5 + 0.11 * d);
  }

  char name[256];
  int cus = 0, lds = 0;
  int64_t hbm = 0;
  FUS_CHECK(fus_device_info(0, name, &cus, &hbm, &lds));
  std::printf("device: %s, %d CUs, %.0f GiB, ABI v%d\n", name, cus, hbm / 1073741824.0, fus_abi_version());

  double *d_G = upload(G), *d_detJ = upload(detJ), *d_cc = upload(cc), *d_D = upload(Dv);
  int32_t* d_dm = upload(dofmap);
  double *d_x = nullptr, *d_y = nullptr;
  HIP_OK(hipMalloc(&d_x, ndofs * sizeof(double)));
  HIP_OK(hipMalloc(&d_y, ndofs * sizeof(double)));
  if (!d_G || !d_detJ || !d_cc || !d_D || !d_dm) return 2;
  const int64_t ws_bytes = fus_stiffness_plan_bytes(P, ncell);
  void* ws = nullptr;
  HIP_OK(hipMalloc(&ws, ws_bytes));
  FUS_CHECK(fus_stiffness_plan_build(d_dm, P, ncell, ws, ws_bytes, nullptr));

  auto apply = [&](const std::vector<double>& x, std::vector<double>& y, int mode) -> int {
    if (hipMemcpy(d_x, x.data(), ndofs * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return 2;
    FUS_CHECK(fus_fill_f64(0.0, d_y, ndofs, nullptr));
    if (mode == 0)
      FUS_CHECK(fus_stiffness_apply_f64(d_x, d_cc, d_y, d_G, d_dm, d_D, P, ncell, nullptr));
    else if (mode == 1)
      FUS_CHECK(fus_stiffness_apply_planned_f64(d_x, d_cc, d_y, d_G, ws, d_D, P, ncell, nullptr));
    else
      FUS_CHECK(fus_mass_apply_f64(d_x, d_cc, d_y, d_detJ, d_dm, Nd, ncell, nullptr));
    y.resize(ndofs);
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    if (hipMemcpy(y.data(), d_y, ndofs * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return 2;
    return 0;
  };
  auto dot = [&](const std::vector<double>& p, const std::vector<double>& q) {
    double s = 0;
    for (int d = 0; d < ndofs; ++d) s += p[d] * q[d];
    return s;
  };

  std::vector<double> K1, Ku, Kv, Klin, Ku_planned, M1;
  int rc;
  if ((rc = apply(ones, K1, 0)) || (rc = apply(u, Ku, 0)) || (rc = apply(v, Kv, 0)) || (rc = apply(lin, Klin, 0)) ||
      (rc = apply(u, Ku_planned, 1)) || (rc = apply(ones, M1, 2)))
    return rc;

  const double vol = 1.0 * 0.5 * 0.4;  // (N h_x)(N h_y)(N h_z)
  double maxK1 = 0, maxdiff = 0, scale = 0, summ = 0;
  for (int d = 0; d < ndofs; ++d) {
    maxK1 = std::fmax(maxK1, std::fabs(K1[d]));
    maxdiff = std::fmax(maxdiff, std::fabs(Ku[d] - Ku_planned[d]));
    scale = std::fmax(scale, std::fabs(Ku[d]));
    summ += M1[d];
  }
  const double sym = std::fabs(dot(v, Ku) - dot(u, Kv)) / std::fabs(dot(v, Ku));
  const double energy = dot(lin, Klin), exact = (a[0] * a[0] + a[1] * a[1] + a[2] * a[2]) * vol;
  std::printf("max|K 1| = %.2e   symmetry = %.2e   u.Ku = %.15g (exact %.15g)\n", maxK1, sym, energy, exact);
  std::printf("planned vs plan-free = %.2e (scale %.2e)   sum(M 1) = %.15g (volume %.15g)\n", maxdiff, scale, summ, vol);
  const bool ok = maxK1 < 1e-12 * scale && sym < 1e-13 && std::fabs(energy - exact) < 1e-12 * exact &&
                  maxdiff < 1e-13 * scale && std::fabs(summ - vol) < 1e-13;
  std::printf(ok ? "C_ABI_DEMO_OK\n" : "C_ABI_DEMO_FAILED\n");
  for (void* p : {(void*)d_G, (void*)d_detJ, (void*)d_cc, (void*)d_D, (void*)d_dm, (void*)d_x, (void*)d_y, ws}) (void)hipFree(p);
  return ok ? 0 : 1;
}
