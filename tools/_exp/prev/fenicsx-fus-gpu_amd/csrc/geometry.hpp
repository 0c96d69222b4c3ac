// Geometry precompute on the device: the per-quadrature-point scaled
// geometric factor G = w |det J| J^-T J^-1 (upper triangle) and scaled Jacobian determinant
// |det J| w for P1 (8-vertex) hexahedra, and the boundary-facet surface Jacobian.
// Same inputs, conventions and outputs as the reference's host routines
//   numba-cpu/precompute.py:115-163  compute_scaled_geometrical_factor
//   numba-cpu/precompute.py:76-112   compute_scaled_jacobian_determinant
//   numba-cpu/precompute.py:17-73    compute_boundary_facets_scaled_jacobian_determinant
// (J_[a][d] = sum_v dphi[a][q][v] x[v][d]; G indexed by reference directions).
// One thread per (cell, q); a cell's threads write its 48 n^3-byte G slab contiguously.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fus {

template <typename T>
__device__ __forceinline__ void jacobian_at(const T* __restrict__ x_g, const int32_t* __restrict__ cell_verts,
                                            const T* __restrict__ dphi, int nq, int q, T (&J)[3][3]) {
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int d = 0; d < 3; ++d) J[a][d] = T(0);
#pragma unroll
  for (int v = 0; v < 8; ++v) {
    const int64_t vid = cell_verts[v];
    const T X = x_g[3 * vid], Y = x_g[3 * vid + 1], Z = x_g[3 * vid + 2];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const T d = dphi[((int64_t)a * nq + q) * 8 + v];
      J[a][0] += d * X;
      J[a][1] += d * Y;
      J[a][2] += d * Z;
    }
  }
}

template <typename T>
__global__ void __launch_bounds__(256)
    geometry_kernel(const T* __restrict__ x_g, const int32_t* __restrict__ x_dofs, const T* __restrict__ dphi,
                    const T* __restrict__ weights, int nq, int64_t ncell, T* __restrict__ G, T* __restrict__ detJ) {
  const int64_t total = ncell * nq;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += stride) {
    const int64_t cell = idx / nq;
    const int q = (int)(idx - cell * nq);
    T J[3][3];
    jacobian_at<T>(x_g, x_dofs + cell * 8, dphi, nq, q, J);
    // adjugate: A[d][a] = cofactor => inv(J_)[d][a] = A[d][a] / det
    T A[3][3];
    A[0][0] = J[1][1] * J[2][2] - J[1][2] * J[2][1];
    A[0][1] = J[0][2] * J[2][1] - J[0][1] * J[2][2];
    A[0][2] = J[0][1] * J[1][2] - J[0][2] * J[1][1];
    A[1][0] = J[1][2] * J[2][0] - J[1][0] * J[2][2];
    A[1][1] = J[0][0] * J[2][2] - J[0][2] * J[2][0];
    A[1][2] = J[0][2] * J[1][0] - J[0][0] * J[1][2];
    A[2][0] = J[1][0] * J[2][1] - J[1][1] * J[2][0];
    A[2][1] = J[0][1] * J[2][0] - J[0][0] * J[2][1];
    A[2][2] = J[0][0] * J[1][1] - J[0][1] * J[1][0];
    const T det = J[0][0] * A[0][0] + J[0][1] * A[1][0] + J[0][2] * A[2][0];
    const T w = weights[q];
    const T sdet = (det < T(0) ? -det : det) * w;
    if (detJ) detJ[idx] = sdet;
    if (G) {
      // G_[a][b] = sum_d inv[d][a] inv[d][b] = sum_d A[d][a] A[d][b] / det^2
      const T s = sdet / (det * det);
      T* g = G + idx * 6;
      g[0] = s * (A[0][0] * A[0][0] + A[1][0] * A[1][0] + A[2][0] * A[2][0]);
      g[1] = s * (A[0][0] * A[0][1] + A[1][0] * A[1][1] + A[2][0] * A[2][1]);
      g[2] = s * (A[0][0] * A[0][2] + A[1][0] * A[1][2] + A[2][0] * A[2][2]);
      g[3] = s * (A[0][1] * A[0][1] + A[1][1] * A[1][1] + A[2][1] * A[2][1]);
      g[4] = s * (A[0][1] * A[0][2] + A[1][1] * A[1][2] + A[2][1] * A[2][2]);
      g[5] = s * (A[0][2] * A[0][2] + A[1][2] * A[1][2] + A[2][2] * A[2][2]);
    }
  }
}

// Reference facet Jacobians of the hexahedron, facet order (z=0, y=0, x=0, x=1, y=1, z=1):
// the two in-facet reference axes (numba-cpu/precompute.py:49-59).
__device__ __constant__ int kFacetAxes[6][2] = {{0, 1}, {0, 2}, {1, 2}, {1, 2}, {0, 2}, {0, 1}};

template <typename T>
__global__ void __launch_bounds__(256)
    facet_geometry_kernel(const T* __restrict__ x_g, const int32_t* __restrict__ x_dofs,
                          const int32_t* __restrict__ boundary_data, const T* __restrict__ dphi_f,
                          const T* __restrict__ weights, int nqf, int64_t nfacets, T* __restrict__ detJ_f) {
  const int64_t total = nfacets * nqf;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += stride) {
    const int64_t f = idx / nqf;
    const int q = (int)(idx - f * nqf);
    const int64_t cell = boundary_data[2 * f];
    const int lf = boundary_data[2 * f + 1];
    T J[3][3];
    jacobian_at<T>(x_g, x_dofs + cell * 8, dphi_f + (int64_t)lf * 3 * nqf * 8, nqf, q, J);
    // J_facet[:, t] = J_cell^T[:, axis_t] = J[axis_t][:]
    const int a0 = kFacetAxes[lf][0], a1 = kFacetAxes[lf][1];
    const T cx = J[a0][1] * J[a1][2] - J[a0][2] * J[a1][1];
    const T cy = J[a0][2] * J[a1][0] - J[a0][0] * J[a1][2];
    const T cz = J[a0][0] * J[a1][1] - J[a0][1] * J[a1][0];
    detJ_f[idx] = sqrt(cx * cx + cy * cy + cz * cz) * weights[q];
  }
}

template <typename T>
inline hipError_t launch_geometry(const T* x_g, const int32_t* x_dofs, const T* dphi, const T* weights, int nq,
                                  int64_t ncell, T* G, T* detJ, hipStream_t stream) {
  const int64_t total = ncell * nq;
  if (total <= 0) return hipSuccess;
  int64_t nblocks = (total + 255) / 256;
  if (nblocks > 256 * 64) nblocks = 256 * 64;
  hipLaunchKernelGGL((geometry_kernel<T>), dim3((unsigned)nblocks), dim3(256), 0, stream, x_g, x_dofs, dphi, weights, nq,
                     ncell, G, detJ);
  return hipGetLastError();
}

template <typename T>
inline hipError_t launch_facet_geometry(const T* x_g, const int32_t* x_dofs, const int32_t* boundary_data,
                                        const T* dphi_f, const T* weights, int nqf, int64_t nfacets, T* detJ_f,
                                        hipStream_t stream) {
  const int64_t total = nfacets * nqf;
  if (total <= 0) return hipSuccess;
  int64_t nblocks = (total + 255) / 256;
  if (nblocks > 256 * 64) nblocks = 256 * 64;
  hipLaunchKernelGGL((facet_geometry_kernel<T>), dim3((unsigned)nblocks), dim3(256), 0, stream, x_g, x_dofs,
                     boundary_data, dphi_f, weights, nqf, nfacets, detJ_f);
  return hipGetLastError();
}

}  // namespace fus
