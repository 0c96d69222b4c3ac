// Plain C++ host (no Python, no torch) that grades the C ABI against the REFERENCE'S OWN OUTPUTS.
//
// Reads tests/golden/ops_P4_2x2x2_pert_float64.bin -- the raw-binary twin (tests/golden/export_raw.py)
// of a golden case produced by running the reference's numba-cpu/{operators,precompute}.py -- uploads
// the arrays the reference consumed, and compares what libfusgpu.so computes with what the reference
// produced:
//   stiffness (plan-free, planned, in-kernel geometry)   vs ref_y_stiffness   numba-cpu/operators.py:71-227
//   cell mass (plan-free, planned)                       vs ref_y_mass        numba-cpu/operators.py:19-68
//   boundary-facet mass                                  vs ref_y_facet_mass
//   device geometry precompute                           vs ref_G, ref_detJ   numba-cpu/precompute.py:76-163
// This is the call sequence of MassSpectral3D / StiffnessSpectral3D::operator() in the reference's C++
// flavour (cpp/common/spectral_op.hpp:29-107, 132-284).  Then a halo exchange in a 1-rank world whose
// rank is its own neighbour: fus_comm_* / fus_halo_* with the RCCL transport (what
// cpp/common/Linear.hpp:120,193,196,212 would bind to).  Tolerance: rel. l2 <= 1e-12.
//
//   usage: c_abi_golden <path to .bin>          exit code 0 = all checks passed
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "fus_gpu.h"
#include "fus_gpu.hpp"

struct Entry {
  int dtype = 0;
  int64_t count = 0;
  std::vector<char> bytes;
  const double* f64() const { return reinterpret_cast<const double*>(bytes.data()); }
  const int32_t* i32() const { return reinterpret_cast<const int32_t*>(bytes.data()); }
  const int64_t* i64() const { return reinterpret_cast<const int64_t*>(bytes.data()); }
};

static bool load(const char* path, std::map<std::string, Entry>& out) {
  FILE* f = std::fopen(path, "rb");
  if (!f) return false;
  char magic[8];
  int64_t n = 0;
  bool ok = std::fread(magic, 1, 8, f) == 8 && std::memcmp(magic, "FUSGOLD1", 8) == 0 && std::fread(&n, 8, 1, f) == 1;
  for (int64_t i = 0; ok && i < n; ++i) {
    char name[33] = {0};
    int32_t hdr[2];
    Entry e;
    ok = std::fread(name, 1, 32, f) == 32 && std::fread(hdr, 4, 2, f) == 2 && std::fread(&e.count, 8, 1, f) == 1;
    if (!ok) break;
    e.dtype = hdr[0];
    const size_t nbytes = (size_t)e.count * (e.dtype == 1 ? 4 : 8);
    const size_t padded = (nbytes + 7) / 8 * 8;
    e.bytes.resize(padded);
    ok = std::fread(e.bytes.data(), 1, padded, f) == padded;
    out[name] = std::move(e);
  }
  std::fclose(f);
  return ok;
}

#define CHECK_HIP(e)                                                                            \
  do {                                                                                          \
    hipError_t err_ = (e);                                                                      \
    if (err_ != hipSuccess) {                                                                   \
      std::fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(err_), __LINE__);     \
      return 2;                                                                                 \
    }                                                                                           \
  } while (0)
#define CHECK_FUS(e)                                                                            \
  do {                                                                                          \
    int rc_ = (e);                                                                              \
    if (rc_ != FUS_OK) {                                                                        \
      std::fprintf(stderr, "fus error %d (%s; %s) at line %d\n", rc_, fus_error_string(rc_),    \
                   fus_comm_last_error(nullptr), __LINE__);                                     \
      return 3;                                                                                 \
    }                                                                                           \
  } while (0)

static void* upload(const Entry& e) {
  void* d = nullptr;
  const size_t nbytes = (size_t)e.count * (e.dtype == 1 ? 4 : 8);
  if (hipMalloc(&d, nbytes ? nbytes : 8) != hipSuccess) return nullptr;
  if (nbytes && hipMemcpy(d, e.bytes.data(), nbytes, hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  return d;
}

static double rel_l2(const std::vector<double>& got, const double* ref) {
  double num = 0, den = 0;
  for (size_t i = 0; i < got.size(); ++i) {
    num += (got[i] - ref[i]) * (got[i] - ref[i]);
    den += ref[i] * ref[i];
  }
  return std::sqrt(num / (den > 0 ? den : 1));
}

int main(int argc, char** argv) {
  if (fus_abi_version() != FUS_ABI_VERSION) {
    std::fprintf(stderr, "libfusgpu.so has ABI version %d, this host was built against %d\n", fus_abi_version(), FUS_ABI_VERSION);
    return 2;
  }
  if (argc < 2) {
    std::fprintf(stderr, "usage: %s tests/golden/ops_P4_2x2x2_pert_float64.bin\n", argv[0]);
    return 64;
  }
  std::map<std::string, Entry> g;
  if (!load(argv[1], g)) {
    std::fprintf(stderr, "cannot read %s\n", argv[1]);
    return 65;
  }
  const int P = (int)g["P"].i64()[0], n = P + 1, Nd = n * n * n;
  const int64_t ndofs = g["x"].count, ncell = g["dofmap"].count / Nd;
  const int64_t nfacet = g["bfacet_dofmap"].count / (n * n);
  std::printf("golden case: P=%d, %lld cells, %lld dofs, %lld boundary facets\n", P, (long long)ncell, (long long)ndofs,
              (long long)nfacet);

  double* d_x = (double*)upload(g["x"]);
  double* d_cc = (double*)upload(g["cell_constants"]);
  double* d_G = (double*)upload(g["ref_G"]);
  double* d_detJ = (double*)upload(g["ref_detJ"]);
  double* d_D = (double*)upload(g["dphi_1d"]);
  double* d_pts = (double*)upload(g["pts"]);
  double* d_wts = (double*)upload(g["wts"]);
  double* d_xg = (double*)upload(g["x_g"]);
  int32_t* d_xd = (int32_t*)upload(g["x_dofs"]);
  int32_t* d_dm = (int32_t*)upload(g["dofmap"]);
  double* d_fc = (double*)upload(g["facet_constants"]);
  double* d_dJf = (double*)upload(g["ref_detJ_f"]);
  int32_t* d_fdm = (int32_t*)upload(g["bfacet_dofmap"]);
  double* d_y = nullptr;
  CHECK_HIP(hipMalloc(&d_y, ndofs * sizeof(double)));
  if (!d_x || !d_cc || !d_G || !d_detJ || !d_D || !d_pts || !d_wts || !d_xg || !d_xd || !d_dm || !d_fc || !d_dJf || !d_fdm) return 2;

  const int64_t ws_bytes = fus_stiffness_plan_bytes(P, ncell);
  void* ws = nullptr;
  CHECK_HIP(hipMalloc(&ws, ws_bytes));
  CHECK_FUS(fus_stiffness_plan_build(d_dm, P, ncell, ws, ws_bytes, nullptr));

  std::vector<double> y(ndofs);
  auto reset_y = [&]() { return hipMemcpy(d_y, g["y0"].bytes.data(), ndofs * sizeof(double), hipMemcpyHostToDevice); };
  auto fetch_y = [&]() {
    if (hipDeviceSynchronize() != hipSuccess) return hipErrorUnknown;
    return hipMemcpy(y.data(), d_y, ndofs * sizeof(double), hipMemcpyDeviceToHost);
  };
  const double tol = 1e-12;
  bool ok = true;
  auto report = [&](const char* what, double err, double bar) {
    std::printf("  %-44s rel l2 = %.3e  %s\n", what, err, err < bar ? "ok" : "FAIL");
    ok = ok && err < bar;
  };

  // ---- stiffness: three entry points, one reference output
  CHECK_HIP(reset_y());
  CHECK_FUS(fus_stiffness_apply_f64(d_x, d_cc, d_y, d_G, d_dm, d_D, P, ncell, nullptr));
  CHECK_HIP(fetch_y());
  report("stiffness, plan-free", rel_l2(y, g["ref_y_stiffness"].f64()), tol);
  CHECK_HIP(reset_y());
  CHECK_FUS(fus_stiffness_apply_planned_f64(d_x, d_cc, d_y, d_G, ws, d_D, P, ncell, nullptr));
  CHECK_HIP(fetch_y());
  report("stiffness, planned", rel_l2(y, g["ref_y_stiffness"].f64()), tol);
  CHECK_HIP(reset_y());
  CHECK_FUS(fus_stiffness_apply_planned_geom_f64(d_x, d_cc, d_y, d_xg, d_xd, d_pts, d_wts, ws, d_D, P, ncell, nullptr));
  CHECK_HIP(fetch_y());
  report("stiffness, geometry formed in the kernel", rel_l2(y, g["ref_y_stiffness"].f64()), tol);

  // ---- cell mass and boundary-facet mass
  CHECK_HIP(reset_y());
  CHECK_FUS(fus_mass_apply_f64(d_x, d_cc, d_y, d_detJ, d_dm, Nd, ncell, nullptr));
  CHECK_HIP(fetch_y());
  report("cell mass, plan-free", rel_l2(y, g["ref_y_mass"].f64()), tol);
  CHECK_HIP(reset_y());
  CHECK_FUS(fus_mass_apply_planned_f64(d_x, d_cc, d_y, d_detJ, ws, Nd, fus_plan_entities_per_batch(Nd), ncell, nullptr));
  CHECK_HIP(fetch_y());
  report("cell mass, planned (shares the stiffness plan)", rel_l2(y, g["ref_y_mass"].f64()), tol);
  CHECK_HIP(reset_y());
  CHECK_FUS(fus_mass_apply_f64(d_x, d_fc, d_y, d_dJf, d_fdm, n * n, nfacet, nullptr));
  CHECK_HIP(fetch_y());
  report("boundary-facet mass", rel_l2(y, g["ref_y_facet_mass"].f64()), tol);

  // ---- device geometry precompute vs the reference's precompute.py
  {
    const int nq = Nd;
    std::vector<double> dphi(3 * (size_t)nq * 8), w3(nq);
    const double* pts = g["pts"].f64();
    const double* wts = g["wts"].f64();
    for (int q = 0; q < nq; ++q) {
      const double X[3] = {pts[q / (n * n)], pts[(q / n) % n], pts[q % n]};
      w3[q] = wts[q / (n * n)] * wts[(q / n) % n] * wts[q % n];
      for (int v = 0; v < 8; ++v) {
        const int b[3] = {v & 1, (v >> 1) & 1, (v >> 2) & 1};
        double f[3], df[3];
        for (int a = 0; a < 3; ++a) {
          f[a] = b[a] ? X[a] : 1.0 - X[a];
          df[a] = b[a] ? 1.0 : -1.0;
        }
        dphi[(0 * (size_t)nq + q) * 8 + v] = df[0] * f[1] * f[2];
        dphi[(1 * (size_t)nq + q) * 8 + v] = f[0] * df[1] * f[2];
        dphi[(2 * (size_t)nq + q) * 8 + v] = f[0] * f[1] * df[2];
      }
    }
    double *d_dphi = nullptr, *d_w3 = nullptr, *d_G2 = nullptr, *d_dJ2 = nullptr;
    CHECK_HIP(hipMalloc(&d_dphi, dphi.size() * 8));
    CHECK_HIP(hipMalloc(&d_w3, w3.size() * 8));
    CHECK_HIP(hipMalloc(&d_G2, (size_t)ncell * nq * 6 * 8));
    CHECK_HIP(hipMalloc(&d_dJ2, (size_t)ncell * nq * 8));
    CHECK_HIP(hipMemcpy(d_dphi, dphi.data(), dphi.size() * 8, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_w3, w3.data(), w3.size() * 8, hipMemcpyHostToDevice));
    CHECK_FUS(fus_geometry_factors_f64(d_xg, d_xd, d_dphi, d_w3, nq, ncell, d_G2, d_dJ2, nullptr));
    CHECK_HIP(hipDeviceSynchronize());
    std::vector<double> G2((size_t)ncell * nq * 6), dJ2((size_t)ncell * nq);
    CHECK_HIP(hipMemcpy(G2.data(), d_G2, G2.size() * 8, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(dJ2.data(), d_dJ2, dJ2.size() * 8, hipMemcpyDeviceToHost));
    report("device precompute: G", rel_l2(G2, g["ref_G"].f64()), 1e-13);
    report("device precompute: detJ", rel_l2(dJ2, g["ref_detJ"].f64()), 1e-13);
    // ---- the C++ functor twins of cpp/common/spectral_op.hpp (include/fus_gpu.hpp): constructed from the dofmap and the
    // P1 geometry (they compute G / detJ on the device and own their batch plan), applied like the reference's operator()
    if (P == 4) {
      try {
        const fus_gpu::Geometry<double> geo{d_xg, d_xd, d_dphi, d_w3};
        fus_gpu::StiffnessSpectral3D<double, 4> stiffness(d_dm, ncell, geo, d_D);
        fus_gpu::MassSpectral3D<double, 4> mass(d_dm, ncell, geo);
        CHECK_HIP(reset_y());
        stiffness(d_x, d_cc, d_y);
        CHECK_HIP(fetch_y());
        report("StiffnessSpectral3D<double,4>::operator()", rel_l2(y, g["ref_y_stiffness"].f64()), tol);
        CHECK_HIP(reset_y());
        mass(d_x, d_cc, d_y);
        CHECK_HIP(fetch_y());
        report("MassSpectral3D<double,4>::operator()", rel_l2(y, g["ref_y_mass"].f64()), tol);
        // the same functor on the atomic-free kernel (transposed dofmap, fus_mass_apply_gather_*)
        mass.enable_gather(d_dm, ndofs, nullptr, /*static_detJ=*/false);
        if (!mass.gather_enabled() || mass.static_detJ_enabled()) throw std::runtime_error("the gather plan was declined for a hexahedral dofmap");
        CHECK_HIP(reset_y());
        mass(d_x, d_cc, d_y);
        CHECK_HIP(fetch_y());
        report("MassSpectral3D<double,4>::operator(), atomic-free kernel", rel_l2(y, g["ref_y_mass"].f64()), tol);
        // ... and with its detJ (computed once in the constructor, like the reference's) streamed in row order
        fus_gpu::MassSpectral3D<double, 4> mass_s(d_dm, ncell, geo);
        mass_s.enable_gather(d_dm, ndofs);
        if (!mass_s.static_detJ_enabled()) throw std::runtime_error("the static companion was declined for a hexahedral dofmap");
        const std::vector<double> y_gather = y;
        CHECK_HIP(reset_y());
        mass_s(d_x, d_cc, d_y);
        CHECK_HIP(fetch_y());
        report("MassSpectral3D<double,4>::operator(), atomic-free kernel, detJ in row order", rel_l2(y, g["ref_y_mass"].f64()), tol);
        if (y != y_gather) throw std::runtime_error("static-detJ mass apply differs bitwise from the gather apply");
        // a CALLER-OWNED detJ is never snapshotted by default (the caller may update it between applies: ADVICE r5) -- only on request;
        // and a second enable_gather() replaces the plans of the first
        fus_gpu::MassSpectral3D<double, 4> mass_c(d_dm, ncell, mass_s.detJ());
        mass_c.enable_gather(d_dm, ndofs);
        if (!mass_c.gather_enabled() || mass_c.static_detJ_enabled()) throw std::runtime_error("a caller-owned detJ was snapshotted without being asked");
        mass_c.enable_gather(d_dm, ndofs, nullptr, fus_gpu::MassSpectral3D<double, 4>::kStaticAlways);
        if (!mass_c.static_detJ_enabled()) throw std::runtime_error("kStaticAlways did not build the static companion");
        CHECK_HIP(reset_y());
        mass_c(d_x, d_cc, d_y);
        CHECK_HIP(fetch_y());
        if (y != y_gather) throw std::runtime_error("caller-owned detJ, kStaticAlways: differs bitwise from the gather apply");
      } catch (const std::exception& e) {
        std::fprintf(stderr, "functor twins: %s\n", e.what());
        ok = false;
      }
    }
    for (void* p : {(void*)d_dphi, (void*)d_w3, (void*)d_G2, (void*)d_dJ2}) (void)hipFree(p);
  }

  // ---- halo exchange in a 1-rank world whose rank ghosts 100 of its own dofs: the RCCL transport (unique id ->
  // fus_comm_create) and the PEER transport (fus_comm_create_peer; the arena handles go through the blob round trip a
  // multi-rank host does with MPI_Allgatherv: export -> connect), with the exchanges of the PEER pass issued on the
  // communicator's own stream between fus_comm_fork and fus_comm_join, as HaloApply's concurrent schedule does
  for (int transport = 0; transport < 2; ++transport) {
    const char* tname = transport == 0 ? "RCCL send/recv to self" : "PEER transport, own arena";
    fus_comm_t comm = nullptr;
    if (transport == 0) {
      unsigned char id[FUS_UNIQUE_ID_BYTES];
      CHECK_FUS(fus_comm_unique_id(id));
      CHECK_FUS(fus_comm_create(id, 1, 0, &comm));
    } else {
      CHECK_FUS(fus_comm_create_peer(1, 0, &comm));
    }
    const int64_t N = 1000, ng = 100;
    std::vector<int64_t> o_idx(ng), g_idx(ng);
    for (int64_t i = 0; i < ng; ++i) {
      o_idx[i] = (i * 37) % ng;  // a permutation of the ghost block (37 is coprime to 100): the non-direct path
      g_idx[i] = (i * 7 + 3) % N;
    }
    const int32_t rank0 = 0;
    const int64_t size = ng;
    fus_halo_t halo = nullptr;
    CHECK_FUS(fus_halo_create(comm, 8, N, ng, 1, &rank0, &size, o_idx.data(), 1, &rank0, &size, g_idx.data(), &halo));
    if (transport == 1) {
      std::vector<char> blob((size_t)fus_halo_ipc_blob_bytes(halo));
      CHECK_FUS(fus_halo_ipc_export(halo, blob.data()));
      const void* blobs[1] = {blob.data()};  // every rank's blob, here: one rank, its own neighbour
      CHECK_FUS(fus_halo_ipc_connect(halo, 1, blobs));
    }
    std::vector<double> v(N + ng), ref;
    for (int64_t i = 0; i < N + ng; ++i) v[i] = std::sin(0.1 * i) + 2.0;
    double* d_v = nullptr;
    CHECK_HIP(hipMalloc(&d_v, v.size() * 8));
    hipStream_t s;
    CHECK_HIP(hipStreamCreate(&s));
    // forward: ghosts take their owner's value
    CHECK_HIP(hipMemcpyAsync(d_v, v.data(), v.size() * 8, hipMemcpyHostToDevice, s));
    if (transport == 0) {
      CHECK_FUS(fus_halo_forward_begin(halo, d_v, s));
      CHECK_FUS(fus_halo_forward_end(halo, d_v, s));
    } else {
      void* cs = fus_comm_stream(comm);
      CHECK_FUS(fus_comm_fork(comm, s));  // the communicator's stream after everything on s so far (the upload)
      CHECK_FUS(fus_halo_forward_begin(halo, d_v, cs));
      CHECK_FUS(fus_halo_forward_end(halo, d_v, cs));
      CHECK_FUS(fus_comm_join(comm, s));  // s after the exchange
    }
    std::vector<double> got(v.size());
    CHECK_HIP(hipMemcpyAsync(got.data(), d_v, v.size() * 8, hipMemcpyDeviceToHost, s));
    CHECK_HIP(hipStreamSynchronize(s));
    ref = v;
    for (int64_t i = 0; i < ng; ++i) ref[N + o_idx[i]] = v[g_idx[i]];
    report((std::string("halo forward (") + tname + ")").c_str(), rel_l2(got, ref.data()), 1e-15);
    // reverse: owners accumulate their ghosts' partial sums
    CHECK_HIP(hipMemcpyAsync(d_v, v.data(), v.size() * 8, hipMemcpyHostToDevice, s));
    CHECK_FUS(fus_halo_reverse(halo, d_v, s));
    CHECK_HIP(hipMemcpyAsync(got.data(), d_v, v.size() * 8, hipMemcpyDeviceToHost, s));
    CHECK_HIP(hipStreamSynchronize(s));
    ref = v;
    for (int64_t i = 0; i < ng; ++i) ref[g_idx[i]] += v[N + o_idx[i]];
    report((std::string("halo reverse (") + tname + ")").c_str(), rel_l2(got, ref.data()), 1e-15);
    if (transport == 1) {
      int64_t st[8] = {-1, -1, -1, -1, -1, -1, -1, -1}, sync_to = -1;
      CHECK_FUS(fus_halo_ipc_status(halo, st));
      CHECK_FUS(fus_comm_sync_timeouts(comm, &sync_to));
      report("PEER transport: device-side waits that timed out", (double)(st[0] + sync_to), 0.5);
    }
    CHECK_FUS(fus_halo_destroy(halo));
    CHECK_FUS(fus_comm_destroy(comm));
    (void)hipFree(d_v);
    (void)hipStreamDestroy(s);
  }

  std::printf(ok ? "C_ABI_GOLDEN_OK\n" : "C_ABI_GOLDEN_FAILED\n");
  for (void* p : {(void*)d_x, (void*)d_cc, (void*)d_G, (void*)d_detJ, (void*)d_D, (void*)d_pts, (void*)d_wts, (void*)d_xg,
                  (void*)d_xd, (void*)d_dm, (void*)d_fc, (void*)d_dJf, (void*)d_fdm, (void*)d_y, ws})
    (void)hipFree(p);
  return ok ? 0 : 1;
}
