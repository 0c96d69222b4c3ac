/*
 * fus_gpu.h -- C ABI of libfusgpu.so: MI355X (gfx950) matrix-free operator-application path
 * for the FEniCSx-FUS acoustic wave solver.
 *
 * The reference has no FFI of its own: its "boundary" is the numba JIT call of Python closures.
 * Every entry point below names the reference interface it replaces (file:line, relative to the
 * reference checkout).  All pointers are DEVICE pointers to ordinary (coarse-grained) device
 * allocations -- hipMalloc / torch tensors; the scatter-adds use the hardware floating-point atomics
 * (global_atomic_add_f64/f32), which are not defined on fine-grained or host-mapped memory.  All
 * buffers are caller-owned; nothing is allocated, freed or synchronised per call; every launch is asynchronous
 * on ``stream`` (a hipStream_t passed as void*, NULL = the default stream).
 *
 * Return value: FUS_OK (0) or a negative error code; fus_error_string() describes it.
 * Layouts are exactly the reference's:
 *   x, y            T[nlocal + nghost]                  dof vectors (owned first, then ghosts)
 *   cell_constants  T[ncell]
 *   G               T[ncell][n^3][6]  C-contiguous      (G00,G01,G02,G11,G12,G22) * w * |detJ|
 *   detJ            T[nent][N]
 *   dofmap          int32[ncell][n^3]                   tensor-product local order l = i n^2 + j n + k
 *   dphi            T[n][n]  ([q][i], row-major; the flat and the 2-D form are the same bytes)
 * with n = P + 1.
 */
#ifndef FUS_GPU_H
#define FUS_GPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FUS_OK 0
#define FUS_ERR_INVALID_ARGUMENT (-1) /* null pointer, negative size, misaligned buffer */
#define FUS_ERR_UNSUPPORTED_DEGREE (-2) /* P outside [FUS_MIN_DEGREE, FUS_MAX_DEGREE] */
#define FUS_ERR_UNSUPPORTED_ENTITY (-3)
#define FUS_ERR_NO_DEVICE (-4)
#define FUS_ERR_PLAN_MISMATCH (-6) /* workspace was not built (through this library, at this address) for this shape */
#define FUS_ERR_COMM (-5) /* RCCL / communicator failure: fus_comm_last_error() has the text */
#define FUS_ERR_HIP_BASE (-1000) /* -(1000 + hipError_t) for launch/runtime failures */

#define FUS_MIN_DEGREE 1
#define FUS_MAX_DEGREE 10 /* the reference's quadrature-degree map covers P = 2..10 */

/*
 * ABI version, bumped on every incompatible change (a host checks fus_abi_version() == FUS_ABI_VERSION at load):
 *   1  rounds 1-2
 *   2  FUS_TUNE_PLAN_THREADS removed, FUS_TUNE_PLAN_RUNS is an apply-time setting, the last argument of
 *      fus_rk4_stage_* is a 4-valued stage kind, planned applies return FUS_ERR_PLAN_MISMATCH for a workspace that
 *      was not built and registered through this library at that address, fus_comm_destroy refuses while halo
 *      objects of the communicator are alive; new: the PEER halo transport (fus_comm_create_peer, fus_halo_ipc_*).
 *   3  fus_halo_ipc_status fills EIGHT words (time-outs and poisoned flags separately); a failed PEER exchange poisons the
 *      flags it publishes, so its neighbours fail too instead of consuming stale data; fus_comm_fork / fus_comm_join
 *      enforce their one-caller-stream contract (FUS_ERR_INVALID_ARGUMENT); new: fus_comm_fork_lazy, fus_comm_arm_join,
 *      fus_comm_health; the PEER blob identifies the exporting process by a random token and its device by PCI bus id.
 *      Added since without a bump (new symbols only): fus_mass_gather_plan_bytes / _build / _info, fus_mass_apply_gather_*,
 *      fus_mass_gather_plan_build_rows, fus_mass_gather_static_bytes / _build_* , fus_mass_apply_gather_static_*.
 * There are deliberately NO fus_cpu_* twins of the entry points (SURVEY.md 8b proposed them): a CPU path inside the
 * product would be a silent fallback; the CPU restatement of the reference is test infrastructure and lives outside the product tree.
 */
#define FUS_ABI_VERSION 3
/* Library / device queries. */
int fus_abi_version(void);
/* First 16 hex digits of the SHA-256 of the sources (the .hip and .hpp files of csrc/ and this header, concatenated in sorted path order)
 * the library was built from; "unknown" for a build outside csrc/Makefile. */
const char* fus_source_hash(void);
const char* fus_error_string(int code);
/* name: >= 256 bytes or NULL; returns FUS_OK or FUS_ERR_NO_DEVICE. */
int fus_device_info(int device, char* name, int* compute_units, int64_t* hbm_bytes, int* lds_bytes_per_cu);

/* Tuning knobs (process-global, not part of the numerical contract).  Keys: */
#define FUS_TUNE_STIFFNESS_VARIANT 1 /* 0 = default; see DESIGN.md for the variants */
#define FUS_TUNE_XCD_REMAP 2         /* 1 = give each XCD a contiguous range of cell batches */
#define FUS_TUNE_MASS_VARIANT 3       /* rows per thread of the atomic-free mass apply (fus_mass_apply_gather_*): 1, 2, 4; 0 (default) = by size and type */
#define FUS_TUNE_PLAN_VARIANT 4      /* planned stiffness kernel build: see csrc/fus_gpu.hip */
#define FUS_TUNE_PLAN_RUNS 5         /* which encoding of a plan's dof lists the apply kernels read: 0 the lists, 2 the run tables, 1 auto (default: fp64 run tables; fp32 run tables up to 125 dofs per entity, lists above); 0 at plan build = no run tables are built */
#define FUS_TUNE_VECTOR_STREAM 6      /* non-temporal accesses in the streaming vector kernels (fus_axpy ... fus_rk4_stage_*): 0 never, 1 auto (default: non-temporal loads and stores for operands > 24 MB), 2 always, 3 / 4 the same with non-temporal stores only -- a plain store leaves its line dirty in the memory-side cache, to be written back while the NEXT kernel runs (csrc/vecops.hpp) */
int fus_set_tuning(int key, int value);
int fus_get_tuning(int key);

/*
 * Stiffness operator apply   y += sum_cells P_c^T D^T ( c_c G_c (D x_c) )
 * replaces  numba-cpu/operators.py:71-227  stiffness_operator(P, dphi, float_type) -> operator(x, cell_constants, y, G, dofmap)
 *      and  cuda/operators.py:73-192       stiffness_operator(P, float_type) -> operator[grid, block](x, consts, y, G, dofmap, dphi)
 * y is accumulated into (caller zeroes it).  Out-of-range dofmap entries are undefined behaviour, as
 * in the reference (no bounds checks under njit / CUDA).
 */
int fus_stiffness_apply_f64(const double* x, const double* cell_constants, double* y, const double* G,
                            const int32_t* dofmap, const double* dphi, int P, int64_t ncell, void* stream);
int fus_stiffness_apply_f32(const float* x, const float* cell_constants, float* y, const float* G,
                            const int32_t* dofmap, const float* dphi, int P, int64_t ncell, void* stream);

/*
 * Batch plan for the gather/scatter side (optional fast path; same numerical contract).
 * Built once per dofmap on the device into a caller-owned workspace; see csrc/plan.hpp for the
 * layout.  Workspace contract: a planned apply accepts only a workspace that was BUILT through this library AT THAT
 * ADDRESS (host-side registry; anything else is FUS_ERR_PLAN_MISMATCH): do not copy or relocate a built workspace,
 * and call fus_plan_release(workspace) before freeing or reusing its memory.  The reference has no counterpart: its CUDA kernel issues one atomic per (cell, dof)
 * (cuda/operators.py:190).  The planned apply reads the plan INSTEAD of ``dofmap``.
 *   fus_stiffness_plan_bytes : workspace size in bytes for (P, ncell), or a negative error code
 *   fus_stiffness_plan_build : fill ``workspace`` (256-byte aligned) from ``dofmap``; asynchronous
 *   fus_stiffness_apply_planned_* : y += K x using a workspace built for the same (P, ncell) dofmap
 */
int64_t fus_stiffness_plan_bytes(int P, int64_t ncell);
int fus_stiffness_plan_build(const int32_t* dofmap, int P, int64_t ncell, void* workspace, int64_t workspace_bytes,
                             void* stream);
int fus_stiffness_apply_planned_f64(const double* x, const double* cell_constants, double* y, const double* G,
                                    const void* workspace, const double* dphi, int P, int64_t ncell, void* stream);
int fus_stiffness_apply_planned_f32(const float* x, const float* cell_constants, float* y, const float* G,
                                    const void* workspace, const float* dphi, int P, int64_t ncell, void* stream);

/*
 * Mass operator apply (cells: N = n^3; boundary facets: N = n^2; any N >= 1)
 *   y[dofmap[e][i]] += x[dofmap[e][i]] * detJ[e][i] * entity_constants[e]
 * replaces  numba-cpu/operators.py:19-68  mass_operator(N, float_type) -> operator(x, entity_constants, y, entity_detJ, entity_dofmap)
 *      and  cuda/operators.py:18-70       mass_operator[grid, block](x, entity_constants, y, detJ_entity, entity_dofmap)
 */
int fus_mass_apply_f64(const double* x, const double* entity_constants, double* y, const double* entity_detJ,
                       const int32_t* entity_dofmap, int ndof_per_entity, int64_t nent, void* stream);
int fus_mass_apply_f32(const float* x, const float* entity_constants, float* y, const float* entity_detJ,
                       const int32_t* entity_dofmap, int ndof_per_entity, int64_t nent, void* stream);

/*
 * The boundary-facet mass terms of one RK4 stage in one launch (ndof_per_entity = n^2 for both sets):
 *   set A (source facets, x = 1):  y[dmA[e][i]] += (sA1 cA1[e] + sA2 cA2[e]) detJA[e][i]      cA2 may be NULL
 *   set B (absorbing facets):      y[dmB[e][i]] += xB[dmB[e][i]] cB[e] detJB[e][i]
 * replaces  mass_operator[..](g, facet_coeff1, b, detJ_f1, dofmap_f1) (+ the dg term of the Westervelt solver)
 *      and  mass_operator[..](v_n, facet_coeff2, b, detJ_f2, dofmap_f2) with g, dg filled into vectors:
 *           cuda/demo_linear_box.py:510-530,546-549; cuda/demo_nonlinear_bowl.py:560-595,633-641
 */
int fus_facet_terms_f64(double* y, const double* cA1, double sA1, const double* cA2, double sA2, const double* detJA,
                        const int32_t* dmA, int64_t nentA, const double* xB, const double* cB, const double* detJB,
                        const int32_t* dmB, int64_t nentB, int ndof_per_entity, void* stream);
int fus_facet_terms_f32(float* y, const float* cA1, float sA1, const float* cA2, float sA2, const float* detJA,
                        const int32_t* dmA, int64_t nentA, const float* xB, const float* cB, const float* detJB,
                        const int32_t* dmB, int64_t nentB, int ndof_per_entity, void* stream);
/*
 * The same launch with (sA1, sA2) = scalars[0], scalars[1] read from DEVICE memory: every argument of a time
 * step is then a fixed pointer or a constant, so the launches of a step can be captured once in a hipGraph
 * (hipStreamBeginCapture on ``stream``; all entry points only enqueue on ``stream``) and replayed with new
 * source values g(t), dg/dt written to ``scalars`` -- for meshes small enough to be launch-bound.
 */
int fus_facet_terms_dev_f64(double* y, const double* cA1, const double* cA2, const double* scalars, const double* detJA,
                            const int32_t* dmA, int64_t nentA, const double* xB, const double* cB, const double* detJB,
                            const int32_t* dmB, int64_t nentB, int ndof_per_entity, void* stream);
int fus_facet_terms_dev_f32(float* y, const float* cA1, const float* cA2, const float* scalars, const float* detJA,
                            const int32_t* dmA, int64_t nentA, const float* xB, const float* cB, const float* detJB,
                            const int32_t* dmB, int64_t nentB, int ndof_per_entity, void* stream);

/*
 * Opt-in fast path for AFFINE cells (reported separately from the headline, whose
 * bytes contract is the general per-quadrature-point G): on an affine cell
 * G[c][q] = G[c][0] * (w_q / w_0), so the apply reads only the first 6-value record of each cell of
 * the SAME ``G`` array the reference builds, plus ``wratio`` = T[n^3] = w_q / w_0 (tensor GLL weights).
 * The caller asserts affinity (every box mesh of the reference's demos is affine);
 * results equal the general path up to round-off.
 */
int fus_stiffness_apply_planned_affine_f64(const double* x, const double* cell_constants, double* y, const double* G,
                                           const double* wratio, const void* workspace, const double* dphi, int P,
                                           int64_t ncell, void* stream);
int fus_stiffness_apply_planned_affine_f32(const float* x, const float* cell_constants, float* y, const float* G,
                                           const float* wratio, const void* workspace, const float* dphi, int P,
                                           int64_t ncell, void* stream);

/*
 * Stiffness apply with the geometric factor formed in the kernel from the 8 vertices of each cell
 * (its own bytes contract -- no G array is read -- and its own bench line).  The
 * formulas are the reference's host precompute, numba-cpu/precompute.py:115-163
 * compute_scaled_geometrical_factor, for what its callers pass: P1 (trilinear) hexahedra with
 * vertex v = vx + 2 vy + 4 vz and the tensor GLL rule:
 *   x_g T[nvert][3], x_dofs int32[ncell][8]  (the reference's ``(x_dofs, x_g)`` mesh pair, same cell order as the dofmap),
 *   pts T[n], wts T[n]                       1-D GLL points on [0, 1] and weights (quadrature point q = qx n^2 + qy n + qz)
 * Results equal fus_stiffness_apply_planned_* on G = compute_scaled_geometrical_factor(...) up to round-off.
 */
int fus_stiffness_apply_planned_geom_f64(const double* x, const double* cell_constants, double* y, const double* x_g,
                                         const int32_t* x_dofs, const double* pts, const double* wts,
                                         const void* workspace, const double* dphi, int P, int64_t ncell, void* stream);
int fus_stiffness_apply_planned_geom_f32(const float* x, const float* cell_constants, float* y, const float* x_g,
                                         const int32_t* x_dofs, const float* pts, const float* wts,
                                         const void* workspace, const float* dphi, int P, int64_t ncell, void* stream);

/*
 * Generic batch plan (any entity kind: cells N = n^3, boundary facets N = n^2) and planned mass
 * apply.  fus_plan_entities_per_batch(N) returns the preferred batch size (for cells it equals
 * the stiffness plan's, so ONE workspace built from the cell dofmap serves both operators:
 * fus_stiffness_plan_build(dofmap, P, ...) == fus_plan_build(dofmap, n^3, fus_plan_entities_per_batch(n^3), ...)).
 *
 * SYNCHRONISATION: fus_plan_build / fus_plan_build_ordered / fus_stiffness_plan_build enqueue the build kernels on ``stream`` and then
 * BLOCK THE HOST until they have run (one 8-byte device-to-host copy + hipStreamSynchronize: the number of batches that carry a run
 * table decides which list encoding the applies read).  They are set-up calls: not stream-asynchronous, and NOT legal inside a hipGraph
 * stream capture (they return a HIP error there) -- build every plan before capturing (the Python side's StepGraphMixin warms its
 * plans first).
 */
int fus_plan_entities_per_batch(int ndof_per_entity);
int64_t fus_plan_bytes(int ndof_per_entity, int entities_per_batch, int64_t nent);
int fus_plan_build(const int32_t* entity_dofmap, int ndof_per_entity, int entities_per_batch, int64_t nent,
                   void* workspace, int64_t workspace_bytes, void* stream);
/*
 * The same with an entity order (device array int32[nent], a permutation; NULL = natural order): batch b
 * then holds the entities entity_order[b * entities_per_batch ...].  The order is copied into the
 * workspace; the planned apply kernels index G / detJ / the constants through it, so a mesh whose cell
 * order has no locality (random cell order: 0.243 -> 0.365 ms at P = 4, 10 M dofs,
 * profiles/r02c_numbering.log) gets it back at set-up time without moving any array (cells sorted by
 * their smallest dof: 0.243 ms).  No reference counterpart (its kernel has no batches).
 *
 * Every planned apply entry point checks that its workspace was built through one of the *_plan_build
 * calls, at this address, for the (ndof_per_entity, entities_per_batch, nent) it is called with, and
 * returns FUS_ERR_PLAN_MISMATCH otherwise.
 */
int fus_plan_build_ordered(const int32_t* entity_dofmap, const int32_t* entity_order, int ndof_per_entity,
                           int entities_per_batch, int64_t nent, void* workspace, int64_t workspace_bytes,
                           void* stream);
/*
 * Optional second pass over a BUILT plan: mark the dofs that exactly one batch of the plan touches.  The planned mass apply
 * then finishes the sums of marked dofs with a plain load + store instead of a memory-side float atomic (the low-intensity
 * kernels are bound by the chip's float-atomic request rate, not by HBM: csrc/plan.hpp).
 *   dof_use_count  device int32[ndofs]; on entry what ELSE adds into each dof while a launch with this plan runs (0
 *                  everywhere for a launch that runs alone; >= 1 for dofs that another launch on another stream, or a
 *                  halo receive, adds into concurrently); on exit increased by this plan's (batch, dof) incidences.
 * A dof is marked iff its count is exactly 1 afterwards.  THE CALLER GUARANTEES that, while a launch with a marked plan
 * runs, nothing he did not declare adds into the marked dofs of y (consecutive launches on ONE stream are always fine).
 * No reference counterpart (its kernels issue one atomic per (entity, dof), cuda/operators.py:66-70).
 */
int fus_plan_mark_exclusive(void* workspace, int ndof_per_entity, int entities_per_batch, int64_t nent,
                            int32_t* dof_use_count, int64_t ndofs, void* stream);
/* Forget a workspace (before freeing it): drops its entry of the registry the planned applies check. */
int fus_plan_release(const void* workspace);
/* What a built plan holds: its number of batches, how many of them carry a run-length table of their distinct dofs (the
 * others kept the raw list: more than 128 runs, or no gain), and whether the planned applies will read the run tables
 * (FUS_TUNE_PLAN_RUNS auto: only for plans where at least half of the batches carry one -- a numbering whose lists do not
 * compress is read through the lists).  FUS_ERR_PLAN_MISMATCH for a workspace fus_plan_build* has not seen.  No reference
 * counterpart (its kernels read the dofmap itself, cuda/operators.py:108-125). */
int fus_plan_encoding(const void* workspace, int64_t* batches, int64_t* batches_with_runs, int* reads_runs_f64, int* reads_runs_f32);
int fus_mass_apply_planned_f64(const double* x, const double* entity_constants, double* y, const double* entity_detJ,
                               const void* workspace, int ndof_per_entity, int entities_per_batch, int64_t nent,
                               void* stream);
int fus_mass_apply_planned_f32(const float* x, const float* entity_constants, float* y, const float* entity_detJ,
                               const void* workspace, int ndof_per_entity, int entities_per_batch, int64_t nent,
                               void* stream);

/*
 * Atomic-free mass apply over the TRANSPOSED dofmap (csrc/mass_gather.hpp): one thread per touched dof sums its entries
 * (entity, local index) in ascending order -- the order of the reference's serial loop (numba-cpu/operators.py:60-68) -- and
 * finishes y[dof] with one plain load + store.  Same result as fus_mass_apply_* up to the summation order of the atomics
 * there; bitwise reproducible from run to run; same algorithmic bytes (detJ once, 4 index bytes per entry, x once, y
 * read-modify-write once) + 1 byte per dof.  The plan is built on the device (radix sort of the entries by dof):
 *   ndofs   length of the dof vectors x / y; every dofmap value must lie in [0, ndofs)
 *   returns FUS_ERR_UNSUPPORTED_ENTITY for a dofmap value out of range or a dof with more than 255 entries (use
 *   fus_mass_apply_planned_* / fus_mass_apply_* then), FUS_ERR_INVALID_ARGUMENT for nent * ndof_per_entity >= 2^31.
 * fus_mass_gather_plan_info: out4 = {touched dofs, 1 if they are exactly 0 .. touched - 1, largest number of entries of a dof,
 * workspace bytes}.  fus_plan_release forgets the workspace.  No reference counterpart (its kernels issue one atomic per
 * (entity, dof), cuda/operators.py:66-70).
 */
int64_t fus_mass_gather_plan_bytes(int ndof_per_entity, int64_t nent, int64_t ndofs);
int fus_mass_gather_plan_build(const int32_t* entity_dofmap, int ndof_per_entity, int64_t nent, int64_t ndofs,
                               void* workspace, int64_t workspace_bytes, void* stream);
int fus_mass_gather_plan_info(const void* workspace, int64_t* out4);
/*
 * The plan of a ROW SUBSET: only the dofs d with row_set[d] == which (row_set: device uint8[ndofs]) get a row; every row still
 * sums ALL its entries.  What a partitioned apply needs (scatterer.HaloApply; the reference applies the operator between its two
 * scatters, cuda/demo_linear_box.py:546-553): the rows whose dof neither needs the forward exchange (an owned dof) nor is added
 * into by the reverse exchange run next to the exchanges, the rest between them -- no launch and no receive ever adds into
 * the same y[d] concurrently, so the atomic-free kernel stays valid at N > 1.  Same workspace size as the full plan.
 */
int fus_mass_gather_plan_build_rows(const int32_t* entity_dofmap, int ndof_per_entity, int64_t nent, int64_t ndofs,
                                    const uint8_t* row_set, int which, void* workspace, int64_t workspace_bytes, void* stream);
/*
 * STATIC companion of a transposed-dofmap plan (opt-in; the caller declares entity_detJ constant across applies, as the
 * reference's drivers keep it: cuda/demo_nonlinear_bowl.py:603-632 re-applies the mass operator with the same detJ every stage):
 * a copy of detJ in ROW order plus a 16-bit entity offset per entry, in a second caller-owned workspace.  The apply then
 * streams detJ contiguously instead of gathering it through the entry ids (the dependent gather that keeps the texture
 * addresser 82 % busy in fus_mass_apply_gather_*); the entity constants are still read per apply and may change.  Same sums
 * in the same order as fus_mass_apply_gather_* (bitwise identical results).  FUS_ERR_UNSUPPORTED_ENTITY: some block of 256
 * consecutive dofs touches entities more than 65 535 apart (a numbering without locality) -- use fus_mass_apply_gather_*.
 * fus_plan_release(static_workspace) forgets it.  Rebuild after changing detJ.
 */
int64_t fus_mass_gather_static_bytes(int ndof_per_entity, int64_t nent, int elem_bytes);
int fus_mass_gather_static_build_f64(const void* workspace, const double* entity_detJ, void* static_workspace,
                                     int64_t static_workspace_bytes, void* stream);
int fus_mass_gather_static_build_f32(const void* workspace, const float* entity_detJ, void* static_workspace,
                                     int64_t static_workspace_bytes, void* stream);
int fus_mass_apply_gather_static_f64(const double* x, const double* entity_constants, double* y, const void* workspace,
                                     const void* static_workspace, int ndof_per_entity, int64_t nent, void* stream);
int fus_mass_apply_gather_static_f32(const float* x, const float* entity_constants, float* y, const void* workspace,
                                     const void* static_workspace, int ndof_per_entity, int64_t nent, void* stream);
int fus_mass_apply_gather_f64(const double* x, const double* entity_constants, double* y, const double* entity_detJ,
                              const void* workspace, int ndof_per_entity, int64_t nent, void* stream);
int fus_mass_apply_gather_f32(const float* x, const float* entity_constants, float* y, const float* entity_detJ,
                              const void* workspace, int ndof_per_entity, int64_t nent, void* stream);

/*
 * Streaming vector kernels of the RK4 stage.
 * replace  cuda/operators.py:195-274 (axpy, copy, fill, pointwise_divide, square)
 *     and  numba-cpu/operators.py:230-300
 */
int fus_axpy_f64(double alpha, const double* x, double* y, int64_t n, void* stream);           /* y = alpha x + y */
int fus_axpy_f32(float alpha, const float* x, float* y, int64_t n, void* stream);
int fus_copy_f64(const double* a, double* b, int64_t n, void* stream);                         /* b = a */
int fus_copy_f32(const float* a, float* b, int64_t n, void* stream);
int fus_fill_f64(double alpha, double* x, int64_t n, void* stream);                            /* x = alpha */
int fus_fill_f32(float alpha, float* x, int64_t n, void* stream);
int fus_pointwise_divide_f64(const double* a, const double* b, double* c, int64_t n, void* stream); /* c = a / b */
int fus_pointwise_divide_f32(const float* a, const float* b, float* c, int64_t n, void* stream);
int fus_scale_f64(double alpha, const double* a, double* b, int64_t n, void* stream);          /* b = alpha a (no reference kernel: fill + axpy there) */
int fus_scale_f32(float alpha, const float* a, float* b, int64_t n, void* stream);
int fus_square_f64(const double* a, double* b, int64_t n, void* stream);                       /* b = a^2 */
int fus_square_f32(const float* a, float* b, int64_t n, void* stream);
/*
 * y += w (.) x  -- the cell mass apply in cached-diagonal form (opt-in; no counterpart in the reference, whose drivers
 * call mass_operator on every use, cuda/demo_nonlinear_bowl.py:603-632).  With GLL collocation M(c) x = (M(c) 1) (.) x:
 * assemble w = M(c) 1 once with fus_mass_apply_* (x = 1), then 3 vector touches per dof per apply instead of the
 * gather-scale-scatter's 47.6 B/dof (P = 4, fp64).  w must be re-assembled when constants, detJ or the dofmap change.
 */
int fus_muladd_f64(const double* w, const double* x, double* y, int64_t n, void* stream);
int fus_muladd_f32(const float* w, const float* x, float* y, int64_t n, void* stream);

/*
 * Geometry precompute on the device, same inputs / conventions / outputs as the reference's host
 * routines (numba-cpu/precompute.py:76-163 and :17-73):
 *   x_g T[nvert][3], x_dofs int32[ncell][8] (P1 hex), dphi T[3][nq][8], weights T[nq]
 *   -> G T[ncell][nq][6] and/or detJ T[ncell][nq]   (either may be NULL)
 *   boundary_data int32[nfacets][2] = (cell, local facet), dphi_f T[6][3][nqf][8], weights T[nqf]
 *   -> detJ_f T[nfacets][nqf]
 */
int fus_geometry_factors_f64(const double* x_g, const int32_t* x_dofs, const double* dphi, const double* weights,
                             int nq, int64_t ncell, double* G, double* detJ, void* stream);
int fus_geometry_factors_f32(const float* x_g, const int32_t* x_dofs, const float* dphi, const float* weights, int nq,
                             int64_t ncell, float* G, float* detJ, void* stream);
int fus_facet_jacobian_f64(const double* x_g, const int32_t* x_dofs, const int32_t* boundary_data,
                           const double* dphi_f, const double* weights, int nqf, int64_t nfacets, double* detJ_f,
                           void* stream);
int fus_facet_jacobian_f32(const float* x_g, const int32_t* x_dofs, const int32_t* boundary_data, const float* dphi_f,
                           const float* weights, int nqf, int64_t nfacets, float* detJ_f, void* stream);

/*
 * Fused RK4 stage update: everything the reference does between scatter_rev(b) of one stage and
 * scatter_fwd of the next (cuda/demo_linear_box.py:556-563 then :491-508,541 -- 12 launches, 216
 * B/dof) in one pass:
 *   kv = b * minv;  u += bw ku;  v += bw kv;  [new_step: u0 = u; v0 = v;]
 *   un = u0 + aw ku;  ku = v0 + aw kv  (= vn, which is also f0 of the next stage);  b = 0
 * over the owned dofs [0, nlocal); b is zeroed over [0, ntotal).  bw = b_runge[i] dt,
 * aw = a_runge[i+1] dt (0 with new_step = 1 after the last stage).  minv = 1 / m.
 * new_step selects the stage kind: 0 middle stage; 1 last stage as written above; and, for drivers that
 * hand (u0, v0) themselves to the operator as the inputs of a step's first stage instead of copies,
 * 2 = first stage (u, v, ku are read from u0, v0, v0: u = u0 + bw v0; v = v0 + bw kv; un = u0 + aw v0;
 * ku = v0 + aw kv) and 3 = last stage (u0 = u + bw ku; v0 = v + bw kv; nothing else written): 41
 * instead of 48 vector touches per step, same arithmetic.
 * LEAN set new_step = 4, 5, 6, 7 (added in round 6, additive: no ABI bump; the four stages of ONE step, used together, with bw = b_runge[0] dt = dt / 6 and
 * aw = a_runge[1] dt = dt / 2 in all four calls): 34 vector touches per step -- u's accumulator runs one stage ahead (its
 * increments are the vn's, known one pass early), the first pass writes no accumulator, the third writes the new u straight
 * into u0, the fourth touches v only; between the calls ``u`` holds an intermediate that is NOT the reference's u.  After the
 * fourth call the new solution is in (u0, v0), as with 2, 0, 0, 3.  u bitwise as the sequence above; v differs in the rounding
 * of ONE term (csrc/rk4.hpp).  Any other value: FUS_ERR_INVALID_ARGUMENT.
 */
int fus_rk4_stage_f64(double bw, double aw, int new_step, const double* minv, double* b, double* u, double* v,
                      double* u0, double* v0, double* ku, double* un, int64_t nlocal, int64_t ntotal, void* stream);
int fus_rk4_stage_f32(float bw, float aw, int new_step, const float* minv, float* b, float* u, float* v, float* u0,
                      float* v0, float* ku, float* un, int64_t nlocal, int64_t ntotal, void* stream);

/*
 * Fused Westervelt cell pass (needs the cell batch plan): what the reference does with four
 * launches over the same cells per RK4 stage (cuda/demo_nonlinear_bowl.py:612-632 + square :603)
 *   b += K(c3) u + K(c4) v + M(c5) v^2 ;   m += M(c2) u
 * reading G and detJ once and gathering u, v once.  c2..c5: per-cell constants T[ncell].
 * fus_rk4_stage_nl_*: fus_rk4_stage_* for a stage-dependent lumped mass: kv = b / m, then m = m0.
 *
 * Diagonal form (what the Westervelt solver of this repo runs): with GLL collocation the mass operator
 * is diagonal, M(c) x = diag(M(c) 1) x, so the two mass terms can be applied pointwise from diagonals
 * assembled once, w2 = M(c2) 1 and w5 = M(c5) 1:
 *   fus_westervelt_cell_apply_planned[_geom]_* with c2 = c5 = m = detJ = NULL computes the stiffness part
 *   alone, b += K(c3) u + K(c4) v  (no detJ stream, one atomic flush, no m array);
 *   fus_rk4_stage_nl2_*: kv = (b + w5 v_n^2) / (m0 + w2 u_n) with (u_n, v_n) the stage's inputs ((u0, v0)
 *   for new_step = 2 or 4, else (un, ku)), then the updates of fus_rk4_stage_* (new_step 0 ... 7).  If w != NULL it also writes
 *   w = u_n' + kappa v_n' of the NEXT stage's inputs: where c4 = kappa c3 in every cell the next cell pass is
 *   then ONE plain stiffness apply K(c3) w (one gather).
 * Same result as the four reference launches up to summation order.
 */
int fus_rk4_stage_nl2_f64(double bw, double aw, int new_step, const double* m0, const double* w2, const double* w5,
                          double* b, double* u, double* v, double* u0, double* v0, double* ku, double* un,
                          double kappa, double* w, int64_t nlocal, int64_t ntotal, void* stream);
int fus_rk4_stage_nl2_f32(float bw, float aw, int new_step, const float* m0, const float* w2, const float* w5,
                          float* b, float* u, float* v, float* u0, float* v0, float* ku, float* un, float kappa,
                          float* w, int64_t nlocal, int64_t ntotal, void* stream);
int fus_westervelt_cell_apply_planned_f64(const double* u, const double* v, const double* c2, const double* c3,
                                          const double* c4, const double* c5, double* b, double* m, const double* G,
                                          const double* detJ, const void* workspace, const double* dphi, int P,
                                          int64_t ncell, void* stream);
int fus_westervelt_cell_apply_planned_f32(const float* u, const float* v, const float* c2, const float* c3,
                                          const float* c4, const float* c5, float* b, float* m, const float* G,
                                          const float* detJ, const void* workspace, const float* dphi, int P,
                                          int64_t ncell, void* stream);
/* The same pass with G and detJ formed in the kernel from the cell vertices (arguments as
 * fus_stiffness_apply_planned_geom_*; own bytes contract: neither array is read). */
int fus_westervelt_cell_apply_planned_geom_f64(const double* u, const double* v, const double* c2, const double* c3,
                                               const double* c4, const double* c5, double* b, double* m,
                                               const double* x_g, const int32_t* x_dofs, const double* pts,
                                               const double* wts, const void* workspace, const double* dphi, int P,
                                               int64_t ncell, void* stream);
int fus_westervelt_cell_apply_planned_geom_f32(const float* u, const float* v, const float* c2, const float* c3,
                                               const float* c4, const float* c5, float* b, float* m, const float* x_g,
                                               const int32_t* x_dofs, const float* pts, const float* wts,
                                               const void* workspace, const float* dphi, int P, int64_t ncell,
                                               void* stream);
int fus_rk4_stage_nl_f64(double bw, double aw, int new_step, const double* m0, double* m, double* b, double* u,
                         double* v, double* u0, double* v0, double* ku, double* un, int64_t nlocal, int64_t ntotal,
                         void* stream);
int fus_rk4_stage_nl_f32(float bw, float aw, int new_step, const float* m0, float* m, float* b, float* u, float* v,
                         float* u0, float* v0, float* ku, float* un, int64_t nlocal, int64_t ntotal, void* stream);

/*
 * Halo pack / unpack (all neighbours in ONE launch: ``index`` is the concatenation of the
 * per-neighbour index lists, the send/recv buffer is the concatenation of the per-neighbour
 * messages).  N = nlocal (offset of the ghost block in a dof vector).
 * replace  cuda/scatterer.py:18-101 (pack_fwd, unpack_fwd, pack_rev, unpack_rev; one launch per neighbour there)
 *     and  numba-cpu/scatterer.py:18-75 (pack, unpack_fwd, unpack_rev)
 *   pack_fwd   : out[i] = in[index[i]]
 *   unpack_fwd : out[index[i] + N] = in[i]
 *   pack_rev   : out[i] = in[index[i] + N]
 *   unpack_rev : out[index[i]] += in[i]          (atomic: two neighbours may hit one owned dof)
 */
int fus_pack_fwd_f64(const double* in, double* out, const int64_t* index, int64_t count, void* stream);
int fus_pack_fwd_f32(const float* in, float* out, const int64_t* index, int64_t count, void* stream);
int fus_unpack_fwd_f64(const double* in, double* out, const int64_t* index, int64_t count, int64_t N, void* stream);
int fus_unpack_fwd_f32(const float* in, float* out, const int64_t* index, int64_t count, int64_t N, void* stream);
int fus_pack_rev_f64(const double* in, double* out, const int64_t* index, int64_t count, int64_t N, void* stream);
int fus_pack_rev_f32(const float* in, float* out, const int64_t* index, int64_t count, int64_t N, void* stream);
int fus_unpack_rev_f64(const double* in, double* out, const int64_t* index, int64_t count, void* stream);
int fus_unpack_rev_f32(const float* in, float* out, const int64_t* index, int64_t count, void* stream);

/*
 * Ghost-dof halo exchange (one process per GPU, RCCL over xGMI).
 * replaces  cuda/scatterer.py:104-188  scatter_reverse(comm, owners_data, ghosts_data, N, float_type) -> scatter(buffer)
 *      and  cuda/scatterer.py:191-277  scatter_forward(...)                                           -> scatter(buffer)
 *      and  numba-cpu/scatterer.py:78-141, 144-207 (the same closures over host arrays)
 *      and  the C++ driver's scatter calls, cpp/common/Linear.hpp:120,193,196,212
 * The reference packs with one kernel per neighbour, device-synchronises, posts MPI Isend/Irecv on
 * device pointers and synchronises again; here an exchange is pack -> grouped ncclSend/ncclRecv ->
 * unpack on a library-owned high-priority stream, ordered against the caller's stream by events only
 * (no host synchronisation), and split in begin / end so interior-cell work overlaps it.
 *
 * Communicator.  fus_comm_unique_id: rank 0 obtains FUS_UNIQUE_ID_BYTES bytes and broadcasts them by any
 * means the host has (MPI_Bcast in the reference's drivers, torch.distributed in this repo's);
 * fus_comm_create: collective over all ranks, binds the CURRENT HIP device (ncclCommInitRank).
 * librccl.so.1 is resolved with dlopen on first use; libfusgpu.so does not link against it.
 * fus_comm_create_local: all ranks live in one process (tests on a one-GPU box; one process driving
 * several GPUs); ranks that pass the same world_id form a world.  Host-side contract of this transport:
 * every rank's *_begin of an exchange is called before any rank's *_end of it.
 * fus_comm_create_peer: no RCCL.  Every halo object of such a communicator owns a receive arena in uncached (fallback: fine-grained)
 * device memory; the ranks exchange the arenas' HIP IPC handles once (fus_halo_ipc_export -> any all-gather the host
 * has -> fus_halo_ipc_connect) and an exchange is then two small kernels per rank: a send kernel that stores straight
 * into the neighbours' arenas (xGMI stores) and publishes a sequence flag, a receive kernel that waits for the flag
 * and stores / adds into the vector (csrc/halo_ipc.hpp).  Unlike RCCL's send/recv kernel they fit next to a
 * chip-filling operator launch, so the exchange really runs under interior-cell work.
 * Host-side contract of this transport (the one MPI's non-blocking collectives and RCCL have): ALL RANKS POST THE
 * EXCHANGES OF A COMMUNICATOR IN THE SAME ORDER -- send and receive kernels share the communicator's one stream, so two
 * ranks posting (halo 1, halo 2) and (halo 2, halo 1) wait for each other (FUS_IPC_TWO_STREAMS=1 lifts this: receive
 * kernels on a stream of their own).  Nothing else: no per-exchange hand-shake between the hosts.
 * Every device-side wait is bounded (FUS_IPC_SPIN_SECONDS, default 20 s).  A wait that gives up is counted, the halo
 * object stops waiting (is "dead") and every flag it publishes from then on is POISONED: a neighbour that reads one does
 * not consume the (stale) arena, counts it and dies too -- a failed exchange spreads to every rank connected to it within a
 * few exchanges instead of producing a wrong field.  fus_halo_ipc_status / fus_comm_health report it; a time loop MUST
 * check fus_comm_health() == 0 before it trusts its result (the reference would block in MPI Waitall,
 * cuda/scatterer.py:175).
 * fus_comm_destroy fails (FUS_ERR_COMM) while halo objects of the communicator are alive.
 */
#define FUS_UNIQUE_ID_BYTES 128
typedef struct fus_comm* fus_comm_t;
typedef struct fus_halo* fus_halo_t;
int fus_comm_unique_id(void* id /* FUS_UNIQUE_ID_BYTES */);
int fus_comm_create(const void* id, int nranks, int rank, fus_comm_t* comm);
int fus_comm_create_local(int world_id, int nranks, int rank, fus_comm_t* comm);
int fus_comm_create_peer(int nranks, int rank, fus_comm_t* comm);
int fus_comm_rank(fus_comm_t comm);
int fus_comm_size(fus_comm_t comm);
void* fus_comm_stream(fus_comm_t comm); /* the hipStream_t the exchanges run on */
/*
 * Event-free ordering between a caller's stream and the communicator's stream (any transport), for hosts that put
 * work of their own on fus_comm_stream() -- e.g. the boundary-cell kernels between a forward and a reverse exchange,
 * next to ONE launch over the interior cells on the caller's stream (scatterer.HaloApply, schedule "concurrent").
 *   fus_comm_fork: what is enqueued on the communicator's stream from now on starts after everything enqueued on
 *                  ``stream`` so far (a one-thread signal kernel on ``stream``, a one-wave bounded wait kernel on the
 *                  communicator's stream: 2.4 us on ``stream`` where an event record costs 7 next to chip-filling launches);
 *   fus_comm_join: the reverse direction.
 * fus_comm_sync_timeouts: waits of these kernels that gave up (FUS_IPC_SPIN_SECONDS); synchronises the communicator's stream.
 * There is ONE sequence flag per direction and communicator, so consecutive forks (and their joins) of a communicator must
 * come from ONE caller stream, as a time loop's do: a fork from another stream is accepted only once the communicator's
 * stream has drained, a join from another stream than its fork never (FUS_ERR_INVALID_ARGUMENT, nothing launched).
 *
 * PEER transport, two kernels fewer per apply in the exchange chain:
 *   fus_comm_fork_lazy: as fus_comm_fork, but no wait kernel -- the FIRST send kernel of the next fus_halo_*_begin[_group]
 *                  posted on the communicator's stream waits for the fork flag itself.  The caller must post that
 *                  exchange next, before anything else on the communicator's stream (where no send kernel can carry the
 *                  wait -- no neighbours on that side, another transport -- a wait kernel is launched after all).
 *   fus_comm_arm_join:  the LAST receive kernel of the next fus_halo_*_begin[_group] of this communicator publishes the
 *                  join flag; the fus_comm_join after it then launches only the wait kernel on the caller's stream.  That
 *                  exchange must be the last work on the communicator's stream before the join (falls back to the signal
 *                  kernel where no receive kernel can carry it).
 *   fus_comm_fork_ex(flags): FUS_FORK_LAZY as fus_comm_fork_lazy; FUS_FORK_ATTACH (any transport): no signal kernel on
 *                  ``stream`` either -- the NEXT PLANNED operator launch on ``stream`` (fus_stiffness_apply_planned*,
 *                  fus_mass_apply_planned_*, fus_westervelt_cell_apply_planned*) publishes the fork flag when its first
 *                  workgroup starts, which is when everything enqueued on ``stream`` before it has completed: 2.4 us less on
 *                  the caller's stream.  If no such launch follows (an empty cell range, a plan-free kernel) call
 *                  fus_comm_fork_flush (the next fus_comm_fork* / fus_comm_join / fus_comm_destroy does it too).
 * fus_comm_health: failed device-side waits (time-outs + poisoned flags) of every live halo object of the communicator
 *                  and of its fork / join kernels; 0 = every exchange so far delivered.  Synchronises the device.
 */
int fus_comm_fork(fus_comm_t comm, void* stream);
int fus_comm_fork_lazy(fus_comm_t comm, void* stream);
#define FUS_FORK_LAZY 1
#define FUS_FORK_ATTACH 2
int fus_comm_fork_ex(fus_comm_t comm, void* stream, int flags);
int fus_comm_fork_flush(fus_comm_t comm);
int fus_comm_join(fus_comm_t comm, void* stream);
int fus_comm_arm_join(fus_comm_t comm);
int fus_comm_sync_timeouts(fus_comm_t comm, int64_t* out);
int fus_comm_health(fus_comm_t comm, int64_t* failures);
/* the same, split: out3 = {time-outs of exchange waits, poisoned flags read (a neighbour had failed), time-outs of fork / join waits} */
int fus_comm_health_detail(fus_comm_t comm, int64_t* out3);
const char* fus_comm_last_error(fus_comm_t comm /* NULL: errors raised before a communicator existed */);
int fus_comm_destroy(fus_comm_t comm);

/*
 * Halo plan = the reference's (owners_data, ghosts_data) of cuda/utils.py:8-78 compute_scatterer_data,
 * passed as HOST arrays (copied at creation):
 *   owners side: my ghost dofs grouped by owning rank -- owner_ranks[n_owner_ranks], owner_sizes[...],
 *                owners_idx = concatenated positions inside my ghost block (vector index = nlocal + idx)
 *   ghosts side: my owned dofs that other ranks ghost -- ghost_ranks, ghost_sizes,
 *                ghosts_idx = concatenated local indices, in the order the ghosting rank packs them
 * elem_bytes: 8 (double) or 4 (float).  Indices are range-checked here (the reference does not).
 * If owners_idx == 0,1,2,... (ghosts numbered owner by owner) the ghost block of the vector is used as
 * the message buffer of that side (fus_halo_is_direct() == 1): no unpack_fwd / pack_rev launch.
 */
int fus_halo_create(fus_comm_t comm, int elem_bytes, int64_t nlocal, int64_t nghost, int n_owner_ranks,
                    const int32_t* owner_ranks, const int64_t* owner_sizes, const int64_t* owners_idx,
                    int n_ghost_ranks, const int32_t* ghost_ranks, const int64_t* ghost_sizes,
                    const int64_t* ghosts_idx, fus_halo_t* halo);
int fus_halo_is_direct(fus_halo_t halo);
int fus_halo_destroy(fus_halo_t halo); /* before fus_comm_destroy of its communicator */
/*
 * PEER transport only: connect a halo object to its neighbours' arenas.  Collective in the sense that every rank
 * exports the blob of ITS halo object number k and connects with the blobs of the other ranks' object number k
 * (any order; the blobs of non-neighbours are ignored; a rank that is its own neighbour passes its own blob).
 * The blobs are plain bytes (valid on this host only: they hold HIP IPC handles).  Destroy the halo objects only
 * after the last exchange has completed on every rank.
 * fus_halo_ipc_status: out8 = {failed device-side waits so far = time-outs + poisoned flags read (0 = healthy), forward
 * exchanges posted, reverse exchanges posted, arena memory kind (0 fine-grained, 1 uncached, 2 ordinary), time-outs,
 * poisoned flags read (a neighbour's halo object had failed), dead (0 / 1), 0}; synchronises the device.
 */
int64_t fus_halo_ipc_blob_bytes(fus_halo_t halo);
int fus_halo_ipc_export(fus_halo_t halo, void* blob);
int fus_halo_ipc_connect(fus_halo_t halo, int nblobs, const void* const* blobs);
int fus_halo_ipc_status(fus_halo_t halo, int64_t* out8);
/*
 * forward: buffer[nlocal + g] = owner's value, for every ghost g          (scatter_forward, overwrite)
 * reverse: owner's buffer[i] += every ghosting rank's partial sum of i     (scatter_reverse, add)
 * fus_halo_forward / fus_halo_reverse (the whole exchange in one call = the reference's scatter(buffer)): on the PEER
 * transport both kernels run on ``stream`` itself, in stream order with what precedes and follows them (22 us per call
 * instead of 45 through the communicator's stream and back); a host that drives SEVERAL ranks from one thread uses
 * *_begin for every rank, then *_end for every rank, instead.
 * in place on ``buffer`` (device pointer, nlocal + nghost elements).  *_begin orders the exchange after
 * everything already enqueued on ``stream`` and returns at once; *_end makes ``stream`` wait for its
 * completion.  Between the two, work on ``stream`` must not touch what the exchange touches: the ghost
 * entries (forward), or -- other than by atomic adds -- the owned entries being added to (reverse).
 * One exchange per halo object may be in flight; use one object per vector exchanged concurrently.
 */
int fus_halo_forward_begin(fus_halo_t halo, void* buffer, void* stream);
int fus_halo_forward_end(fus_halo_t halo, void* buffer, void* stream);
int fus_halo_reverse_begin(fus_halo_t halo, void* buffer, void* stream);
int fus_halo_reverse_end(fus_halo_t halo, void* buffer, void* stream);
/*
 * Several vectors at once (the RK4 stage forward-scatters u_n and v_n; n <= 8, halos of one communicator, one
 * per vector): one event edge and ONE RCCL group for all of them instead of one per vector.  Each is then
 * completed with its own fus_halo_*_end.
 */
int fus_halo_forward_begin_group(const fus_halo_t* halos, void* const* buffers, int n, void* stream);
int fus_halo_reverse_begin_group(const fus_halo_t* halos, void* const* buffers, int n, void* stream);
int fus_halo_forward(fus_halo_t halo, void* buffer, void* stream); /* begin + end */
int fus_halo_reverse(fus_halo_t halo, void* buffer, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FUS_GPU_H */
