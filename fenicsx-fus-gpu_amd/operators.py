"""
Operator call surface of the reference, over libfusgpu.so (hand-written HIP for
gfx950).  Both flavours of the reference are served by the same objects:

numba-cpu flavour (numba-cpu/operators.py):
    mass_operator(N, float_type)            -> op(x, entity_constants, y, entity_detJ, entity_dofmap)   :19-68
    stiffness_operator(P, dphi, float_type) -> op(x, cell_constants, y, G, dofmap)                      :71-227
    axpy(local_size)                        -> kernel(alpha, x, y)                                      :230-251
    copy(a, b); fill(alpha, x); pointwise_divide(a, b, c)                                               :254-300

cuda flavour (cuda/operators.py), launched as ``kernel[grid, block](args...)``:
    mass_operator[g, b](x, entity_constants, y, detJ_entity, entity_dofmap)                             :18-70
    stiffness_operator(P, float_type)       -> op[g, b](x, consts, y, G, dofmap, dphi)                  :73-192
    axpy[g, b](alpha, x, y); copy[g, b](a, b); fill[g, b](alpha, x);
    pointwise_divide[g, b](a, b, c); square[g, b](a, b)                                                 :195-274

The launch configuration given in ``[grid, block]`` is accepted and ignored: the
library chooses its own geometry for CDNA4.  Arrays are device arrays (torch
tensors on the GPU, e.g. from ``device.to_device``); ``y`` / outputs are
modified in place; launches are asynchronous on torch's current HIP stream.
Argument errors raise (TypeError / ValueError) like a numba dispatch failure;
there is no CPU fallback.
"""

from __future__ import annotations

import os

import numpy as np
import torch

from . import _lib

_req = _lib.require_device_tensor

# The stiffness operator builds a batch plan (csrc/plan.hpp) the first time it sees a dofmap and
# reuses it while that dofmap array is unchanged.  FUS_STIFFNESS_PLAN=0 (or use_plan(False))
# selects the plan-free kernel that reads ``dofmap`` directly.
_USE_PLAN = os.environ.get("FUS_STIFFNESS_PLAN", "1") != "0"


def use_plan(flag: bool):
    global _USE_PLAN
    _USE_PLAN = bool(flag)


# Set-up-time locality ordering of the plan's batches (FUS_PLAN_LOCALITY_ORDER=0 / use_locality_order(False)
# turns it off): when a dofmap's cells are not already sorted by their smallest dof, a second plan
# is built with the cells taken in that order (an index indirection inside the plan: G, detJ and the
# constants stay where they are) and kept if its batches touch fewer distinct dofs.  A mesh with
# consecutive cells adjacent (BoxMesh, a bandwidth-reordered dolfinx mesh) is left alone; a random cell
# order goes from 0.365 back to 0.243 ms per apply at P = 4, 10 M dofs (profiles/r02c_numbering.log).
_LOCALITY_ORDER = os.environ.get("FUS_PLAN_LOCALITY_ORDER", "1") != "0"
# Two-row strip order for the plans of the scatter-bound kernels (plan_tiles.py).  OFF by default -- a measured negative
# (profiles/r05d_geom_variance_probe.log, r05e_*): 2 x 5 pieces touch 8 % fewer distinct dofs than 10 cells in a row (94.8 against 102.9
# per cell at P = 4) but flush them in MORE 64-byte atomic requests (45 runs of 21 dofs instead of 25 runs of 41), and the request
# count is what bounds these kernels: 156 us against 153.5 us for the in-kernel-geometry kernel at config 3.
# FUS_PLAN_STRIP_ORDER=1 / use_strip_order(True) turns it on (experiments).
_STRIP_ORDER = os.environ.get("FUS_PLAN_STRIP_ORDER", "0") == "1"


def use_strip_order(flag: bool):
    global _STRIP_ORDER
    _STRIP_ORDER = bool(flag)



def use_locality_order(flag: bool):
    global _LOCALITY_ORDER
    _LOCALITY_ORDER = bool(flag)


class _PlanCache:
    """Batch-plan workspaces keyed on the identity of the dofmap array (pointer, shape, version).
    One cache for the whole module: the cell mass operator and the stiffness operator share a
    plan when they are given the same dofmap."""

    def __init__(self, capacity: int = 16):
        self._plans = {}
        self.capacity = capacity
        self.last_order = None  # cell order of the plan built last (None: natural order)
        self._recording = None  # while a hipGraph is captured: every (workspace, dofmap) handed out

    def start_recording(self):
        self._recording = []

    def stop_recording(self):
        """-> the (workspace, dofmap) tensors handed out since ``start_recording``: whoever baked their addresses
        into a captured graph holds this list, so eviction from the cache cannot free them."""
        held, self._recording = self._recording or [], None
        return held

    def get(self, dofmap: torch.Tensor, exclusive_ndofs=None, external_use=None, strips=False):
        """-> (workspace tensor, entities_per_batch).  ``strips``: the plan of a kernel bound by its scatter side (in-kernel
        geometry, affine cells): its cell order interleaves adjacent rows of cells (``plan_tiles.two_row_strip_order``: 2 x 5
        pieces instead of 10 cells in a row at P = 4, -8 % distinct dofs per batch) when that lowers the number of distinct dofs
        the batches touch -- a separate cache entry from the row-ordered plan of the same dofmap, which the general-G kernels keep.  ``exclusive_ndofs`` (length of the vectors the plan is applied to):
        the plan also carries EXCLUSIVE-DOF MARKS (``fus_plan_mark_exclusive``: a dof touched by exactly one batch is finished
        with a plain load + store instead of a float atomic), a separate cache entry from the unmarked plan of the same
        dofmap.  ``external_use``: device int32[ndofs], what else adds into each dof while a launch with this plan runs
        (default: nothing -- the launch runs alone or only next to launches of the same stream)."""
        lib = _lib.load()
        nent, N = dofmap.shape
        strips = bool(strips) and _STRIP_ORDER and exclusive_ndofs is None
        key = (dofmap.data_ptr(), nent, N, dofmap._version, dofmap.device.index,
               None if exclusive_ndofs is None else (int(exclusive_ndofs), None if external_use is None else (external_use.data_ptr(), external_use._version)))
        if strips:
            key = key + ("strips",)
        hit = self._plans.get(key)
        if hit is None:
            epb = lib.fus_plan_entities_per_batch(N)
            if epb < 0:
                _lib.check(epb, "fus_plan_entities_per_batch")
            nbytes = lib.fus_plan_bytes(N, epb, nent)
            if nbytes < 0:
                _lib.check(int(nbytes), "fus_plan_bytes")

            def build(order):
                w = torch.empty(int(nbytes), dtype=torch.uint8, device=dofmap.device)
                _lib.check(
                    lib.fus_plan_build_ordered(dofmap.data_ptr(), order.data_ptr() if order is not None else None, N, epb, nent,
                                               w.data_ptr(), int(nbytes), _lib.stream_ptr()),
                    "fus_plan_build_ordered",
                )
                return w

            def distinct_dofs(w):  # sum over batches of the distinct dofs a batch touches
                nbatch = (nent + epb - 1) // epb
                return int((w[256:256 + 4 * nbatch].view(torch.int32) & 0xFFFF).sum().item())

            ws = build(None)
            self.last_order = None
            if _LOCALITY_ORDER and nent > 2 * epb:
                mins = dofmap.min(dim=1).values
                if not bool((mins[1:] >= mins[:-1]).all().item()):  # not already in that order
                    order = torch.argsort(mins, stable=True).to(torch.int32)
                    ws2 = build(order)
                    if distinct_dofs(ws2) < 0.97 * distinct_dofs(ws):
                        ws, ws2, self.last_order = ws2, ws, order
                    lib.fus_plan_release(ws2.data_ptr())  # the plan that was not kept
            if strips and nent > 4 * epb:
                n = int(round(N ** (1.0 / 3.0)))
                if n >= 3 and n**3 == N:  # cells of degree >= 2 in tensor-product local order
                    from . import plan_tiles

                    faces = torch.from_numpy(plan_tiles.face_interior_local_dofs(n)).to(dofmap.device)
                    cand = plan_tiles.two_row_strip_order(dofmap[:, faces].cpu().numpy(),
                                                          None if self.last_order is None else self.last_order.cpu().numpy())
                    if cand is not None:
                        order = torch.from_numpy(cand.astype("int32")).to(dofmap.device)
                        ws2 = build(order)
                        if distinct_dofs(ws2) < 0.97 * distinct_dofs(ws):
                            ws, ws2, self.last_order = ws2, ws, order
                        lib.fus_plan_release(ws2.data_ptr())
            if exclusive_ndofs is not None:
                use = (external_use.to(torch.int32).clone() if external_use is not None
                       else torch.zeros(int(exclusive_ndofs), dtype=torch.int32, device=dofmap.device))
                if use.numel() != int(exclusive_ndofs):
                    raise ValueError("external_use must have one entry per dof")
                _lib.check(lib.fus_plan_mark_exclusive(ws.data_ptr(), N, epb, nent, use.data_ptr(), int(exclusive_ndofs), _lib.stream_ptr()),
                           "fus_plan_mark_exclusive")
                del use  # a temporary: the caching allocator hands its memory out again in stream order
            if len(self._plans) >= self.capacity:  # bounded: drop the oldest plan
                old = self._plans.pop(next(iter(self._plans)))
                lib.fus_plan_release(old[0].data_ptr())
            # the entry holds the dofmap tensor itself: while a plan is cached its memory cannot be
            # freed and handed to another array with the same address / shape / version
            hit = (ws, epb, dofmap)
            self._plans[key] = hit
        if self._recording is not None:
            self._recording.append((hit[0], hit[2]))
        return hit[0], hit[1]

    def has(self, dofmap: torch.Tensor) -> bool:
        """True if an (unmarked) plan of ``dofmap`` is cached -- row-ordered or strip-ordered: an operator uses one of the two
        consistently, and ``HaloApply`` asks only for a cell range the operator has been applied to before -- so an apply with
        it does no set-up work (no allocation, no host synchronisation): what ``HaloApply`` needs to know before it lets a
        launch carry a fork signal."""
        nent, N = dofmap.shape
        key = (dofmap.data_ptr(), nent, N, dofmap._version, dofmap.device.index, None)
        return key in self._plans or key + ("strips",) in self._plans

    def clear(self):
        lib = _lib.load()
        for ws, _, _ in self._plans.values():
            lib.fus_plan_release(ws.data_ptr())
        self._plans.clear()
        gather = globals().get("_GATHER_PLANS")  # the transposed-dofmap plans of the mass apply go with them
        if gather is not None:
            gather.clear()
        static = globals().get("_STATIC_DETJ")
        if static is not None:
            static.clear()


_PLANS = _PlanCache()
_MASS_PLAN_MIN_ENTRIES = 1 << 15  # below this the plan-free kernel is already launch-bound

# The mass apply WITHOUT atomics (csrc/mass_gather.hpp): one thread per touched dof sums its (entity, local index) entries
# from the transposed dofmap.  FUS_MASS_GATHER=0 / use_mass_gather(False) keeps the atomic kernels everywhere.
_USE_GATHER = os.environ.get("FUS_MASS_GATHER", "1") != "0"
# mean entries per touched dof above which the gather loses to the atomic batch plan: at P = 2 (27 / 8 = 3.4 entries per dof,
# one-element detJ segments) 0.262 against 0.200 ms, at P = 3 (2.4) 0.142 against 0.158 (profiles/r04t_ab_mass_gather.log)
_GATHER_MAX_MEAN_ENTRIES = 2.6
# ... the STATIC-detJ form of the gather (detJ streamed in row order instead of gathered through one-element segments) wins up to P = 2:
# 0.135 against 0.176 ms fp64 (0.60 of the roofline instead of 0.46), 0.071 against 0.134 ms fp32, where the plain gather takes 0.242 / 0.139
# (profiles/r06g_ab_mass_low_degree.log).  P = 1 (8 entries per dof) stays on the atomic plan.
_GATHER_STATIC_MAX_MEAN_ENTRIES = 4.0


def use_mass_gather(flag: bool):
    global _USE_GATHER
    _USE_GATHER = bool(flag)


class _GatherPlanCache:
    """Transposed-dofmap plans of the atomic-free mass apply, keyed like the batch plans (identity of the dofmap array) plus
    the length of the dof vectors.  An entry is ``None`` when the library refused the dofmap (a dof with more than 255
    entries) or the gather would lose (``_GATHER_MAX_MEAN_ENTRIES``): the caller then takes the atomic path."""

    def __init__(self, capacity: int = 16):
        self._plans = {}
        self._static_only = set()  # keys of plans kept for the static-detJ form alone (the plain gather would lose: P = 2)
        self.capacity = capacity

    def get(self, dofmap: torch.Tensor, ndofs: int, rows=None, static=False):
        """``rows = (row_set, which)``: the plan of the dofs d with ``row_set[d] == which`` only (device uint8[ndofs]; the
        partitioned apply's split into rows next to the exchanges and rows between them).  ``static``: the caller will apply the
        static-detJ form, which pays up to a higher mean number of entries per dof than the plain gather."""
        lib = _lib.load()
        nent, N = dofmap.shape
        key = (dofmap.data_ptr(), nent, N, dofmap._version, dofmap.device.index, int(ndofs))
        if rows is not None:
            row_set, which = rows
            _req(row_set, torch.uint8, "row_set")
            if row_set.numel() != int(ndofs):
                raise ValueError("row_set must have one mark per dof")
            key = key + (row_set.data_ptr(), row_set._version, int(which))
        if key in self._plans:
            hit = self._plans[key]
        else:
            import ctypes as C

            hit = None
            nbytes = lib.fus_mass_gather_plan_bytes(N, nent, int(ndofs))
            if nbytes > 0:
                ws = torch.empty(int(nbytes), dtype=torch.uint8, device=dofmap.device)
                if rows is None:
                    rc = lib.fus_mass_gather_plan_build(dofmap.data_ptr(), N, nent, int(ndofs), ws.data_ptr(), int(nbytes), _lib.stream_ptr())
                else:
                    rc = lib.fus_mass_gather_plan_build_rows(dofmap.data_ptr(), N, nent, int(ndofs), rows[0].data_ptr(), int(rows[1]),
                                                             ws.data_ptr(), int(nbytes), _lib.stream_ptr())
                if rc == 0:
                    info = (C.c_int64 * 4)()
                    _lib.check(lib.fus_mass_gather_plan_info(ws.data_ptr(), info), "fus_mass_gather_plan_info")
                    # (a row subset is judged by the full plan: the caller asks for it only when the full plan was kept)
                    if rows is not None or (info[0] > 0 and nent * N <= _GATHER_STATIC_MAX_MEAN_ENTRIES * info[0]):
                        hit = (ws, dofmap, tuple(int(v) for v in info)) + ((rows[0],) if rows is not None else ())
                        if rows is None and nent * N > _GATHER_MAX_MEAN_ENTRIES * info[0]:
                            self._static_only.add(key)
                    else:
                        lib.fus_plan_release(ws.data_ptr())
                elif rc != _lib.ERR_UNSUPPORTED_ENTITY:
                    _lib.check(rc, "fus_mass_gather_plan_build")
            if len(self._plans) >= self.capacity:
                oldest = next(iter(self._plans))
                old = self._plans.pop(oldest)
                self._static_only.discard(oldest)
                if old is not None:
                    lib.fus_plan_release(old[0].data_ptr())
            self._plans[key] = hit
        if hit is not None and not static and key in self._static_only:
            return None  # the plain gather would lose on this dofmap: the caller takes the atomic batch plan
        if hit is not None and _PLANS._recording is not None:
            _PLANS._recording.append((hit[0], hit[1]))  # a captured graph keeps the workspace alive
        return hit

    def clear(self):
        lib = _lib.load()
        static = globals().get("_STATIC_DETJ")  # row-ordered detJ copies belong to these plans
        if static is not None:
            static.clear()
        for hit in self._plans.values():
            if hit is not None:
                lib.fus_plan_release(hit[0].data_ptr())
        self._plans.clear()
        self._static_only.clear()


_GATHER_PLANS = _GatherPlanCache()


class _Launchable:
    """``obj[grid, block](...)`` == ``obj.launch(...)`` (launch config ignored)."""

    def __getitem__(self, launch_config):
        return self.launch


# --------------------------------------------------------------------------- mass
def _mass_gather_usable(nent, n_per):
    return _USE_GATHER and nent * n_per >= _MASS_PLAN_MIN_ENTRIES and n_per <= 2048 and nent * n_per < 2**31


def mass_rows_available(entity_dofmap, ndofs, row_set):
    """True if ``mass_operator``'s apply can run as two row-subset launches of the atomic-free kernel for this dofmap (the
    full transposed plan is one the operator would use, and both subsets build): what ``HaloApply`` asks before it splits a
    mass apply by dof instead of by cell.  Builds and caches the three plans."""
    nent, n_per = entity_dofmap.shape
    if nent == 0 or not _mass_gather_usable(nent, n_per):
        return False
    if _GATHER_PLANS.get(entity_dofmap, int(ndofs)) is None:
        return False
    return all(_GATHER_PLANS.get(entity_dofmap, int(ndofs), (row_set, w)) is not None for w in (0, 1))


def _mass_apply_rows(x, entity_constants, y, entity_detJ, entity_dofmap, row_set, which, N=None):
    """The rows ``row_set[d] == which`` of ``y += M(c) x`` with the atomic-free kernel (every row sums all its entries)."""
    lib = _lib.load()
    dt = x.dtype if isinstance(x, torch.Tensor) else None
    for name, t in (("x", x), ("entity_constants", entity_constants), ("y", y), ("entity_detJ", entity_detJ)):
        _req(t, dt, name)
    _req(entity_dofmap, torch.int32, "entity_dofmap")
    if entity_dofmap.dim() != 2 or entity_detJ.shape != entity_dofmap.shape:
        raise ValueError("entity_dofmap must be [num_entities, N] and entity_detJ must have the same shape")
    nent, n_per = entity_dofmap.shape
    if (N is not None and n_per != N) or entity_constants.numel() != nent:
        raise ValueError("dofs per entity / number of constants do not match the dofmap")
    if nent == 0:
        return
    hit = _GATHER_PLANS.get(entity_dofmap, min(x.numel(), y.numel()), (row_set, int(which)))
    if hit is None:
        raise _lib.FusGpuError("no row-subset plan for this dofmap (mass_rows_available() says when there is one)")
    fn = getattr(lib, f"fus_mass_apply_gather_{_lib.suffix(dt)}")
    _lib.check(fn(x.data_ptr(), entity_constants.data_ptr(), y.data_ptr(), entity_detJ.data_ptr(), hit[0].data_ptr(), int(n_per), int(nent),
                  _lib.stream_ptr()), "fus_mass_apply_gather (row subset)")


def _mass_apply(x, entity_constants, y, entity_detJ, entity_dofmap, N=None, exclusive=False, atomic=False):
    lib = _lib.load()
    dt = x.dtype if isinstance(x, torch.Tensor) else None
    _req(x, dt, "x")
    _req(entity_constants, dt, "entity_constants")
    _req(y, dt, "y")
    _req(entity_detJ, dt, "entity_detJ")
    _req(entity_dofmap, torch.int32, "entity_dofmap")
    if entity_dofmap.dim() != 2 or entity_detJ.shape != entity_dofmap.shape:
        raise ValueError("entity_dofmap must be [num_entities, N] and entity_detJ must have the same shape")
    nent, n_per = entity_dofmap.shape
    if N is not None and n_per != N:
        raise ValueError(f"operator was built for N={N} dofs per entity, dofmap has {n_per}")
    if entity_constants.numel() != nent:
        raise ValueError("entity_constants must have one value per entity")
    if nent == 0:
        return
    if not atomic and _mass_gather_usable(nent, n_per):
        hit = _GATHER_PLANS.get(entity_dofmap, min(x.numel(), y.numel()))  # every dofmap value is checked against both vectors
        if hit is not None:
            fn = getattr(lib, f"fus_mass_apply_gather_{_lib.suffix(dt)}")
            _lib.check(
                fn(x.data_ptr(), entity_constants.data_ptr(), y.data_ptr(), entity_detJ.data_ptr(), hit[0].data_ptr(),
                   int(n_per), int(nent), _lib.stream_ptr()),
                "fus_mass_apply_gather",
            )
            return
    if _USE_PLAN and 2 <= n_per <= 4096 and nent * n_per >= _MASS_PLAN_MIN_ENTRIES:  # plan batches hold <= 4096 entries
        ws, epb = _PLANS.get(entity_dofmap, exclusive_ndofs=y.numel() if exclusive else None)
        fn = getattr(lib, f"fus_mass_apply_planned_{_lib.suffix(dt)}")
        _lib.check(
            fn(x.data_ptr(), entity_constants.data_ptr(), y.data_ptr(), entity_detJ.data_ptr(), ws.data_ptr(),
               int(n_per), int(epb), int(nent), _lib.stream_ptr()),
            "fus_mass_apply_planned",
        )
        return
    fn = getattr(lib, f"fus_mass_apply_{_lib.suffix(dt)}")
    _lib.check(
        fn(x.data_ptr(), entity_constants.data_ptr(), y.data_ptr(), entity_detJ.data_ptr(), entity_dofmap.data_ptr(),
           int(n_per), int(nent), _lib.stream_ptr()),
        "fus_mass_apply",
    )


def mass_kernel_name(entity_dofmap, ndofs, atomic=False, static=False):
    """Which kernel ``mass_operator``'s apply launches for this dofmap and vector length (bench.py reports it); ``static``: of an operator made
    with ``static_detJ=True`` (it keeps the gather kernel up to more entries per dof: P = 2)."""
    nent, n_per = entity_dofmap.shape
    big = nent * n_per >= _MASS_PLAN_MIN_ENTRIES
    if (_USE_GATHER and not atomic and big and n_per <= 2048 and nent * n_per < 2**31
            and _GATHER_PLANS.get(entity_dofmap, int(ndofs), static=bool(static)) is not None):
        return "fus::mass_gather_kernel"
    return "fus::mass_plan_kernel" if (_USE_PLAN and 2 <= n_per <= 4096 and big) else "fus::mass_kernel"


class _StaticDetJCache:
    """Static companions of transposed-dofmap plans (``fus_mass_gather_static_*``): detJ in row order, keyed on the plan's
    workspace and on the identity of the detJ array (pointer, shape, torch version counter).  An entry is ``None`` when the library
    refused (a block of 256 dofs spanning more than 65 535 entities)."""

    def __init__(self, capacity: int = 8):
        self._entries = {}
        self.capacity = capacity

    def get(self, plan_ws, detJ, n_per, nent):
        lib = _lib.load()
        key = (plan_ws.data_ptr(), detJ.data_ptr(), tuple(detJ.shape), detJ._version, detJ.dtype)
        if key not in self._entries:
            hit = None
            nbytes = lib.fus_mass_gather_static_bytes(int(n_per), int(nent), detJ.element_size())
            if nbytes > 0:
                sws = torch.empty(int(nbytes), dtype=torch.uint8, device=detJ.device)
                fn = getattr(lib, f"fus_mass_gather_static_build_{_lib.suffix(detJ.dtype)}")
                rc = fn(plan_ws.data_ptr(), detJ.data_ptr(), sws.data_ptr(), int(nbytes), _lib.stream_ptr())
                if rc == 0:
                    hit = (sws, plan_ws, detJ)  # holds the plan and detJ: their addresses cannot be re-used while this lives
                elif rc != _lib.ERR_UNSUPPORTED_ENTITY:
                    _lib.check(rc, "fus_mass_gather_static_build")
            if len(self._entries) >= self.capacity:
                old = self._entries.pop(next(iter(self._entries)))
                if old is not None:
                    lib.fus_plan_release(old[0].data_ptr())
            self._entries[key] = hit
        hit = self._entries[key]
        if hit is not None and _PLANS._recording is not None:
            _PLANS._recording.append((hit[0], hit[2]))
        return hit

    def clear(self):
        lib = _lib.load()
        for hit in self._entries.values():
            if hit is not None:
                lib.fus_plan_release(hit[0].data_ptr())
        self._entries.clear()


_STATIC_DETJ = _StaticDetJCache()


class _MassApply:
    """``operator(x, entity_constants, y, entity_detJ, entity_dofmap)`` returned by ``mass_operator(N, float_type)``;
    ``.atomic``: the same operator on the float-atomic kernels (safe next to concurrent writers of ``y``)."""

    def __init__(self, N, tdt, exclusive, atomic, static_detJ=False):
        self.N, self.dtype, self._exclusive, self._atomic, self._static = N, tdt, exclusive, atomic, bool(static_detJ) and not atomic
        self.atomic = self if atomic else _MassApply(N, tdt, exclusive, True)

    def __call__(self, x, entity_constants, y, entity_detJ, entity_dofmap):
        if isinstance(x, torch.Tensor) and x.dtype != self.dtype:
            raise TypeError(f"x: expected dtype {self.dtype}, got {x.dtype}")
        if self._static and self._apply_static(x, entity_constants, y, entity_detJ, entity_dofmap):
            return
        _mass_apply(x, entity_constants, y, entity_detJ, entity_dofmap, self.N, exclusive=self._exclusive, atomic=self._atomic)

    def _apply_static(self, x, entity_constants, y, entity_detJ, entity_dofmap):
        """The apply with detJ streamed in row order (``static_detJ=True``); False when this dofmap has no transposed plan or no
        static companion (the caller then takes the default path: same result)."""
        dt = self.dtype
        for name, t in (("x", x), ("entity_constants", entity_constants), ("y", y), ("entity_detJ", entity_detJ)):
            _req(t, dt, name)
        _req(entity_dofmap, torch.int32, "entity_dofmap")
        if entity_dofmap.dim() != 2 or entity_detJ.shape != entity_dofmap.shape:
            raise ValueError("entity_dofmap must be [num_entities, N] and entity_detJ must have the same shape")
        nent, n_per = entity_dofmap.shape
        if n_per != self.N or entity_constants.numel() != nent:
            raise ValueError("dofs per entity / number of constants do not match the dofmap")
        if nent == 0 or not _mass_gather_usable(nent, n_per):
            return False
        plan = _GATHER_PLANS.get(entity_dofmap, min(x.numel(), y.numel()), static=True)
        if plan is None:
            return False
        st = _STATIC_DETJ.get(plan[0], entity_detJ, n_per, nent)
        if st is None:
            return False
        fn = getattr(_lib.load(), f"fus_mass_apply_gather_static_{_lib.suffix(dt)}")
        _lib.check(fn(x.data_ptr(), entity_constants.data_ptr(), y.data_ptr(), plan[0].data_ptr(), st[0].data_ptr(), int(n_per), int(nent),
                      _lib.stream_ptr()), "fus_mass_apply_gather_static")
        return True

    def refresh(self):
        """``static_detJ=True``: forget the row-ordered copies of detJ.  REQUIRED after changing a detJ array in place through
        anything that writes through ``data_ptr()`` -- which includes THIS package's own vector ops (``fill`` / ``copy`` / ``axpy`` /
        ``pointwise_divide``) and every other C-ABI kernel (``compute_scaled_jacobian_determinant_device`` into the same array):
        they do not bump torch's version counter, the only change the cache notices by itself (torch's own in-place operations)."""
        _STATIC_DETJ.clear()

    def apply_rows(self, x, entity_constants, y, entity_detJ, entity_dofmap, row_set, which):
        """``y[d] += (M(c) x)[d]`` for the dofs with ``row_set[d] == which`` only (atomic-free kernel): the two halves of a
        partitioned apply (``HaloApply``), which never add into one ``y[d]`` concurrently."""
        if isinstance(x, torch.Tensor) and x.dtype != self.dtype:
            raise TypeError(f"x: expected dtype {self.dtype}, got {x.dtype}")
        _mass_apply_rows(x, entity_constants, y, entity_detJ, entity_dofmap, row_set, which, self.N)

    def rows_available(self, entity_dofmap, ndofs, row_set):
        return (not self._atomic) and mass_rows_available(entity_dofmap, ndofs, row_set)


class _MassOperator(_Launchable):
    def __call__(self, N: int, float_type, exclusive=False, atomic=False, static_detJ=False):
        """``mass_operator(N, float_type)`` -> ``operator(x, entity_constants, y, entity_detJ, entity_dofmap)``
        (numba-cpu/operators.py:19-68).

        Default kernel (csrc/mass_gather.hpp): ONE thread per touched dof sums the dof's (entity, local index) entries from
        the transposed dofmap in the order of the reference's serial loop and finishes ``y[dof]`` with a plain load +
        store -- no float atomics (their request rate bounds the atomic kernels at 0.44-0.45 of the HBM roofline), bitwise
        reproducible.  Like the reference's CPU operator (a plain ``+=``), such a launch assumes NOTHING ELSE adds into ``y``
        while it runs (earlier and later launches of the same stream are fine).  ``atomic=True`` (keyword, or the ``.atomic``
        attribute of the returned operator): the float-atomic kernels of cuda/operators.py:66-70's behaviour, safe next to
        another stream's launch or a halo receive adding into the same ``y`` -- what ``HaloApply`` uses for its overlapped
        sub-launches.  Dofmaps the gather does not pay for (P = 2: 3.4 entries per dof) or cannot hold (a dof in more than 255
        entities) take the atomic batch plan by themselves.
        ``exclusive=True`` (atomic batch plan with exclusive-dof marks, round 4's first attempt): kept for the atomic path,
        measured slower than the unmarked plan.
        ``static_detJ=True`` (opt-in): the caller declares ``entity_detJ`` constant across applies -- it is what the reference's
        drivers do (one detJ array for the whole run, cuda/demo_nonlinear_bowl.py:603-632) -- and the operator keeps a copy of it
        in ROW order next to the transposed dofmap: the kernel streams detJ instead of gathering it through the entry ids
        (bitwise the same result).  The copy is a SNAPSHOT: call ``op.refresh()`` after writing into detJ with anything but torch's
        own in-place operations -- this package's vector ops and device precompute write through ``data_ptr()`` and are NOT noticed.
        The constants are read per apply and may change."""
        return _MassApply(int(N), _lib.torch_dtype(float_type), exclusive, atomic, static_detJ)

    @staticmethod
    def launch(x, entity_constants, y, detJ_entity, entity_dofmap):
        _mass_apply(x, entity_constants, y, detJ_entity, entity_dofmap)


mass_operator = _MassOperator()


class DiagonalMassOperator:
    """The cell mass apply in CACHED-DIAGONAL form (opt-in; own bytes contract: 3 vector touches per dof).

    With GLL collocation the operator of numba-cpu/operators.py:19-68 is diagonal:  M(c) x = (M(c) 1) (.) x.
    ``DiagonalMassOperator(entity_constants, entity_detJ, entity_dofmap, ndofs)`` assembles ``w = M(c) 1`` once with the
    reference-compatible ``mass_operator`` (so ``w`` carries exactly its rounding) and ``op(x, y)`` then does
    ``y += w * x`` (``fus_muladd_*``).  The reference's drivers re-apply the gather / scatter form on every use
    (cuda/demo_nonlinear_bowl.py:603-632); the Westervelt solver of this package uses the same identity for its two
    mass terms.  ``w`` is valid for the (constants, detJ, dofmap) it was built from: ``refresh()`` after changing them.
    On a partitioned mesh ``w`` holds this rank's cells' contributions, like ``y`` after the cell operator (reverse-scatter
    either ``w`` once or ``y`` every time)."""

    def __init__(self, entity_constants, entity_detJ, entity_dofmap, ndofs, float_type=None):
        dt = entity_detJ.dtype if float_type is None else _lib.torch_dtype(float_type)
        _req(entity_constants, dt, "entity_constants")
        _req(entity_detJ, dt, "entity_detJ")
        _req(entity_dofmap, torch.int32, "entity_dofmap")
        self.dtype, self.ndofs = dt, int(ndofs)
        self._args = (entity_constants, entity_detJ, entity_dofmap)
        self.w = torch.zeros(self.ndofs, dtype=dt, device=entity_detJ.device)
        self._fn = getattr(_lib.load(), f"fus_muladd_{_lib.suffix(dt)}")
        self.refresh()

    def refresh(self):
        cc, detJ, dm = self._args
        self.w.zero_()
        _mass_apply(torch.ones(self.ndofs, dtype=self.dtype, device=self.w.device), cc, self.w, detJ, dm)

    def __call__(self, x, y):
        _req(x, self.dtype, "x")
        _req(y, self.dtype, "y")
        if x.numel() != self.ndofs or y.numel() != self.ndofs:
            raise ValueError(f"x and y must have {self.ndofs} entries")
        _lib.check(self._fn(self.w.data_ptr(), x.data_ptr(), y.data_ptr(), self.ndofs, _lib.stream_ptr()), "fus_muladd")


def diagonal_mass_operator(entity_constants, entity_detJ, entity_dofmap, ndofs, float_type=None):
    return DiagonalMassOperator(entity_constants, entity_detJ, entity_dofmap, ndofs, float_type)


def facet_terms(y, source, field, scalars=None):
    """The boundary-facet terms of one RK4 stage in one launch (csrc/mass.hpp, ``fus_facet_terms_*``):

        source = (c1, s1, c2, s2, detJ_f, facet_dofmap)   y += M_f(s1 c1 + s2 c2) 1      (c2 may be None)
        field  = (x, c, detJ_f, facet_dofmap)             y += M_f(c) x

    ``scalars``: device tensor holding (s1, s2) -- they are then read from device memory by the kernel
    (``fus_facet_terms_dev_*``; the s1, s2 of ``source`` are ignored), which is what lets a captured time
    step be replayed as a hipGraph with new source values.

    i.e. ``mass_operator(g, facet_coeff1, b, ...)`` [+ the dg term] and ``mass_operator(v_n, facet_coeff2, b, ...)``
    of cuda/demo_linear_box.py:546-549 / cuda/demo_nonlinear_bowl.py:633-641 without filling g into a vector."""
    c1, s1, c2, s2, dA, dmA = source
    xB, cB, dB, dmB = field
    dt = y.dtype if isinstance(y, torch.Tensor) else None
    _req(y, dt, "y")
    for name, t in (("c1", c1), ("detJ_source", dA), ("x", xB), ("c", cB), ("detJ_field", dB)):
        _req(t, dt, name)
    if c2 is not None:
        _req(c2, dt, "c2")
    _req(dmA, torch.int32, "source dofmap")
    _req(dmB, torch.int32, "field dofmap")
    nA, nB = dmA.shape[0], dmB.shape[0]
    N = dmA.shape[1] if nA else (dmB.shape[1] if nB else 1)
    if (nA and (dA.shape != dmA.shape or c1.numel() != nA)) or (nB and (dB.shape != dmB.shape or cB.numel() != nB)):
        raise ValueError("facet arrays: detJ must have the dofmap's shape, one constant per facet")
    if nA and nB and dmA.shape[1] != dmB.shape[1]:
        raise ValueError("both facet sets must have the same number of dofs per facet")
    if nA + nB == 0:
        return
    if scalars is not None:
        _req(scalars, dt, "scalars")
        if scalars.numel() < 2:
            raise ValueError("scalars must hold (s1, s2)")
        fn = getattr(_lib.load(), f"fus_facet_terms_dev_{_lib.suffix(dt)}")
        _lib.check(
            fn(y.data_ptr(), c1.data_ptr() if nA else None, c2.data_ptr() if (nA and c2 is not None) else None, scalars.data_ptr(),
               dA.data_ptr() if nA else None, dmA.data_ptr() if nA else None, int(nA), xB.data_ptr() if nB else None,
               cB.data_ptr() if nB else None, dB.data_ptr() if nB else None, dmB.data_ptr() if nB else None, int(nB), int(N),
               _lib.stream_ptr()),
            "fus_facet_terms_dev",
        )
        return
    fn = getattr(_lib.load(), f"fus_facet_terms_{_lib.suffix(dt)}")
    _lib.check(
        fn(y.data_ptr(), c1.data_ptr() if nA else None, float(s1), c2.data_ptr() if (nA and c2 is not None) else None, float(s2),
           dA.data_ptr() if nA else None, dmA.data_ptr() if nA else None, int(nA), xB.data_ptr() if nB else None,
           cB.data_ptr() if nB else None, dB.data_ptr() if nB else None, dmB.data_ptr() if nB else None, int(nB), int(N),
           _lib.stream_ptr()),
        "fus_facet_terms",
    )


# ---------------------------------------------------------------------- stiffness
class _StiffnessOperator(_Launchable):
    """Returned by ``stiffness_operator``; callable both ways."""

    def __init__(self, P: int, float_type, dphi=None, affine_weights=None, geometry=None):
        self.P = int(P)
        if not (1 <= self.P <= 10):
            raise ValueError(f"polynomial degree {P} outside the supported range 1..10")
        self.n = self.P + 1
        self.dtype = _lib.torch_dtype(float_type)
        self._fn = getattr(_lib.load(), f"fus_stiffness_apply_{_lib.suffix(self.dtype)}")
        self._fn_planned = getattr(_lib.load(), f"fus_stiffness_apply_planned_{_lib.suffix(self.dtype)}")
        self._dphi = None
        self._dphi_src = None
        if dphi is not None:
            self._dphi = self._table(dphi)
        # opt-in affine-cell fast path: tensor quadrature weights [n^3] of the rule G was built with
        self._wratio = None
        if affine_weights is not None:
            w = np.asarray(affine_weights, dtype=np.float64).reshape(-1)
            if w.size != self.n**3:
                raise ValueError(f"affine_weights must hold the {self.n ** 3} tensor quadrature weights")
            dev = torch.device("cuda", torch.cuda.current_device())
            self._wratio = torch.from_numpy(w / w[0]).to(device=dev, dtype=self.dtype).contiguous()
            self._fn_affine = getattr(_lib.load(), f"fus_stiffness_apply_planned_affine_{_lib.suffix(self.dtype)}")

        # opt-in in-kernel geometry: (x_dofs int32[ncell, 8], x_g T[nvert, 3], pts T[n], wts T[n]);
        # the G argument of the apply is then ignored (may be None)
        self._geom = None
        if geometry is not None:
            x_dofs, x_g, pts, wts = geometry
            dev = torch.device("cuda", torch.cuda.current_device())

            def dev_t(a, dtype):
                t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(a)))
                return t.to(device=dev, dtype=dtype).contiguous()

            xd, xg = dev_t(x_dofs, torch.int32), dev_t(x_g, self.dtype)
            pt, wt = dev_t(pts, self.dtype).reshape(-1), dev_t(wts, self.dtype).reshape(-1)
            if xd.dim() != 2 or xd.shape[1] != 8 or xg.dim() != 2 or xg.shape[1] != 3:
                raise ValueError("geometry: x_dofs must be [ncell, 8] and x_g [nvert, 3] (P1 hexahedra)")
            if pt.numel() != self.n or wt.numel() != self.n:
                raise ValueError(f"geometry: pts / wts must hold the {self.n} 1-D GLL points / weights")
            self._geom = (xd, xg, pt, wt)
            self._fn_geom = getattr(_lib.load(), f"fus_stiffness_apply_planned_geom_{_lib.suffix(self.dtype)}")

    def _table(self, dphi):
        """Accept the flat ``[q*n+i]`` (numba-cpu) or 2-D ``[q, i]`` (cuda) table, host or device."""
        if isinstance(dphi, torch.Tensor):
            t = dphi
        else:
            t = torch.from_numpy(np.ascontiguousarray(np.asarray(dphi)))
        if t.numel() != self.n * self.n:
            raise ValueError(f"dphi must have {self.n * self.n} entries for P={self.P}, got {t.numel()}")
        dev = torch.device("cuda", torch.cuda.current_device())
        return t.to(device=dev, dtype=self.dtype).contiguous().reshape(-1)

    def _apply(self, x, cell_constants, y, G, dofmap, dphi_t):
        dt = self.dtype
        _req(x, dt, "x")
        _req(cell_constants, dt, "cell_constants")
        _req(y, dt, "y")
        if self._geom is None:
            _req(G, dt, "G")  # with geometry=: G is ignored (None), or carries the x_dofs rows of a cell sub-range
        _req(dofmap, torch.int32, "dofmap")
        nd = self.n**3
        if dofmap.dim() != 2 or dofmap.shape[1] != nd:
            raise ValueError(f"dofmap must be [ncell, {nd}] for P={self.P}")
        ncell = dofmap.shape[0]
        if self._geom is None and G.numel() != ncell * nd * 6:
            raise ValueError(f"G must be [ncell, {nd}, 6]")
        if cell_constants.numel() != ncell:
            raise ValueError("cell_constants must have one value per cell")
        if ncell == 0:
            return
        if self._geom is not None:
            xd, xg, pt, wt = self._geom
            # a cell sub-range hands its rows of x_dofs in the G position (int32 [ncell, 8]: cannot be
            # mistaken for a geometric-factor array)
            if isinstance(G, torch.Tensor) and G.dtype == torch.int32:
                _req(G, torch.int32, "x_dofs")
                if tuple(G.shape) != (ncell, 8):
                    raise ValueError(f"x_dofs must be [{ncell}, 8]")
                xd = G
            if xd.shape[0] != ncell:
                raise ValueError(f"geometry: x_dofs has {xd.shape[0]} cells, dofmap has {ncell}")
            ws, _ = _PLANS.get(dofmap, strips=True)
            _lib.check(
                self._fn_geom(x.data_ptr(), cell_constants.data_ptr(), y.data_ptr(), xg.data_ptr(), xd.data_ptr(),
                              pt.data_ptr(), wt.data_ptr(), ws.data_ptr(), dphi_t.data_ptr(), self.P, int(ncell),
                              _lib.stream_ptr()),
                "fus_stiffness_apply_planned_geom",
            )
        elif self._wratio is not None:
            ws, _ = _PLANS.get(dofmap, strips=True)
            _lib.check(
                self._fn_affine(x.data_ptr(), cell_constants.data_ptr(), y.data_ptr(), G.data_ptr(),
                                self._wratio.data_ptr(), ws.data_ptr(), dphi_t.data_ptr(), self.P, int(ncell),
                                _lib.stream_ptr()),
                "fus_stiffness_apply_planned_affine",
            )
        elif _USE_PLAN:
            ws, _ = _PLANS.get(dofmap)
            _lib.check(
                self._fn_planned(x.data_ptr(), cell_constants.data_ptr(), y.data_ptr(), G.data_ptr(), ws.data_ptr(),
                                 dphi_t.data_ptr(), self.P, int(ncell), _lib.stream_ptr()),
                "fus_stiffness_apply_planned",
            )
        else:
            _lib.check(
                self._fn(x.data_ptr(), cell_constants.data_ptr(), y.data_ptr(), G.data_ptr(), dofmap.data_ptr(),
                         dphi_t.data_ptr(), self.P, int(ncell), _lib.stream_ptr()),
                "fus_stiffness_apply",
            )

    def prepare(self, dofmap):
        """Set-up, not an apply: build (and cache) the batch plan for ``dofmap`` now instead of
        lazily inside the first apply."""
        _req(dofmap, torch.int32, "dofmap")
        if _USE_PLAN and dofmap.shape[0] > 0:
            _PLANS.get(dofmap, strips=self._geom is not None or self._wratio is not None)

    # numba-cpu flavour: op(x, cell_constants, y, G, dofmap)
    def __call__(self, x, cell_constants, y, G, dofmap):
        if self._dphi is None:
            raise TypeError("this operator was built cuda-style (no dphi); launch it as op[grid, block](..., dphi)")
        self._apply(x, cell_constants, y, G, dofmap, self._dphi)

    # cuda flavour: op[grid, block](x, consts, y, G, dofmap, dphi)
    def launch(self, x, entity_constants, y, G_entity, entity_dofmap, dphi):
        if dphi is not self._dphi_src:  # convert/cache the table once per distinct object
            self._dphi_cuda = self._table(dphi)
            self._dphi_src = dphi
        self._apply(x, entity_constants, y, G_entity, entity_dofmap, self._dphi_cuda)


def stiffness_operator(P, *args, affine_weights=None, geometry=None):
    """``stiffness_operator(P, dphi, float_type)`` (numba-cpu/operators.py:71) or
    ``stiffness_operator(P, float_type)`` (cuda/operators.py:73).

    ``affine_weights`` (keyword, no reference counterpart): opt into the affine-cell fast path by
    passing the tensor quadrature weights ``[n^3]``; the operator then reads only ``G[c, 0, :]`` of
    each cell.  Only valid when every cell is affine (``is_affine_geometry`` checks).

    ``geometry`` (keyword, no reference counterpart): ``(x_dofs, x_g, pts, wts)`` -- the reference's
    mesh pair of numba-cpu/precompute.py:115 plus the 1-D GLL points / weights; the operator then
    forms G in the kernel from the 8 vertices of each (trilinear) cell and ignores its ``G``
    argument.  ``x_dofs`` rows must be in the dofmap's cell order."""
    if len(args) == 2:
        dphi, float_type = args
        return _StiffnessOperator(P, float_type, dphi, affine_weights, geometry)
    if len(args) == 1:
        return _StiffnessOperator(P, args[0], None, affine_weights, geometry)
    raise TypeError("stiffness_operator(P, dphi, float_type) or stiffness_operator(P, float_type)")


class _WesterveltCellOperator:
    """Fused Westervelt cell pass (csrc/westervelt.hpp): ``b += K(c3) u + K(c4) v + M(c5) v^2`` and
    ``m += M(c2) u`` in one sweep over the cells -- the four cell launches of
    cuda/demo_nonlinear_bowl.py:612-632 (+ square :603).  No reference counterpart as a single call."""

    def __init__(self, P, dphi, float_type):
        self._st = _StiffnessOperator(P, float_type, dphi)  # reuses table conversion + checks
        self.P, self.n, self.dtype = self._st.P, self._st.n, self._st.dtype
        self._fn = getattr(_lib.load(), f"fus_westervelt_cell_apply_planned_{_lib.suffix(self.dtype)}")

    def __call__(self, u, v, c2, c3, c4, c5, b, m, G, detJ, dofmap):
        dt = self.dtype
        for name, t in (("u", u), ("v", v), ("c2", c2), ("c3", c3), ("c4", c4), ("c5", c5), ("b", b), ("m", m),
                        ("G", G), ("detJ", detJ)):
            _req(t, dt, name)
        _req(dofmap, torch.int32, "dofmap")
        nd = self.n**3
        ncell = dofmap.shape[0]
        if dofmap.dim() != 2 or dofmap.shape[1] != nd or G.numel() != ncell * nd * 6 or detJ.numel() != ncell * nd:
            raise ValueError(f"dofmap [ncell, {nd}], G [ncell, {nd}, 6], detJ [ncell, {nd}] expected")
        for name, t in (("c2", c2), ("c3", c3), ("c4", c4), ("c5", c5)):
            if t.numel() != ncell:
                raise ValueError(f"{name} must have one value per cell")
        if ncell == 0:
            return
        ws, _ = _PLANS.get(dofmap)
        _lib.check(
            self._fn(u.data_ptr(), v.data_ptr(), c2.data_ptr(), c3.data_ptr(), c4.data_ptr(), c5.data_ptr(),
                     b.data_ptr(), m.data_ptr(), G.data_ptr(), detJ.data_ptr(), ws.data_ptr(),
                     self._st._dphi.data_ptr(), self.P, int(ncell), _lib.stream_ptr()),
            "fus_westervelt_cell_apply_planned",
        )


class _WesterveltCellGeomOperator:
    """The fused Westervelt cell pass with G and detJ formed in the kernel from the cell vertices
    (csrc/westervelt_geom.hpp): ``op(u, v, c2, c3, c4, c5, b, m, x_dofs, dofmap)``.  ``x_dofs`` (int32
    [ncell, 8], dofmap cell order) is a call argument so that cell sub-ranges can be passed as views;
    ``x_g``, ``pts``, ``wts`` are fixed at construction."""

    def __init__(self, P, dphi, float_type, x_g, pts, wts):
        self._st = _StiffnessOperator(P, float_type, dphi)
        self.P, self.n, self.dtype = self._st.P, self._st.n, self._st.dtype
        dev = torch.device("cuda", torch.cuda.current_device())
        conv = lambda a: (a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(a)))).to(  # noqa: E731
            device=dev, dtype=self.dtype).contiguous()
        self.x_g, self.pts, self.wts = conv(x_g), conv(pts).reshape(-1), conv(wts).reshape(-1)
        if self.x_g.dim() != 2 or self.x_g.shape[1] != 3 or self.pts.numel() != self.n or self.wts.numel() != self.n:
            raise ValueError("x_g must be [nvert, 3]; pts / wts the 1-D GLL points / weights")
        self._fn = getattr(_lib.load(), f"fus_westervelt_cell_apply_planned_geom_{_lib.suffix(self.dtype)}")

    def __call__(self, u, v, c2, c3, c4, c5, b, m, x_dofs, dofmap):
        dt = self.dtype
        for name, t in (("u", u), ("v", v), ("c2", c2), ("c3", c3), ("c4", c4), ("c5", c5), ("b", b), ("m", m)):
            _req(t, dt, name)
        _req(dofmap, torch.int32, "dofmap")
        _req(x_dofs, torch.int32, "x_dofs")
        nd = self.n**3
        ncell = dofmap.shape[0]
        if dofmap.dim() != 2 or dofmap.shape[1] != nd or tuple(x_dofs.shape) != (ncell, 8):
            raise ValueError(f"dofmap [ncell, {nd}] and x_dofs [ncell, 8] expected")
        for name, t in (("c2", c2), ("c3", c3), ("c4", c4), ("c5", c5)):
            if t.numel() != ncell:
                raise ValueError(f"{name} must have one value per cell")
        if ncell == 0:
            return
        ws, _ = _PLANS.get(dofmap, strips=True)
        _lib.check(
            self._fn(u.data_ptr(), v.data_ptr(), c2.data_ptr(), c3.data_ptr(), c4.data_ptr(), c5.data_ptr(),
                     b.data_ptr(), m.data_ptr(), self.x_g.data_ptr(), x_dofs.data_ptr(), self.pts.data_ptr(),
                     self.wts.data_ptr(), ws.data_ptr(), self._st._dphi.data_ptr(), self.P, int(ncell), _lib.stream_ptr()),
            "fus_westervelt_cell_apply_planned_geom",
        )


def westervelt_cell_operator(P, dphi, float_type, geometry=None):
    """``geometry=(x_g, pts, wts)``: the variant that forms G and detJ in the kernel (its call takes
    ``x_dofs`` in place of ``G, detJ``)."""
    if geometry is not None:
        return _WesterveltCellGeomOperator(P, dphi, float_type, *geometry)
    return _WesterveltCellOperator(P, dphi, float_type)


def _westervelt_stiffness_only(self, u, v, c3, c4, b, G_or_xdofs, dofmap, geom):
    dt = self.dtype
    for name, t in (("u", u), ("v", v), ("c3", c3), ("c4", c4), ("b", b)):
        _req(t, dt, name)
    _req(dofmap, torch.int32, "dofmap")
    nd = self.n**3
    ncell = dofmap.shape[0]
    if dofmap.dim() != 2 or dofmap.shape[1] != nd or c3.numel() != ncell or c4.numel() != ncell:
        raise ValueError(f"dofmap [ncell, {nd}] and one c3 / c4 value per cell expected")
    if ncell == 0:
        return
    ws, _ = _PLANS.get(dofmap, strips=geom)
    if geom:
        _req(G_or_xdofs, torch.int32, "x_dofs")
        if tuple(G_or_xdofs.shape) != (ncell, 8):
            raise ValueError("x_dofs [ncell, 8] expected")
        rc = self._fn(u.data_ptr(), v.data_ptr(), None, c3.data_ptr(), c4.data_ptr(), None, b.data_ptr(), None,
                      self.x_g.data_ptr(), G_or_xdofs.data_ptr(), self.pts.data_ptr(), self.wts.data_ptr(), ws.data_ptr(),
                      self._st._dphi.data_ptr(), self.P, int(ncell), _lib.stream_ptr())
    else:
        _req(G_or_xdofs, dt, "G")
        if G_or_xdofs.numel() != ncell * nd * 6:
            raise ValueError(f"G [ncell, {nd}, 6] expected")
        rc = self._fn(u.data_ptr(), v.data_ptr(), None, c3.data_ptr(), c4.data_ptr(), None, b.data_ptr(), None,
                      G_or_xdofs.data_ptr(), None, ws.data_ptr(), self._st._dphi.data_ptr(), self.P, int(ncell),
                      _lib.stream_ptr())
    _lib.check(rc, "fus_westervelt_cell_apply_planned (stiffness part)")


def _stiffness_only_general(self, u, v, c3, c4, b, G, dofmap):
    """``b += K(c3) u + K(c4) v`` in one pass over the cells (G read once, u and v gathered once): the
    stiffness part of the Westervelt stage; the mass terms are applied pointwise by the driver from
    precomputed diagonals (``fus_rk4_stage_nl2_*``)."""
    _westervelt_stiffness_only(self, u, v, c3, c4, b, G, dofmap, False)


def _stiffness_only_geom(self, u, v, c3, c4, b, x_dofs, dofmap):
    """As above with G formed in the kernel from the cell vertices."""
    _westervelt_stiffness_only(self, u, v, c3, c4, b, x_dofs, dofmap, True)


_WesterveltCellOperator.stiffness_only = _stiffness_only_general
_WesterveltCellGeomOperator.stiffness_only = _stiffness_only_geom


def locality_cell_order(dofmap):
    """Set-up helper: permutation of the cells (int64 tensor on the dofmap's device) that puts cells
    with nearby dofs next to each other -- cells sorted by their smallest dof.  The planned kernels
    handle ANY cell order correctly; their speed depends on how many distinct dofs the 256 // n^2
    consecutive cells of a batch touch (tools/exp_numbering.py), which a mesh whose cell order is
    unrelated to its dof numbering loses.  The plan cache applies this order by itself, as an index
    indirection inside the plan (``use_locality_order``); a driver may instead apply it once to every
    per-cell array (``dofmap[perm]``, ``G[perm]``, ``cell_constants[perm]``, ``detJ[perm]``)."""
    _req(dofmap, torch.int32, "dofmap")
    return torch.argsort(dofmap.min(dim=1).values, stable=True)


def is_affine_geometry(G, weights, rtol=1e-12):
    """True if ``G[c, q, :] == G[c, 0, :] * w_q / w_0`` for every cell (set-up check for the
    affine fast path; one pass over G with torch, not part of the apply)."""
    w = torch.as_tensor(np.asarray(weights, dtype=np.float64).reshape(-1), device=G.device).to(G.dtype)
    Gv = G.reshape(G.shape[0], -1, 6)
    ref = Gv[:, :1, :] * (w / w[0]).reshape(1, -1, 1)
    scale = Gv.abs().amax()
    return bool(((Gv - ref).abs().amax() <= rtol * scale).item())


# -------------------------------------------------------------------- vector ops
def _vec(name, *tensors):
    dt = tensors[0].dtype if isinstance(tensors[0], torch.Tensor) else None
    for i, t in enumerate(tensors):
        _req(t, dt, f"arg{i}")
    n = tensors[0].numel()
    return getattr(_lib.load(), f"fus_{name}_{_lib.suffix(dt)}"), n


def _axpy(alpha, x, y, n=None):
    fn, size = _vec("axpy", x, y)
    n = min(x.numel(), y.numel()) if n is None else int(n)
    if n > x.numel() or n > y.numel():
        raise ValueError("axpy: n exceeds the vector length")
    _lib.check(fn(float(alpha), x.data_ptr(), y.data_ptr(), n, _lib.stream_ptr()), "fus_axpy")


class _Axpy(_Launchable):
    def __call__(self, local_size: int):
        n = int(local_size)

        def kernel(alpha, x, y):
            _axpy(alpha, x, y, n)

        return kernel

    @staticmethod
    def launch(alpha, x, y):
        _axpy(alpha, x, y)


class _Copy(_Launchable):
    @staticmethod
    def launch(a, b):
        fn, n = _vec("copy", a, b)
        if b.numel() < n:
            raise ValueError("copy: output shorter than input")
        _lib.check(fn(a.data_ptr(), b.data_ptr(), n, _lib.stream_ptr()), "fus_copy")

    __call__ = launch


class _Fill(_Launchable):
    @staticmethod
    def launch(alpha, x):
        fn, n = _vec("fill", x)
        _lib.check(fn(float(alpha), x.data_ptr(), n, _lib.stream_ptr()), "fus_fill")

    __call__ = launch


class _PointwiseDivide(_Launchable):
    @staticmethod
    def launch(a, b, c):
        fn, _ = _vec("pointwise_divide", a, b, c)
        n = c.numel()
        if a.numel() < n or b.numel() < n:
            raise ValueError("pointwise_divide: inputs shorter than output")
        _lib.check(fn(a.data_ptr(), b.data_ptr(), c.data_ptr(), n, _lib.stream_ptr()), "fus_pointwise_divide")

    __call__ = launch


class _Square(_Launchable):
    @staticmethod
    def launch(a, b):
        fn, n = _vec("square", a, b)
        if b.numel() < n:
            raise ValueError("square: output shorter than input")
        _lib.check(fn(a.data_ptr(), b.data_ptr(), n, _lib.stream_ptr()), "fus_square")

    __call__ = launch


class _Scale(_Launchable):
    """b = alpha a  (no reference kernel; the drivers here use it for per-stage facet constants)."""

    @staticmethod
    def launch(alpha, a, b):
        fn, n = _vec("scale", a, b)
        if b.numel() < n:
            raise ValueError("scale: output shorter than input")
        _lib.check(fn(float(alpha), a.data_ptr(), b.data_ptr(), n, _lib.stream_ptr()), "fus_scale")

    __call__ = launch


axpy = _Axpy()
scale = _Scale()
copy = _Copy()
fill = _Fill()
pointwise_divide = _PointwiseDivide()
square = _Square()
