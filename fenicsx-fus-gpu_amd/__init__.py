"""
fenicsx-fus-gpu_amd: MI355X-native (gfx950) matrix-free operator-application
path for the FEniCSx-FUS acoustic wave solver.

The directory name is not a Python identifier; load it through
``fusgpu_loader.load()`` at the repo root (or put this directory on
``sys.path`` and ``from operators import ...`` exactly like the reference's
flat scripts do, e.g. cuda/demo_linear_box.py:27-37).

Modules
-------
operators   mass / stiffness / vector-op call surface (numba-cpu factories and
            cuda ``kernel[grid, block](...)`` launch style) over the C ABI
scatterer   scatter_forward / scatter_reverse (halo exchange)
utils       compute_scatterer_data
precompute  detJ / G / facet detJ (host)
boxmesh     synthetic structured hex meshes + block partitioning
gll         GLL nodes, weights, derivative tables
device      numba.cuda-like shim over torch tensors (to_device, ...)
_lib        ctypes binding of csrc/libfusgpu.so (fails loudly if missing)
"""

__version__ = "0.1.0"
