"""``compute_scatterer_data`` against the reference's own function.

tests/golden/halo_plan_*.npz hold what /root/reference/cuda/utils.py:8-78 returned, rank by rank, on
BoxMesh index maps (owner-grouped, lexicographic and randomly permuted ghost numberings); the
generator (tests/golden/generate_golden.py --only plan) imports that file unmodified with stub
``mpi4py`` / ``dolfinx`` modules and an in-process MPI.COMM_WORLD.

Integer work: the bar is bit-exact.
  * ``stable=False`` (the reference's ``np.argsort(owners)``, cuda/utils.py:28): every field, element for
    element -- ``unique_*``, ``*_size``, ``owners_idx``, ``ghosts_idx``.
  * ``stable=True`` (this repo's default: a stable sort, so owner-grouped ghosts yield the identity list
    and the exchange can use the ghost block as its message buffer): ``unique_*`` and ``*_size``
    element for element, the index lists group by group as SETS.  The one freedom the reference leaves
    is the order inside one owner's group (numpy's default introsort is not stable): both sides of an
    exchange use the same order (the ghosting rank sends its order to the owner, :54-73), so any
    within-group order is a valid plan; ``test_pairing`` checks the invariant that makes it so."""
import numpy as np
import pytest

from conftest import golden_files, pkg


def _meshes(d):
    boxmesh = pkg("boxmesh")
    P, shape, grid = int(d["P"]), tuple(int(v) for v in d["shape"]), tuple(int(v) for v in d["grid"])
    go = str(d["ghost_order"])
    go = go if go in ("owner", "lex") else int(go)
    R = int(np.prod(grid))
    return [boxmesh.BoxMesh(P, shape, grid=grid, rank=r, ghost_order=go) for r in range(R)]


@pytest.mark.parametrize("stable", [False, True], ids=["reference-sort", "stable-sort"])
@pytest.mark.parametrize("path", golden_files("halo_plan_"), ids=lambda p: p.split("/")[-1][:-4])
def test_plan_equals_reference(path, stable):
    utils = pkg("utils")
    d = np.load(path)
    meshes = _meshes(d)
    od, gd = utils.compute_scatterer_data_all([m.index_map for m in meshes], stable=stable)
    for r in range(len(meshes)):
        o_idx, o_size, o_off, o_ranks = od[r]
        g_idx, g_size, g_off, g_ranks = gd[r]
        assert np.array_equal(o_ranks, d[f"unique_owners_{r}"])
        assert np.array_equal(o_size, d[f"owners_size_{r}"])
        assert np.array_equal(g_ranks, d[f"unique_ghosts_{r}"])
        assert np.array_equal(g_size, d[f"ghosts_size_{r}"])
        if not stable:
            assert np.array_equal(o_idx, d[f"owners_idx_{r}"]), f"rank {r}: owners_idx differs from the reference"
            assert np.array_equal(g_idx, d[f"ghosts_idx_{r}"]), f"rank {r}: ghosts_idx differs from the reference"
        else:
            for i in range(len(o_ranks)):
                assert np.array_equal(np.sort(o_idx[o_off[i]:o_off[i + 1]]), np.sort(d[f"owners_idx_{r}"][o_off[i]:o_off[i + 1]]))
            for i in range(len(g_ranks)):
                assert np.array_equal(np.sort(g_idx[g_off[i]:g_off[i + 1]]), np.sort(d[f"ghosts_idx_{r}"][g_off[i]:g_off[i + 1]]))
        # the cuda-style 3-element list form the reference returns
        (ol, osz, orr), (gl, gsz, grr) = utils.to_lists(od[r]), utils.to_lists(gd[r])
        assert len(ol) == len(orr) and all(len(a) == s for a, s in zip(ol, osz))
        assert len(gl) == len(grr) and all(len(a) == s for a, s in zip(gl, gsz))


@pytest.mark.parametrize("path", golden_files("halo_plan_"), ids=lambda p: p.split("/")[-1][:-4])
def test_pairing(path):
    """Message i of rank r to owner o lists ghosts in the order o's ``ghosts_idx`` group for r lists its
    owned dofs: global index of r's k-th ghost in the group == o's local_range[0] + ghosts_idx[k]."""
    d = np.load(path)
    meshes = _meshes(d)
    for r, m in enumerate(meshes):
        im = m.index_map
        off = np.concatenate(([0], np.cumsum(d[f"owners_size_{r}"])))
        for i, o in enumerate(d[f"unique_owners_{r}"]):
            mine = np.asarray(im.ghosts)[d[f"owners_idx_{r}"][off[i]:off[i + 1]]]
            o = int(o)
            goff = np.concatenate(([0], np.cumsum(d[f"ghosts_size_{o}"])))
            k = list(d[f"unique_ghosts_{o}"]).index(r)
            theirs = d[f"ghosts_idx_{o}"][goff[k]:goff[k + 1]] + meshes[o].index_map.local_range[0]
            assert np.array_equal(mine, theirs)
