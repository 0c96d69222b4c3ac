#!/usr/bin/env python3
"""Host-side issue time per apply (Python + ctypes + torch launch path), measured by issuing K steps
without synchronising: if it exceeds the GPU time per step the path is host-bound."""
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import fusgpu_loader  # noqa: E402
from conftest import build_problem  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
ops, scat = fusgpu_loader.submodule("operators"), fusgpu_loader.submodule("scatterer")
pb = build_problem(4, 54, perturb=0.16)
mesh = pb["mesh"]
dev = torch.device("cuda", 0)
x, cc, G = (torch.from_numpy(pb[k]).to(dev) for k in ("x", "cc", "G"))
dm = torch.from_numpy(mesh.dofmap).to(dev)
y = torch.zeros(mesh.ndofs, dtype=torch.float64, device=dev)
op = ops.stiffness_operator(4, pb["D"].flatten(), np.float64)
halo = scat.HaloApply(mesh, op, scat.TorchComm(), np.float64)
halo.prepare(x, cc, G, dm)
op.prepare(dm)
send = torch.zeros(47089 * 3, dtype=torch.float64, device=dev)
recv = torch.zeros_like(send)


def issue(fn, K=200):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6


# a halo plan of config-4 size (3 faces + 3 edges + 1 corner of a 2x2x2 block partition: 141 919 ghost
# dofs) in a 1-rank world whose rank is its own neighbour, through both transports
rng = np.random.default_rng(0)
ng = 3 * 47089 + 3 * 217 + 1
N = mesh.ndofs - ng
od = [np.arange(ng, dtype=np.int64), np.array([ng]), np.array([0, ng]), np.array([0], dtype=np.int32)]
gd = [rng.choice(N, size=ng, replace=False).astype(np.int64), np.array([ng]), np.array([0, ng]), np.array([0], dtype=np.int32)]
od_perm = [rng.permutation(ng).astype(np.int64)] + od[1:]
ncomm = scat.NativeComm()
nat_fwd, nat_rev = scat.scatter_forward(ncomm, od, gd, N, np.float64), scat.scatter_reverse(ncomm, od, gd, N, np.float64)
nat_fwd_p, nat_rev_p = scat.scatter_forward(ncomm, od_perm, gd, N, np.float64), scat.scatter_reverse(ncomm, od_perm, gd, N, np.float64)
tcomm = scat.TorchComm()
t_fwd, t_rev = scat.scatter_forward(tcomm, od, gd, N, np.float64), scat.scatter_reverse(tcomm, od, gd, N, np.float64)
t_fwd_p, t_rev_p = scat.scatter_forward(tcomm, od_perm, gd, N, np.float64), scat.scatter_reverse(tcomm, od_perm, gd, N, np.float64)
print(f"halo plan: {ng} ghost dofs ({ng * 8 / 1e6:.2f} MB per direction), 1-rank world, rank 0 <-> rank 0")

for name, fn in (("single launch op(...)", lambda: op(x, cc, y, G, dm)),
                 ("native halo fwd+rev, direct (fus_halo_*: 1 pack, 2 RCCL groups, 1 unpack)", lambda: (nat_fwd(y), nat_rev(y))),
                 ("torch  halo fwd+rev, direct (all_to_all_single + Python launches)", lambda: (t_fwd(y), t_rev(y))),
                 ("native halo fwd+rev, permuted ghosts (2 pack, 2 groups, 2 unpack)", lambda: (nat_fwd_p(y), nat_rev_p(y))),
                 ("torch  halo fwd+rev, permuted ghosts", lambda: (t_fwd_p(y), t_rev_p(y))),
                 ("HaloApply.apply (1 rank: 3 launches, no exchange)", lambda: halo.apply(x, cc, y, G, dm)),
                 ("all_to_all_single to self, async + wait (377 kB x3)", lambda: dist.all_to_all_single(recv, send, [send.numel()], [send.numel()], async_op=True).wait()),
                 ("fill kernel via ctypes", lambda: ops.fill(0.0, recv))):
    h, tot = issue(fn)
    print(f"{name:78s} host issue {h:7.1f} us/step   wall {tot:7.1f} us/step")
dist.destroy_process_group()
