import cProfile, pstats, sys, os, io
import numpy as np
sys.path.insert(0, os.getcwd())
import torch, fusgpu_loader
boxmesh, ls = fusgpu_loader.submodule("boxmesh"), fusgpu_loader.submodule("linear_solver")
P, N, L = 2, 18, 0.12
mesh = boxmesh.BoxMesh(P, N, length=L)
h = ls.time_step_parameters(mesh, P, 1500.0, 0.5e6, L)
dt, tf, nstep = ls.snap_time_step(h, P, 1500.0, 0.5e6, L)
s = ls.LinearSpectral3D(mesh, np.float64)
s.init()
s.rk4(0.0, tf, dt, max_steps=5)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
s.rk4(5 * dt, tf, dt, max_steps=60)
torch.cuda.synchronize()
pr.disable()
st = io.StringIO()
pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(22)
print(st.getvalue()[:6000])
