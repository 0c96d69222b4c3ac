import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import fusgpu_loader
ops = fusgpu_loader.submodule("operators")
torch.cuda.set_device(0)
for nbytes in (1 << 30, 82 * 10**6):
    nel = nbytes // 8
    a = torch.ones(nel, dtype=torch.float64, device="cuda"); b = torch.empty_like(a); c = torch.empty_like(a)
    for name, fn, vol in (("copy", lambda: ops.copy(a, b), 16), ("axpy", lambda: ops.axpy[1,1](0.5, a, b), 24), ("fill", lambda: ops.fill(1.0, b), 8), ("divide", lambda: ops.pointwise_divide(a, b, c), 24)):
        fn(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"{nbytes/1e6:7.0f} MB {name:7s} {vol*nel*10/(e0.elapsed_time(e1)*1e-3)/1e9:7.0f} GB/s")
