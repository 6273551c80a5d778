import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import common, nrc_amd
import test_train as T
rc = common.make_rc()
for level in (0, 1, 2):
    for wf in (False, True):
        n = 1000
        pts = T._points_off_kinks(level, n); dd, df = T._upstream(n); df = df if wf else None
        layout, total = rc.density_grad_layout(level)
        flat, dens = rc.density_backward(level, pts, dd, df)
        ref, dref = T._oracle_flat(level, pts, dd, df, layout)
        r32, _ = T._oracle_flat(level, pts, dd, df, layout, torch.float32)
        got = flat.cpu().numpy().astype(np.float64)
        print("level", level, "feat", wf, "dens max", dref.max())
        for name, off, shape in layout:
            sz = int(np.prod(shape)); a, b, c = got[off:off+sz], ref[off:off+sz], r32[off:off+sz]
            sc = np.abs(b).max()
            print(f"  {name.split('MLP_')[1]:45s} scale {sc:10.3e} hip_err {np.abs(a-b).max()/max(sc,1e-30):9.2e} o32_err {np.abs(c-b).max()/max(sc,1e-30):9.2e} nnz {np.count_nonzero(a)}/{np.count_nonzero(b)}")
