import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_deform_patch import _inputs, _run
from sipmask_amd import _lib
B, C, Co, G = 1, 256, 256, 4
for sizes in ([(19, 45)], [(8, 32)], [(5, 12)]):
    xs, offs, wt, x_rows, off_rows, lv = _inputs(B, sizes, C, Co, G, 0.0, 5)
    got, _ = _run(B, sizes, C, Co, G, off_rows, x_rows, wt, _lib.SM_CONV_OUT_F32, gather=False)
    old, _ = _run(B, sizes, C, Co, G, off_rows, x_rows, wt, _lib.SM_CONV_OUT_F32, gather=True)
    h, w = sizes[0]
    bad = ((got - old).abs() > 1e-3).view(h, w, Co)
    print(sizes, "bad frac", bad.float().mean().item())
    print(" rows(y) with bad:", bad.any(2).any(1).nonzero().flatten().tolist())
    print(" cols(x) with bad:", bad.any(2).any(0).nonzero().flatten().tolist())
    print(" couts with bad:", bad.any(0).any(0).nonzero().flatten().tolist()[:64])
    yx = bad.any(2).nonzero()[:12].tolist()
    print(" first (y,x):", yx)
    if yx:
        y, x = yx[0]
        print(" got", got.view(h, w, Co)[y, x, :8].tolist(), "\n old", old.view(h, w, Co)[y, x, :8].tolist())
