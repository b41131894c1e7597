#!/usr/bin/env python3
"""Phase timeline of ONE mi355_decode_layer launch from the in-kernel s_memrealtime stamps (100 MHz)."""
import os, sys
import ctypes as C
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.argv = [sys.argv[0], "4", "8b"]
import bench_layer as BL  # builds 4 layers and runs the comparison once
K = BL.K
st = torch.zeros(256 * 16 * 16, dtype=torch.int64, device="cuda")
K.lib.mi355_decode_layer_debug_stamps.argtypes = [C.c_void_p]
K.lib.mi355_decode_layer_debug_stamps(st.data_ptr())
for rep in range(2):
    for li in (1, 2):
        st.zero_(); BL.sync.zero_(); torch.cuda.synchronize()
        K.check(K.lib.mi355_decode_layer(BL.fused[li], BL.sync.data_ptr(), torch.cuda.current_stream().cuda_stream), "dl")
        torch.cuda.synchronize()
        s = st.view(256, 16, 16).cpu().numpy().astype("float64")
        import numpy as _np
        _np.save(os.path.join(ROOT, "gpurun_out", f"dl_stamps_l{li}_r{rep}.npy"), s)
        t0 = s[:, :, 0][s[:, :, 0] > 0].min()
        rel = (s - t0) / 100.0   # us
        import numpy as np
        def row(name, x):
            x = x[x > -1e5]
            print(f"  {name:34s} min {x.min():6.2f}  med {np.median(x):6.2f}  max {x.max():6.2f}")
        print(f"layer {li} (down {'q6_K' if li % 2 == 0 else 'q4_K'}), rep {rep}: us since the first wave's first stamp")
        names_s = ["start", "ph0 acts ready", "ph0 done", "ph1 acts ready", "ph1 done", "ph2 acts ready", "ph2 done", "ph3 acts ready", "ph3 done"]
        for i, n in enumerate(names_s):
            row("stream " + n, rel[:, 4:, i].ravel())
        names_c = ["start", "ph0 staged", "e0 results", "e0 arrived", "e0 all arrived", "e0 staged", "e1 results", "e1 arrived", "e1 all arrived", "e1 staged",
                   "e2 results", "e2 arrived", "e2 all arrived", "e2 staged"]
        for i, n in enumerate(names_c):
            row("comm0 " + n, rel[:, 0, i].ravel())
