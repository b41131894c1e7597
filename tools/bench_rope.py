import sys; sys.path.insert(0, "/root/repo")
import torch, llm_amd.kernels as K, llm_amd.ops as ops
for nt, nh in ((512, 32), (512, 8), (64, 32)):
    x = torch.randn(nt, nh, 128, device="cuda"); pos = torch.arange(nt, dtype=torch.int32, device="cuda")
    for _ in range(3): ops.rope(x, pos, 128, 0, freq_base=500000.0)
    torch.cuda.synchronize(); K.lib.mi355_timing_reset(); K.lib.mi355_timing_enable(1)
    for _ in range(20): ops.rope(x, pos, 128, 0, freq_base=500000.0)
    torch.cuda.synchronize(); K.lib.mi355_timing_enable(0)
    print(nt, nh, {t: round(us / n, 2) for t, (n, us) in K.timing_report().items() if n})
