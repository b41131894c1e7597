#!/usr/bin/env python3
"""Decode-sized small kernels, per-dispatch execution time (library timing mode = what rocprofv3 reports)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import llm_amd.kernels as K, llm_amd.ops as ops

n = 4096
s = torch.cuda.current_stream().cuda_stream
x, y, w = (torch.randn(1, n, device="cuda") for _ in range(3))
su, d = torch.empty_like(x), torch.empty_like(x)
aq = torch.empty(n, dtype=torch.int8, device="cuda"); ads = torch.empty(n // 32 * 2, device="cuda")
T = K.Tensor.of
def t(name, fn, tag):
    K.lib.mi355_timing_reset(); K.lib.mi355_timing_enable(1)
    for _ in range(200): fn()
    torch.cuda.synchronize(); K.lib.mi355_timing_enable(0)
    c, us = K.timing_read(tag)
    print(f"{name:40s} {us / max(c, 1):7.2f} us  ({c} launches)", flush=True)
t("add+norm+mul+quant", lambda: K.lib.mi355_add_rms_norm_mul_f32(T(x), T(y), T(su), w.data_ptr(), T(d), 1e-5, aq.data_ptr(), ads.data_ptr(), s), "add_rms_norm_mul")
t("add+norm+mul", lambda: K.lib.mi355_add_rms_norm_mul_f32(T(x), T(y), T(su), w.data_ptr(), T(d), 1e-5, None, None, s), "add_rms_norm_mul")
t("norm+mul+quant", lambda: K.lib.mi355_rms_norm_mul_quant_f32(T(x), w.data_ptr(), T(d), 1e-5, aq.data_ptr(), ads.data_ptr(), s), "rms_norm_mul")
t("norm+mul", lambda: K.lib.mi355_rms_norm_mul_f32(T(x), w.data_ptr(), T(d), 1e-5, s), "rms_norm_mul")
t("norm", lambda: K.lib.mi355_rms_norm_f32(T(x), T(d), 1e-5, s), "rms_norm")
t("quantize_act", lambda: K.lib.mi355_quantize_act(x.data_ptr(), aq.data_ptr(), ads.data_ptr(), n, 1, 1, 1, n * 4, 0, 0, s), "quantize_q8_1")
t("scale", lambda: K.lib.mi355_scale_f32(T(x), T(d), 2.0, 0.0, s), "scale")
