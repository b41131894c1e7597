"""llm_amd -- MI355X-native (gfx950) kernels for llama.cpp's quantized MUL_MAT / MUL_MAT_ID /
FLASH_ATTN_EXT hot path, plus the host-side mirror of the reference's operator interface.

The compute lives in two shared libraries built from llm_amd/csrc (see llm_amd/csrc/Makefile):
  lib/libmi355_kernels.so  hand-written HIP kernels behind the C ABI of include/mi355_kernels.h
  lib/libggml-mi355.so     the ggml backend plugin (ggml_backend_init) that llama.cpp tools load
This package binds the first one with ctypes for tests and benchmarks.  There is no CPU fallback:
importing llm_amd.kernels without the built library raises.
"""
from . import blocks  # noqa: F401

__all__ = ["blocks", "kernels", "ops"]
