"""CPU test: the C-ABI library loads and exports every symbol include/*.h declares (no compute)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return re.findall(r"MI355_API\s+[\w\s\*]+?\b(mi355_\w+)\s*\(", src)


def test_kernel_library_exports_every_declared_symbol():
    names = _declared("mi355_kernels.h")
    assert len(names) >= 10
    lib = ctypes.CDLL(os.path.join(ROOT, "llm_amd", "lib", "libmi355_kernels.so"))
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_python_binding_lists_the_same_symbols():
    import llm_amd.kernels as K
    assert sorted(K.EXPORTS) == sorted(_declared("mi355_kernels.h"))
    assert K.lib.mi355_abi_version() == K.ABI_VERSION == 6


def test_sizes_match_block_formats():
    import llm_amd.blocks as B
    import llm_amd.kernels as K
    for t in B.TYPE_NAMES:
        assert K.lib.mi355_blck_size(t) == B.BLCK_SIZE[t]
        assert K.lib.mi355_type_size(t) == B.TYPE_SIZE[t]
    assert K.lib.mi355_row_size(B.Q4_K, 4096) == 2304
    assert K.lib.mi355_row_size(B.Q4_K, 100) == -1
    assert K.lib.mi355_blck_size(99) == 0
    # config 1 of BASELINE.json: q4_0 4096x4096 mat-vec = 9 469 952 algorithmic bytes
    assert B.matvec_bytes(B.Q4_0, 4096, 4096) == 9469952


def test_argument_errors_need_no_gpu():
    import llm_amd.kernels as K
    a = K.MmvqArgs()
    assert K.lib.mi355_mul_mat_vec_q(a, None) == -4          # NULL pointers
    assert K.lib.mi355_dequantize_row(12, None, None, 256, None) == -4
    assert K.lib.mi355_quantize_row_q8_1(16, 16, 33, 1, 1, 1, 132, 132, 132, None) == -2  # k % 32
    assert K.lib.mi355_dequantize_row(77, 16, 16, 256, None) == -1
    assert b"alignment" in K.lib.mi355_error_string(-3)


def test_plugin_exports_the_loader_symbols():
    """include/ggml_mi355.h: the two symbols ggml's loader dlsym()s (ggml-backend-reg.cpp:234-250).
    The plugin cannot be dlopen()ed without libggml-base, so read its dynamic symbol table."""
    import subprocess
    src = open(os.path.join(ROOT, "include", "ggml_mi355.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"GGML_MI355_API\s+[\w\s\*]+?\b(ggml_backend_\w+)\s*\(", src)
    assert sorted(names) == ["ggml_backend_init", "ggml_backend_score"]
    so = os.path.join(ROOT, "llm_amd", "lib", "libggml-mi355.so")
    assert os.path.exists(so), "build the plugin: make -C llm_amd/csrc plugin"
    out = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    defined = {ln.split()[-1] for ln in out.splitlines() if " T " in ln}
    assert set(names) <= defined, defined
    # nothing else leaks: the plugin's only defined text symbols are the loader's two
    assert defined == set(names), defined
