"""GPU acceptance tests of the drop-in boundary: the REFERENCE's own conformance harness
(oracle/_ref/test-backend-ops, compiled from the reference's sources) runs against the plugin
libggml-mi355.so loaded through GGML_BACKEND_PATH, comparing every supported case with ggml-cpu
(NMSE thresholds of tests/test-backend-ops.cpp: 5e-4 MUL_MAT / MUL_MAT_ID / FLASH_ATTN_EXT, 1e-7 default)."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TBO = os.path.join(ROOT, "oracle", "_ref", "test-backend-ops")
PLUGIN = os.path.join(ROOT, "llm_amd", "lib", "libggml-mi355.so")

OPS = ["MUL_MAT", "MUL_MAT_ID", "FLASH_ATTN_EXT", "ADD", "MUL", "RMS_NORM", "ROPE", "GLU", "CPY", "CONT", "GET_ROWS",
       "SCALE", "SOFT_MAX"]
# lower bounds on the number of cases that must actually RUN on the device (not "not supported")
MIN_RUN = {"MUL_MAT": 100, "MUL_MAT_ID": 100, "FLASH_ATTN_EXT": 1000, "ROPE": 50, "SOFT_MAX": 100}


def _run(op):
    assert os.path.exists(TBO), "oracle/_ref/test-backend-ops is not built (make -C oracle ref)"
    assert os.path.exists(PLUGIN), "llm_amd/lib/libggml-mi355.so is not built"
    env = dict(os.environ, GGML_BACKEND_PATH=PLUGIN)
    return subprocess.run([TBO, "test", "-o", op], env=env, capture_output=True, text=True, timeout=900)


@pytest.mark.parametrize("op", OPS)
def test_backend_ops_vs_ggml_cpu(op):
    r = _run(op)
    out = re.sub(r"\x1b\[[0-9;]*m", "", r.stdout)
    assert "Backend MI355_0" in out, out[-2000:] + r.stderr[-2000:]
    fails = [l for l in out.splitlines() if "FAIL" in l]
    assert r.returncode == 0 and not fails, "\n".join(fails[:20])
    ran = sum(1 for l in out.splitlines() if l.rstrip().endswith("OK") and "(" in l)
    assert ran >= MIN_RUN.get(op, 1), f"only {ran} {op} cases ran on the device"


def test_llama_graph_end_to_end_has_no_cpu_compute_split():
    """tiny random llama GGUF through the reference's libllama: all layers on MI355_0 (2 splits: the
    CPU-side token embedding lookup + one device split), decode and prefill complete."""
    tool = os.path.join(ROOT, "tools", "run_llama_bench.sh")
    r = subprocess.run([tool, "tiny", "Q4_K_M", "-p", "64", "-n", "8", "-r", "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "graph splits = 2" in r.stderr + r.stdout
    assert '"devices": "CPU, MI355_0"' in r.stdout
