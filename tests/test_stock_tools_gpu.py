"""The drop-in claim on the reference's own tools: llama-bench (tools/llama-bench/llama-bench.cpp) and llama-cli
(tools/main/main.cpp), compiled unmodified from the reference's sources by oracle/Makefile (`make -C oracle tools`), pick
the plugin up through GGML_BACKEND_PATH (ggml_backend_load_all, llama-bench.cpp:1838) and run a random-weight model on
MI355_0.  The model for llama-cli carries the tokenizer metadata of the reference's own vocabulary fixture
(models/ggml-vocab-llama-spm.gguf, kept as data under tests/golden/)."""
import json
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFBIN = os.path.join(ROOT, "oracle", "_ref")
PLUGIN = os.path.join(ROOT, "llm_amd", "lib", "libggml-mi355.so")


def _model(name, ftype, vocab=None):
    path = f"/tmp/random-{name}-{ftype}{'-vocab' if vocab else ''}.gguf"
    if not os.path.exists(path):
        cmd = [os.path.join(REFBIN, "make-random-gguf"), path, name, ftype] + ([vocab] if vocab else [])
        subprocess.run(cmd, check=True, capture_output=True, timeout=600)
    return path


def test_stock_llama_bench_runs_on_the_plugin():
    exe = os.path.join(REFBIN, "llama-bench")
    assert os.path.exists(exe), "oracle/_ref/llama-bench is not built (make -C oracle tools)"
    env = dict(os.environ, GGML_BACKEND_PATH=PLUGIN)
    r = subprocess.run([exe, "-m", _model("tiny", "Q4_K_M"), "-ngl", "99", "-fa", "1", "-p", "64", "-n", "16", "-r", "2", "-o", "json"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    rows = json.loads(r.stdout)
    assert len(rows) == 2 and {(x["n_prompt"], x["n_gen"]) for x in rows} == {(64, 0), (0, 16)}
    for x in rows:
        assert x["avg_ts"] > 0 and x["n_gpu_layers"] == 99 and x["flash_attn"]
        assert "MI355" in x["backends"], x["backends"]
        assert "MI355X" in x["gpu_info"] or "MI355" in x["gpu_info"] or "gfx950" in x["gpu_info"], x["gpu_info"]


def test_stock_llama_bench_row_split_on_virtual_devices():
    """`llama-bench -sm row` (the harness of SURVEY config 4) over two virtual devices"""
    exe = os.path.join(REFBIN, "llama-bench")
    env = dict(os.environ, GGML_BACKEND_PATH=PLUGIN, GGML_MI355_VIRTUAL_DEVICES="2")
    r = subprocess.run([exe, "-m", _model("tiny", "Q4_K_M"), "-ngl", "99", "-fa", "1", "-sm", "row", "-ts", "1/1", "-p", "32", "-n", "8", "-r", "1", "-o", "json"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    rows = json.loads(r.stdout)
    assert len(rows) == 2 and all(x["avg_ts"] > 0 and x["split_mode"] == "row" for x in rows)


def test_stock_llama_cli_generates_on_the_plugin():
    exe = os.path.join(REFBIN, "llama-cli")
    assert os.path.exists(exe), "oracle/_ref/llama-cli is not built (make -C oracle tools)"
    gguf = _model("tiny", "Q4_K_M", os.path.join(ROOT, "tests", "golden", "ggml-vocab-llama-spm.gguf"))
    env = dict(os.environ, GGML_BACKEND_PATH=PLUGIN)
    r = subprocess.run([exe, "-m", gguf, "-ngl", "99", "-fa", "-no-cnv", "-p", "Hello world", "-n", "8", "--temp", "0", "-t", "4"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "MI355_0" in r.stderr and "offloaded 5/5 layers to GPU" in r.stderr, r.stderr[-3000:]
    # the prompt echoed, then 8 tokens of a random-weight model: some text beyond the prompt (how many words is up to the weights)
    assert r.stdout.lstrip().startswith("Hello world") and len(r.stdout.strip()) > len("Hello world") + 3, r.stdout
