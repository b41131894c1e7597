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

OPS = ["MUL_MAT", "MUL_MAT_ID", "FLASH_ATTN_EXT", "ADD", "MUL", "SUB", "DIV", "RMS_NORM", "ROPE", "REGLU", "GEGLU", "SWIGLU", "CPY", "CONT", "DUP",
       "GET_ROWS", "SET_ROWS", "SCALE", "SOFT_MAX", "ARGSORT", "SUM_ROWS",
       # what other model families' graphs add to a Llama / Mixtral graph (Gemma: TANH; Phi / StableLM / Falcon: GELU, NORM; ...)
       "ABS", "SGN", "NEG", "STEP", "TANH", "ELU", "RELU", "SIGMOID", "GELU", "GELU_QUICK", "SILU", "HARDSWISH", "HARDSIGMOID", "EXP", "GELU_ERF",
       "SQR", "SQRT", "SIN", "COS", "LOG", "CLAMP", "NORM", "L2_NORM"]
# lower bounds on the number of cases that must actually RUN on the device (not "not supported")
MIN_RUN = {"MUL_MAT": 640, "SET_ROWS": 80, "CPY": 40, "MUL_MAT_ID": 160, "FLASH_ATTN_EXT": 1900, "ROPE": 50, "SOFT_MAX": 100}


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


@pytest.mark.parametrize("ftype", ["Q4_K_M", "F16"])
def test_llama_graph_without_flash_attention_stays_on_the_device(ftype, tmp_path):
    """`-fa 0`: K.Q and (softmax).V are MUL_MATs on f16 views of the KV cache (permuted, broadcast over the GQA heads) with a
    SOFT_MAX between -- the float-weight MUL_MAT of SURVEY 8 f4.  No CPU compute split; the logits are as close to the exact twin's as
    ggml-cpu's are (see test_llama_logits_match_ggml_cpu)."""
    tool = os.path.join(ROOT, "tools", "run_llama_bench.sh")
    r = subprocess.run([tool, "tiny", ftype, "-fa", "0", "-p", "64", "-n", "8", "-r", "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "graph splits = 2" in r.stderr + r.stdout, (r.stderr + r.stdout)[-3000:]
    e_cpu, e_gpu, d = _logits_vs_exact_twin(tmp_path, "tiny", ftype, extra=("-fa", "0"))
    assert e_cpu < 5e-2 and e_gpu <= 1.25 * e_cpu + 2e-4 and d <= 2.0 * e_cpu + 2e-4, (e_cpu, e_gpu, d)


def test_mixtral_style_graph_runs_entirely_on_the_device(tmp_path):
    """tiny random 8-expert llama GGUF (router MUL_MAT f32 -> SOFT_MAX -> ARGSORT/top-k -> GET_ROWS -> SUM_ROWS/DIV ->
    MUL_MAT_ID x3 -> weighted sum): no CPU compute split.  The routing is discrete -- a near-tie between the 2nd and 3rd expert
    resolves differently under different activation rounding -- so ggml-cpu and the plugin are each measured against the model's
    exact twin and the plugin has to be as close as ggml-cpu is; test_moe_layers_are_as_close_to_the_exact_twin_as_ggml_cpu
    looks at the layers one by one."""
    tool = os.path.join(ROOT, "tools", "run_llama_bench.sh")
    r = subprocess.run([tool, "tinymoe", "Q4_K_M", "-p", "64", "-n", "8", "-r", "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "graph splits = 2" in r.stderr + r.stdout, (r.stderr + r.stdout)[-3000:]
    e_cpu, e_gpu, d = _logits_vs_exact_twin(tmp_path, "tinymoe", "Q4_K_M")
    assert e_cpu < 0.5 and e_gpu <= 1.25 * e_cpu + 1e-3, (e_cpu, e_gpu, d)


def test_moe_router_is_one_launch_for_small_batches_too():
    """Steps of 2..64 tokens: the router chain (MUL_MAT SOFT_MAX ARGSORT GET_ROWS SUM_ROWS DIV) is mi355_moe_router_f32's one launch per
    layer, as for one token.  ggml-alloc lays the weight sums over the dead logits / probabilities, which with several workgroups in
    flight would be a race: the chain's tensors that nobody outside it reads go to scratch (bit-exactness of the fused graph for 2-, 3- and
    12-token batches: test_fusions_are_bit_exact[tinymoe])."""
    tool = os.path.join(ROOT, "tools", "run_llama_bench.sh")
    env = dict(os.environ, GGML_MI355_KERNEL_TIMES="1")
    r = subprocess.run([tool, "tinymoe", "Q4_K_M", "-p", "12", "-n", "0", "-r", "2"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    count = lambda name: sum(int(m.group(1)) for m in re.finditer(r"\[mi355\] kernel %s (\d+) " % name, r.stderr))
    # 4 layers x (1 warm-up + 2 timed) steps.  The last layer of each step routes one token (the only output row) and stays unfused:
    # ggml-alloc puts its gathered weights on the bytes of the layer's dead input there, which the experts still read when the fused
    # launch would write them
    assert count("moe_router") == 9 and count("argsort") == 3 and count("soft_max") == 3, r.stderr[-2000:]
    # and the tail of every layer, MUL(experts, weights) + ADD of its two slot views, is mi355_moe_combine2_f32's one pass (the last layer
    # of a step has one token and takes the decode form, the MoE-down mat-vec with the sum in its epilogue)
    assert count("moe_combine2") == 9, r.stderr[-2000:]
    # decode: the ffn norm runs in the router's launch too (mi355_moe_router_norm_f32) -- in the layers whose router is fused (not the last)
    r = subprocess.run([tool, "tinymoe", "Q4_K_M", "-p", "0", "-n", "4", "-r", "1"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    steps = count("moe_router_norm") // 3
    assert steps >= 4 and count("moe_router_norm") == 3 * steps and count("moe_router") == 0 and count("rms_norm_mul") == steps, r.stderr[-2000:]


def _env_for(ngl):
    """reference runs (-ngl 0) are ggml-cpu alone: with the plugin loaded the scheduler would offload their large batches to it"""
    env = dict(os.environ)
    if ngl == "0":
        env["MI355_NO_PLUGIN"] = "1"
    return env


def _logits_vs_exact_twin(tmp_path, model, ftype, ctk="f16", extra=()):
    """Logits (48-token prefill, 6 decode steps, batches of 3, 2, 12 and 160 tokens) of the model's exact twin on ggml-cpu (the same
    random blocks stored dequantized as F32, make-random-gguf <ftype>+F32: f32 activations, nothing rounded but the f16 KV cache), of
    the model on ggml-cpu alone and of the model on MI355_0 -> NMSE (ggml-cpu vs twin, plugin vs twin, plugin vs ggml-cpu)."""
    import numpy as np
    tool = os.path.join(ROOT, "tools", "run_llama_bench.sh")
    outs = {}
    for name, ft, ngl, kv in (("exact", ftype + "+F32", "0", "f16"), ("cpu", ftype, "0", ctk), ("gpu", ftype, "99", ctk)):
        path = str(tmp_path / f"logits_{name}.bin")
        r = subprocess.run([tool, model, ft, "-ngl", ngl, "-ctk", kv, "-ctv", kv, "--dump-logits", path, *extra],
                           capture_output=True, text=True, timeout=900, env=_env_for(ngl))
        assert r.returncode == 0, r.stderr[-3000:]
        if name == "gpu":
            assert "MI355_0" in r.stdout, r.stdout[-500:]
        outs[name] = np.fromfile(path, dtype=np.float32).reshape(11, -1)
    x, a, b = outs["exact"], outs["cpu"], outs["gpu"]
    assert np.isfinite(b).all()
    nmse = lambda u, v: float(((u - v) ** 2).sum() / (u ** 2).sum())
    e_cpu, e_gpu, d = nmse(x, a), nmse(x, b), nmse(a, b)
    print(f"{model} {ftype} kv {ctk} {' '.join(extra)}: ggml-cpu vs exact {e_cpu:.3e}, plugin vs exact {e_gpu:.3e}, plugin vs ggml-cpu {d:.3e}")
    return e_cpu, e_gpu, d


def _read_nodes(path):
    """--dump-nodes output of tools/llama_bench_driver.cpp -> list of (name, array [ne1*ne2*ne3, ne0]) in graph order"""
    import numpy as np
    raw = np.fromfile(path, dtype=np.uint8)
    out = []
    for line in open(path + ".idx"):
        name, typ, ne0, ne1, ne2, ne3, off = line.rstrip("\n").rsplit(" ", 6)
        if " " in name:             # "... (view)", "... (reshaped)": derived tensors the graph builder did not name itself
            continue
        ne0, rows, off = int(ne0), int(ne1) * int(ne2) * int(ne3), int(off)
        out.append((name, raw[off:off + rows * ne0 * 4].view(np.int32 if typ == "i32" else np.float32).reshape(rows, ne0)))
    return out


def test_moe_layers_are_as_close_to_the_exact_twin_as_ggml_cpu(tmp_path):
    """Layer-by-layer look at the FUSED Mixtral-style graph (the scheduler's eval callback dumps l_out-<layer>, the experts' weighted sum
    and the router's choice for a 48-token prompt, 6 decode steps and batches of 3, 2, 12 and 160 tokens).  Three runs: the model's exact
    twin (F32 weights, ggml-cpu), the model on ggml-cpu alone, the model on MI355_0.  The routing is discrete and the random model is
    chaotic, so ggml-cpu itself drifts from the twin layer by layer; per layer, the plugin's drift (NMSE of l_out and of ffn_moe_out
    over all steps) and its count of expert choices that differ from the twin's must not exceed ggml-cpu's by more than a quarter.
    The device run that dumps only l_out (sub-graphs = whole layers, every fusion active) must equal, bit for bit, the one that
    also stops at the router's tensors."""
    import numpy as np
    tool = os.path.join(ROOT, "tools", "run_llama_bench.sh")
    full = "l_out,ffn_moe_topk,ffn_moe_out"
    runs = {}
    for name, ft, ngl, prefixes in (("exact", "Q4_K_M+F32", "0", full), ("cpu", "Q4_K_M", "0", full), ("gpu", "Q4_K_M", "99", full), ("gpu_layers", "Q4_K_M", "99", "l_out")):
        path = str(tmp_path / f"nodes_{name}.bin")
        r = subprocess.run([tool, "tinymoe", ft, "-ngl", ngl, "--dump-logits", str(tmp_path / f"lg_{name}.bin"), "--dump-nodes", path,
                            "--node-prefix", prefixes], capture_output=True, text=True, timeout=900, env=_env_for(ngl))
        assert r.returncode == 0, r.stderr[-3000:]
        runs[name] = _read_nodes(path)
    names = [n for n, _ in runs["exact"]]
    assert names == [n for n, _ in runs["cpu"]] == [n for n, _ in runs["gpu"]] and len(names) > 100
    fine = [(n, a) for n, a in runs["gpu"] if n.startswith("l_out")]
    assert [n for n, _ in fine] == [n for n, _ in runs["gpu_layers"]]
    for (n, a), (_, b) in zip(fine, runs["gpu_layers"]):
        assert np.array_equal(a, b), n
    num = {}       # (kind, layer) -> [sum |x - cpu|^2, sum |x - gpu|^2, sum |x|^2]  or  [choices differing on cpu, on gpu, choices]
    for (name, x), (_, a), (_, b) in zip(runs["exact"], runs["cpu"], runs["gpu"]):
        kind, layer = name.rsplit("-", 1)
        acc = num.setdefault((kind, int(layer)), [0.0, 0.0, 0.0])
        if kind == "ffn_moe_topk":
            sx, sa, sb = np.sort(x, axis=1), np.sort(a, axis=1), np.sort(b, axis=1)
            acc[0] += int((sx != sa).any(axis=1).sum()); acc[1] += int((sx != sb).any(axis=1).sum()); acc[2] += x.shape[0]
        else:
            assert np.isfinite(b).all(), name
            acc[0] += float(((x - a) ** 2).sum()); acc[1] += float(((x - b) ** 2).sum()); acc[2] += float((x ** 2).sum())
    for (kind, layer), (ea, eb, n) in sorted(num.items()):
        print(f"{kind}-{layer}: ggml-cpu {ea / n:.3e}  plugin {eb / n:.3e}" + ("  (fraction of tokens whose experts differ from the twin's)" if kind == "ffn_moe_topk" else "  (NMSE vs the twin)"))
        if kind == "ffn_moe_topk":
            assert eb <= 1.25 * ea + 3, (kind, layer, ea, eb, n)
        else:
            assert eb / n <= 1.25 * ea / n + 1e-3, (kind, layer, ea / n, eb / n)


@pytest.mark.parametrize("ftype,ctk", [("Q4_K_M", "f16"), ("Q8_0", "q8_0"), ("Q4_0", "f16"), ("Q4_1", "f16"), ("Q5_0", "f16"), ("Q5_1", "f16"), ("F16", "f16"),
                                       ("Q2_K", "f16"), ("Q3_K", "f16"), ("IQ4_NL", "f16"), ("IQ4_XS", "f16")])
def test_llama_logits_match_ggml_cpu(ftype, ctk, tmp_path):
    """Whole-graph parity (covers the backend's fusions, which test-backend-ops cannot see): logits of a 48-token prefill, 6 decode
    steps and batches of 3, 2, 12 and 160 tokens on MI355_0 vs the same model on the reference's ggml-cpu.  Both round the activations
    of every MUL_MAT (ggml-cpu to q8_K / q8_0 / q8_1 per its vec_dot_type, the plugin to q8_1 or bf16), so neither is "the" answer:
    the yardstick is the model's exact twin -- the same random blocks stored dequantized as F32 (make-random-gguf <ftype>+F32), run
    on ggml-cpu with f32 activations.  The plugin has to be as close to the twin as ggml-cpu is, and close to ggml-cpu."""
    e_cpu, e_gpu, d = _logits_vs_exact_twin(tmp_path, "tiny", ftype, ctk)
    assert e_cpu < 5e-2, e_cpu                      # the yardstick itself is sane
    assert e_gpu <= 1.25 * e_cpu + 2e-4, (e_gpu, e_cpu)
    assert d <= 2.0 * e_cpu + 2e-4, (d, e_cpu)        # two roundings of the same exact values


@pytest.mark.parametrize("ctk", ["q4_0", "bf16"])
def test_q4_0_and_bf16_kv_cache_stay_on_the_device(ctk, tmp_path):
    """`-ctk q4_0 -ctv q4_0` / `bf16`: FLASH_ATTN_EXT reads such a cache, and the stores into it -- CPY (or SET_ROWS) f32 -> q4_0
    (quantize_row_q4_0_ref) / bf16 -- run on the device too: no CPU compute split, logits as close to the exact twin's as ggml-cpu's
    with the same cache type."""
    tool = os.path.join(ROOT, "tools", "run_llama_bench.sh")
    r = subprocess.run([tool, "tiny", "Q4_K_M", "-ctk", ctk, "-ctv", ctk, "-p", "64", "-n", "8", "-r", "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "graph splits = 2" in r.stderr + r.stdout, (r.stderr + r.stdout)[-3000:]
    e_cpu, e_gpu, d = _logits_vs_exact_twin(tmp_path, "tiny", "Q4_K_M", ctk)
    # (4-bit K / V rows on a random model: ggml-cpu itself is far from the twin, NMSE ~0.5 -- the plugin has to be no farther)
    assert e_cpu < (1.0 if ctk == "q4_0" else 8e-2), e_cpu
    assert e_gpu <= 1.25 * e_cpu + 2e-4, (e_gpu, e_cpu)
    assert d <= 2.0 * e_cpu + 2e-4, (d, e_cpu)


@pytest.mark.parametrize("model,ctk", [("hd80", "f16"), ("hd96", "f16"), ("hd96", "q8_0"), ("hd256", "f16"), ("hd256", "q8_0")])
def test_other_head_sizes_match_ggml_cpu(model, ctk, tmp_path):
    """Llama graphs with head sizes 80, 96 and 256 (FLASH_ATTN_EXT on the kernels built for 128 / 96 / 256 features per row, the fused Q/K/V +
    rope + KV-store forms per head size, f16 and q8_0 KV cache where a row is whole 32-blocks): no CPU compute split, and the logits are
    as close to the exact twin's as ggml-cpu's are (the yardstick of test_llama_logits_match_ggml_cpu)."""
    tool = os.path.join(ROOT, "tools", "run_llama_bench.sh")
    r = subprocess.run([tool, model, "Q4_K_M", "-ctk", ctk, "-ctv", ctk, "-p", "64", "-n", "8", "-r", "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "graph splits = 2" in r.stderr + r.stdout, (r.stderr + r.stdout)[-3000:]
    e_cpu, e_gpu, d = _logits_vs_exact_twin(tmp_path, model, "Q4_K_M", ctk)
    assert e_cpu < 5e-2, e_cpu
    assert e_gpu <= 1.25 * e_cpu + 2e-4, (e_gpu, e_cpu)
    assert d <= 2.0 * e_cpu + 2e-4, (d, e_cpu)


# mid: 2048 wide, 4 KV heads of 128, ffn 5120; wideffn: Llama-8B's 14336; 70b2: two layers of Llama-3-70B's shapes (K = 8192: norm in
# the grouped gate / up launch, 28672 rows: no one-launch GLU form, q5_K wv)
# hd80 / hd96 / hd256: head sizes 80 (no GQA), 96 (GQA 4) and 256 (GQA 2) -- rope, the K/V store and FLASH_ATTN_EXT per head size
@pytest.mark.parametrize("model", ["tiny", "tinymoe", "mid", "wideffn", "70b2", "hd80", "hd96", "hd256"])
def test_fusions_are_bit_exact(tmp_path, model, kv="f16", extra_env=None):
    """The backend's graph-level fusions (RMS_NORM*weight, fused q8_1 quantization, activation cache, Q/K/V + rope + store,
    and for the 8-expert model the one-launch MoE router) keep the f32 op order of the unfused graph: logits with and
    without them are identical to the bit."""
    import numpy as np
    tool = os.path.join(ROOT, "tools", "run_llama_bench.sh")
    outs = []
    for nofuse in (False, True):
        path = str(tmp_path / f"logits_{int(nofuse)}.bin")
        env = dict(os.environ, **(extra_env or {}))
        if nofuse:
            env["GGML_MI355_NO_FUSE"] = "1"
        r = subprocess.run([tool, model, "Q4_K_M", "-ctk", kv, "-ctv", kv, "--dump-logits", path], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        outs.append(np.fromfile(path, dtype=np.uint32))
    assert np.array_equal(outs[0], outs[1])


def test_norm_in_the_grouped_gate_up_launch_is_bit_exact(tmp_path):
    """Where the one-launch gate/up/GLU form does not apply (Llama-3-70B: 28672 rows; forced here with GGML_MI355_NO_GLU_MATVEC) the ffn norm
    runs in the prologue of the grouped gate + up mat-vec: same bits as the unfused graph"""
    test_fusions_are_bit_exact(tmp_path, "tiny", extra_env={"GGML_MI355_NO_GLU_MATVEC": "1"})


@pytest.mark.parametrize("ftype", ["Q4_K_M", "Q8_0"])
def test_gate_up_glu_gemm_is_bit_exact(tmp_path, ftype):
    """Prefill: MUL_MAT(ffn_gate) MUL_MAT(ffn_up) GLU as ONE launch (mi355_glu_mul_mat_dq; the two f32 products are never written).  A
    512-token batch through a model with Llama-8B's feed-forward width (14336 x 1024, the smallest shape the fused form is offered
    for) gives the logits of the unfused graph to the bit, and the fused launch did run (GGML_MI355_NO_GLU_GEMM changes nothing but time)."""
    import numpy as np
    tool = os.path.join(ROOT, "tools", "run_llama_bench.sh")
    outs = {}
    for name, extra in (("fused", {}), ("no_glu_gemm", {"GGML_MI355_NO_GLU_GEMM": "1"}), ("nofuse", {"GGML_MI355_NO_FUSE": "1"})):
        path = str(tmp_path / f"logits_{name}.bin")
        env = dict(os.environ, GGML_MI355_KERNEL_TIMES="1", **extra)
        r = subprocess.run([tool, "wideffn", ftype, "--big-batch", "512", "--dump-logits", path], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        m = re.search(r"\[mi355\] kernel mmdq_glu_\S+ (\d+) ", r.stderr)
        outs[name] = (np.fromfile(path, dtype=np.uint32), int(m.group(1)) if m else 0)
    # one launch: the first of the two layers (the last one computes the feed-forward of the batch's last token only)
    assert outs["fused"][1] == 1 and outs["no_glu_gemm"][1] == 0 and outs["nofuse"][1] == 0
    assert np.array_equal(outs["fused"][0], outs["nofuse"][0])
    assert np.array_equal(outs["fused"][0], outs["no_glu_gemm"][0])


@pytest.mark.parametrize("kv", ["f16", "q8_0", "q4_0"])
def test_small_batch_forms_are_bit_exact_and_run(tmp_path, kv):
    """The 12-token batch of the parity sequence takes the skinny kernels: the norm and the GLU write the q8_K planes themselves
    (mi355_rms_norm_mul_q8k_f32, mi355_glu_q8k_f32) and Q/K/V + 2 ROPE + 2 CPY run as three products and one mi355_qkv_finish.  Each
    form can be switched off alone; the logits never change by a bit, the launch counts show the forms did run."""
    import numpy as np
    tool = os.path.join(ROOT, "tools", "run_llama_bench.sh")
    outs = {}
    for name, extra in (("fused", {}), ("no_q8k", {"GGML_MI355_NO_NORM_Q8K": "1"}), ("no_qkv_small", {"GGML_MI355_NO_QKV_SMALL": "1"}),
                        ("nofuse", {"GGML_MI355_NO_FUSE": "1"})):
        path = str(tmp_path / f"logits_{name}.bin")
        env = dict(os.environ, GGML_MI355_KERNEL_TIMES="1", **extra)
        r = subprocess.run([tool, "tiny", "Q4_K_M", "-ctk", kv, "-ctv", kv, "--dump-logits", path], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        cnt = {k: int(v) for k, v in re.findall(r"\[mi355\] kernel (\S+) (\d+) ", r.stderr)}
        outs[name] = (np.fromfile(path, dtype=np.uint32), cnt)
    for name in ("no_q8k", "no_qkv_small", "nofuse"):
        assert np.array_equal(outs["fused"][0], outs[name][0]), name
    c = {k: v[1] for k, v in outs.items()}
    assert c["fused"].get("glu_q8k", 0) > 0 and c["no_q8k"].get("glu_q8k", 0) == 0 and c["nofuse"].get("glu_q8k", 0) == 0, c["fused"]
    assert c["fused"].get("quantize_q8_K", 0) < c["no_q8k"].get("quantize_q8_K", 0)
    assert c["fused"].get("qkv_finish", 0) > c["no_qkv_small"].get("qkv_finish", 0) and c["nofuse"].get("qkv_finish", 0) == 0
    assert c["fused"].get("rope", 0) < c["no_qkv_small"].get("rope", 0)


@pytest.mark.parametrize("kv", ["f16", "q8_0"])
def test_decode_layer_launch_is_bit_exact_and_runs(tmp_path, kv):
    """GGML_MI355_DECODE_LAYER=1: MUL_MAT(wo) ADD RMS_NORM MUL MUL_MAT MUL_MAT GLU MUL_MAT ADD and the next layer's RMS_NORM MUL + Q/K/V +
    2 ROPE + 2 CPY of every decode token run as ONE persistent launch per layer (mi355_decode_layer; on by default from n_embd 8192).
    A 3-layer model of 2048 x 5120 matrices, 230 decode steps (graph replay, the KV rows through the pointer table, a re-capture at the
    256-row boundary): logits equal, bit for bit, to the unfused graph's and to the four-launch fusions'; the launch did run (one per
    token for every layer but the last -- llama.cpp puts the GET_ROWS of the output rows between its wo and the residual ADD -- and the
    Q/K/V launch only for the first layer; with a q8_0 cache the launch ends at l_out and Q/K/V keep theirs)."""
    import numpy as np
    tool = os.path.join(ROOT, "tools", "run_llama_bench.sh")
    outs = {}
    for name, extra in (("layer", {"GGML_MI355_DECODE_LAYER": "1"}), ("four", {"GGML_MI355_DECODE_LAYER": "0"}), ("nofuse", {"GGML_MI355_NO_FUSE": "1"}),
                        ("layer_counted", {"GGML_MI355_DECODE_LAYER": "1", "GGML_MI355_KERNEL_TIMES": "1"})):
        path = str(tmp_path / f"logits_{name}.bin")
        r = subprocess.run([tool, "mid", "Q4_K_M", "-ctk", kv, "-ctv", kv, "--dump-logits", path, "--decode-steps", "230"], capture_output=True, text=True,
                           timeout=600, env=dict(os.environ, GGML_MI355_GRAPH_STATS="1", **extra))
        assert r.returncode == 0, r.stderr[-3000:]
        cnt = {k: int(v) for k, v in re.findall(r"\[mi355\] kernel (\S+) (\d+) ", r.stderr)}
        m = re.search(r"decode graphs: (\d+) replays", r.stderr)
        outs[name] = (np.fromfile(path, dtype=np.uint32), cnt, int(m.group(1)) if m else 0)
    assert np.array_equal(outs["layer"][0], outs["nofuse"][0])
    assert np.array_equal(outs["layer"][0], outs["four"][0])
    assert np.array_equal(outs["layer"][0], outs["layer_counted"][0])
    assert outs["layer"][2] >= 200, "the decode steps were not replayed from the captured graph"
    c = outs["layer_counted"][1]
    assert c.get("decode_layer", 0) >= 2 * 230, c
    if kv == "f16":
        assert c.get("mmvq_qkv_rope", 0) <= c["decode_layer"] // 2 + 8, c          # the first layer's only


def test_fusions_are_bit_exact_with_a_q8_0_cache(tmp_path):
    """-ctk q8_0 -ctv q8_0: the fused Q/K/V launch leaves K and V as f32 rows in scratch and mi355_kv_store_q8_0 quantizes both into the
    cache -- the same bits as RMS_NORM, 3 MUL_MAT, 2 ROPE and 2 CPY f32 -> q8_0 launched one by one"""
    test_fusions_are_bit_exact(tmp_path, "tiny", kv="q8_0")


def test_fusions_are_bit_exact_with_a_q4_0_cache(tmp_path):
    """-ctk q4_0 -ctv q4_0: the same path with mi355_kv_store_q(Q4_0) -- the bits of the CPY f32 -> q4_0 nodes launched one by one"""
    test_fusions_are_bit_exact(tmp_path, "tiny", kv="q4_0")


@pytest.mark.parametrize("model,kv", [("tiny", "f16"), ("tinymoe", "f16"), ("tiny", "q8_0"), ("tiny", "q4_0")])
def test_decode_graph_replay_is_bit_exact(tmp_path, model, kv):
    """the plugin captures a decode token's launches into a hipGraph and replays it while the ggml graph keeps
    its shape (the KV-cache rows of the new token go through a device table).  230 decode steps after a 48-token prompt
    cross the 256-entry KV padding boundary (a re-capture); logits with replay equal the eager run's bit for bit, and
    nearly all steps were replays."""
    import numpy as np
    tool = os.path.join(ROOT, "tools", "run_llama_bench.sh")
    outs = []
    for eager in (False, True):
        path = str(tmp_path / f"logits_{int(eager)}.bin")
        env = dict(os.environ, GGML_MI355_GRAPH_STATS="1")
        if eager:
            env["GGML_MI355_NO_GRAPHS"] = "1"
        r = subprocess.run([tool, model, "Q4_K_M", "-ctk", kv, "-ctv", kv, "--dump-logits", path, "--decode-steps", "230"], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        outs.append(np.fromfile(path, dtype=np.uint32))
        if not eager:
            m = re.search(r"decode graphs: (\d+) replays, (\d+) captures", r.stderr)
            assert m, r.stderr[-2000:]
            assert int(m.group(1)) >= 220 and int(m.group(2)) <= 4, m.group(0)
    assert outs[0].size == outs[1].size and np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("ndev", [2, 4, 8])
def test_row_split_mul_mat_bit_exact_on_virtual_devices(ndev):
    """the plugin's split buffer type (`-sm row`).  2, 4 or 8 ggml devices mapped onto the one GPU
    (GGML_MI355_VIRTUAL_DEVICES) exercise helper streams, the activation copy, cross-stream events and row
    offsets; MUL_MAT on a split weight must equal MUL_MAT on the unsplit weight bit for bit, for all five
    weight types, mat-vec / skinny / GEMM kernels (incl. the K-cut GEMM at K >= 1024), skewed splits, any device as
    main, a device in the middle that owns no rows, and Llama-3-70B's row counts (tools/test_split_mul_mat.cpp)."""
    exe = os.path.join(ROOT, "oracle", "_ref", "test-split-mul-mat")
    assert os.path.exists(exe), "oracle/_ref/test-split-mul-mat is not built (make -C oracle drivers)"
    env = dict(os.environ, GGML_BACKEND_PATH=PLUGIN, GGML_MI355_VIRTUAL_DEVICES=str(ndev))
    r = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    assert " 0 failed" in r.stdout


def test_row_split_rccl_all_gather_one_rank(tmp_path):
    """GGML_MI355_SPLIT_GATHER=rccl: the row-split gather as an RCCL all-gather (ncclCommInitAll, one rank per device).  A one-GPU box
    can only give it one rank -- RCCL refuses several ranks on one GPU -- which still runs communicator set-up, the local slice, the
    collective and the placement into dst on hardware; results equal the unsplit ones bit for bit.  With virtual devices the switch
    must fall back to the peer-store path (and say so) rather than fail."""
    exe = os.path.join(ROOT, "oracle", "_ref", "test-split-mul-mat")
    assert os.path.exists(exe), "oracle/_ref/test-split-mul-mat is not built (make -C oracle drivers)"
    env = dict(os.environ, GGML_BACKEND_PATH=PLUGIN, GGML_MI355_SPLIT_GATHER="rccl")
    env.pop("GGML_MI355_VIRTUAL_DEVICES", None)
    r = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "RCCL all-gather over 1 devices" in r.stdout + r.stderr, (r.stdout + r.stderr)[-3000:]
    assert " 0 failed" in r.stdout and "20 cases ran" in r.stdout, r.stdout[-2000:]
    env["GGML_MI355_VIRTUAL_DEVICES"] = "2"
    r = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "using peer stores" in r.stdout + r.stderr, (r.stdout + r.stderr)[-3000:]


def test_llama_row_split_logits_equal_single_device(tmp_path):
    """Whole model with `-sm row -ts 1,1` over two virtual devices (layers 0-1 main on MI355_0, 2-3 on MI355_1,
    every weight matrix row-split over both) gives the same logits as the single-device run, to the bit."""
    import numpy as np
    tool = os.path.join(ROOT, "tools", "run_llama_bench.sh")
    outs = []
    for split in (False, True):
        path = str(tmp_path / f"logits_{int(split)}.bin")
        env = dict(os.environ)
        args = [tool, "tiny", "Q4_K_M", "--dump-logits", path]
        if split:
            env["GGML_MI355_VIRTUAL_DEVICES"] = "2"
            args += ["-sm", "row", "-ts", "1,1"]
        r = subprocess.run(args, capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        if split:
            assert "MI355_0_Split" in r.stderr + r.stdout, (r.stderr + r.stdout)[-3000:]
        outs.append(np.fromfile(path, dtype=np.uint32))
    assert np.array_equal(outs[0], outs[1])


def test_row_split_decode_fused_forms_and_graph_replay_are_bit_exact(tmp_path):
    """`-sm row` decode over two virtual devices with the fused launches running on row slices (Q/K/V + rope + KV store, wo / ffn_down
    + residual, gate / up + GLU; mi355_split_fan_out) and the token replayed from a captured graph whose K / V table entries carry the
    slice offsets: 70 decode steps of the `mid` model (2048 wide, 4 KV heads of 128: every slice is whole heads) give the unsplit run's
    logits to the bit -- with the same-GPU slices on the main stream (default), through helper streams and events
    (GGML_MI355_SPLIT_STREAMS=1: the fork / join path a multi-GPU node takes), and with the split forms off (one MUL_MAT at a time,
    eager) -- and the split graphs did replay."""
    import numpy as np
    tool = os.path.join(ROOT, "tools", "run_llama_bench.sh")
    variants = {"unsplit": None, "split": {}, "split_streams": {"GGML_MI355_SPLIT_STREAMS": "1"}, "split_unfused": {"GGML_MI355_NO_SPLIT_FUSE": "1"}}
    outs = {}
    for name, extra in variants.items():
        path = str(tmp_path / f"logits_{name}.bin")
        env = dict(os.environ, GGML_MI355_GRAPH_STATS="1")
        args = [tool, "mid", "Q4_K_M", "--dump-logits", path, "--decode-steps", "70"]
        if extra is not None:
            env.update(extra, GGML_MI355_VIRTUAL_DEVICES="2")
            args += ["-sm", "row", "-ts", "1,1"]
        r = subprocess.run(args, capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        outs[name] = np.fromfile(path, dtype=np.uint32)
        replays = sum(int(x) for x in re.findall(r"decode graphs: (\d+) replays", r.stderr))
        if name in ("split", "split_streams"):
            assert replays >= 60, r.stderr[-2000:]
        if name == "split_unfused":
            assert replays == 0, r.stderr[-2000:]
    for name in ("split", "split_streams", "split_unfused"):
        assert outs[name].size == outs["unsplit"].size and np.array_equal(outs[name], outs["unsplit"]), name


@pytest.mark.parametrize("ndev,ts", [(2, "1,1"), (8, "1,1,1,1,1,1,1,1")])
def test_row_split_decode_70b_shapes_bit_exact(tmp_path, ndev, ts):
    """Llama-3-70B's shapes (two layers of them) row-split over 2 and 8 virtual devices -- 8 KV heads over 8 devices is one head per
    device, the finest cut the fused Q/K/V form takes -- against the unsplit logits: equal to the bit, and the split tokens replayed."""
    import numpy as np
    tool = os.path.join(ROOT, "tools", "run_llama_bench.sh")
    ref_path, path = str(tmp_path / "ref.bin"), str(tmp_path / "split.bin")
    r = subprocess.run([tool, "70b2", "Q4_K_M", "--dump-logits", ref_path, "--decode-steps", "24"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, GGML_MI355_VIRTUAL_DEVICES=str(ndev), GGML_MI355_GRAPH_STATS="1")
    r = subprocess.run([tool, "70b2", "Q4_K_M", "--dump-logits", path, "--decode-steps", "24", "-sm", "row", "-ts", ts], capture_output=True, text=True,
                       timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    assert np.array_equal(np.fromfile(ref_path, dtype=np.uint32), np.fromfile(path, dtype=np.uint32))
    assert sum(int(x) for x in re.findall(r"decode graphs: (\d+) replays", r.stderr)) >= 20, r.stderr[-2000:]


@pytest.mark.parametrize("ndev,ts", [(3, "2,1,1"), (8, "1,1,1,1,1,1,1,1"), (4, "0,1,1,2"), (2, "3,1")])
def test_row_split_decode_uneven_layouts_are_bit_exact(tmp_path, ndev, ts):
    """The split decode forms over uneven tensor splits, a device that owns nothing, and eight devices (where a device owns rows of wq
    but none of wk: the fused Q/K/V form steps aside for that layer and the token is not captured): 40 decode steps of the `mid`
    model give the unsplit logits to the bit."""
    import numpy as np
    tool = os.path.join(ROOT, "tools", "run_llama_bench.sh")
    ref_path, path = str(tmp_path / "ref.bin"), str(tmp_path / "split.bin")
    r = subprocess.run([tool, "mid", "Q4_K_M", "--dump-logits", ref_path, "--decode-steps", "40"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, GGML_MI355_VIRTUAL_DEVICES=str(ndev))
    r = subprocess.run([tool, "mid", "Q4_K_M", "--dump-logits", path, "--decode-steps", "40", "-sm", "row", "-ts", ts], capture_output=True, text=True,
                       timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    assert np.array_equal(np.fromfile(ref_path, dtype=np.uint32), np.fromfile(path, dtype=np.uint32))
