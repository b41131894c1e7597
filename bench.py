#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X hot path.

Metric (BASELINE.json): "Q4_K mat-vec GB/s (% HBM peak) ...".  One "step" = one decode token's
worth of quantized mat-vecs of Llama-3-8B Q4_K_M: all 225 weight matrices (32 layers x
{wq, wk, wv, wo, gate, up, down} + output; Q4_K with Q6_K where llama.cpp's Q4_K_M recipe uses it)
streamed once from HBM in graph order, each preceded by the q8_1 quantization of the activation
that feeds it (4 per layer).  Weights are random valid GGUF blocks resident in HBM (4.6 GB, so
every launch is cold with respect to the 256 MiB Infinity Cache); activations are synthetic.
value = algorithmic bytes of the whole step / step time (GB/s, whole job).

  python bench.py --gpus N --steps K --warmup W     (N > 1: launched by torch.distributed.run)

N > 1 is the reference's `-sm row`: every matrix is row-split across the ranks
(llm_amd/split.py), each rank runs the mat-vecs on its slices -- a layer's launches replayed as one
graph -- and a layer's slices are combined with one RCCL all-gather; the total work is fixed, so
"scaling" is "strong".
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="llama3-8b")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the llama-bench style tg128 / pp512 leg")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the llama-bench legs of BASELINE.json's configs 3-5 (Q8_0 pp2048 with a q8_0 cache, 70B, 8x7B)")
    ap.add_argument("--no-group", action="store_true", help="one launch per matrix instead of grouped QKV / gate+up launches")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a HIP graph")
    ap.add_argument("--no-glu-fuse", action="store_true", help="grouped gate+up mat-vec and a separate quantization instead of mi355_glu_matvec")
    ap.add_argument("--no-norm-fuse", action="store_true", help="pre-quantized activations instead of the rms-norm prologue / residual epilogue launches")
    return ap.parse_args()


def make_weights(torch, B, t, m, k, gen):
    """random valid blocks of type t on the GPU: random quant bytes, finite fp16 scales."""
    rs = B.row_size(t, k)
    nb = k // B.BLCK_SIZE[t]
    w = torch.randint(0, 256, (m * nb, B.TYPE_SIZE[t]), dtype=torch.uint8, device="cuda", generator=gen)
    scale = {B.Q4_0: 0.05, B.Q8_0: 0.005, B.Q4_K: 0.003, B.Q5_K: 0.002, B.Q6_K: 0.0005}[t]
    for off in B._HALF_FIELDS[t]:
        d = ((torch.rand(m * nb, device="cuda", generator=gen) * 0.75 + 0.25) * scale).to(torch.float16)
        w[:, off:off + 2] = d.view(torch.uint8).view(m * nb, 2)
    return w.view(m, rs)


def _ref_bench(args, timeout):
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_bench")
    if not os.path.exists(exe):
        raise FileNotFoundError("oracle/_ref/ref_bench is not built (make -C oracle ref)")
    env = dict(os.environ, OMP_NUM_THREADS=str(args[4]), OMP_WAIT_POLICY="active")
    out = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, timeout=timeout, check=True, env=env)
    return json.loads(out.stdout.strip().splitlines()[-1])


def cpu_baseline(threads):
    """ggml-cpu (the reference itself, oracle/_ref built from its sources) on the host cores, timed BEFORE this process
    touches the GPU: the bench's dominant shape (q4_K 14336x4096, cold: rotating over 2.1 GB) and BASELINE.json's
    config 1 (q4_0 4096x4096 n=1).  If the reference binary is missing or fails, the 1-core scalar oracle port is timed
    instead and the record says so loudly ("degraded": true, with the reason)."""
    try:
        r = _ref_bench(["q4_K", 14336, 4096, 1, threads, 8, 64], 240)
        base = {"value": round(r["gbps"], 2), "unit": "GB/s", "cores": threads, "kind": "reference", "degraded": False,
                "sample": f"ggml-cpu MUL_MAT q4_K 14336x4096 n=1 rotating over 64 matrices (2.1 GB, cold), "
                          f"{r['runs']} runs in ~8 s, {r['us_per_run']:.1f} us/run"}
        try:
            c = _ref_bench(["q4_0", 4096, 4096, 1, threads, 4, 128], 120)
            base["config1"] = {"value": round(c["gbps"], 2), "unit": "GB/s", "us_per_run": round(c["us_per_run"], 1), "cores": threads,
                               "kind": "reference",
                               "sample": "BASELINE.json configs[0]: ggml-cpu MUL_MAT q4_0 4096x4096 n=1, rotating over 128 matrices "
                                         f"(1.2 GB, cold), {c['runs']} runs in ~4 s"}
        except Exception as e:  # noqa: BLE001
            base["config1"] = {"error": str(e)[:200]}
        return base
    except Exception as e:  # noqa: BLE001 -- fall back to the port, but say so in the record
        reason = f"{type(e).__name__}: {e}"[:300]
        print(f"[bench] oracle/_ref/ref_bench failed ({reason}); timing the 1-core oracle port instead: cpu_baseline is DEGRADED",
              file=sys.stderr)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as O
    import llm_amd.blocks as B
    m, k = 512, 4096
    w = B.random_blocks(B.Q4_K, m, k, seed=1)
    x = np.random.default_rng(0).standard_normal((1, k)).astype(np.float32)
    t0, runs = time.perf_counter(), 0
    while time.perf_counter() - t0 < 10.0:
        O.mul_mat(B.Q4_K, w, x, m, k, 1, "cpu")
        runs += 1
    us = (time.perf_counter() - t0) / runs * 1e6
    return {"value": round(B.matvec_bytes(B.Q4_K, m, k) / us * 1e-3, 3), "unit": "GB/s", "cores": 1, "kind": "port", "degraded": True,
            "degraded_reason": reason,
            "sample": f"oracle (scalar C restatement) q4_K {m}x{k} n=1, {runs} runs in ~10 s -- NOT the 16-thread reference number"}


def end_to_end(budget_s):
    """BASELINE.json's other half of the metric: llama-bench's tg128 / pp512 for a random-weight Llama-3-8B Q4_K_M GGUF
    through the reference's libllama (oracle/_ref, built from its sources) with this plugin as the GPU backend.  The
    harness is the reference's own llama-bench (tools/llama-bench/llama-bench.cpp, compiled unmodified by `make -C oracle
    tools`) when it is built, else tools/llama_bench_driver.cpp (same method, llama-bench.cpp:1747-1795).  Run as a child
    process within a time budget."""
    ref = os.path.join(ROOT, "oracle", "_ref")
    plugin = os.path.join(ROOT, "llm_amd", "lib", "libggml-mi355.so")
    stock = os.path.exists(os.path.join(ref, "llama-bench"))
    need = [os.path.join(ref, x) for x in ("llama-bench" if stock else "mi355-llama-bench", "make-random-gguf", "libllama.so")] + [plugin]
    missing = [os.path.relpath(x, ROOT) for x in need if not os.path.exists(x)]
    if missing:
        return {"error": "not built: " + ", ".join(missing)}
    t0 = time.perf_counter()
    gguf = "/tmp/random-8b-Q4_K_M.gguf"
    try:
        if not os.path.exists(gguf):
            subprocess.run([os.path.join(ref, "make-random-gguf"), gguf, "8b", "Q4_K_M"], check=True, capture_output=True, timeout=budget_s)
        if stock:
            r = subprocess.run([os.path.join(ref, "llama-bench"), "-m", gguf, "-ngl", "99", "-fa", "1", "-p", "512", "-n", "128", "-r", "3", "-o", "json"],
                               capture_output=True, text=True, timeout=budget_s, env=dict(os.environ, GGML_BACKEND_PATH=plugin))
            rows = json.loads(r.stdout)
            pp = [x for x in rows if x["n_prompt"] == 512 and x["n_gen"] == 0][0]
            tg = [x for x in rows if x["n_gen"] == 128 and x["n_prompt"] == 0][0]
            j = {"devices": tg["gpu_info"], "tg_avg_ts": tg["avg_ts"], "tg_stddev_ts": tg["stddev_ts"], "pp_avg_ts": pp["avg_ts"], "pp_stddev_ts": pp["stddev_ts"]}
        else:
            r = subprocess.run([os.path.join(ROOT, "tools", "run_llama_bench.sh"), "8b", "Q4_K_M", "-p", "512", "-n", "128", "-r", "3"],
                               capture_output=True, text=True, timeout=budget_s)
            j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    except Exception as e:  # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"[:300]}
    bytes_per_token = 4.616e9       # weights streamed per decode token
    return {"model": "Llama-3-8B Q4_K_M, random-init GGUF (tools/make_random_gguf.cpp), f16 KV, flash attention", "harness":
            ("the reference's llama-bench (unmodified, oracle/_ref/llama-bench)" if stock else "tools/llama_bench_driver.cpp (llama-bench's method)") +
            " + reference libllama + plugin via GGML_BACKEND_PATH, 3 reps", "devices": j.get("devices"),
            "tg128_tps": round(j["tg_avg_ts"], 1), "tg128_stddev": round(j["tg_stddev_ts"], 1),
            "pp512_tps": round(j["pp_avg_ts"], 1), "pp512_stddev": round(j["pp_stddev_ts"], 1),
            "tg128_frac_of_hbm_bound": round(j["tg_avg_ts"] * bytes_per_token / (HBM_PEAK_GBPS * 1e9), 4),
            "seconds": round(time.perf_counter() - t0, 1)}


def other_configs(budget_s):
    """BASELINE.json configs 3-5 on ONE GPU through the reference's own llama-bench + plugin (random-weight GGUFs written to /tmp
    and removed again): 8B Q8_0 pp2048 with a q8_0 KV cache; 70B Q4_K_M pp512 / tg64 (its 1-GPU row; the 2/4/8-GPU rows need a
    multi-GPU node); Mixtral-8x7B-shaped Q4_K_M batched decode pp8..64 and tg64.  Every leg has its own time limit and the legs
    stop when the budget is used up, so the default run stays within minutes."""
    ref = os.path.join(ROOT, "oracle", "_ref")
    plugin = os.path.join(ROOT, "llm_amd", "lib", "libggml-mi355.so")
    exe, mk = os.path.join(ref, "llama-bench"), os.path.join(ref, "make-random-gguf")
    if not all(os.path.exists(x) for x in (exe, mk, plugin)):
        return {"error": "oracle/_ref/llama-bench, make-random-gguf or the plugin is not built"}
    # (name, model, ftype, llama-bench arguments, extra environment, keep the GGUF for the next leg)
    legs = [("config3_8b_q8_0_q8kv", "8b", "Q8_0", ["-p", "2048", "-n", "0", "-ctk", "q8_0", "-ctv", "q8_0"], {}, False),
            ("config5_8x7b_q4km", "8x7b", "Q4_K_M", ["-p", "8,16,32,64,512", "-n", "64"], {}, False),
            ("config4_70b_q4km_1gpu", "70b", "Q4_K_M", ["-p", "512", "-n", "64"], {}, True),
            # the row-split code path of config 4 (-sm row: fused launches on row slices, the token replayed as a graph) rehearsed on
            # this one GPU as two ggml devices: what the split costs, not what it gains -- the 2 / 4 / 8-GPU rows need a node
            ("config4_70b_q4km_row_split_2_virtual_devices", "70b", "Q4_K_M", ["-p", "0", "-n", "64", "-sm", "row", "-ts", "1/1"],
             {"GGML_MI355_VIRTUAL_DEVICES": "2"}, False)]
    out, t_all = {}, time.perf_counter()
    for name, model, ftype, extra, leg_env, keep in legs:
        left = budget_s - (time.perf_counter() - t_all)
        if left < 45:
            out[name] = {"skipped": "time budget of the default run used up"}
            continue
        gguf = f"/tmp/random-{model}-{ftype}.gguf"
        t0 = time.perf_counter()
        try:
            if not os.path.exists(gguf):
                subprocess.run([mk, gguf, model, ftype], check=True, capture_output=True, timeout=left)
            r = subprocess.run([exe, "-m", gguf, "-ngl", "99", "-fa", "1", "-r", "2", "-o", "json"] + extra, capture_output=True, text=True,
                               timeout=max(30, left - (time.perf_counter() - t0)), env=dict(os.environ, GGML_BACKEND_PATH=plugin, **leg_env))
            rows = json.loads(r.stdout)
            res = {}
            for x in rows:
                key = f"pp{x['n_prompt']}" if x["n_gen"] == 0 else f"tg{x['n_gen']}"
                res[key + "_tps"] = round(x["avg_ts"], 1)
            res["seconds"] = round(time.perf_counter() - t0, 1)
            out[name] = res
        except Exception as e:  # noqa: BLE001
            out[name] = {"error": f"{type(e).__name__}: {e}"[:200]}
        finally:
            if not keep:
                try:
                    os.remove(gguf)
                except OSError:
                    pass
    for leftover in {f"/tmp/random-{m}-{t}.gguf" for _, m, t, _, _, _ in legs}:
        if os.path.exists(leftover):
            os.remove(leftover)
    out["harness"] = "the reference's llama-bench (unmodified) + reference libllama + plugin, 2 reps, random-init GGUFs (tools/make_random_gguf.cpp), flash attention"
    return out


def main():
    args = parse()
    # the CPU baseline runs first, before this process initialises the GPU (and before torch's threads exist)
    base = None
    if int(os.environ.get("RANK", "0")) == 0 and int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_cpu_baseline:
        base = cpu_baseline(min(16, os.cpu_count() or 1))
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            print(f"--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus}", file=sys.stderr)
            sys.exit(2)
    # MI355_BENCH_BACKEND=gloo is a rehearsal switch for boxes with fewer GPUs than ranks (ranks then share devices and
    # the all-gather goes through gloo); the measured configuration is always nccl = RCCL, one rank per GPU
    backend = os.environ.get("MI355_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)

    def say(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)
    say(f"process group ready: world {world}, backend {backend if world > 1 else 'none'}")

    import llm_amd.blocks as B
    import llm_amd.kernels as K
    import llm_amd.model_shapes as MS
    import llm_amd.ops as ops
    from llm_amd.split import all_gather_rows, row_split

    dims = MS.MODELS[args.model]
    mats = MS.decode_matrices(dims, "Q4_K_M")
    gen = torch.Generator(device="cuda")
    gen.manual_seed(1234 + rank)

    # ---- resident state: weights (row slice of this rank), activations, outputs, act scratch
    weights, rows = [], []
    for mt in mats:
        lo, hi = row_split(mt.m, world)[rank]
        rows.append((lo, hi))
        weights.append(make_weights(torch, B, mt.type, hi - lo, mt.k, gen))
    xs = {k: torch.randn((1, k), device="cuda", generator=gen) for k in sorted({mt.k for mt in mats})}
    acts = {k: (torch.empty((1, k), dtype=torch.int8, device="cuda"),
                torch.empty((1, k // 32, 2), dtype=torch.float32, device="cuda")) for k in xs}
    outs = [torch.empty((1, hi - lo), dtype=torch.float32, device="cuda") for lo, hi in rows]

    say(f"{len(mats)} weight matrices resident ({sum(w.numel() for w in weights) / 1e9:.2f} GB on this rank)")

    def quantize(k, stream):
        aq, ads = acts[k]
        K.check(K.lib.mi355_quantize_act(xs[k].data_ptr(), aq.data_ptr(), ads.data_ptr(), k, 1, 1, 1, k * 4, 0, 0, stream),
                "mi355_quantize_act")

    # Launch plan: matrices that read the same activation and have the same block type (wq/wk[/wv], gate/up)
    # share ONE grouped launch (mi355_mul_mat_vec_q_group); --no-group launches every matrix on its own.
    def mm_arg(mt, w, o):
        a = K.MmvqArgs()
        aq, ads = acts[mt.k]
        a.type, a.W, a.w_row_stride, a.w_ch_stride = mt.type, w.data_ptr(), B.row_size(mt.type, mt.k), 0
        a.act_q, a.act_ds, a.act_ch_stride = aq.data_ptr(), ads.data_ptr(), 0
        a.dst, a.dst_col_stride, a.dst_ch_stride = o.data_ptr(), o.shape[1], 0
        a.m, a.k, a.ncols, a.nchannels, a.ch_ratio = o.shape[1], mt.k, 1, 1, 1
        return a

    # ffn gate + up of a layer run as the fused mi355_glu_matvec (both mat-vecs, the gated activation and the q8_1 form
    # of its result in one launch -- what the plugin does for a decode token), so ffn_down's activation quantization
    # is not a launch of its own.  Row-split ranks (N > 1) hold slices of both matrices and keep the grouped mat-vec.
    fuse_glu = world == 1 and not args.no_group and not args.no_glu_fuse
    hidden = torch.empty((1, dims.n_ff), dtype=torch.float32, device="cuda")
    # ... and, as in the plugin's decode graph, the RMS_NORM -> MUL(weight) -> q8_1 of the layer input runs
    # in the prologue of the launch that consumes it (wq/wk/wv with rope and the KV-cache store: mi355_qkv_rope_matvec;
    # gate/up: mi355_glu_matvec) and wo / ffn_down add the residual in their epilogue, so the only activation
    # quantizations left as launches are the attention output's (attention itself is not part of this bench) and the
    # final norm's.
    fuse_norm = fuse_glu and not args.no_norm_fuse and dims.n_embd <= 4096
    norm_w = torch.rand(dims.n_embd, device="cuda", generator=gen) + 0.5
    normed = torch.empty((1, dims.n_embd), dtype=torch.float32, device="cuda")
    resid = torch.randn((1, dims.n_embd), device="cuda", generator=gen)
    resid_out = torch.empty((1, dims.n_embd), dtype=torch.float32, device="cuda")
    kv_dim = dims.n_head_kv * dims.head_dim
    k_row = torch.empty(kv_dim, dtype=torch.float16, device="cuda")
    v_row = torch.empty(kv_dim, dtype=torch.float16, device="cuda")
    pos = torch.tensor([127], dtype=torch.int32, device="cuda")

    def glu_arg(ig, iu):
        mg = mats[ig]
        a = K.GluMatvecArgs()
        aq, ads = acts[mg.k]
        oq, ods = acts[dims.n_ff]
        a.type, a.glu_op, a.w_gate, a.w_up = mg.type, 2, weights[ig].data_ptr(), weights[iu].data_ptr()
        a.gate_row_stride = a.up_row_stride = B.row_size(mg.type, mg.k)
        a.m, a.k, a.act_q, a.act_ds = mg.m, mg.k, aq.data_ptr(), ads.data_ptr()
        a.gate_out, a.up_out, a.dst = outs[ig].data_ptr(), outs[iu].data_ptr(), hidden.data_ptr()
        a.out_q, a.out_ds = oq.data_ptr(), ods.data_ptr()
        if fuse_norm:
            a.act_q = a.act_ds = None
            a.norm_x, a.norm_w, a.norm_out, a.norm_eps = xs[mg.k].data_ptr(), norm_w.data_ptr(), normed.data_ptr(), 1e-5
        return a

    def qkv_arg(iq, ik, iv):
        mq, mk, mv = mats[iq], mats[ik], mats[iv]
        a = K.QkvRopeArgs()
        a.type_qk, a.type_v, a.type_k = mq.type, mv.type, 0
        a.wq, a.wk, a.wv = weights[iq].data_ptr(), weights[ik].data_ptr(), weights[iv].data_ptr()
        a.wq_row_stride = a.wk_row_stride = B.row_size(mq.type, mq.k)
        a.wv_row_stride = B.row_size(mv.type, mv.k)
        a.m_q, a.m_k, a.m_v, a.k = mq.m, mk.m, mv.m, mq.k
        a.q_out, a.k_out, a.v_out = outs[iq].data_ptr(), k_row.data_ptr(), v_row.data_ptr()
        a.k_out_type = a.v_out_type = B.F16
        a.pos, a.freq_factors = pos.data_ptr(), None
        a.head_dim, a.n_dims, a.mode, a.n_ctx_orig = dims.head_dim, dims.head_dim, 0, 8192
        a.freq_base, a.freq_scale, a.ext_factor, a.attn_factor, a.beta_fast, a.beta_slow = 500000.0, 1.0, 0.0, 1.0, 32.0, 1.0
        a.norm_x, a.norm_w, a.norm_out, a.norm_eps = xs[mq.k].data_ptr(), norm_w.data_ptr(), normed.data_ptr(), 1e-5
        return a

    plan = []      # (quantize_k or None, ctypes array of MmvqArgs | GluMatvecArgs, n (0 = fused glu), [matrix indices])
    last_src = None
    for i, mt in enumerate(mats):
        need_q = mt.k if mt.src != last_src else None   # one q8_1 quantization per distinct activation, as in the graph
        last_src = mt.src
        prev = plan[-1] if plan else None
        if fuse_norm and mt.src == "attn_norm" and prev and mats[prev[3][0]].src == "attn_norm" and need_q is None:
            prev[3].append(i)        # wq, wk, wv of a layer: one mi355_qkv_rope_matvec launch, whatever wv's type
            continue
        if (not args.no_group and need_q is None and prev and len(prev[3]) < 4 and mats[prev[3][0]].type == mt.type
                and mats[prev[3][0]].k == mt.k and mats[prev[3][0]].src == mt.src):
            prev[3].append(i)
        else:
            plan.append([need_q, None, 0, [i]])
    # N > 1: the row slices a LAYER produces are exchanged with ONE all-gather (a real row-split decoder needs three exchanges per
    # layer -- attention output, gated activation, layer output; the launches of this bench do not consume each other's results, so
    # one per layer is the lower bound the schedule is priced at).  Slices of a matrix are equal on all ranks whenever M is a
    # multiple of 128 * N (all 8B matrices but output.weight): those outputs are views into the layer's send buffer and the gather
    # needs no staging; the others take the padded path of llm_amd/split.py, one gather each.  A layer's launches are captured
    # into one graph per layer; the collectives stay outside the graphs.
    def layer_of(p):
        name = mats[p[3][0]].name
        return name.split(".")[1] if name.startswith("blk.") else name
    segments = []
    if world > 1:
        for j, p in enumerate(plan):
            if not segments or segments[-1]["key"] != layer_of(p):
                segments.append({"key": layer_of(p), "plan": [], "eq": [], "uneq": [], "graph": None})
            segments[-1]["plan"].append(j)
            for i in p[3]:
                equal = len({hi - lo for lo, hi in row_split(mats[i].m, world)}) == 1
                segments[-1]["eq" if equal else "uneq"].append(i)
        for sg in segments:
            sizes = [rows[i][1] - rows[i][0] for i in sg["eq"]]
            sg["send"] = torch.empty(max(1, sum(sizes)), dtype=torch.float32, device="cuda") if sizes else None
            sg["recv"] = torch.empty(world * sum(sizes), dtype=torch.float32, device="cuda") if sizes else None
            off = 0
            for i, n_ in zip(sg["eq"], sizes):
                outs[i] = sg["send"][off:off + n_].view(1, n_)
                off += n_
    for p in plan:
        names = [mats[i].name.split(".")[-1] for i in p[3]]
        if fuse_glu and names == ["ffn_gate", "ffn_up"] and mats[p[3][0]].m % 32 == 0 and mats[p[3][0]].m <= 16384:
            p[1], p[2] = glu_arg(*p[3]), 0
            if fuse_norm:
                p[0] = None
        elif fuse_norm and names == ["attn_q", "attn_k", "attn_v"]:
            p[0], p[1], p[2] = None, qkv_arg(*p[3]), -1
        else:
            arr = (K.MmvqArgs * len(p[3]))(*[mm_arg(mats[i], weights[i], outs[i]) for i in p[3]])
            if fuse_norm and names in (["attn_output"], ["ffn_down"]):       # + residual ADD in the epilogue
                arr[0].residual, arr[0].sum = resid.data_ptr(), resid_out.data_ptr()
            if fuse_norm and names == ["attn_output"]:
                # the attention output arrives as f32 (attention is not part of this bench): wo quantizes it in its prologue
                arr[0].act_q = arr[0].act_ds = None
                arr[0].src_f32 = xs[mats[p[3][0]].k].data_ptr()
                p[0] = None
            if fuse_norm and names == ["output"]:
                # final norm in the prologue of the output mat-vec
                arr[0].act_q = arr[0].act_ds = None
                arr[0].src_f32, arr[0].norm_w, arr[0].norm_out, arr[0].norm_eps = xs[mats[p[3][0]].k].data_ptr(), norm_w.data_ptr(), normed.data_ptr(), 1e-5
                p[0] = None
            p[1], p[2] = arr, len(p[3])
    for prev, p in zip(plan, plan[1:]):
        if prev[2] == 0 and mats[p[3][0]].src == "ffn_act":
            p[0] = None          # ffn_down reads the q8_1 activations the fused launch just wrote

    # lab: BENCH_PREFETCH_MB=<x> reads the first x MB of the NEXT launch's weights on a second stream while a
    # launch runs, so that its first loads come from the memory-side cache; off by default
    pf_mb = float(os.environ.get("BENCH_PREFETCH_MB", "0"))
    pf_wgs = int(os.environ.get("BENCH_PREFETCH_WGS", "64"))
    pf_stream = torch.cuda.Stream() if pf_mb > 0 else None

    def prefetch_next(j):
        if j + 1 >= len(plan):
            return
        cur = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(cur)
        pf_stream.wait_event(ev)
        left = int(pf_mb * 1e6)
        for i in plan[j + 1][3]:
            nb = min(left // len(plan[j + 1][3]), weights[i].numel()) // 16 * 16
            if nb > 0:
                K.check(K.lib.mi355_prefetch(weights[i].data_ptr(), nb, pf_wgs, pf_stream.cuda_stream), "mi355_prefetch")

    def launch(j):
        need_q, arr, n, idx = plan[j]
        stream = torch.cuda.current_stream().cuda_stream
        if pf_stream is not None:
            prefetch_next(j)
        if need_q is not None:
            quantize(need_q, stream)
        if n == 0:
            K.check(K.lib.mi355_glu_matvec(arr, stream), "mi355_glu_matvec")
        elif n < 0:
            K.check(K.lib.mi355_qkv_rope_matvec(arr, stream), "mi355_qkv_rope_matvec")
        else:
            K.check(K.lib.mi355_mul_mat_vec_q_group(arr, n, stream), "mi355_mul_mat_vec_q_group")

    def step(eager=False):
        """one decode token's mat-vecs on the current stream (eager: no per-layer graphs -- the timing mode's event pairs are not replayed)"""
        if world == 1:
            for j in range(len(plan)):
                launch(j)
        for sg in segments:
            if sg["graph"] is not None and not eager:
                sg["graph"].replay()
            else:
                for j in sg["plan"]:
                    launch(j)
            # rehearsal backends (gloo) gather through host memory; nccl = RCCL gathers device buffers over xGMI
            if sg["send"] is not None and backend == "nccl":
                dist.all_gather_into_tensor(sg["recv"], sg["send"])
            elif sg["send"] is not None:
                dist.all_gather_into_tensor(sg["recv"].cpu(), sg["send"].cpu())
            for i in sg["uneq"]:
                all_gather_rows(outs[i] if backend == "nccl" else outs[i].cpu(), mats[i].m, rank, world, dist)
        if pf_stream is not None:      # join the second stream (a captured graph must end on one stream)
            ev = torch.cuda.Event()
            ev.record(pf_stream)
            torch.cuda.current_stream().wait_event(ev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    use_graph = world == 1 and not args.no_graph
    run = step
    if world > 1 and not args.no_graph and pf_stream is None:
        # one graph per layer: its launches replay without the host in between; the all-gathers are issued between the graphs
        step()
        torch.cuda.synchronize()
        for sg in segments:
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                for j in sg["plan"]:
                    launch(j)
            sg["graph"] = gr
        say(f"{len(segments)} per-layer graphs captured")
    if use_graph:
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            step()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step()
        run = g.replay

    for _ in range(args.warmup):
        run()
    barrier()
    say("warm-up done, timing")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    alg_bytes = MS.decode_algorithmic_bytes(dims, "Q4_K_M")     # whole job (all ranks together)
    ms_per_step = dt / args.steps * 1e3
    value = alg_bytes / (dt / args.steps) * 1e-9

    # ---- roofline of the dominant kernel: mmvq q4_K (n=1).  Duration = the dispatch's own start/stop
    # events on the launch stream (hipExtLaunchKernelGGL, the library's timing mode), i.e. kernel
    # execution time without launch gaps -- the same quantity rocprofv3 --kernel-trace reports.
    K.lib.mi355_timing_reset()
    K.lib.mi355_timing_enable(1)
    nroof = max(1, min(args.steps, 5))
    for _ in range(nroof):
        step(eager=True)
    torch.cuda.synchronize()
    K.lib.mi355_timing_enable(0)
    # dominant kernel by time: the fused gate/up launch (mmvq_glu_kernel<q4_K>) when it is in the plan, else the grouped
    # mat-vec (mmvq_group_kernel<q4_K, 1>).  Algorithmic bytes = SURVEY 8(d)'s M*(K/256)*144 + 4K + 4M per matrix.
    fused_idx = [i for p in plan if p[2] == 0 for i in p[3]]
    if fused_idx:
        tag, kname, roof_idx = "mmvq_glu", "mmvq_glu_kernel<q4_K> (ffn gate + up + GLU + q8_1)", fused_idx
    else:
        tag, kname = "mmvq_q4_K_n1", "mmvq_group_kernel<q4_K, ncols=1>"
        roof_idx = [i for i, mt in enumerate(mats) if mt.type == B.Q4_K]
    n_launch, tot_us = K.timing_read(tag)
    q4k_bytes = nroof * sum(B.matvec_bytes(mats[i].type, outs[i].shape[1], mats[i].k) for i in roof_idx)
    avg_us = tot_us / n_launch
    achieved = q4k_bytes / tot_us * 1e-3
    kernels = {t_: {"launches": n, "avg_us": round(us / n, 3)} for t_, (n, us) in K.timing_report().items() if n}
    # HBM bytes per launch of that kernel from the PMC counters: collected in a run of its own (rocprofv3 --pmc cannot be
    # combined with this timed run) by tools/pmc_traffic.sh on this same workload and committed under profiles/.
    traffic, tr_name = None, ""
    if world == 1 and not args.no_group:
        try:
            tr_name = "r03_bench_traffic.json" if os.path.exists(os.path.join(ROOT, "profiles", "r03_bench_traffic.json")) else "r02_bench_traffic.json"
            tr = json.load(open(os.path.join(ROOT, "profiles", tr_name)))
            if tr["kernel"].startswith("mmvq_glu") == bool(fused_idx):
                traffic = tr["traffic_bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            traffic = None
    roofline = {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                "traffic": traffic, "traffic_source": f"profiles/{tr_name} (rocprofv3 --pmc FETCH_SIZE x2, separate pass)" if traffic else None,
                "launches": n_launch, "avg_launch_us": round(avg_us, 3),
                "algorithmic_bytes_per_launch": round(q4k_bytes / n_launch), "all_kernels": kernels}
    # ---- the two denominators: the data sheet's 8 TB/s and what this box streams.  The ceiling is measured with the weight
    # stream's own instruction (LDS-DMA, non-temporal: mi355_stream_read) over 2 GiB, i.e. cold with respect to the 256 MiB Infinity Cache.
    ceiling = None
    single_op = []
    if world == 1:
        def timed(fn, reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            fn()
            torch.cuda.synchronize()
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / reps      # us
        big = torch.empty(2 << 30, dtype=torch.uint8, device="cuda")
        big.random_(0, 256)
        stream0 = torch.cuda.current_stream().cuda_stream
        best = 0.0
        for wgs in (256, 512):
            us = timed(lambda: K.check(K.lib.mi355_stream_read(big.data_ptr(), big.numel(), wgs, stream0), "mi355_stream_read"), 5)
            best = max(best, big.numel() / us * 1e-3)
        ceiling = {"value": round(best, 1), "unit": "GB/s", "what": "mi355_stream_read: 2 GiB read once by LDS-DMA (global_load_lds_dwordx4 nt, 8 KiB in "
                   "flight per wave, 16 waves per CU), best of 256 / 512 workgroups", "frac_of_8TBps": round(best / HBM_PEAK_GBPS, 4)}
        del big
        # single-op Q4_K mat-vec, cold (rotating over >= 1 GB of matrices), eager launches back to back: per launch, launch cost included
        def single(m, k, pool):
            outv = torch.empty((1, m), dtype=torch.float32, device="cuda")
            x1 = torch.randn((1, k), device="cuda", generator=gen)
            aq1, ads1 = ops.quantize_act(x1)
            args = []
            for w in pool:
                a = K.MmvqArgs()
                a.type, a.W, a.w_row_stride, a.w_ch_stride = B.Q4_K, w.data_ptr(), B.row_size(B.Q4_K, k), 0
                a.act_q, a.act_ds, a.act_ch_stride = aq1.data_ptr(), ads1.data_ptr(), 0
                a.dst, a.dst_col_stride, a.dst_ch_stride = outv.data_ptr(), m, 0
                a.m, a.k, a.ncols, a.nchannels, a.ch_ratio = m, k, 1, 1, 1
                args.append(a)
            def sweep():
                for a in args:
                    K.check(K.lib.mi355_mul_mat_vec_q(a, stream0), "mi355_mul_mat_vec_q")
            us = timed(sweep, 5) / len(args)
            gbps = B.matvec_bytes(B.Q4_K, m, k) / us * 1e-3
            # the same launches as pure reads: mi355_stream_read over each matrix's bytes, same pool, same order -- what a launch of
            # this size can stream at all, its launch cost and ramp included
            wb = B.row_size(B.Q4_K, k) * m
            def sweep_read():
                for w in pool:
                    K.check(K.lib.mi355_stream_read(w.data_ptr(), wb, 256, stream0), "mi355_stream_read")
            us_r = timed(sweep_read, 5) / len(pool)
            return {"shape": f"q4_K {m}x{k} n=1", "us": round(us, 2), "GB/s": round(gbps, 1), "frac_of_8TBps": round(gbps / HBM_PEAK_GBPS, 4),
                    "frac_of_measured_ceiling": round(gbps / best, 4), "cold_pool_MB": round(len(pool) * wb / 1e6),
                    "read_only_launch_us": round(us_r, 2), "read_only_GB/s": round(wb / us_r * 1e-3, 1), "frac_of_read_only_launch": round(us_r / us, 4)}
        gates = [weights[i] for i, mt in enumerate(mats) if mt.name.endswith("ffn_gate") and mt.type == B.Q4_K]
        single_op.append(single(dims.n_ff, dims.n_embd, gates))
        pool2 = [make_weights(torch, B, B.Q4_K, 28672, 8192, gen) for _ in range(8)]
        single_op.append(single(28672, 8192, pool2))
        del pool2
        roofline["peak_measured"] = ceiling["value"]
        roofline["frac_of_measured"] = round(achieved / best, 4)
    n_launches = sum(1 for p in plan) + sum(1 for p in plan if p[0] is not None)

    if rank == 0:
        line = {
            "metric": "Q4_K mat-vec GB/s", "value": round(value, 1), "unit": "GB/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak" if world == 1 else "strong", "vs_baseline": None,
            "dtype": "int8", "data": "synthetic",
            "config": {"workload": f"{dims.name} Q4_K_M decode mat-vec sweep: 225 weight matrices "
                                   f"({MS.decode_weight_bytes(dims) / 1e9:.3f} GB, Q4_K + Q6_K per llama.cpp's recipe) streamed once per "
                                   "step" + (", in the launches the plugin uses for a decode token (rms-norm + q8_1 in the prologue of the fused "
                                             "QKV+rope and gate/up+GLU launches, residual add in the wo / down epilogue)" if fuse_norm else
                                             " + q8_1 activation quantization") + ", random-init GGUF blocks",
                       "parallelism": "single GPU" if world == 1 else f"row split x{world} + one RCCL all-gather per layer",
                       "launch": "hipGraph replay" if use_graph else ("one hipGraph per layer, collectives between them" if segments and segments[0]["graph"] is not None else "eager"),
                       "launches_per_step": n_launches, "grouped": not args.no_group,
                       "pct_hbm_peak": round(value / (HBM_PEAK_GBPS * world) * 100, 2),
                       "decode_tokens_per_s_matvec_only": round(1e3 / ms_per_step, 1)},
            "roofline": roofline,
        }
        if ceiling is not None:
            line["stream_ceiling"] = ceiling
            line["single_op"] = single_op
            line["config"]["pct_of_measured_ceiling"] = round(value / ceiling["value"] * 100, 2)
        if base is not None:
            line["cpu_baseline"] = base
        if world == 1 and not args.no_e2e:
            line["e2e"] = end_to_end(240)      # a child process with its own copy of the model (288 GB of HBM: no contention)
            if not args.no_other_configs:
                line["e2e_other_configs"] = other_configs(300)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
