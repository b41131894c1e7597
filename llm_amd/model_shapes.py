"""Weight-matrix inventory of the models BASELINE.json names, with the per-tensor types that
llama.cpp's Q4_K_M recipe assigns (src/llama-quant.cpp:178-404 of the reference).

A decode step streams every matrix once (mat-vec), so this list *is* the tg workload:
`decode_matrices()` returns them in graph order (src/llama-model.cpp:4898-5052, llm_build_llama).
"""
from dataclasses import dataclass

from . import blocks as B


@dataclass(frozen=True)
class ModelDims:
    name: str
    n_embd: int
    n_layer: int
    n_head: int
    n_head_kv: int
    head_dim: int
    n_ff: int
    n_vocab: int
    is_70b: bool = False


LLAMA3_8B = ModelDims("llama3-8b", 4096, 32, 32, 8, 128, 14336, 128256)
LLAMA3_70B = ModelDims("llama3-70b", 8192, 80, 64, 8, 128, 28672, 128256, is_70b=True)
MODELS = {m.name: m for m in (LLAMA3_8B, LLAMA3_70B)}


def use_more_bits(i_layer, n_layers):
    """src/llama-quant.cpp:185-187."""
    return i_layer < n_layers // 8 or i_layer >= 7 * n_layers // 8 or (i_layer - n_layers // 8) % 3 == 2


@dataclass(frozen=True)
class Matrix:
    name: str
    type: int
    m: int      # output rows
    k: int      # input features
    src: str    # which activation feeds it: "attn_norm" | "attn_out" | "ffn_norm" | "ffn_act" | "final_norm"

    @property
    def bytes(self):
        return self.m * B.row_size(self.type, self.k)


def decode_matrices(dims, ftype="Q4_K_M"):
    """All MUL_MAT weight matrices of one decode step, in graph order."""
    assert ftype in ("Q4_K_M", "Q8_0", "Q4_0")
    base = {"Q4_K_M": B.Q4_K, "Q8_0": B.Q8_0, "Q4_0": B.Q4_0}[ftype]
    kv_dim = dims.n_head_kv * dims.head_dim
    out = []
    for il in range(dims.n_layer):
        t_v = t_down = base
        if ftype == "Q4_K_M":
            more = use_more_bits(il, dims.n_layer)
            t_v = B.Q6_K if more else (B.Q5_K if dims.is_70b else B.Q4_K)   # llama-quant.cpp:291-299
            t_down = B.Q6_K if more else B.Q4_K                            # llama-quant.cpp:347-353
        p = f"blk.{il}."
        out += [
            Matrix(p + "attn_q", base, dims.n_embd, dims.n_embd, "attn_norm"),
            Matrix(p + "attn_k", base, kv_dim, dims.n_embd, "attn_norm"),
            Matrix(p + "attn_v", t_v, kv_dim, dims.n_embd, "attn_norm"),
            Matrix(p + "attn_output", base, dims.n_embd, dims.n_embd, "attn_out"),
            Matrix(p + "ffn_gate", base, dims.n_ff, dims.n_embd, "ffn_norm"),
            Matrix(p + "ffn_up", base, dims.n_ff, dims.n_embd, "ffn_norm"),
            Matrix(p + "ffn_down", t_down, dims.n_embd, dims.n_ff, "ffn_act"),
        ]
    t_out = B.Q6_K if ftype == "Q4_K_M" else base                          # output.weight, llama-quant.cpp:229-240
    out.append(Matrix("output", t_out, dims.n_vocab, dims.n_embd, "final_norm"))
    return out


def decode_weight_bytes(dims, ftype="Q4_K_M"):
    return sum(m.bytes for m in decode_matrices(dims, ftype))


def decode_algorithmic_bytes(dims, ftype="Q4_K_M"):
    """Sum over launches of M*(K/bs)*B + 4K + 4M."""
    return sum(B.matvec_bytes(m.type, m.m, m.k) for m in decode_matrices(dims, ftype))
