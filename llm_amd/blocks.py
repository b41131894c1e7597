"""GGUF block formats on the host (numpy): sizes and synthetic-weight generation.

Mirrors ggml/src/ggml-common.h:167-334 of the reference.  Used to build random but *valid*
quantized tensors for tests and benchmarks without the reference's quantizer: random quant bytes,
fp16 scales drawn from a sane range (no inf/nan), every packed 6-bit scale pattern allowed.
"""
import numpy as np

F32, F16, Q4_0, Q8_0, Q8_1, Q4_K, Q5_K, Q6_K = 0, 1, 2, 8, 9, 12, 13, 14
Q4_1, Q5_0, Q5_1, I32, I64, BF16 = 3, 6, 7, 26, 27, 30
Q2_K, Q3_K, IQ4_NL, IQ4_XS = 10, 11, 20, 23
TYPE_NAMES = {F32: "f32", F16: "f16", Q4_0: "q4_0", Q8_0: "q8_0", Q8_1: "q8_1", Q4_K: "q4_K", Q5_K: "q5_K", Q6_K: "q6_K",
              Q4_1: "q4_1", Q5_0: "q5_0", Q5_1: "q5_1", BF16: "bf16", Q2_K: "q2_K", Q3_K: "q3_K", IQ4_NL: "iq4_nl", IQ4_XS: "iq4_xs"}
TYPE_BY_NAME = {v: k for k, v in TYPE_NAMES.items()}
BLCK_SIZE = {F32: 1, F16: 1, BF16: 1, Q4_0: 32, Q4_1: 32, Q5_0: 32, Q5_1: 32, Q8_0: 32, Q8_1: 32, Q4_K: 256, Q5_K: 256, Q6_K: 256,
             Q2_K: 256, Q3_K: 256, IQ4_NL: 32, IQ4_XS: 256}
TYPE_SIZE = {F32: 4, F16: 2, BF16: 2, Q4_0: 18, Q4_1: 20, Q5_0: 22, Q5_1: 24, Q8_0: 34, Q8_1: 36, Q4_K: 144, Q5_K: 176, Q6_K: 210,
             Q2_K: 84, Q3_K: 110, IQ4_NL: 18, IQ4_XS: 136}
WEIGHT_TYPES = (Q4_0, Q8_0, Q4_K, Q5_K, Q6_K)
WEIGHT_TYPES2 = (Q4_1, Q5_0, Q5_1)          # mat-vec + bf16 GEMM only (no skinny / fused kernels)
WEIGHT_TYPES3 = (Q2_K, Q3_K, IQ4_NL, IQ4_XS)  # likewise (ggml-common.h:256-279, 405-418)
FLOAT_WEIGHT_TYPES = (F16, BF16, F32)
# byte offsets of the fp16 scale fields inside a block
_HALF_FIELDS = {Q4_0: (0,), Q8_0: (0,), Q4_K: (0, 2), Q5_K: (0, 2), Q6_K: (208,), Q4_1: (0, 2), Q5_0: (0,), Q5_1: (0, 2),
                Q2_K: (80, 82), Q3_K: (108,), IQ4_NL: (0,), IQ4_XS: (0,)}


def row_size(t, k):
    assert k % BLCK_SIZE[t] == 0, (t, k)
    return k // BLCK_SIZE[t] * TYPE_SIZE[t]


def matvec_bytes(t, m, k, n=1):
    """ALGORITHMIC bytes of one mat-vec launch: M*(K/bs)*B + 4K*n + 4M*n."""
    return m * row_size(t, k) + 4 * k * n + 4 * m * n


def random_blocks(t, nrows, k, seed=0, scale=None):
    """uint8 array [nrows, row_size] of random valid blocks of type t."""
    rng = np.random.default_rng(seed)
    rs = row_size(t, k)
    nb = k // BLCK_SIZE[t]
    raw = rng.integers(0, 256, size=(nrows, nb, TYPE_SIZE[t]), dtype=np.uint8)
    if scale is None:
        scale = {Q4_0: 0.05, Q8_0: 0.005, Q4_K: 0.003, Q5_K: 0.002, Q6_K: 0.0005, Q4_1: 0.05, Q5_0: 0.03, Q5_1: 0.03,
                 Q2_K: 0.02, Q3_K: 0.003, IQ4_NL: 0.001, IQ4_XS: 0.0001}[t]
    for off in _HALF_FIELDS[t]:
        d = (rng.uniform(0.25, 1.0, size=(nrows, nb)) * scale).astype(np.float16)
        if off == 0 and t in (Q4_0, Q8_0, Q6_K, Q5_0, IQ4_NL, IQ4_XS) or off in (208, 108) or off == 2 and t in (Q4_1, Q5_1):
            d = d * rng.choice(np.array([-1.0, 1.0], dtype=np.float16), size=d.shape)
        raw[:, :, off:off + 2] = d.view(np.uint8).reshape(nrows, nb, 2)
    return raw.reshape(nrows, rs)
