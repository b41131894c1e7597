/*
 * oracle.h -- CPU restatement of the reference's (llama.cpp / ggml) arithmetic for the
 * quantized MUL_MAT / MUL_MAT_ID / FLASH_ATTN_EXT hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  The product path (llm_amd/, the HIP kernels and the
 * ggml backend plugin) never links, loads or calls anything in oracle/.
 *
 * Every function cites the reference file:line (relative to /root/reference) it restates.
 * Pinning: tests/test_oracle_golden.py checks every function here against fixtures in
 * tests/golden/ that were produced by the reference itself (oracle/_ref, built from the reference's
 * own sources by oracle/Makefile; generator oracle/gen_golden.cpp).
 */
#ifndef MI355_ORACLE_H
#define MI355_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ggml type ids that this path uses (ggml/include/ggml.h:363-400) */
enum {
    ORC_TYPE_F32  = 0,
    ORC_TYPE_F16  = 1,
    ORC_TYPE_Q4_0 = 2,
    ORC_TYPE_Q4_1 = 3,
    ORC_TYPE_Q5_0 = 6,
    ORC_TYPE_Q5_1 = 7,
    ORC_TYPE_Q8_0 = 8,
    ORC_TYPE_Q8_1 = 9,
    ORC_TYPE_Q2_K = 10,
    ORC_TYPE_Q3_K = 11,
    ORC_TYPE_Q4_K = 12,
    ORC_TYPE_Q5_K = 13,
    ORC_TYPE_Q6_K = 14,
    ORC_TYPE_Q8_K = 15,
    ORC_TYPE_IQ4_NL = 20,
    ORC_TYPE_IQ4_XS = 23,
    ORC_TYPE_BF16 = 30,
};

/* block geometry: elements per block / bytes per block (ggml/src/ggml-common.h:167-334) */
int64_t orc_blck_size(int type);
int64_t orc_type_size(int type);
int64_t orc_row_size(int type, int64_t k);

/* IEEE binary16 <-> binary32, round-to-nearest-even (ggml/src/ggml-impl.h:342-389) */
uint16_t orc_fp32_to_fp16(float f);
float    orc_fp16_to_fp32(uint16_t h);

/* activation quantizers, bit-exact restatements of the *_ref functions */
void orc_quantize_row_q8_0(const float *x, void *y, int64_t k); /* ggml-quants.c:188-211 */
void orc_quantize_row_q4_0(const float *x, void *y, int64_t k); /* ggml-quants.c:25-60 (the KV-cache store format of -ctk q4_0) */
void orc_fp32_to_bf16_row(const float *x, uint16_t *y, int64_t k); /* ggml-impl.h:453-466, ggml.c ggml_fp32_to_bf16_row_ref */
void orc_quantize_row_q8_1(const float *x, void *y, int64_t k); /* ggml-quants.c:214-247 */
void orc_quantize_row_q8_K(const float *x, void *y, int64_t k); /* ggml-quants.c:2471-2508 */

/* block dequantizers, bit-exact (ggml-quants.c:249-267, 343-357, 1274-1296, 1476-1501, 1684-1713) */
int  orc_dequantize_row(int type, const void *x, float *y, int64_t k);

/* CPU-backend dot products (ggml-cpu/quants.c:111-145, 270-298, 515-588, 590-668, 670-723).
 * `act` is q8_0 blocks for Q4_0/Q8_0 weights and q8_K blocks for the K-quants
 * (type_traits_cpu, ggml-cpu/ggml-cpu.c:196-370). */
float orc_vec_dot_cpu(int type, int64_t k, const void *w, const void *act);

/* MUL_MAT as ggml-cpu computes it (ggml-cpu/ggml-cpu.c:1196-1388): activations are quantized to
 * the weight type's vec_dot_type with the *_ref quantizer, then vec_dot per (row, col).
 * w: [m rows][k] blocks, row stride w_row_bytes; x: [n][k] f32 contiguous; dst: [n][m] f32. */
int orc_mul_mat_cpu(int type, const void *w, int64_t w_row_bytes, const float *x, float *dst,
                    int64_t m, int64_t k, int64_t n);

/* MUL_MAT in the q8_1 contract of the GPU path (what ggml-cuda's mmvq/mmq compute,
 * ggml-cuda/vecdotq.cuh): dst = sum_k dequant(W)[m,k] * dequant_q8_1(quantize_q8_1_ref(x))[n,k],
 * evaluated exactly per 32-block in integers and accumulated in double. */
int orc_mul_mat_q8_1(int type, const void *w, int64_t w_row_bytes, const float *x, float *dst,
                     int64_t m, int64_t k, int64_t n);

/* MUL_MAT_ID (ggml.c:3000-3036; ggml-cpu/ggml-cpu.c:1470-1648):
 * as: [n_expert][m][k] blocks; b: [n_tokens][b_ne1][k] f32; ids: [n_tokens][n_used] i32;
 * dst: [n_tokens][n_used][m] f32; dst[t][e] = as[ids[t][e]] . b[t][e % b_ne1].
 * semantics: 0 = ggml-cpu (q8_0/q8_K activations), 1 = q8_1 contract. */
int orc_mul_mat_id(int type, const void *as, int64_t w_row_bytes, int64_t expert_bytes,
                   const float *b, const int32_t *ids, float *dst,
                   int64_t m, int64_t k, int64_t n_expert, int64_t n_used, int64_t n_tokens,
                   int64_t b_ne1, int semantics);

/* FLASH_ATTN_EXT as ggml-cpu computes it (ggml-cpu/ops.cpp:7975-8192), one (batch=1) call.
 * q: [n_head][n_q][dk] f32 contiguous; k: [n_head_kv][n_kv][dk] of type_k (rows contiguous);
 * v: [n_head_kv][n_kv][dv] of type_v; mask: [n_q_pad][n_kv] fp16 or NULL;
 * dst: [n_q][n_head][dv] f32 (the op's permuted output layout). */
int orc_flash_attn_ext(const float *q, const void *k, const void *v, const uint16_t *mask,
                       float *dst, int type_k, int type_v,
                       int64_t dk, int64_t dv, int64_t n_q, int64_t n_kv,
                       int64_t n_head, int64_t n_head_kv,
                       float scale, float max_bias, float logit_softcap);

/* Same op evaluated in double precision on dequantized K/V and unquantized Q ("true" value);
 * used for tolerance checks that do not depend on ggml-cpu's f16 accumulators. */
int orc_flash_attn_ext_f64(const float *q, const void *k, const void *v, const uint16_t *mask,
                           float *dst, int type_k, int type_v,
                           int64_t dk, int64_t dv, int64_t n_q, int64_t n_kv,
                           int64_t n_head, int64_t n_head_kv,
                           float scale, float max_bias, float logit_softcap);

/* NMSE as test-backend-ops defines it (tests/test-backend-ops.cpp:177-190) */
double orc_nmse(const float *a, const float *b, int64_t n);

#ifdef __cplusplus
}
#endif
#endif
