/*
 * ggml_mi355.h -- the drop-in boundary of libggml-mi355.so, the MI355X (gfx950 / CDNA4) ggml backend
 * plugin.  These are the ONLY symbols the reference binds: ggml's loader looks them up with dlsym
 * after dlopen (ggml/src/ggml-backend-reg.cpp:234-250; declared by GGML_BACKEND_DL_IMPL /
 * GGML_BACKEND_DL_SCORE_IMPL in ggml/src/ggml-backend-impl.h:214-251).  Everything else is reached
 * through the function tables of the object ggml_backend_init() returns:
 *
 *   ggml_backend_reg_i          ggml-backend-impl.h:181-207   name "MI355", devices "MI355_<i>"
 *   ggml_backend_device_i       ggml-backend-impl.h:137-179   type GPU, caps {async, host_buffer, events}
 *   ggml_backend_buffer_type_i  ggml-backend-impl.h:17-29     hipMalloc buffers (alignment 128), pinned host
 *   ggml_backend_buffer_i       ggml-backend-impl.h:41-58     synchronous set/get/memset/cpy/clear
 *   ggml_backend_i              ggml-backend-impl.h:87-117    one in-order HIP stream; async graph_compute
 *
 * It replaces the reference's ggml-cuda plugin entry (ggml_backend_cuda_reg,
 * ggml/src/ggml-cuda/ggml-cuda.cu:3660-3715, GGML_BACKEND_DL_IMPL at :3748) for the ops of the hot
 * path: MUL_MAT / MUL_MAT_ID on Q4_0, Q8_0, Q4_K, Q5_K, Q6_K weights, FLASH_ATTN_EXT with f16 /
 * q8_0 / q4_0 KV, and the small ops of a Llama / Mixtral graph so the scheduler has no CPU split.
 *
 * Loading (no change to the reference):  GGML_BACKEND_PATH=/path/libggml-mi355.so llama-bench ...
 * (ggml_backend_load_all_from_path, ggml-backend-reg.cpp:578-582), or ggml_backend_load(path).
 * The plugin leaves the ggml-base symbols undefined and resolves them against the host executable's
 * libggml-base.so, so the reference must be built with shared libraries.
 *
 * Proc addresses served by reg->iface.get_proc_address:
 *   "ggml_backend_get_features"        ggml_backend_feature * (*)(ggml_backend_reg_t)
 *   "ggml_backend_split_buffer_type"   ggml_backend_buffer_type_t (*)(int main_device, const float * tensor_split)
 *                                      (ggml/include/ggml-backend.h:188; row split, ggml-cuda.cu:1044-1084)
 *
 * Environment switches (diagnostics, read by the plugin):
 *   GGML_MI355_NO_FUSE=1     run every graph node with its own kernel, no activation-quantization cache
 *   GGML_MI355_NO_PINNED=1   do not offer the pinned host buffer type
 *   GGML_MI355_NO_SKINNY=1   MUL_MAT with 4..64 columns: mat-vec / bf16 GEMM kernels instead of the skinny int8-MFMA ones
 *   GGML_MI355_NO_KQ_ONLY=1  33..64 columns: only matrices of >= 32 M weights (GGML_MI355_SKINNY_WIDE_MIN) take the skinny kernels, whatever
 *                            the types of the other matrices that read the same activations
 *   GGML_MI355_NO_SKINNY_GROUP=1  steps of 5..32 tokens: one skinny launch per matrix instead of one for wq + wk (+ wv) and one for
 *                            ffn_up + ffn_gate (mi355_mul_mat_skinny_k_group); the results are the same bits either way
 *   GGML_MI355_NO_GLU_MATVEC=1  decode ffn: grouped gate+up mat-vec and a separate GLU kernel instead of mi355_glu_matvec
 *   GGML_MI355_NO_MOE_DOWN=1    MoE decode: MUL_MAT_ID(down_exps), the weight MUL and the expert / residual ADDs as separate nodes
 *   GGML_MI355_NO_NORM_ROUTER=1 MoE decode: the ffn norm as a launch of its own in front of the router's (mi355_moe_router_f32) instead of inside it
 *   GGML_MI355_NO_MOE_COMBINE=1 MoE batches: MUL(experts, weights) and the ADD of its two slot views as two launches instead of mi355_moe_combine2_f32
 *   GGML_MI355_NO_MOE_READY=1   MoE steps of 2..64 tokens: every MUL_MAT_ID quantizes its own input and buckets its own ids (no q8_K planes from
 *                            the norm / GLU launches, no routing tables kept from the up / gate launch for ffn_down)
 *   GGML_MI355_NO_MATVEC_ADD=1  decode: wo / ffn_down mat-vec and the residual ADD as separate nodes
 *   GGML_MI355_NO_NORM_GLU=1, GGML_MI355_NO_NORM_QKV=1  decode: ffn_norm / attn_norm as a launch of its own instead of the
 *                            prologue of mi355_glu_matvec / mi355_qkv_rope_matvec
 *   GGML_MI355_MMQ_INT8=1    prefill MUL_MAT on the int8-MFMA tile kernel (the reference's MMQ arithmetic) instead of
 *                            the dequantize-in-LDS bf16 GEMM
 *   GGML_MI355_DUMP_GRAPH=1  print the node list of every graph handed to graph_compute
 *   GGML_MI355_TIME_GRAPH=1  every 64 calls, print the host time spent inside graph_compute and between calls
 *   GGML_MI355_VIRTUAL_DEVICES=n   expose n ggml devices mapped round-robin onto the physical GPUs (tests the row-split
 *                            path on a one-GPU machine)
 *   GGML_MI355_NO_GRAPHS=1   decode tokens are launched eagerly, not replayed from a captured hipGraph (GGML_MI355_GRAPH_STATS=1
 *                            prints replay / capture / eager counts when the backend is freed, GGML_MI355_GRAPH_DEBUG=1 why a graph
 *                            was captured again)
 *   GGML_MI355_NO_SPLIT_FUSE=1   -sm row decode: split MUL_MATs one at a time (quantize, copy, launch per device) instead of the fused
 *                            decode launches on row slices; GGML_MI355_NO_SPLIT_GRAPH=1 keeps those but never captures a split token;
 *                            GGML_MI355_SPLIT_GRAPH=1 captures split tokens also when the devices are several physical GPUs (default:
 *                            only when they are one, i.e. virtual devices -- untested otherwise); GGML_MI355_SPLIT_STREAMS=1 issues a
 *                            same-GPU device's slice on its helper stream instead of the main stream
 *   GGML_MI355_SPLIT_GATHER=rccl   row split: gather the devices' row slices with an RCCL all-gather (ncclCommInitAll, one rank per
 *                            physical GPU; librccl.so is opened only then) instead of peer stores into the main device's dst; falls
 *                            back to peer stores, with a warning, when two devices share a GPU or RCCL cannot be set up
 */
#ifndef GGML_MI355_H
#define GGML_MI355_H

#ifdef __cplusplus
extern "C" {
#endif

struct ggml_backend_reg;   /* ggml/src/ggml-backend-impl.h:191-207 */

#define GGML_MI355_API __attribute__((visibility("default")))

/* The registry object (static lifetime, api_version == GGML_BACKEND_API_VERSION).  Required. */
GGML_MI355_API struct ggml_backend_reg * ggml_backend_init(void);

/* 0 when no gfx950 device is visible (the loader then skips the plugin), 100 otherwise.  Optional. */
GGML_MI355_API int ggml_backend_score(void);

#ifdef __cplusplus
}
#endif
#endif /* GGML_MI355_H */
