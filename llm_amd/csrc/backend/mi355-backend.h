// mi355-backend.h -- internal header of the ggml-mi355 plugin: what its translation units share.
//   ggml-mi355.cpp     registry, devices, buffer types (device / pinned host / row-split), the backend object, graph capture and replay
//   mi355-matmul.cpp   MUL_MAT / MUL_MAT_ID / FLASH_ATTN_EXT dispatch: kernel choice per (type, N, K), activation formats, row-split execution
//   mi355-fusions.cpp  the graph-level pattern matchers and fused launches, and the loop that issues a graph (mi355_issue_graph)
//   ops.cpp            the remaining ggml ops of a Llama / Mixtral graph
// Nothing here is exported: the plugin is built with -fvisibility=hidden and exports ggml_backend_init / ggml_backend_score only.
#pragma once

#include "ggml.h"
#include "ggml-backend.h"
#include "ggml-backend-impl.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>      // types only: the library is opened with dlopen when GGML_MI355_SPLIT_GATHER=rccl asks for it
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "mi355_kernels.h"
#include "mi355_ops.h"

#define MI355_NAME "MI355"
#define MI355_DEV_PREFIX "MI355_"
#define MI355_MAX_DEVICES 16

// ggml's logger lives in ggml-base (ggml-impl.h is internal; the symbol is exported)
extern "C" void ggml_log_internal(enum ggml_log_level level, const char * format, ...);
#define MI_LOG_INFO(...)  ggml_log_internal(GGML_LOG_LEVEL_INFO, __VA_ARGS__)
#define MI_LOG_WARN(...)  ggml_log_internal(GGML_LOG_LEVEL_WARN, __VA_ARGS__)
#define MI_LOG_ERROR(...) ggml_log_internal(GGML_LOG_LEVEL_ERROR, __VA_ARGS__)
#define MI_LOG_DEBUG(...) ggml_log_internal(GGML_LOG_LEVEL_DEBUG, __VA_ARGS__)

// unrecoverable device errors abort with a message, as CUDA_CHECK does (ggml-cuda.cu:70-81)
#define HIP_CHECK(expr)                                                                              \
    do {                                                                                             \
        hipError_t err_ = (expr);                                                                    \
        if (err_ != hipSuccess) {                                                                    \
            ggml_abort(__FILE__, __LINE__, "HIP error: %s\n  in %s", hipGetErrorString(err_), #expr); \
        }                                                                                            \
    } while (0)

#define MI_CHECK(expr)                                                                               \
    do {                                                                                             \
        int rc_ = (expr);                                                                            \
        if (rc_ != 0) {                                                                              \
            ggml_abort(__FILE__, __LINE__, "mi355 kernel error: %s\n  in %s", mi355_error_string(rc_), #expr); \
        }                                                                                            \
    } while (0)


#define MI355_ROW_ROUNDING 128   // rows per MMQ tile (mmq.hip BM); the reference uses mmq_y = 128 on AMD (mmq.cuh:126-129)
#define MI355_KV_SLOTS 512          // fused Q/K/V launches per graph whose K / V destinations go through the table
#define MI355_KV_RING  8
#define MI355_ENV_SET(name) ([]() -> bool { static const bool v_ = getenv(name) != nullptr; return v_; }())

struct mi355_device_ctx {
    int         device = 0;
    std::string name;
    std::string description;
};

// ---- devices (ggml-mi355.cpp)
int  mi355_physical_count();
int  mi355_device_count();
int  mi355_phys(int device);
void mi355_set_device(int device);
void mi355_copy_d2d_async(void * dst, int dst_dev, const void * src, int src_dev, size_t n, hipStream_t stream);

struct mi355_rccl {
    bool tried = false, ok = false;
    void * lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char * (*GetErrorString)(ncclResult_t) = nullptr;
    ncclComm_t comm[MI355_MAX_DEVICES] = {};
    // (the staging buffers of the gather belong to the backend that drives it: mi355_backend_ctx::split_bufs)
};
mi355_rccl & mi355_rccl_state();
bool mi355_split_gather_rccl();
#define RCCL_CHECK(call) do { const ncclResult_t rc_ = (call); if (rc_ != ncclSuccess) { \
    MI_LOG_ERROR("%s: %s failed: %s\n", MI355_NAME, #call, mi355_rccl_state().GetErrorString(rc_)); GGML_ABORT("RCCL error"); } } while (0)

struct mi355_split_extra {
    void *  data[MI355_MAX_DEVICES]     = {};
    int64_t row_low[MI355_MAX_DEVICES]  = {};
    int64_t row_high[MI355_MAX_DEVICES] = {};
};
bool tensor_is_split(const ggml_tensor * t);

struct mi355_copy_stats { double us[2] = {0, 0}; long n[2] = {0, 0}; size_t bytes[2] = {0, 0}; };
extern mi355_copy_stats g_copy_stats;

// ------------------------------------------------------------------------------------------------
// backend (stream)
// ------------------------------------------------------------------------------------------------
struct mi355_backend_ctx {
    int         device;
    std::string name;
    hipStream_t stream = nullptr;
    // device scratch, grown on demand and reused across ops:
    //   [0] q8_1 activations of the MUL_MAT / MUL_MAT_ID in flight   [1] split-KV partials of FLASH_ATTN_EXT
    //   [2] spare activation buffer: a kernel that reads activations AND emits the next ones (mi355_glu_matvec) writes
    //       here, then [0] and [2] trade places      [3] partial tiles of a split-K prefill GEMM (mi355_mul_mat_dq_slice_ws)
    //   [4] a decode token's K and V rows in f32, between the fused Q/K/V launch and their quantization into a q8_0 cache
    //   [5] the sync blocks of a token's persistent decode-layer launches (mi355_decode_layer), cleared by one memset per graph
    // [0] activations of the current MUL_MAT, [1] attention workspace, [2] the other activation buffer (swap_act_scratch), [3] split-K workspace
    // of the prefill GEMM / its Q-K-V planes, [4] f32 K / V rows on their way into a q8_0 cache, [5] decode-layer sync blocks, [6] a row nobody
    // reads (mi355_try_fuse_glu_matvec), [7] the Q / K / V planes of a small batch (whose members may run a GEMM that uses [3]),
    // [8] the routing tables of a MoE layer's ids (built by the up / gate launch, reused by the down launch)
    void * scratch[9] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t scratch_size[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int dl_used = 0;                // decode-layer launches issued for the graph in flight (= sync blocks handed out)
    void swap_act_scratch() { std::swap(scratch[0], scratch[2]); std::swap(scratch_size[0], scratch_size[2]); }

    // the activation tensor whose q8_1 form currently sits in scratch[0]: consecutive MUL_MATs that read
    // the same src1 (wq/wk/wv, gate/up) quantize it once.  Dropped as soon as any node writes into it.
    struct {
        bool valid = false;
        int fmt = 0;                // 0: q8_1 SoA (mi355_quantize_act), 1: bf16 rows (mi355_act_bf16), 2: q8_K planes (mi355_quantize_act_q8k)
        const void * data = nullptr;
        size_t span = 0;
        int64_t ne[4] = {0, 0, 0, 0};
        size_t nb[4] = {0, 0, 0, 0};
    } act_cache;
    // the ids tensor whose routing tables (moe_bucket) currently sit in scratch[8]: the MUL_MAT_IDs of one MoE layer share their ids.
    // Dropped as soon as any node writes into the ids, with every new graph, and when scratch[8] moves.
    struct {
        bool valid = false;
        const void * data = nullptr;
        size_t span = 0;
        int64_t n_tokens = 0, n_used = 0, n_expert = 0, stride = 0;
    } moe_tables;
    bool moe_tables_cached(const ggml_tensor * ids, int64_t n_expert) const {
        return moe_tables.valid && moe_tables.data == ids->data && moe_tables.n_tokens == ids->ne[1] && moe_tables.n_used == ids->ne[0] &&
               moe_tables.n_expert == n_expert && moe_tables.stride == (int64_t)ids->nb[1];
    }
    void moe_tables_remember(const ggml_tensor * ids, int64_t n_expert) {
        moe_tables.valid = true; moe_tables.data = ids->data; moe_tables.span = ggml_nbytes(ids);
        moe_tables.n_tokens = ids->ne[1]; moe_tables.n_used = ids->ne[0]; moe_tables.n_expert = n_expert; moe_tables.stride = (int64_t)ids->nb[1];
    }
    // the activation planes of `rows` rows of K values starting at `data`, whatever shape the reader's view of them has
    bool act_cached_rows(const void * data, int64_t K, int64_t rows, int fmt) const {
        if (!act_cache.valid || act_cache.data != data || act_cache.fmt != fmt || act_cache.ne[0] != K || act_cache.nb[0] != sizeof(float)) return false;
        if (act_cache.ne[1] * act_cache.ne[2] * act_cache.ne[3] != rows || act_cache.span != (size_t)(rows * K) * sizeof(float)) return false;
        return true;        // span == rows * K * 4: the cached tensor was contiguous
    }

    // row split: this stream is the "main" one of a split MUL_MAT; every other device that owns rows gets a
    // helper stream, a copy of the quantized activations and an event the main stream waits on
    struct peer_t {
        hipStream_t stream = nullptr;
        hipEvent_t  done = nullptr;
        void *      act = nullptr;
        size_t      act_size = 0;
        uint64_t    act_version = 0;
        void *      gemm_ws = nullptr;      // split-K workspace of the prefill GEMM on this device's helper stream
        size_t      gemm_ws_size = 0;
    } peers[MI355_MAX_DEVICES];
    // RCCL form of the gather: per device d (this backend's own included) its slice [cols][rows_max] and everybody's slices
    // [ndev][cols][rows_max].  They belong to THIS backend -- under -sm row the main device alternates by layer, so two
    // backends' helper streams run on one GPU, and buffers shared between them would have no ordering.
    struct split_buf_t { float * part = nullptr; size_t part_size = 0; float * gath = nullptr; size_t gath_size = 0; } split_bufs[MI355_MAX_DEVICES];
    // device memory on `dev` that grows on demand, outside any capture (split graphs are never captured); `st`: the stream whose
    // launches may still use the old block
    static void grow_on(int dev, void ** buf, size_t * have, size_t need, hipStream_t st) {
        if (*have >= need) return;
        mi355_set_device(dev);
        HIP_CHECK(hipStreamSynchronize(st));
        if (*buf) HIP_CHECK(hipFree(*buf));
        *have = (need + (8u << 20)) & ~((size_t)(1u << 20) - 1);
        HIP_CHECK(hipMalloc(buf, *have));
    }
    hipEvent_t ev_ready = nullptr;
    uint64_t   act_version = 0;     // bumped whenever scratch[0] receives new activations

    peer_t & peer(int d) {
        peer_t & p = peers[d];
        if (!p.stream) {
            mi355_set_device(d);
            HIP_CHECK(hipStreamCreateWithFlags(&p.stream, hipStreamNonBlocking));
            HIP_CHECK(hipEventCreateWithFlags(&p.done, hipEventDisableTiming));
            if (mi355_phys(d) != mi355_phys(device)) {   // the helper writes its rows straight into our dst over xGMI
                hipError_t err = hipDeviceEnablePeerAccess(mi355_phys(device), 0);
                if (err != hipSuccess && err != hipErrorPeerAccessAlreadyEnabled) HIP_CHECK(err);
                (void)hipGetLastError();
            }
            mi355_set_device(device);
        }
        return p;
    }

    // ---- decode graphs (reference analogue: the CUDA-graph path of ggml-cuda.cu:2579-2905) -------------------------
    // A decode token's launches (~165) are captured once into a hipGraph and replayed while the ggml graph keeps its
    // shape: node ops, shapes, strides, parameters and addresses.  The one thing that moves with every token is the
    // KV-cache row the new K / V are stored to; the fused Q/K/V launch reads those two addresses from a device table
    // (kv_dev) that is refreshed before each replay, the way ggml-cuda refreshes its cpy destination pointers.
    struct node_sig {
        int32_t op, type, flags;
        const void * data;
        int64_t ne[4]; size_t nb[4];
        const void * src[GGML_MAX_SRC];
        int32_t params[GGML_MAX_OP_PARAMS / sizeof(int32_t)];
        bool kv_dst;                 // CPY whose destination goes through the table: its address may change
    };
    struct graph_cache_t {
        std::vector<node_sig> sig;   // the last graph seen
        hipGraphExec_t exec = nullptr;
        std::vector<int> kv_nodes;   // CPY node indices, two per table slot (K, V), in launch order
        std::vector<size_t> kv_off;  // byte offset of the entry's destination inside that node's row (a row-split slice), else 0
        void kv_push(int node, size_t off) { kv_nodes.push_back(node); kv_off.push_back(off); }
        void kv_clear() { kv_nodes.clear(); kv_off.clear(); }
        void ** kv_dev = nullptr;    // device table, MI355_KV_SLOTS * 2 pointers
        void ** kv_host = nullptr;   // pinned staging ring, MI355_KV_RING copies of the table
        int  ring = 0;
        int  recaptures = 0;         // consecutive graphs that had to be captured again
        bool disabled = false;
        bool capturing = false, capture_failed = false;
        int  n_slots = 0;            // slots handed out during the capture in progress
        long n_replay = 0, n_capture = 0, n_eager = 0;       // GGML_MI355_GRAPH_STATS=1 prints them when the backend is freed
        long n_dropped = 0;          // captured graphs thrown away because a scratch block they point into moved
        hipEvent_t ring_ev[MI355_KV_RING] = {};    // recorded behind the table copy that reads ring slot i: the slot is reused only after it
        void drop() {
            if (exec) { HIP_CHECK(hipGraphExecDestroy(exec)); exec = nullptr; n_dropped++; }
            sig.clear(); kv_clear(); recaptures = 0;
        }
    } gc;

    void * get_scratch(int which, size_t need) {
        if (need <= scratch_size[which]) return scratch[which];
        if (gc.capturing) {          // no allocation or synchronisation under capture: give up on this capture,
            gc.capture_failed = true;    // the eager run that follows grows the buffer (nothing captured is ever executed)
            return scratch[which];
        }
        if (scratch[which]) {
            HIP_CHECK(hipStreamSynchronize(stream)); // earlier kernels may still read the old block
            HIP_CHECK(hipFree(scratch[which]));
            // a captured decode graph has this block's address baked into its launches: drop it, so the next decode graph
            // runs eagerly (first sighting) and is captured again against the new block
            gc.drop();
        }
        if (which == 0) act_cache.valid = false;
        if (which == 8) moe_tables.valid = false;
        scratch_size[which] = (need + (8u << 20)) & ~((size_t)(1u << 20) - 1);
        HIP_CHECK(hipMalloc(&scratch[which], scratch_size[which]));
        // the flash-attention workspace starts with arrival counters that must be zero before the first use
        if (which == 1) HIP_CHECK(hipMemsetAsync(scratch[which], 0, scratch_size[which], stream));
        return scratch[which];
    }

    // split-K workspace of the prefill GEMM for helper device d's slice (the kernel library allocates nothing: every stream that
    // launches a split-K GEMM brings its own)
    void peer_gemm_ws(int d, const ggml_tensor * src0, const mi355_split_extra * extra, int64_t K, int64_t N, int path, void ** ws, int64_t * bytes);

    bool act_cached(const ggml_tensor * t, int fmt = 0) const {
        if (!act_cache.valid || act_cache.data != t->data || act_cache.fmt != fmt) return false;
        for (int i = 0; i < 4; i++) if (act_cache.ne[i] != t->ne[i] || act_cache.nb[i] != t->nb[i]) return false;
        return true;
    }
    void act_remember(const ggml_tensor * t, int fmt = 0) {
        act_version++;
        act_cache.fmt = fmt;
        act_cache.valid = true; act_cache.data = t->data; act_cache.span = ggml_nbytes(t);
        for (int i = 0; i < 4; i++) { act_cache.ne[i] = t->ne[i]; act_cache.nb[i] = t->nb[i]; }
    }
    void act_invalidate_if_written(const ggml_tensor * node) {
        if (moe_tables.valid) {
            const char * w0 = (const char *)node->data, * w1 = w0 + ggml_nbytes(node);
            const char * c0 = (const char *)moe_tables.data, * c1 = c0 + moe_tables.span;
            if (w0 < c1 && c0 < w1) moe_tables.valid = false;
        }
        if (!act_cache.valid) return;
        const char * w0 = (const char *)node->data, * w1 = w0 + ggml_nbytes(node);
        const char * c0 = (const char *)act_cache.data, * c1 = c0 + act_cache.span;
        if (w0 < c1 && c0 < w1) act_cache.valid = false;
    }
};

// ---- MUL_MAT paths (mi355-matmul.cpp) ----
#define MI355_BF16_ROWS_ABOVE 32
enum mi355_mm_path { MM_VEC = 0, MM_DQ = 1, MM_SKINNY_K = 2, MM_SKINNY = 3, MM_Q_INT8 = 4, MM_FLOAT = 5 };
// activation format each path consumes (act_cache.fmt): 0 q8_1 SoA, 1 bf16 rows, 2 q8_K planes
static inline int mm_path_fmt(int path) { return path == MM_DQ ? 1 : path == MM_SKINNY_K ? 2 : 0; }
struct mi355_act_layout { size_t b0, b1, b2; size_t total() const { return b0 + b1 + b2; } };

// ---- mi355-matmul.cpp ----
bool is_weight_type(enum ggml_type t);              // the five types with skinny / fused kernels: q4_0, q8_0, q4_K, q5_K, q6_K
bool is_block_weight_type(enum ggml_type t);        // + q4_1, q5_0, q5_1, q2_K, q3_K, iq4_nl, iq4_xs (mat-vec + bf16 GEMM)
bool is_float_weight_type(enum ggml_type t);
size_t align_up(size_t x, size_t a);
int  mi355_mul_mat_path(const ggml_tensor * src0, const ggml_tensor * src1, const ggml_tensor * dst = nullptr);
void mi355_note_graph_activations(ggml_cgraph * cgraph);     // what mi355_mul_mat_path may know about the graph in flight
mi355_act_layout mi355_act_planes(int path, int64_t K, int64_t rows);
void mi355_mul_mat(mi355_backend_ctx * ctx, ggml_tensor * dst);
void mi355_mul_mat_id(mi355_backend_ctx * ctx, ggml_tensor * dst);
bool mi355_mul_mat_id_pair(mi355_backend_ctx * ctx, ggml_tensor * d0, ggml_tensor * d1);      // up_exps + gate_exps of a small batch in one launch
bool mi355_flash_attn_supported(const ggml_tensor * op);
bool mi355_compute_forward(mi355_backend_ctx * ctx, ggml_tensor * node, const ggml_tensor * next = nullptr);
bool mi355_try_fuse_rms_norm_mul(mi355_backend_ctx * ctx, ggml_cgraph * cgraph, int i, ggml_tensor * add = nullptr, const ggml_tensor * gemm = nullptr);

// ---- mi355-fusions.cpp ----
enum ggml_status mi355_issue_graph(mi355_backend_ctx * ctx, ggml_cgraph * cgraph);
uint16_t * prefill_act_bf16(mi355_backend_ctx * ctx, const ggml_tensor * src1);   // bf16 rows of a prefill MUL_MAT's activations in scratch[0]
