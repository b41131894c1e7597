// ggml-mi355.cpp -- ggml backend plugin for AMD Instinct MI355X (gfx950).
//
// Host side of the drop-in boundary: implements ggml's own plugin interface
//   ggml_backend_reg_i / ggml_backend_device_i / ggml_backend_buffer_type_i / ggml_backend_buffer_i /
//   ggml_backend_i            (reference: ggml/src/ggml-backend-impl.h:17-207)
// and exports ggml_backend_init / ggml_backend_score (ggml-backend-impl.h:214-251) so that the
// reference's unmodified llama-cli, llama-bench and test-backend-ops load it through
// GGML_BACKEND_PATH (ggml/src/ggml-backend-reg.cpp:578-582).  The compute is in
// libmi355_kernels.so (include/mi355_kernels.h); this file holds no kernels and no CPU fallback:
// an op the kernels do not implement is reported by supports_op() == false and the scheduler keeps
// it on another backend.
//
// Built against the reference's public headers (ggml.h, ggml-backend.h, ggml-backend-impl.h); the
// ggml-base symbols it calls are resolved from the host executable at dlopen time.
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

static ggml_backend_reg_t ggml_backend_mi355_reg(void);

// ------------------------------------------------------------------------------------------------
// device info
// ------------------------------------------------------------------------------------------------
struct mi355_device_ctx {
    int         device = 0;
    std::string name;
    std::string description;
};

// Physical gfx950 devices.  GGML_MI355_VIRTUAL_DEVICES=n (diagnostics / tests) presents n ggml devices
// that all map round-robin onto the physical ones, so the multi-device code paths (row split, peer
// copies, cross-device events) can be exercised on a single-GPU machine.
static int mi355_physical_count() {
    static int count = [] {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
        int ok = 0;
        for (int i = 0; i < n && i < MI355_MAX_DEVICES; i++) {
            hipDeviceProp_t p;
            if (hipGetDeviceProperties(&p, i) != hipSuccess) break;
            // the kernels are compiled for gfx950 only
            if (strncmp(p.gcnArchName, "gfx950", 6) != 0) {
                MI_LOG_WARN("%s: device %d is %s, not gfx950 -- ignored\n", MI355_NAME, i, p.gcnArchName);
                break;
            }
            ok++;
        }
        return ok;
    }();
    return count;
}

static int mi355_device_count() {
    static int count = [] {
        const int phys = mi355_physical_count();
        const char * v = getenv("GGML_MI355_VIRTUAL_DEVICES");
        if (phys > 0 && v && atoi(v) > 0) return atoi(v) < MI355_MAX_DEVICES ? atoi(v) : MI355_MAX_DEVICES;
        return phys;
    }();
    return count;
}

static int mi355_phys(int device) { return device % mi355_physical_count(); }

static void mi355_set_device(int device) {
    int cur = -1;
    HIP_CHECK(hipGetDevice(&cur));
    if (cur != mi355_phys(device)) HIP_CHECK(hipSetDevice(mi355_phys(device)));
}

// device-to-device copy between two ggml devices (which may share a physical GPU)
static void mi355_copy_d2d_async(void * dst, int dst_dev, const void * src, int src_dev, size_t n, hipStream_t stream) {
    if (mi355_phys(dst_dev) == mi355_phys(src_dev)) {
        HIP_CHECK(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, stream));
    } else {
        HIP_CHECK(hipMemcpyPeerAsync(dst, mi355_phys(dst_dev), src, mi355_phys(src_dev), n, stream));
    }
}

// ------------------------------------------------------------------------------------------------
// RCCL all-gather form of the row-split gather (SURVEY 8e phase A; the reference's "collective" is N-1 peer 2-D copies,
// ggml-cuda.cu:1659-1690).  GGML_MI355_SPLIT_GATHER=rccl selects it; the default stays the peer-store path (rows written
// straight into the main device's dst by the mat-vec epilogue: nothing to launch after the kernels, which is what a
// decode token's 4 KB..14 KB slices want), RCCL is the bandwidth option for prefill-sized results and the reference form
// of the exchange.  One process, one communicator rank per device (ncclCommInitAll), every device sends its [cols][rows_max]
// slice and receives all of them; the main device then places each slice's rows into dst.  librccl is opened with dlopen only
// when the switch is on, so the plugin has no load-time dependency on it.
// ------------------------------------------------------------------------------------------------
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
static mi355_rccl & mi355_rccl_state() { static mi355_rccl r; return r; }

static bool mi355_split_gather_rccl() {
    static const bool want = [] { const char * e = getenv("GGML_MI355_SPLIT_GATHER"); return e && strcmp(e, "rccl") == 0; }();
    if (!want) return false;
    mi355_rccl & r = mi355_rccl_state();
    if (r.tried) return r.ok;
    r.tried = true;
    const int ndev = mi355_device_count();
    for (int d = 0; d < ndev; d++)
        for (int e = 0; e < d; e++)
            if (mi355_phys(d) == mi355_phys(e)) {
                MI_LOG_WARN("%s: GGML_MI355_SPLIT_GATHER=rccl needs one physical GPU per device (virtual devices share one) -- using peer stores\n", MI355_NAME);
                return false;
            }
    r.lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!r.lib) r.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!r.lib) { MI_LOG_WARN("%s: cannot open librccl.so (%s) -- using peer stores\n", MI355_NAME, dlerror()); return false; }
    r.CommInitAll    = (decltype(r.CommInitAll))dlsym(r.lib, "ncclCommInitAll");
    r.AllGather      = (decltype(r.AllGather))dlsym(r.lib, "ncclAllGather");
    r.GroupStart     = (decltype(r.GroupStart))dlsym(r.lib, "ncclGroupStart");
    r.GroupEnd       = (decltype(r.GroupEnd))dlsym(r.lib, "ncclGroupEnd");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
    if (!r.CommInitAll || !r.AllGather || !r.GroupStart || !r.GroupEnd || !r.GetErrorString) {
        MI_LOG_WARN("%s: librccl.so lacks an expected symbol -- using peer stores\n", MI355_NAME);
        return false;
    }
    int devs[MI355_MAX_DEVICES];
    for (int d = 0; d < ndev; d++) devs[d] = mi355_phys(d);
    const ncclResult_t rc = r.CommInitAll(r.comm, ndev, devs);
    if (rc != ncclSuccess) {
        MI_LOG_WARN("%s: ncclCommInitAll over %d devices failed (%s) -- using peer stores\n", MI355_NAME, ndev, r.GetErrorString(rc));
        return false;
    }
    MI_LOG_INFO("%s: row-split gather by RCCL all-gather over %d devices\n", MI355_NAME, ndev);
    r.ok = true;
    return true;
}
#define RCCL_CHECK(call) do { const ncclResult_t rc_ = (call); if (rc_ != ncclSuccess) { \
    MI_LOG_ERROR("%s: %s failed: %s\n", MI355_NAME, #call, mi355_rccl_state().GetErrorString(rc_)); GGML_ABORT("RCCL error"); } } while (0)

// ------------------------------------------------------------------------------------------------
// device buffer  (reference behaviour: ggml-cuda.cu:543-751)
// ------------------------------------------------------------------------------------------------
struct mi355_buffer_ctx {
    int    device;
    void * base;
};

struct mi355_buft_ctx {
    int         device;
    std::string name;
};

static bool buft_is_mi355(ggml_backend_buffer_type_t buft);

static void mi355_buffer_free(ggml_backend_buffer_t buffer) {
    mi355_buffer_ctx * ctx = (mi355_buffer_ctx *)buffer->context;
    mi355_set_device(ctx->device);
    HIP_CHECK(hipFree(ctx->base));
    delete ctx;
}

static void * mi355_buffer_get_base(ggml_backend_buffer_t buffer) {
    return ((mi355_buffer_ctx *)buffer->context)->base;
}

static enum ggml_status mi355_buffer_init_tensor(ggml_backend_buffer_t, ggml_tensor *) {
    // rows are not padded (the kernels never read past a row), so there is nothing to initialise
    return GGML_STATUS_SUCCESS;
}

static void mi355_buffer_memset_tensor(ggml_backend_buffer_t buffer, ggml_tensor * tensor, uint8_t value, size_t offset, size_t size) {
    mi355_buffer_ctx * ctx = (mi355_buffer_ctx *)buffer->context;
    mi355_set_device(ctx->device);
    HIP_CHECK(hipMemset((char *)tensor->data + offset, value, size));
    HIP_CHECK(hipDeviceSynchronize());
}

// GGML_MI355_TIME_GRAPH=1 (diagnostics): host time spent in the synchronous copies between graphs, printed with the graph times.
// (Measured with it: a decode token's four input uploads -- embedding row, positions, mask, output ids, 82 KB -- cost 65 us of host
// time, 10-16 us each whatever the size (lab/h2d_lab.cpp).  Moving them to a pinned ring and an upload stream that the compute stream
// waits on cut the host time to 13 us and the token rate not at all (512 vs 510 t/s): the four copies then queue in front of the
// graph on the device instead.  Not kept.)
struct mi355_copy_stats { double us[2] = {0, 0}; long n[2] = {0, 0}; size_t bytes[2] = {0, 0}; };
static mi355_copy_stats g_copy_stats;
struct mi355_copy_timer {
    int which; size_t size; bool on; std::chrono::steady_clock::time_point t0;
    mi355_copy_timer(int w, size_t sz) : which(w), size(sz), on(getenv("GGML_MI355_TIME_GRAPH") != nullptr) { if (on) t0 = std::chrono::steady_clock::now(); }
    ~mi355_copy_timer() {
        if (!on) return;
        g_copy_stats.us[which] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        g_copy_stats.n[which]++; g_copy_stats.bytes[which] += size;
    }
};

// set/get are synchronous host<->device copies (ggml-cuda.cu:604-618)
static void mi355_buffer_set_tensor(ggml_backend_buffer_t buffer, ggml_tensor * tensor, const void * data, size_t offset, size_t size) {
    mi355_buffer_ctx * ctx = (mi355_buffer_ctx *)buffer->context;
    mi355_copy_timer tm(0, size);
    mi355_set_device(ctx->device);
    HIP_CHECK(hipMemcpy((char *)tensor->data + offset, data, size, hipMemcpyHostToDevice));
}

static void mi355_buffer_get_tensor(ggml_backend_buffer_t buffer, const ggml_tensor * tensor, void * data, size_t offset, size_t size) {
    mi355_buffer_ctx * ctx = (mi355_buffer_ctx *)buffer->context;
    mi355_copy_timer tm(1, size);
    mi355_set_device(ctx->device);
    HIP_CHECK(hipMemcpy(data, (const char *)tensor->data + offset, size, hipMemcpyDeviceToHost));
}

static bool mi355_buffer_cpy_tensor(ggml_backend_buffer_t buffer, const ggml_tensor * src, ggml_tensor * dst) {
    ggml_backend_buffer_t sbuf = src->view_src ? src->view_src->buffer : src->buffer;
    if (!sbuf || !buft_is_mi355(sbuf->buft)) return false; // foreign source: caller falls back to get/set
    mi355_buffer_ctx * sctx = (mi355_buffer_ctx *)sbuf->context;
    mi355_buffer_ctx * dctx = (mi355_buffer_ctx *)buffer->context;
    mi355_set_device(dctx->device);
    mi355_copy_d2d_async(dst->data, dctx->device, src->data, sctx->device, ggml_nbytes(src), nullptr);
    HIP_CHECK(hipDeviceSynchronize());
    return true;
}

static void mi355_buffer_clear(ggml_backend_buffer_t buffer, uint8_t value) {
    mi355_buffer_ctx * ctx = (mi355_buffer_ctx *)buffer->context;
    mi355_set_device(ctx->device);
    HIP_CHECK(hipMemset(ctx->base, value, buffer->size));
    HIP_CHECK(hipDeviceSynchronize());
}

static const ggml_backend_buffer_i mi355_buffer_iface = {
    /* .free_buffer   = */ mi355_buffer_free,
    /* .get_base      = */ mi355_buffer_get_base,
    /* .init_tensor   = */ mi355_buffer_init_tensor,
    /* .memset_tensor = */ mi355_buffer_memset_tensor,
    /* .set_tensor    = */ mi355_buffer_set_tensor,
    /* .get_tensor    = */ mi355_buffer_get_tensor,
    /* .cpy_tensor    = */ mi355_buffer_cpy_tensor,
    /* .clear         = */ mi355_buffer_clear,
    /* .reset         = */ NULL,
};

// buffer type
static const char * mi355_buft_get_name(ggml_backend_buffer_type_t buft) {
    return ((mi355_buft_ctx *)buft->context)->name.c_str();
}

static bool buft_is_mi355(ggml_backend_buffer_type_t buft) {
    return buft->iface.get_name == mi355_buft_get_name; // identity by function pointer (ggml-cuda.cu:566-568)
}

static ggml_backend_buffer_t mi355_buft_alloc_buffer(ggml_backend_buffer_type_t buft, size_t size) {
    mi355_buft_ctx * bctx = (mi355_buft_ctx *)buft->context;
    mi355_set_device(bctx->device);
    void * ptr = nullptr;
    const size_t alloc = size > 0 ? size : 1;
    hipError_t err = hipMalloc(&ptr, alloc);
    if (err != hipSuccess) {
        (void)hipGetLastError(); // allocation failure is the one soft error: NULL to the caller (ggml-cuda.cu:683-689)
        MI_LOG_ERROR("%s: allocating %.2f MiB on device %d: hipMalloc failed: %s\n", __func__, size / 1024.0 / 1024.0,
                     bctx->device, hipGetErrorString(err));
        return nullptr;
    }
    mi355_buffer_ctx * ctx = new mi355_buffer_ctx{bctx->device, ptr};
    return ggml_backend_buffer_init(buft, mi355_buffer_iface, ctx, size);
}

static size_t mi355_buft_get_alignment(ggml_backend_buffer_type_t) { return 128; }

static ggml_backend_dev_t mi355_reg_get_device(ggml_backend_reg_t reg, size_t index);

static ggml_backend_buffer_type_t mi355_buffer_type(int device) {
    static std::mutex mu;
    static ggml_backend_buffer_type bufts[MI355_MAX_DEVICES];
    static bool inited = false;
    std::lock_guard<std::mutex> lk(mu);
    if (device < 0 || device >= mi355_device_count()) return nullptr;
    if (!inited) {
        for (int i = 0; i < mi355_device_count(); i++) {
            bufts[i].iface = {
                /* .get_name       = */ mi355_buft_get_name,
                /* .alloc_buffer   = */ mi355_buft_alloc_buffer,
                /* .get_alignment  = */ mi355_buft_get_alignment,
                /* .get_max_size   = */ NULL,
                /* .get_alloc_size = */ NULL,
                /* .is_host        = */ NULL,
            };
            bufts[i].device  = mi355_reg_get_device(ggml_backend_mi355_reg(), i);
            bufts[i].context = new mi355_buft_ctx{i, std::string(MI355_DEV_PREFIX) + std::to_string(i)};
        }
        inited = true;
    }
    return &bufts[device];
}

// ------------------------------------------------------------------------------------------------
// pinned host buffer (ggml-cuda.cu:1086-1146)
// ------------------------------------------------------------------------------------------------
static const char * mi355_host_buft_name(ggml_backend_buffer_type_t) { return MI355_NAME "_Host"; }

static void mi355_host_buffer_free(ggml_backend_buffer_t buffer) { HIP_CHECK(hipHostFree(buffer->context)); }

static ggml_backend_buffer_t mi355_host_buft_alloc(ggml_backend_buffer_type_t buft, size_t size) {
    void * ptr = nullptr;
    if (getenv("GGML_MI355_NO_PINNED") == nullptr) {
        hipError_t err = hipHostMalloc(&ptr, size > 0 ? size : 1, hipHostMallocDefault);
        if (err != hipSuccess) { (void)hipGetLastError(); ptr = nullptr; }
    }
    if (!ptr) return ggml_backend_buft_alloc_buffer(ggml_backend_cpu_buffer_type(), size); // pageable fallback
    ggml_backend_buffer_t buffer = ggml_backend_cpu_buffer_from_ptr(ptr, size);
    buffer->buft = buft;
    buffer->iface.free_buffer = mi355_host_buffer_free;
    return buffer;
}

static ggml_backend_buffer_type_t mi355_host_buffer_type() {
    static ggml_backend_buffer_type buft = {
        /* .iface = */ {
            /* .get_name       = */ mi355_host_buft_name,
            /* .alloc_buffer   = */ mi355_host_buft_alloc,
            /* .get_alignment  = */ ggml_backend_cpu_buffer_type()->iface.get_alignment,
            /* .get_max_size   = */ NULL,
            /* .get_alloc_size = */ ggml_backend_cpu_buffer_type()->iface.get_alloc_size,
            /* .is_host        = */ ggml_backend_cpu_buffer_type()->iface.is_host,
        },
        /* .device  = */ mi355_reg_get_device(ggml_backend_mi355_reg(), 0),
        /* .context = */ nullptr,
    };
    return &buft;
}

// ------------------------------------------------------------------------------------------------
// row-split buffer type: llama.cpp's `-sm row` (reference: ggml_backend_cuda_split_buffer_type and its
// buffer, ggml-cuda.cu:753-1084).  A 2-D weight matrix is cut by rows across all devices; device d
// holds rows [row_low, row_high) contiguously in its own allocation, made per tensor in init_tensor
// (tensor->extra keeps the per-device pointers); tensor->data is a dummy address that is never
// dereferenced.  Only MUL_MAT may read such a tensor (supports_op), and set/get move whole tensors.
// ------------------------------------------------------------------------------------------------
#define MI355_ROW_ROUNDING 128   // rows per MMQ tile (mmq.hip BM); the reference uses mmq_y = 128 on AMD (mmq.cuh:126-129)

struct mi355_split_buft_ctx {
    int         main_device;
    float       split[MI355_MAX_DEVICES];   // cumulative, normalised: device d starts at nrows * split[d]
    std::string name;
};

struct mi355_split_extra {
    void *  data[MI355_MAX_DEVICES]     = {};
    int64_t row_low[MI355_MAX_DEVICES]  = {};
    int64_t row_high[MI355_MAX_DEVICES] = {};
};

struct mi355_split_buffer_ctx {
    std::vector<mi355_split_extra *> extras;
    ~mi355_split_buffer_ctx() {
        for (mi355_split_extra * e : extras) {
            for (int d = 0; d < MI355_MAX_DEVICES; d++) {
                if (e->data[d]) { mi355_set_device(d); HIP_CHECK(hipFree(e->data[d])); }
            }
            delete e;
        }
    }
};

// boundaries exactly as get_row_split (ggml-cuda.cu:768-781): float product truncated, rounded down to the
// tile height, the last device takes the remainder
static void mi355_row_split(const float * split, int64_t nrows, int d, int64_t * lo, int64_t * hi) {
    const int n = mi355_device_count();
    *lo = d == 0 ? 0 : (int64_t)(nrows * split[d]);
    *lo -= *lo % MI355_ROW_ROUNDING;
    if (d == n - 1) {
        *hi = nrows;
    } else {
        *hi = (int64_t)(nrows * split[d + 1]);
        *hi -= *hi % MI355_ROW_ROUNDING;
    }
}

static const char * mi355_split_buft_get_name(ggml_backend_buffer_type_t buft) {
    return ((mi355_split_buft_ctx *)buft->context)->name.c_str();
}

static bool buft_is_mi355_split(ggml_backend_buffer_type_t buft) {
    return buft->iface.get_name == mi355_split_buft_get_name;
}

static bool tensor_is_split(const ggml_tensor * t) { return t && t->buffer && buft_is_mi355_split(t->buffer->buft); }

static void mi355_split_buffer_free(ggml_backend_buffer_t buffer) { delete (mi355_split_buffer_ctx *)buffer->context; }

static void * mi355_split_buffer_get_base(ggml_backend_buffer_t) { return (void *)0x1000; }

static enum ggml_status mi355_split_buffer_init_tensor(ggml_backend_buffer_t buffer, ggml_tensor * tensor) {
    GGML_ASSERT(tensor->view_src == nullptr && "views of split tensors are not supported");
    GGML_ASSERT(ggml_is_contiguous(tensor) && "split buffers only hold contiguous tensors");
    mi355_split_buffer_ctx * ctx  = (mi355_split_buffer_ctx *)buffer->context;
    mi355_split_buft_ctx *   bctx = (mi355_split_buft_ctx *)buffer->buft->context;
    mi355_split_extra * extra = new mi355_split_extra;
    ctx->extras.push_back(extra);
    const size_t row_bytes = ggml_row_size(tensor->type, tensor->ne[0]);
    for (int d = 0; d < mi355_device_count(); d++) {
        mi355_row_split(bctx->split, ggml_nrows(tensor), d, &extra->row_low[d], &extra->row_high[d]);
        const int64_t rows = extra->row_high[d] - extra->row_low[d];
        if (rows == 0) continue;
        mi355_set_device(d);
        hipError_t err = hipMalloc(&extra->data[d], rows * row_bytes);
        if (err != hipSuccess) {
            (void)hipGetLastError();
            MI_LOG_ERROR("%s: allocating %.2f MiB of %s on device %d failed: %s\n", __func__, rows * row_bytes / 1048576.0,
                         tensor->name, d, hipGetErrorString(err));
            // give back the slices the earlier devices already hold for this tensor: it stays without an extra
            for (int e = 0; e < d; e++) {
                if (extra->data[e]) { mi355_set_device(e); HIP_CHECK(hipFree(extra->data[e])); }
            }
            ctx->extras.pop_back();
            delete extra;
            tensor->extra = nullptr;
            return GGML_STATUS_ALLOC_FAILED;
        }
    }
    tensor->extra = extra;
    return GGML_STATUS_SUCCESS;
}

// split tensors are written and read whole (ggml-cuda.cu:878-954)
static void mi355_split_buffer_set_tensor(ggml_backend_buffer_t, ggml_tensor * tensor, const void * data, size_t offset, size_t size) {
    GGML_ASSERT(offset == 0 && size == ggml_nbytes(tensor));
    mi355_split_extra * extra = (mi355_split_extra *)tensor->extra;
    GGML_ASSERT(extra != nullptr && "split tensor was not initialised (init_tensor failed?)");
    const size_t row_bytes = tensor->nb[1];
    for (int d = 0; d < mi355_device_count(); d++) {
        if (!extra->data[d]) continue;
        mi355_set_device(d);
        HIP_CHECK(hipMemcpy(extra->data[d], (const char *)data + extra->row_low[d] * row_bytes,
                            (extra->row_high[d] - extra->row_low[d]) * row_bytes, hipMemcpyHostToDevice));
    }
}

static void mi355_split_buffer_get_tensor(ggml_backend_buffer_t, const ggml_tensor * tensor, void * data, size_t offset, size_t size) {
    GGML_ASSERT(offset == 0 && size == ggml_nbytes(tensor));
    const mi355_split_extra * extra = (const mi355_split_extra *)tensor->extra;
    GGML_ASSERT(extra != nullptr && "split tensor was not initialised (init_tensor failed?)");
    const size_t row_bytes = tensor->nb[1];
    for (int d = 0; d < mi355_device_count(); d++) {
        if (!extra->data[d]) continue;
        mi355_set_device(d);
        HIP_CHECK(hipMemcpy((char *)data + extra->row_low[d] * row_bytes, extra->data[d],
                            (extra->row_high[d] - extra->row_low[d]) * row_bytes, hipMemcpyDeviceToHost));
    }
}

// a split buffer owns no memory of its own (slices are allocated per tensor and written whole by set_tensor), so there is
// nothing to clear; get_max_size is left NULL (no limit: every tensor is its own set of allocations)
static void mi355_split_buffer_clear(ggml_backend_buffer_t, uint8_t) {}

static const ggml_backend_buffer_i mi355_split_buffer_iface = {
    /* .free_buffer   = */ mi355_split_buffer_free,
    /* .get_base      = */ mi355_split_buffer_get_base,
    /* .init_tensor   = */ mi355_split_buffer_init_tensor,
    /* .memset_tensor = */ NULL,
    /* .set_tensor    = */ mi355_split_buffer_set_tensor,
    /* .get_tensor    = */ mi355_split_buffer_get_tensor,
    /* .cpy_tensor    = */ NULL,
    /* .clear         = */ mi355_split_buffer_clear,
    /* .reset         = */ NULL,
};

static ggml_backend_buffer_t mi355_split_buft_alloc_buffer(ggml_backend_buffer_type_t buft, size_t size) {
    // device memory is allocated per tensor in init_tensor; `size` is the cap ggml-alloc enforces
    return ggml_backend_buffer_init(buft, mi355_split_buffer_iface, new mi355_split_buffer_ctx, size);
}

static size_t mi355_split_buft_get_alloc_size(ggml_backend_buffer_type_t buft, const ggml_tensor * tensor) {
    GGML_ASSERT(ggml_is_contiguous(tensor) && "split buffers only hold contiguous tensors");
    mi355_split_buft_ctx * bctx = (mi355_split_buft_ctx *)buft->context;
    size_t total = 0;
    for (int d = 0; d < mi355_device_count(); d++) {
        int64_t lo, hi;
        mi355_row_split(bctx->split, ggml_nrows(tensor), d, &lo, &hi);
        total += (hi - lo) * ggml_row_size(tensor->type, tensor->ne[0]);
    }
    return total;
}

static bool mi355_split_buft_is_host(ggml_backend_buffer_type_t) { return false; }

// exported through get_proc_address("ggml_backend_split_buffer_type") -- ggml-backend.h:188, caller
// src/llama-model.cpp:351-363.  tensor_split has one weight per device (all zero / NULL: split by memory).
static ggml_backend_buffer_type_t mi355_split_buffer_type(int main_device, const float * tensor_split) {
    static std::mutex mu;
    static std::vector<ggml_backend_buffer_type *> made;   // never freed: buffer types have static lifetime
    std::lock_guard<std::mutex> lk(mu);
    const int n = mi355_device_count();
    if (main_device < 0 || main_device >= n) return nullptr;
    float w[MI355_MAX_DEVICES] = {};
    bool all_zero = true;
    for (int d = 0; d < n && tensor_split; d++) if (tensor_split[d] != 0.0f) all_zero = false;
    for (int d = 0; d < n; d++) {
        if (all_zero) {
            size_t free_b, total_b;
            mi355_set_device(d);
            HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
            w[d] = (float)total_b;
        } else {
            w[d] = tensor_split[d];
        }
    }
    float cum[MI355_MAX_DEVICES] = {}, sum = 0.0f;
    for (int d = 0; d < n; d++) { cum[d] = sum; sum += w[d]; }
    for (int d = 0; d < n; d++) cum[d] /= sum;
    for (ggml_backend_buffer_type * b : made) {
        mi355_split_buft_ctx * c = (mi355_split_buft_ctx *)b->context;
        if (c->main_device == main_device && memcmp(c->split, cum, sizeof(cum)) == 0) return b;
    }
    mi355_split_buft_ctx * c = new mi355_split_buft_ctx;
    c->main_device = main_device;
    memcpy(c->split, cum, sizeof(cum));
    c->name = std::string(MI355_DEV_PREFIX) + std::to_string(main_device) + "_Split";
    ggml_backend_buffer_type * b = new ggml_backend_buffer_type{
        /* .iface = */ {
            /* .get_name       = */ mi355_split_buft_get_name,
            /* .alloc_buffer   = */ mi355_split_buft_alloc_buffer,
            /* .get_alignment  = */ mi355_buft_get_alignment,
            /* .get_max_size   = */ NULL,
            /* .get_alloc_size = */ mi355_split_buft_get_alloc_size,
            /* .is_host        = */ mi355_split_buft_is_host,
        },
        /* .device  = */ mi355_reg_get_device(ggml_backend_mi355_reg(), main_device),
        /* .context = */ c,
    };
    made.push_back(b);
    return b;
}

// environment switches are read once per process (graph_compute consults them per node)
#define MI355_KV_SLOTS 512          // fused Q/K/V launches per graph whose K / V destinations go through the table
#define MI355_KV_RING  8
#define MI355_ENV_SET(name) ([]() -> bool { static const bool v_ = getenv(name) != nullptr; return v_; }())

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
    void * scratch[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};     // [6]: a row nobody reads (see mi355_try_fuse_glu_matvec)
    size_t scratch_size[7] = {0, 0, 0, 0, 0, 0, 0};
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
        if (!act_cache.valid) return;
        const char * w0 = (const char *)node->data, * w1 = w0 + ggml_nbytes(node);
        const char * c0 = (const char *)act_cache.data, * c1 = c0 + act_cache.span;
        if (w0 < c1 && c0 < w1) act_cache.valid = false;
    }
};

static ggml_guid_t mi355_guid() {
    static ggml_guid guid = {0x4d, 0x49, 0x33, 0x35, 0x35, 0x58, 0x67, 0x66, 0x78, 0x39, 0x35, 0x30, 0x67, 0x67, 0x6d, 0x6c};
    return &guid;
}

static const char * mi355_backend_get_name(ggml_backend_t backend) { return ((mi355_backend_ctx *)backend->context)->name.c_str(); }

static void mi355_backend_free(ggml_backend_t backend) {
    mi355_backend_ctx * ctx = (mi355_backend_ctx *)backend->context;
    mi355_set_device(ctx->device);
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 7; i++) if (ctx->scratch[i]) HIP_CHECK(hipFree(ctx->scratch[i]));
    if (MI355_ENV_SET("GGML_MI355_KERNEL_TIMES")) {      // per-kernel launch counts and device time of the whole process so far
        std::vector<char> rep((size_t)mi355_timing_report(nullptr, 0));
        mi355_timing_report(rep.data(), (int64_t)rep.size());
        std::string line;
        for (char c : rep) {
            if (c == '\n') { fprintf(stderr, "[mi355] kernel %s\n", line.c_str()); line.clear(); }
            else if (c) line += c;
        }
    }
    if (MI355_ENV_SET("GGML_MI355_GRAPH_STATS"))
        fprintf(stderr, "[mi355] %s decode graphs: %ld replays, %ld captures, %ld eager runs of capturable graphs%s\n", ctx->name.c_str(),
                ctx->gc.n_replay, ctx->gc.n_capture, ctx->gc.n_eager, ctx->gc.disabled ? " (disabled)" : "");
    if (MI355_ENV_SET("GGML_MI355_GRAPH_STATS") && ctx->gc.n_dropped)
        fprintf(stderr, "[mi355] %s decode graphs dropped because a scratch block moved: %ld\n", ctx->name.c_str(), ctx->gc.n_dropped);
    if (ctx->gc.exec) HIP_CHECK(hipGraphExecDestroy(ctx->gc.exec));
    for (hipEvent_t ev : ctx->gc.ring_ev) if (ev) HIP_CHECK(hipEventDestroy(ev));
    if (ctx->gc.kv_dev) HIP_CHECK(hipFree(ctx->gc.kv_dev));
    if (ctx->gc.kv_host) HIP_CHECK(hipHostFree(ctx->gc.kv_host));
    if (ctx->ev_ready) HIP_CHECK(hipEventDestroy(ctx->ev_ready));
    for (int d = 0; d < MI355_MAX_DEVICES; d++) {
        mi355_backend_ctx::peer_t & p = ctx->peers[d];
        if (!p.stream) continue;
        mi355_set_device(d);
        HIP_CHECK(hipStreamSynchronize(p.stream));
        if (p.act) HIP_CHECK(hipFree(p.act));
        if (p.gemm_ws) HIP_CHECK(hipFree(p.gemm_ws));
        HIP_CHECK(hipEventDestroy(p.done));
        HIP_CHECK(hipStreamDestroy(p.stream));
    }
    for (int d = 0; d < MI355_MAX_DEVICES; d++) {
        mi355_backend_ctx::split_buf_t & sb = ctx->split_bufs[d];
        if (!sb.part && !sb.gath) continue;
        mi355_set_device(d);
        if (sb.part) HIP_CHECK(hipFree(sb.part));
        if (sb.gath) HIP_CHECK(hipFree(sb.gath));
    }
    mi355_set_device(ctx->device);
    HIP_CHECK(hipStreamDestroy(ctx->stream));
    delete ctx;
    delete backend;
}

static bool tensor_on_device(const ggml_tensor * t, int device) {
    ggml_backend_buffer_t buf = t->view_src ? t->view_src->buffer : t->buffer;
    return buf && buft_is_mi355(buf->buft) && ((mi355_buffer_ctx *)buf->context)->device == device;
}

static void mi355_backend_set_tensor_async(ggml_backend_t backend, ggml_tensor * tensor, const void * data, size_t offset, size_t size) {
    mi355_backend_ctx * ctx = (mi355_backend_ctx *)backend->context;
    GGML_ASSERT(tensor_on_device(tensor, ctx->device) && "unsupported buffer type");
    mi355_set_device(ctx->device);
    HIP_CHECK(hipMemcpyAsync((char *)tensor->data + offset, data, size, hipMemcpyHostToDevice, ctx->stream));
}

static void mi355_backend_get_tensor_async(ggml_backend_t backend, const ggml_tensor * tensor, void * data, size_t offset, size_t size) {
    mi355_backend_ctx * ctx = (mi355_backend_ctx *)backend->context;
    GGML_ASSERT(tensor_on_device(tensor, ctx->device) && "unsupported buffer type");
    mi355_set_device(ctx->device);
    HIP_CHECK(hipMemcpyAsync(data, (const char *)tensor->data + offset, size, hipMemcpyDeviceToHost, ctx->stream));
}

static bool backend_is_mi355(ggml_backend_t backend) { return backend && ggml_guid_matches(backend->guid, mi355_guid()); }

static bool mi355_backend_cpy_tensor_async(ggml_backend_t backend_src, ggml_backend_t backend_dst, const ggml_tensor * src, ggml_tensor * dst) {
    if (!backend_is_mi355(backend_src) || !backend_is_mi355(backend_dst)) return false;
    mi355_backend_ctx * sctx = (mi355_backend_ctx *)backend_src->context;
    mi355_backend_ctx * dctx = (mi355_backend_ctx *)backend_dst->context;
    if (!tensor_on_device(src, sctx->device) || !tensor_on_device(dst, dctx->device)) return false;
    if (backend_src != backend_dst) {
        // copy on the source stream, then make the destination stream wait for it (ggml-cuda.cu:2516-2569)
        mi355_set_device(sctx->device);
        mi355_copy_d2d_async(dst->data, dctx->device, src->data, sctx->device, ggml_nbytes(dst), sctx->stream);
        hipEvent_t ev;
        HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        HIP_CHECK(hipEventRecord(ev, sctx->stream));
        mi355_set_device(dctx->device);
        HIP_CHECK(hipStreamWaitEvent(dctx->stream, ev, 0));
        HIP_CHECK(hipEventDestroy(ev));
    } else {
        mi355_set_device(sctx->device);
        HIP_CHECK(hipMemcpyAsync(dst->data, src->data, ggml_nbytes(dst), hipMemcpyDeviceToDevice, sctx->stream));
    }
    return true;
}

static void mi355_backend_synchronize(ggml_backend_t backend) {
    mi355_backend_ctx * ctx = (mi355_backend_ctx *)backend->context;
    mi355_set_device(ctx->device);
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
}

// ------------------------------------------------------------------------------------------------
// ops
// ------------------------------------------------------------------------------------------------
static bool is_weight_type(enum ggml_type t) {
    return t == GGML_TYPE_Q4_0 || t == GGML_TYPE_Q8_0 || t == GGML_TYPE_Q4_K || t == GGML_TYPE_Q5_K || t == GGML_TYPE_Q6_K;
}

// MUL_MAT also takes the rest of the 32-value block types, the 2- / 3-bit K-quants and the non-linear 4-bit types (mat-vec + bf16
// GEMM, no fused / skinny kernels) and float weights
static bool is_block_weight_type(enum ggml_type t) {
    return is_weight_type(t) || t == GGML_TYPE_Q4_1 || t == GGML_TYPE_Q5_0 || t == GGML_TYPE_Q5_1 ||
           t == GGML_TYPE_Q2_K || t == GGML_TYPE_Q3_K || t == GGML_TYPE_IQ4_NL || t == GGML_TYPE_IQ4_XS;
}
static bool is_float_weight_type(enum ggml_type t) { return t == GGML_TYPE_F32 || t == GGML_TYPE_F16 || t == GGML_TYPE_BF16; }

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// MUL_MAT on block-quantized src0 (reference dispatch: ggml_cuda_mul_mat, ggml-cuda.cu:1988-2063;
// contract ggml.c:2971-2986; oracle ggml_compute_forward_mul_mat, ggml-cpu.c:1196-1388)
//
// Kernel per batch size N (columns of src1), from tools/bench_skinny.py on MI355X (weights 4096..14336 squared-ish):
//   N <= 3 (q4_K, q5_K) / 4 (q6_K) / 8 (others)   mat-vec (mmvq.hip), q8_1 activations, HBM-bound at N = 1
//   up to 32 (K-quants; GGML_MI355_SKINNY_MAX) skinny int8-MFMA kernels (mmsk.hip), one pass of 32 columns: the K-quants on
//                                         q8_K activations with the integer sub-block scales kept on the matrix core,
//                                         the other types on q8_1 with a per-block f32 rescale (up to 32 / 64)
//   beyond                                dequantize-in-LDS bf16 MFMA GEMM (mmdq.hip), bf16 activations
// GGML_MI355_MMQ_INT8=1 keeps prefill on the int8 MFMA tile kernel (mmq.hip), GGML_MI355_NO_SKINNY=1 removes the
// middle row.  The choice depends on (type, N, K) only, never on the row count, so a row-split matrix takes the same
// path (and produces the same bits) as the unsplit one.
// Float weights (f16 / bf16 / f32: an unquantized GGUF, or K / V of a graph without flash attention) take MM_FLOAT -- the
// strided, broadcasting mat-vec of mmf.hip straight on src1, one pass over the weights per 8 columns -- up to 16 columns
// and whenever the GEMM's preconditions fail (K % 64, alignment, f16 src1); beyond, the bf16 GEMM per channel.
// batches of more rows than this are "prefill-sized": the K-quant MUL_MATs that read them take the bf16 GEMM, so fused producers (norm,
// GLU, attention) store the bf16 rows alongside their f32 result.  A consumer that wants another format converts for itself, as always.
#define MI355_BF16_ROWS_ABOVE 32
enum mi355_mm_path { MM_VEC = 0, MM_DQ = 1, MM_SKINNY_K = 2, MM_SKINNY = 3, MM_Q_INT8 = 4, MM_FLOAT = 5 };
// activation format each path consumes (act_cache.fmt): 0 q8_1 SoA, 1 bf16 rows, 2 q8_K planes
static int mm_path_fmt(int path) { return path == MM_DQ ? 1 : path == MM_SKINNY_K ? 2 : 0; }

// dst: the MUL_MAT node when the caller has it (its op_params carry the precision request)
static int mi355_mul_mat_path(const ggml_tensor * src0, const ggml_tensor * src1, const ggml_tensor * dst = nullptr) {
    static const bool int8_mmq = MI355_ENV_SET("GGML_MI355_MMQ_INT8");
    static const bool no_skinny = MI355_ENV_SET("GGML_MI355_NO_SKINNY");
    const int64_t K = src0->ne[0], N = src1->ne[1];
    const bool aligned = (((uintptr_t)src1->data | src1->nb[1] | src1->nb[2] | src1->nb[3]) & 15) == 0;
    if (is_float_weight_type(src0->type)) {
        // The bf16 GEMM rounds both operands to bf16 (8 significant bits).  f32 weights, and products the graph asks f32
        // precision for (GGML_PREC_F32: llama.cpp sets it on K*Q without flash attention, src/llama-graph.cpp build_attn_mha), stay
        // on the exact f32-accumulating mat-vec whatever the batch -- the reference keeps f16 inputs with f32 accumulation there
        // (ggml-cuda.cu:1205-1351) and ggml-cpu is f32 throughout.
        if (src0->type == GGML_TYPE_F32) return MM_FLOAT;
        if (dst && dst->op == GGML_OP_MUL_MAT && (enum ggml_prec)dst->op_params[0] == GGML_PREC_F32) return MM_FLOAT;
        return N > 16 && K % 64 == 0 && aligned && src1->type == GGML_TYPE_F32 ? MM_DQ : MM_FLOAT;
    }
    if (!is_weight_type(src0->type)) {      // q4_1 / q5_0 / q5_1 / q2_K / q3_K / iq4_nl / iq4_xs
        return N > 8 && K % 64 == 0 && aligned ? MM_DQ : MM_VEC;
    }
    if (!no_skinny && K % 256 == 0 && aligned && !(N > 8 && int8_mmq)) {
        // q8_K activations, integer sub-block scales on the matrix core: q4_K / q5_K digit-split (mmsk2), q6_K one 32x32x16 MFMA per
        // 16-weight sub-block (mmsk6; up to 4 columns the column-group mat-vec is still ahead: 15.8 against 20.5 us on 4096 x 14336)
        const bool kq = src0->type == GGML_TYPE_Q4_K || src0->type == GGML_TYPE_Q5_K || src0->type == GGML_TYPE_Q6_K;
        // One pass of 32 columns.  A second pass streams the weights again, and since the GEMM splits K its time is flat up to 128
        // columns (4096 x 14336 q4_K: 41 us at 33..128 columns against 37 / 56 / 74 for 2 / 3 / 4 skinny passes; 4096 x 4096: 18 against
        // 24 for 2), and the GEMM path brings the fused Q/K/V finish with it: 8B pp64 6889 -> 7545 t/s, pp128 12.3k -> 13.8k.
        static const int64_t skinny_max = [] { const char * e = getenv("GGML_MI355_SKINNY_MAX"); return e ? atoll(e) : 32; }();
        if (kq  && N >= (src0->type == GGML_TYPE_Q6_K ? 5 : 4) && N <= skinny_max) return MM_SKINNY_K;
        if (!kq && N > 8  && N <= (K >= 8192 ? 64 : 32))  return MM_SKINNY;
    }
    if (N > 8 && int8_mmq && K % 128 == 0) return MM_Q_INT8;
    if (N > 8 && K % 64 == 0 && aligned && !int8_mmq) return MM_DQ;
    return MM_VEC;
}

// planes of the activation scratch for `rows` rows of length K in the format of `path` (each 256-byte aligned)
struct mi355_act_layout { size_t b0, b1, b2; size_t total() const { return b0 + b1 + b2; } };
static mi355_act_layout mi355_act_planes(int path, int64_t K, int64_t rows) {
    mi355_act_layout l = {0, 0, 0};
    switch (mm_path_fmt(path)) {
        case 1: l.b0 = align_up((size_t)rows * K * 2, 256); break;
        case 2: l.b0 = align_up((size_t)rows * K, 256); l.b1 = align_up((size_t)rows * (K / 256) * 4, 256); l.b2 = align_up((size_t)rows * (K / 16) * 2, 256); break;
        default: l.b0 = align_up((size_t)mi355_act_q_bytes(K, rows), 256); l.b1 = align_up((size_t)mi355_act_ds_bytes(K, rows), 256); break;
    }
    return l;
}

// the launches of one MUL_MAT over rows [row_low, row_low + M) of the weight matrix whose first owned row is at W;
// `act` holds all of src1 in the format of `path`
// gemm_ws / gemm_ws_bytes: split-K workspace of the prefill GEMM for THIS stream (NULL: the kernel library's own, which it allocates)
static void mi355_mul_mat_launch(const ggml_tensor * dst, const char * W, int64_t row_low, int64_t M,
                                 const char * act, int path, hipStream_t stream, void * gemm_ws = nullptr, int64_t gemm_ws_bytes = 0) {
    const ggml_tensor * src0 = dst->src[0];
    const ggml_tensor * src1 = dst->src[1];
    const int64_t K = src0->ne[0], ne02 = src0->ne[2], ne03 = src0->ne[3];
    const int64_t N = src1->ne[1], ne12 = src1->ne[2], ne13 = src1->ne[3];
    const int64_t r2 = ne12 / ne02, r3 = ne13 / ne03;
    const mi355_act_layout L = mi355_act_planes(path, K, N * ne12 * ne13);
    const int8_t * aq = (const int8_t *)act;
    const float * ads = (const float *)(act + L.b0);
    char * dbase = (char *)dst->data + row_low * sizeof(float);
    const int64_t dcol = (int64_t)(dst->nb[1] / sizeof(float));
    if (path != MM_VEC) {
        // one launch (or pass of 32 columns) per broadcast channel
        for (int64_t i13 = 0; i13 < ne13; i13++) {
            for (int64_t i12 = 0; i12 < ne12; i12++) {
                const char * Wc = W + (i13 / r3) * src0->nb[3] + (i12 / r2) * src0->nb[2];
                const int64_t arow = (i13 * ne12 + i12) * N;
                float * d = (float *)(dbase + i13 * dst->nb[3] + i12 * dst->nb[2]);
                if (path == MM_DQ) {
                    MI_CHECK(mi355_mul_mat_dq_slice_ws((int)src0->type, Wc, (int64_t)src0->nb[1], (const uint16_t *)act + arow * K, K, d, dcol, M, src0->ne[1], K, N,
                                                       gemm_ws, gemm_ws_bytes, stream));
                } else if (path == MM_Q_INT8) {
                    MI_CHECK(mi355_mul_mat_q((int)src0->type, Wc, (int64_t)src0->nb[1], aq + arow * K, ads + arow * (K / 32) * 2, d, dcol, M, K, N, stream));
                } else {
                    for (int64_t c0 = 0; c0 < N; c0 += 32) {
                        const int64_t r = arow + c0, nc = N - c0 < 32 ? N - c0 : 32;
                        if (path == MM_SKINNY_K) {
                            const int16_t * bs = (const int16_t *)(act + L.b0 + L.b1);
                            MI_CHECK(mi355_mul_mat_skinny_k((int)src0->type, Wc, (int64_t)src0->nb[1], aq + r * K, ads + r * (K / 256), bs + r * (K / 16),
                                                            d + c0 * dcol, dcol, M, K, nc, stream));
                        } else {
                            MI_CHECK(mi355_mul_mat_skinny((int)src0->type, Wc, (int64_t)src0->nb[1], aq + r * K, ads + r * (K / 32) * 2,
                                                          d + c0 * dcol, dcol, M, K, nc, stream));
                        }
                    }
                }
            }
        }
        return;
    }
    for (int64_t i13 = 0; i13 < ne13; i13++) {
        for (int64_t c0 = 0; c0 < N; c0 += 8) {
            mi355_mmvq_args a;
            memset(&a, 0, sizeof(a));
            a.type = (int32_t)src0->type;
            a.W = W + (i13 / r3) * src0->nb[3];
            a.w_row_stride = (int64_t)src0->nb[1];
            a.w_ch_stride  = (int64_t)src0->nb[2];
            a.act_q  = aq + (i13 * ne12 * N + c0) * K;
            a.act_ds = ads + (i13 * ne12 * N + c0) * (K / 32) * 2;
            a.act_ch_stride = N;
            a.dst = (float *)(dbase + i13 * dst->nb[3]) + c0 * dcol;
            a.dst_col_stride = dcol;
            a.dst_ch_stride  = (int64_t)(dst->nb[2] / sizeof(float));
            a.m = M; a.k = K;
            a.ncols = (int32_t)(N - c0 < 8 ? N - c0 : 8);
            a.nchannels = (int32_t)ne12;
            a.ch_ratio  = (int32_t)r2;
            MI_CHECK(mi355_mul_mat_vec_q(&a, stream));
        }
    }
}

static mi355_tensor to_mi355_tensor(const ggml_tensor * t);

void mi355_backend_ctx::peer_gemm_ws(int d, const ggml_tensor * src0, const mi355_split_extra * extra, int64_t K, int64_t N, int path, void ** ws, int64_t * bytes) {
    *ws = nullptr; *bytes = 0;
    if (path != MM_DQ) return;
    const int64_t own = extra->row_high[d] - extra->row_low[d];
    const int64_t need = mi355_mul_mat_dq_workspace_bytes(own > 0 ? own : 0, src0->ne[1], K, N);
    if (need <= 0) return;
    peer_t & p = peer(d);
    grow_on(d, &p.gemm_ws, &p.gemm_ws_size, (size_t)need, p.stream);
    mi355_set_device(device);
    *ws = p.gemm_ws; *bytes = need;
}

static void mi355_mul_mat(mi355_backend_ctx * ctx, ggml_tensor * dst) {
    const ggml_tensor * src0 = dst->src[0];
    const ggml_tensor * src1 = dst->src[1];
    if (src0->type == GGML_TYPE_F32 && src0->ne[1] <= 1024 && src1->type == GGML_TYPE_F32) {
        // small f32 matrices (the MoE router): the kernel whose summation order the fused router reproduces
        mi355_tensor ta = to_mi355_tensor(src0), tb = to_mi355_tensor(src1), td = to_mi355_tensor(dst);
        MI_CHECK(mi355_mul_mat_f32(&ta, &tb, &td, ctx->stream));
        return;
    }
    if (is_float_weight_type(src0->type) && mi355_mul_mat_path(src0, src1, dst) == MM_FLOAT) {
        mi355_tensor ta = to_mi355_tensor(src0), tb = to_mi355_tensor(src1), td = to_mi355_tensor(dst);
        MI_CHECK(mi355_mul_mat_f(&ta, &tb, &td, ctx->stream));
        return;
    }
    const int64_t K = src0->ne[0], M = src0->ne[1];
    const int64_t N = src1->ne[1], ne12 = src1->ne[2], ne13 = src1->ne[3];
    const int64_t rows = N * ne12 * ne13;
    if (rows == 0 || M == 0) return;

    // Activations for all of src1 (strided / permuted src1 is handled by the converters), once per distinct src1 and
    // format: q8_1 for the mat-vec, q8_K for the K-quant skinny kernel, bf16 for the prefill GEMM.
    const int path = mi355_mul_mat_path(src0, src1, dst), fmt = mm_path_fmt(path);
    const mi355_act_layout L = mi355_act_planes(path, K, rows);
    const size_t qb = L.total(), db = 0;
    char * ws = (char *)ctx->get_scratch(0, qb);
    if (!ctx->act_cached(src1, fmt) || MI355_ENV_SET("GGML_MI355_NO_FUSE")) {
        const float * x = (const float *)src1->data;
        if (fmt == 1)      MI_CHECK(mi355_act_bf16(x, (uint16_t *)ws, K, N, ne12, ne13, src1->nb[1], src1->nb[2], src1->nb[3], ctx->stream));
        else if (fmt == 2) MI_CHECK(mi355_quantize_act_q8k(x, (int8_t *)ws, (float *)(ws + L.b0), (int16_t *)(ws + L.b0 + L.b1), K, N, ne12, ne13,
                                                           src1->nb[1], src1->nb[2], src1->nb[3], ctx->stream));
        else               MI_CHECK(mi355_quantize_act(x, (int8_t *)ws, (float *)(ws + L.b0), K, N, ne12, ne13, src1->nb[1], src1->nb[2], src1->nb[3], ctx->stream));
        ctx->act_remember(src1, fmt);
    }

    // split-K workspace of the prefill GEMM: from this stream's scratch, so the kernel library allocates nothing
    void * gemm_ws = nullptr;
    int64_t gemm_ws_bytes = 0;
    if (path == MM_DQ) {
        const int64_t own_rows = tensor_is_split(src0) ? ((const mi355_split_extra *)src0->extra)->row_high[ctx->device] - ((const mi355_split_extra *)src0->extra)->row_low[ctx->device] : M;
        gemm_ws_bytes = mi355_mul_mat_dq_workspace_bytes(own_rows > 0 ? own_rows : 0, M, K, N);
        if (gemm_ws_bytes > 0) gemm_ws = ctx->get_scratch(3, (size_t)gemm_ws_bytes);
        if (gemm_ws && ctx->scratch_size[3] < (size_t)gemm_ws_bytes) { gemm_ws = nullptr; gemm_ws_bytes = 0; }    // under capture the buffer could not grow
    }
    if (!tensor_is_split(src0)) {
        mi355_mul_mat_launch(dst, (const char *)src0->data, 0, M, ws, path, ctx->stream, gemm_ws, gemm_ws_bytes);
        return;
    }

    // Row split (reference: ggml_cuda_op_mul_mat, ggml-cuda.cu:1431-1749).  This stream's device is the main
    // one: it quantized src1 and owns dst.  Every other device with rows gets the quantized activations by one
    // peer copy (skipped when it already holds this version), runs the same kernels on its row slice and
    // stores its rows *directly into dst on the main device* (peer stores over xGMI) -- no staging buffer and
    // no gather copy.  One event forks the helpers off the main stream, one per helper joins them.
    const mi355_split_extra * extra = (const mi355_split_extra *)src0->extra;
    GGML_ASSERT(extra && src0->ne[2] == 1 && src0->ne[3] == 1);
    if (!ctx->ev_ready) HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_ready, hipEventDisableTiming));
    HIP_CHECK(hipEventRecord(ctx->ev_ready, ctx->stream));
    const int ndev = mi355_device_count();
    // all of dst's columns one after the other (nb[2] = N nb[1], ...): what the 2-D placement after the all-gather assumes
    const bool flat_cols = dst->nb[2] == (size_t)N * dst->nb[1] && dst->nb[3] == (size_t)ne12 * dst->nb[2];
    if (flat_cols && mi355_split_gather_rccl()) {
        // ---- RCCL form: every device computes its rows into a local [cols][rows_max] slice, one all-gather hands every device all
        // slices, the main device places each slice's rows into dst.  Same kernels on the same row slices as the peer-store form:
        // the same bits.
        mi355_rccl & r = mi355_rccl_state();
        int64_t rows_max = 1;
        for (int d = 0; d < ndev; d++) rows_max = std::max(rows_max, extra->row_high[d] - extra->row_low[d]);
        const size_t slice = (size_t)rows * rows_max;                 // floats per device
        for (int d = 0; d < ndev; d++) {
            const bool main = d == ctx->device;
            hipStream_t st = main ? ctx->stream : ctx->peer(d).stream;
            mi355_backend_ctx::split_buf_t & sb = ctx->split_bufs[d];
            mi355_backend_ctx::grow_on(d, (void **)&sb.part, &sb.part_size, slice * sizeof(float), st);
            mi355_backend_ctx::grow_on(d, (void **)&sb.gath, &sb.gath_size, slice * sizeof(float) * ndev, st);
            mi355_set_device(d);
            const char * act = ws;
            if (!main) {
                mi355_backend_ctx::peer_t & p = ctx->peers[d];
                HIP_CHECK(hipStreamWaitEvent(p.stream, ctx->ev_ready, 0));
                if (extra->data[d]) {
                    if (p.act_size < qb + db) {
                        HIP_CHECK(hipStreamSynchronize(p.stream));
                        if (p.act) HIP_CHECK(hipFree(p.act));
                        p.act_size = (qb + db + (1u << 20)) & ~((size_t)(1u << 20) - 1);
                        HIP_CHECK(hipMalloc(&p.act, p.act_size));
                        p.act_version = 0;
                    }
                    if (p.act_version != ctx->act_version) {
                        mi355_copy_d2d_async(p.act, d, ws, ctx->device, qb + db, p.stream);
                        p.act_version = ctx->act_version;
                    }
                    act = (const char *)p.act;
                }
            }
            if (extra->data[d]) {
                // dst as the kernels see it: column stride rows_max, rows [row_low, row_high) of the matrix at [0, rows_d)
                ggml_tensor local = *dst;
                local.nb[1] = (size_t)rows_max * sizeof(float);
                local.nb[2] = (size_t)N * local.nb[1];
                local.nb[3] = (size_t)ne12 * local.nb[2];
                local.data  = (char *)sb.part - extra->row_low[d] * (int64_t)sizeof(float);
                void * lws = gemm_ws; int64_t lws_bytes = gemm_ws_bytes;
                if (!main) ctx->peer_gemm_ws(d, src0, extra, K, N, path, &lws, &lws_bytes);
                mi355_mul_mat_launch(&local, (const char *)extra->data[d], extra->row_low[d], extra->row_high[d] - extra->row_low[d], act, path, st, lws, lws_bytes);
            }
        }
        RCCL_CHECK(r.GroupStart());
        for (int d = 0; d < ndev; d++) {
            mi355_set_device(d);
            RCCL_CHECK(r.AllGather(ctx->split_bufs[d].part, ctx->split_bufs[d].gath, slice, ncclFloat, r.comm[d], d == ctx->device ? ctx->stream : ctx->peers[d].stream));
        }
        RCCL_CHECK(r.GroupEnd());
        mi355_set_device(ctx->device);
        for (int d = 0; d < ndev; d++) {
            const int64_t nr = extra->row_high[d] - extra->row_low[d];
            if (nr <= 0) continue;
            HIP_CHECK(hipMemcpy2DAsync((char *)dst->data + extra->row_low[d] * sizeof(float), dst->nb[1],
                                       ctx->split_bufs[ctx->device].gath + (size_t)d * slice, (size_t)rows_max * sizeof(float),
                                       (size_t)nr * sizeof(float), (size_t)rows, hipMemcpyDeviceToDevice, ctx->stream));
        }
        return;
    }
    for (int d = 0; d < ndev; d++) {
        if (d == ctx->device || !extra->data[d]) continue;
        mi355_backend_ctx::peer_t & p = ctx->peer(d);
        mi355_set_device(d);
        HIP_CHECK(hipStreamWaitEvent(p.stream, ctx->ev_ready, 0));
        if (p.act_size < qb + db && ctx->gc.capturing) {      // no allocation under capture: this capture is given up (nothing captured
            ctx->gc.capture_failed = true;                    // ever runs), the eager run that follows grows the buffer
            HIP_CHECK(hipEventRecord(p.done, p.stream));
            continue;
        }
        if (p.act_size < qb + db) {
            HIP_CHECK(hipStreamSynchronize(p.stream));
            if (p.act) HIP_CHECK(hipFree(p.act));
            p.act_size = (qb + db + (1u << 20)) & ~((size_t)(1u << 20) - 1);
            HIP_CHECK(hipMalloc(&p.act, p.act_size));
            p.act_version = 0;
        }
        if (p.act_version != ctx->act_version) {
            mi355_copy_d2d_async(p.act, d, ws, ctx->device, qb + db, p.stream);
            p.act_version = ctx->act_version;
        }
        void * lws = nullptr; int64_t lws_bytes = 0;
        ctx->peer_gemm_ws(d, src0, extra, K, N, path, &lws, &lws_bytes);
        mi355_set_device(d);
        mi355_mul_mat_launch(dst, (const char *)extra->data[d], extra->row_low[d], extra->row_high[d] - extra->row_low[d],
                             (const char *)p.act, path, p.stream, lws, lws_bytes);
        HIP_CHECK(hipEventRecord(p.done, p.stream));
    }
    mi355_set_device(ctx->device);
    if (extra->data[ctx->device]) {
        mi355_mul_mat_launch(dst, (const char *)extra->data[ctx->device], extra->row_low[ctx->device],
                             extra->row_high[ctx->device] - extra->row_low[ctx->device], ws, path, ctx->stream, gemm_ws, gemm_ws_bytes);
    }
    for (int d = 0; d < ndev; d++) {
        if (d == ctx->device || !extra->data[d]) continue;
        HIP_CHECK(hipStreamWaitEvent(ctx->stream, ctx->peers[d].done, 0));
    }
}

// MUL_MAT_ID (contract ggml.c:3000-3036; reference ggml_cuda_mul_mat_id, ggml-cuda.cu:2065-2206;
// oracle ggml_compute_forward_mul_mat_id, ggml-cpu.c:1470-1648).  Expert routing stays on the
// device: the kernel reads ids itself, so there is no stream sync + host-side sort per call
// (the reference does both, ggml-cuda/mmq.cu:157-158,188-189).
static void mi355_mul_mat_id(mi355_backend_ctx * ctx, ggml_tensor * dst) {
    const ggml_tensor * as  = dst->src[0];
    const ggml_tensor * b   = dst->src[1];
    const ggml_tensor * ids = dst->src[2];
    const int64_t K = as->ne[0], M = as->ne[1], n_expert = as->ne[2];
    const int64_t b_ne1 = b->ne[1], n_tokens = b->ne[2];
    const int64_t n_used = ids->ne[0];
    if (n_tokens == 0 || M == 0) return;
    // a few to a few hundred pairs (SURVEY config 5, T = 8..64): bucket by expert on the device, then every expert's matrix
    // is streamed once per 32 of its columns through the skinny int8-MFMA kernels
    static const int64_t sk_max = getenv("GGML_MI355_MOE_SKINNY_MAX") ? atoll(getenv("GGML_MI355_MOE_SKINNY_MAX")) : 256;
    static const int64_t sk_min = getenv("GGML_MI355_MOE_SKINNY_MIN") ? atoll(getenv("GGML_MI355_MOE_SKINNY_MIN")) : 12;
    const bool dense_dst = dst->nb[1] == (size_t)M * sizeof(float) && dst->nb[2] == (size_t)M * n_used * sizeof(float) &&
                           ((((uintptr_t)b->data) | b->nb[1] | b->nb[2]) & 15) == 0;
    // (measured on 8x7B Q4_K_M, profiles/r03_moe_skinny_min.txt: 16 pairs 642 t/s here against 605 on the per-pair mat-vec since the
    // skinny kernels stage only the columns that exist -- it was 521 against 548 before --, 8 pairs 371 against 444; above 256 the GEMM)
    // Many experts: the skinny launch costs a pass per expert in use whatever its count, the per-pair mat-vec a pass per pair -- it stays
    // ahead up to about 1.5 pairs per expert (tools/moe_lab.py, q4_K: 64 experts x 2048 x 4096 top-6, 48 pairs 43 us against 66, 96 pairs
    // 74 against 70; 128 experts top-8, 128 pairs 83 against 113, 256 pairs 156 against 122; 8 experts: 16 pairs 89 against 83)
    const int64_t sk_from = std::max(sk_min, n_expert * 3 / 2);
    if (n_tokens * n_used >= sk_from && n_tokens * n_used <= sk_max && K % 256 == 0 && n_expert <= 256 && dense_dst && is_weight_type(as->type) &&
        !MI355_ENV_SET("GGML_MI355_NO_SKINNY")) {
        const int64_t need = mi355_mul_mat_id_skinny_workspace_bytes((int)as->type, K, n_tokens, b_ne1, n_used, n_expert);
        void * ws = ctx->get_scratch(0, (size_t)need);
        ctx->act_cache.valid = false;
        const int rc = mi355_mul_mat_id_skinny((int)as->type, as->data, (int64_t)as->nb[1], (int64_t)as->nb[2], (const float *)b->data,
                                               (int64_t)b->nb[1], (int64_t)b->nb[2], (const int32_t *)ids->data,
                                               (int64_t)(ids->nb[1] / sizeof(int32_t)), (float *)dst->data, M, K, n_expert, n_used,
                                               n_tokens, b_ne1, ws, need, ctx->stream);
        if (rc == MI355_OK) return;
        if (rc != MI355_E_UNSUPPORTED) MI_CHECK(rc);
    }
    // prefill-sized batches: bucket the (token, slot) pairs by expert on the device and run one bf16-MFMA GEMM over all
    // experts, so every touched expert matrix is read once.  Below ~64 pairs the per-pair mat-vec (each pair streams
    // its expert's rows, mostly from the Infinity Cache) is faster than dequantizing whole tiles for a few columns.
    if (n_tokens * n_used >= std::max((int64_t)64, sk_from) && K % 64 == 0 && n_expert <= 256 && dense_dst && !MI355_ENV_SET("GGML_MI355_MMQ_INT8")) {
        const int64_t need = mi355_mul_mat_id_dq_workspace_bytes(K, n_tokens, b_ne1, n_used, n_expert);
        void * ws = ctx->get_scratch(0, (size_t)need);
        ctx->act_cache.valid = false;     // the workspace replaces whatever activations were cached in scratch[0]
        const int rc = mi355_mul_mat_id_dq((int)as->type, as->data, (int64_t)as->nb[1], (int64_t)as->nb[2], (const float *)b->data,
                                           (int64_t)b->nb[1], (int64_t)b->nb[2], (const int32_t *)ids->data,
                                           (int64_t)(ids->nb[1] / sizeof(int32_t)), (float *)dst->data, M, K, n_expert, n_used, n_tokens,
                                           b_ne1, ws, need, ctx->stream);
        if (rc == MI355_OK) return;
        if (rc != MI355_E_UNSUPPORTED) MI_CHECK(rc);
    }
    const int64_t rows = b_ne1 * n_tokens;
    const size_t qb = align_up((size_t)mi355_act_q_bytes(K, rows), 256);
    const size_t db = align_up((size_t)mi355_act_ds_bytes(K, rows), 256);
    char * ws = (char *)ctx->get_scratch(0, qb + db);
    int8_t * aq  = (int8_t *)ws;
    float  * ads = (float *)(ws + qb);
    if (!ctx->act_cached(b)) {   // gate / up experts read the same b
        MI_CHECK(mi355_quantize_act((const float *)b->data, aq, ads, K, b_ne1, n_tokens, 1, b->nb[1], b->nb[2], b->nb[3], ctx->stream));
        ctx->act_remember(b);
    }

    // (token, slot) pairs are the launch's channels; at most 65535 per launch
    const int64_t tok_per_launch = 65535 / n_used;
    for (int64_t t0 = 0; t0 < n_tokens; t0 += tok_per_launch) {
        const int64_t nt = n_tokens - t0 < tok_per_launch ? n_tokens - t0 : tok_per_launch;
        mi355_mmvq_args a;
        memset(&a, 0, sizeof(a));
        a.type = (int32_t)as->type;
        a.W = as->data;
        a.w_row_stride = (int64_t)as->nb[1];
        a.w_ch_stride  = (int64_t)as->nb[2];
        a.act_q  = aq + t0 * b_ne1 * K;
        a.act_ds = ads + t0 * b_ne1 * (K / 32) * 2;
        a.act_ch_stride = b_ne1;
        a.dst = (float *)((char *)dst->data + t0 * dst->nb[2]);
        a.dst_col_stride = M;
        a.dst_ch_stride  = (int64_t)(dst->nb[1] / sizeof(float));
        a.m = M; a.k = K; a.ncols = 1;
        a.nchannels = (int32_t)(nt * n_used);
        a.ch_ratio = 1;
        a.ids = (const int32_t *)((const char *)ids->data + t0 * ids->nb[1]);
        a.ids_stride = (int32_t)(ids->nb[1] / sizeof(int32_t));
        a.ids_nused  = (int32_t)n_used;
        a.act_ne1    = (int32_t)b_ne1;
        a.n_expert   = (int32_t)n_expert;
        MI_CHECK(mi355_mul_mat_vec_q(&a, ctx->stream));
    }
}

// FLASH_ATTN_EXT (contract ggml.h:2012-2033; reference dispatch ggml-cuda/fattn.cu:271-346;
// oracle ggml_compute_forward_flash_attn_ext_f16, ggml-cpu/ops.cpp:7975-8192)
static void fattn_args_from(const ggml_tensor * dst, mi355_fattn_args * a) {
    const ggml_tensor * q = dst->src[0], * k = dst->src[1], * v = dst->src[2], * mask = dst->src[3];
    memset(a, 0, sizeof(*a));
    a->q = (const float *)q->data; a->q_nb1 = q->nb[1]; a->q_nb2 = q->nb[2]; a->q_nb3 = q->nb[3];
    a->k = k->data; a->type_k = (int32_t)k->type; a->k_nb1 = k->nb[1]; a->k_nb2 = k->nb[2]; a->k_nb3 = k->nb[3];
    a->v = v->data; a->type_v = (int32_t)v->type; a->v_nb1 = v->nb[1]; a->v_nb2 = v->nb[2]; a->v_nb3 = v->nb[3];
    if (mask) { a->mask = mask->data; a->m_nb1 = mask->nb[1]; a->m_nb2 = mask->nb[2]; a->m_nb3 = mask->nb[3]; a->m_ne2 = mask->ne[2]; a->m_ne3 = mask->ne[3]; }
    a->dst = (float *)dst->data; a->d_nb1 = dst->nb[1]; a->d_nb2 = dst->nb[2]; a->d_nb3 = dst->nb[3];
    a->d = q->ne[0]; a->n_q = q->ne[1]; a->n_head = q->ne[2]; a->ne3 = q->ne[3]; a->n_kv = k->ne[1]; a->n_head_kv = k->ne[2];
    memcpy(&a->scale, (const float *)dst->op_params + 0, 4);
    memcpy(&a->max_bias, (const float *)dst->op_params + 1, 4);
    memcpy(&a->logit_softcap, (const float *)dst->op_params + 2, 4);
}

static bool mi355_flash_attn_supported(const ggml_tensor * op) {
    const ggml_tensor * q = op->src[0], * k = op->src[1], * v = op->src[2], * mask = op->src[3];
    if (q->type != GGML_TYPE_F32 || op->type != GGML_TYPE_F32) return false;
    if (k->type != v->type || (k->type != GGML_TYPE_F16 && k->type != GGML_TYPE_BF16 && k->type != GGML_TYPE_Q8_0 && k->type != GGML_TYPE_Q4_0)) return false;
    const int64_t d = q->ne[0];
    if ((d != 64 && d != 128 && d != 256) || k->ne[0] != d || v->ne[0] != d) return false;
    if (k->ne[3] != q->ne[3] || v->ne[3] != q->ne[3] || k->ne[2] != v->ne[2] || q->ne[2] % k->ne[2] != 0) return false;
    if (q->nb[0] != sizeof(float) || k->nb[0] != ggml_type_size(k->type) || v->nb[0] != ggml_type_size(v->type)) return false;
    if ((q->nb[1] | q->nb[2] | q->nb[3]) & 15) return false;
    if ((k->type == GGML_TYPE_F16 || k->type == GGML_TYPE_BF16) && ((k->nb[1] | k->nb[2] | k->nb[3] | v->nb[1] | v->nb[2] | v->nb[3]) & 15)) return false;
    if (mask && (mask->type != GGML_TYPE_F16 || mask->nb[0] != sizeof(ggml_fp16_t))) return false;
    if (!ggml_is_contiguous(op)) return false;
    if (q->ne[1] > 65535 || q->ne[3] * 64 > 65535) return false;
    return true;
}

static void mi355_flash_attn(mi355_backend_ctx * ctx, ggml_tensor * dst) {
    mi355_fattn_args a;
    fattn_args_from(dst, &a);
    // a workspace lets the KV range be split; only worth having for decode-sized batches
    void * ws = nullptr;
    int64_t ws_bytes = 0;
    // (decode-sized batches; and the 512-token ubatches of a long prompt, whose KV blocks are dealt to two workgroups per unit group)
    if ((a.n_kv >= 128 && a.n_head_kv * a.n_q * a.ne3 < 512) || (a.n_q >= 32 && a.n_kv >= 1024 && (a.n_q + 31) / 32 * a.n_head * a.ne3 < 1024)) {
        ws_bytes = mi355_flash_attn_ext_workspace_bytes(&a);
        ws = ctx->get_scratch(1, (size_t)ws_bytes);
    }
    // decode: the result (all heads of a token = one activation row) feeds the attention-output MUL_MAT; emit its
    // q8_1 form from the same kernel
    const int64_t K = a.d * a.n_head, rows = a.n_q * a.ne3;
    if (rows <= 8 && ggml_is_contiguous(dst) && !MI355_ENV_SET("GGML_MI355_NO_FUSE")) {
        const size_t qb = align_up((size_t)mi355_act_q_bytes(K, rows), 256), db = align_up((size_t)mi355_act_ds_bytes(K, rows), 256);
        char * act = (char *)ctx->get_scratch(0, qb + db);
        const int rc = mi355_flash_attn_ext_quant(&a, ws, ws_bytes, (int8_t *)act, (float *)(act + qb), ctx->stream);
        if (rc == MI355_OK) {
            // remembered under the shape its consumer uses: kqv_out = [n_head * d, n_tokens]
            ggml_tensor as_rows = *dst;
            as_rows.ne[0] = K; as_rows.ne[1] = a.n_q; as_rows.ne[2] = a.ne3; as_rows.ne[3] = 1;
            as_rows.nb[0] = sizeof(float); as_rows.nb[1] = K * sizeof(float); as_rows.nb[2] = as_rows.nb[1] * a.n_q; as_rows.nb[3] = as_rows.nb[2] * a.ne3;
            ctx->act_remember(&as_rows);
            return;
        }
        if (rc != MI355_E_UNSUPPORTED) MI_CHECK(rc);
    }
    // prefill: the bf16 rows for the attention-output GEMM from the same kernel
    if (rows > MI355_BF16_ROWS_ABOVE && ggml_is_contiguous(dst) && K % 8 == 0 && !MI355_ENV_SET("GGML_MI355_NO_FUSE") && !MI355_ENV_SET("GGML_MI355_MMQ_INT8")) {
        uint16_t * abf = (uint16_t *)ctx->get_scratch(0, align_up((size_t)rows * K * 2, 256));
        const int rc = mi355_flash_attn_ext_bf16(&a, ws, ws_bytes, abf, ctx->stream);
        if (rc == MI355_OK) {
            ggml_tensor as_rows = *dst;
            as_rows.ne[0] = K; as_rows.ne[1] = a.n_q; as_rows.ne[2] = a.ne3; as_rows.ne[3] = 1;
            as_rows.nb[0] = sizeof(float); as_rows.nb[1] = K * sizeof(float); as_rows.nb[2] = as_rows.nb[1] * a.n_q; as_rows.nb[3] = as_rows.nb[2] * a.ne3;
            ctx->act_remember(&as_rows, 1);
            return;
        }
        if (rc != MI355_E_UNSUPPORTED) MI_CHECK(rc);
    }
    MI_CHECK(mi355_flash_attn_ext(&a, ws, ws_bytes, ctx->stream));
}

static bool mi355_glu_with_quant(mi355_backend_ctx * ctx, ggml_tensor * node, const ggml_tensor * next);

// next: the node after `node` in the graph (NULL at the end), for producers that write the activation format of their reader
static bool mi355_compute_forward(mi355_backend_ctx * ctx, ggml_tensor * node, const ggml_tensor * next = nullptr) {
    switch (node->op) {
        case GGML_OP_MUL_MAT:        mi355_mul_mat(ctx, node);    return true;
        case GGML_OP_MUL_MAT_ID:     mi355_mul_mat_id(ctx, node); return true;
        case GGML_OP_FLASH_ATTN_EXT: mi355_flash_attn(ctx, node); return true;
        case GGML_OP_GLU:            if (mi355_glu_with_quant(ctx, node, next)) return true; break;
        default: break;
    }
    return mi355_ops_forward(node, ctx->stream);
}

// A fused kernel reads and writes per element index, so an output may share memory with an input only when the
// two coincide exactly (in-place); a shifted overlap (ggml-alloc reusing part of a freed block) would let one row's
// store land on another row's not-yet-read input.
static bool alias_safe(const ggml_tensor * out, const ggml_tensor * in) {
    const char * o0 = (const char *)out->data, * o1 = o0 + ggml_nbytes(out);
    const char * i0 = (const char *)in->data,  * i1 = i0 + ggml_nbytes(in);
    if (o1 <= i0 || i1 <= o0) return true;
    if (o0 != i0) return false;
    for (int d = 0; d < 4; d++) if (out->nb[d] != in->nb[d] || out->ne[d] != in->ne[d]) return false;
    return true;
}

static mi355_tensor to_mi355_tensor(const ggml_tensor * t) {
    mi355_tensor r;
    r.data = t->data; r.type = (int32_t)t->type;
    for (int d = 0; d < 4; d++) { r.ne[d] = t->ne[d]; r.nb[d] = (int64_t)t->nb[d]; }
    return r;
}

// RMS_NORM at node i followed by MUL with the per-feature norm weight; when `add` is given it is the ADD at node
// i-1 whose result the norm reads (residual add), and all three nodes run as one kernel.
// `gemm` (with `add`): the MUL_MAT whose product is one addend of `add` has not run yet; it runs here as mi355_mul_mat_dq_parts and the
// norm kernel adds its K-slice planes on the way in (prefill: wo / ffn_down -> residual ADD -> RMS_NORM -> MUL, no slice-sum launch).
// Nothing is launched unless the whole chain can be.
static uint16_t * prefill_act_bf16(mi355_backend_ctx * ctx, const ggml_tensor * src1);
static bool mi355_try_fuse_rms_norm_mul(mi355_backend_ctx * ctx, ggml_cgraph * cgraph, int i, ggml_tensor * add = nullptr, const ggml_tensor * gemm = nullptr) {
    ggml_tensor * norm = ggml_graph_node(cgraph, i);
    ggml_tensor * mul  = ggml_graph_node(cgraph, i + 1);
    if (mul->op != GGML_OP_MUL || MI355_ENV_SET("GGML_MI355_NO_FUSE")) return false;
    const ggml_tensor * w = mul->src[0] == norm ? mul->src[1] : (mul->src[1] == norm ? mul->src[0] : nullptr);
    if (!w || w == norm) return false;
    if (w->type != GGML_TYPE_F32 || mul->type != GGML_TYPE_F32 || norm->type != GGML_TYPE_F32) return false;
    if (w->ne[0] != norm->ne[0] || ggml_nrows(w) != 1 || w->nb[0] != sizeof(float)) return false;
    if (!ggml_are_same_shape(norm, mul) || norm->nb[0] != sizeof(float) || mul->nb[0] != sizeof(float)) return false;
    if (norm->src[0]->nb[0] != sizeof(float)) return false;
    if (norm->flags & GGML_TENSOR_FLAG_OUTPUT) return false;
    const int n = ggml_graph_n_nodes(cgraph);
    for (int j = i + 2; j < n; j++) {           // any other reader of the un-multiplied norm?
        const ggml_tensor * t = ggml_graph_node(cgraph, j);
        for (int s = 0; s < GGML_MAX_SRC; s++) if (t->src[s] == norm || (t->src[s] && t->src[s]->view_src == norm)) return false;
    }
    const ggml_tensor * x = norm->src[0];
    if (!alias_safe(mul, x) || !alias_safe(mul, w)) return false;
    if (add) {
        const ggml_tensor * a = add->src[0], * b = add->src[1];
        if (x != add || add->type != GGML_TYPE_F32 || a->type != GGML_TYPE_F32 || b->type != GGML_TYPE_F32) return false;
        if (!ggml_are_same_shape(a, add) || !ggml_are_same_shape(b, add)) return false;
        if (a->nb[0] != sizeof(float) || b->nb[0] != sizeof(float) || add->nb[0] != sizeof(float)) return false;
        if (!alias_safe(add, a) || !alias_safe(add, b) || !alias_safe(mul, a) || !alias_safe(mul, b)) return false;
        const char * m0 = (const char *)mul->data, * s0 = (const char *)add->data;
        if (m0 < s0 + ggml_nbytes(add) && s0 < m0 + ggml_nbytes(mul)) return false;   // the two outputs must be disjoint
    }
    float eps;
    memcpy(&eps, norm->op_params, sizeof(float));
    mi355_tensor tx = to_mi355_tensor(x), td = to_mi355_tensor(mul);
    const int64_t rows = ggml_nrows(mul), K = mul->ne[0];
    // batches whose MUL_MATs take the K-quant skinny kernels (5..32 rows; the first MUL_MAT that reads the result tells): also emit
    // the q8_K planes they read, instead of a quantization launch per batch and norm (126 such launches in a 16-token 8B step)
    if (!gemm && rows > 1 && rows <= 64 && K % 256 == 0 && K <= 8192 && ggml_is_contiguous(mul) && !MI355_ENV_SET("GGML_MI355_NO_NORM_Q8K")) {
        const int nn = ggml_graph_n_nodes(cgraph);
        int fmt = -1;
        for (int j = i + 2; j < nn && j < i + 16 && fmt < 0; j++) {
            const ggml_tensor * t = ggml_graph_node(cgraph, j);
            if (t->op == GGML_OP_MUL_MAT && t->src[1] == mul && !tensor_is_split(t->src[0])) fmt = mm_path_fmt(mi355_mul_mat_path(t->src[0], mul, t));
        }
        if (fmt == 2) {
            const mi355_act_layout L = mi355_act_planes(MM_SKINNY_K, K, rows);
            char * ws = (char *)ctx->get_scratch(0, L.total());
            mi355_tensor ta = to_mi355_tensor(add ? add->src[0] : x), tb, ts;
            if (add) { tb = to_mi355_tensor(add->src[1]); ts = to_mi355_tensor(add); }
            const int rc = mi355_rms_norm_mul_q8k_f32(&ta, add ? &tb : nullptr, add ? &ts : nullptr, (const float *)w->data, &td, eps,
                                                      (int8_t *)ws, (float *)(ws + L.b0), (int16_t *)(ws + L.b0 + L.b1), ctx->stream);
            if (rc == MI355_OK) {
                if (add) ctx->act_invalidate_if_written(add);
                ctx->act_invalidate_if_written(mul);
                ctx->act_remember(mul, 2);
                return true;
            }
            if (rc != MI355_E_UNSUPPORTED) MI_CHECK(rc);
        }
    }
    // decode-sized rows: also emit the q8_1 form for the MUL_MATs that read the result
    const bool quant = rows <= 8 && K % 32 == 0 && K <= 8192 && ggml_is_contiguous(mul);
    int8_t * aq = nullptr; float * ads = nullptr;
    if (quant) {
        const size_t qb = align_up((size_t)mi355_act_q_bytes(K, rows), 256), db = align_up((size_t)mi355_act_ds_bytes(K, rows), 256);
        char * ws = (char *)ctx->get_scratch(0, qb + db);
        aq = (int8_t *)ws; ads = (float *)(ws + qb);
    }
    // prefill-sized rows (every MUL_MAT that reads them takes the bf16 GEMM): also emit the bf16 rows, so those MUL_MATs
    // need no conversion launch
    if (gemm) {
        // everything the norm kernel could refuse is checked before the GEMM is launched
        const ggml_tensor * other = add->src[0] == gemm ? add->src[1] : add->src[0];
        uintptr_t al = (uintptr_t)other->data | (uintptr_t)add->data | (uintptr_t)mul->data | (uintptr_t)w->data;
        for (int d = 1; d < 4; d++) al |= (uintptr_t)other->nb[d] | (uintptr_t)add->nb[d] | (uintptr_t)mul->nb[d];
        if (!(rows > MI355_BF16_ROWS_ABOVE && K % 16 == 0 && K <= 8192 && ggml_is_contiguous(mul) && ggml_is_contiguous(add) && (al & 15) == 0) || MI355_ENV_SET("GGML_MI355_MMQ_INT8")) return false;
        const ggml_tensor * W = gemm->src[0], * src1 = gemm->src[1];
        const int64_t pb = mi355_mul_mat_dq_parts_bytes(W->ne[1], W->ne[1], W->ne[0], src1->ne[1]);
        if (pb <= 0 || W->ne[1] != K || src1->ne[1] != rows) return false;
        float * parts = (float *)ctx->get_scratch(3, (size_t)pb);
        if (ctx->scratch_size[3] < (size_t)pb) return false;
        const uint16_t * act = prefill_act_bf16(ctx, src1);
        int32_t kz = 0;
        const int rc0 = mi355_mul_mat_dq_parts((int)W->type, W->data, (int64_t)W->nb[1], act, W->ne[0], parts, pb, W->ne[1], W->ne[1], W->ne[0], src1->ne[1], &kz, ctx->stream);
        if (rc0 == MI355_E_UNSUPPORTED || rc0 == MI355_E_ALIGN) return false;
        MI_CHECK(rc0);
        uint16_t * abf = (uint16_t *)ctx->get_scratch(0, align_up((size_t)rows * K * 2, 256));
        mi355_tensor tb = to_mi355_tensor(other), ts = to_mi355_tensor(add);
        MI_CHECK(mi355_parts_add_rms_norm_mul_bf16_f32(parts, kz, &tb, &ts, (const float *)w->data, &td, eps, abf, ctx->stream));
        ctx->act_invalidate_if_written(add);
        ctx->act_invalidate_if_written(mul);
        ctx->act_remember(mul, 1);
        return true;
    }
    if (rows > MI355_BF16_ROWS_ABOVE && K % 16 == 0 && K <= 8192 && ggml_is_contiguous(mul) && !MI355_ENV_SET("GGML_MI355_MMQ_INT8")) {
        uint16_t * abf = (uint16_t *)ctx->get_scratch(0, align_up((size_t)rows * K * 2, 256));
        mi355_tensor ta = to_mi355_tensor(add ? add->src[0] : x), tb, ts;
        if (add) { tb = to_mi355_tensor(add->src[1]); ts = to_mi355_tensor(add); }
        const int rc = mi355_rms_norm_mul_bf16_f32(&ta, add ? &tb : nullptr, add ? &ts : nullptr, (const float *)w->data, &td, eps, abf, ctx->stream);
        if (rc == MI355_OK) {
            if (add) ctx->act_invalidate_if_written(add);
            ctx->act_invalidate_if_written(mul);
            ctx->act_remember(mul, 1);
            return true;
        }
        if (rc != MI355_E_UNSUPPORTED) MI_CHECK(rc);
    }
    if (add) {
        mi355_tensor ta = to_mi355_tensor(add->src[0]), tb = to_mi355_tensor(add->src[1]), ts = to_mi355_tensor(add);
        int rc = mi355_add_rms_norm_mul_f32(&ta, &tb, &ts, (const float *)w->data, &td, eps, aq, ads, ctx->stream);
        if (rc == MI355_E_UNSUPPORTED && quant) {
            aq = nullptr; ads = nullptr;
            rc = mi355_add_rms_norm_mul_f32(&ta, &tb, &ts, (const float *)w->data, &td, eps, nullptr, nullptr, ctx->stream);
        }
        if (rc == MI355_E_UNSUPPORTED) return false;
        MI_CHECK(rc);
        ctx->act_invalidate_if_written(add);
        ctx->act_invalidate_if_written(mul);
        if (aq) ctx->act_remember(mul);
        return true;
    }
    ctx->act_invalidate_if_written(mul);
    if (quant) {
        const int rc = mi355_rms_norm_mul_quant_f32(&tx, (const float *)w->data, &td, eps, aq, ads, ctx->stream);
        if (rc == MI355_OK) { ctx->act_remember(mul); return true; }
        if (rc != MI355_E_UNSUPPORTED) MI_CHECK(rc);
    }
    MI_CHECK(mi355_rms_norm_mul_f32(&tx, (const float *)w->data, &td, eps, ctx->stream));
    return true;
}

// GLU whose result feeds ffn_down: the activation format ffn_down reads (q8_1 at decode, q8_K planes for 5..32 rows over a K-quant,
// bf16 rows for the prefill GEMM) is written by the same launch
static bool mi355_glu_with_quant(mi355_backend_ctx * ctx, ggml_tensor * node, const ggml_tensor * next) {
    const ggml_tensor * a = node->src[0], * b = node->src[1];
    const int64_t rows = ggml_nrows(node), K = node->ne[0];
    if (!ggml_is_contiguous(node) || MI355_ENV_SET("GGML_MI355_NO_FUSE")) return false;
    auto T = [](const ggml_tensor * t) { mi355_tensor r; r.data = t->data; r.type = (int32_t)t->type;
                                         for (int d = 0; d < 4; d++) { r.ne[d] = t->ne[d]; r.nb[d] = (int64_t)t->nb[d]; } return r; };
    mi355_tensor ta = T(a), tb, td = T(node);
    if (b) tb = T(b);
    if (rows > MI355_BF16_ROWS_ABOVE && b && K % 4 == 0 && !MI355_ENV_SET("GGML_MI355_MMQ_INT8")) {   // prefill: bf16 rows for the ffn_down GEMM
        uint16_t * abf = (uint16_t *)ctx->get_scratch(0, align_up((size_t)rows * K * 2, 256));
        const int rc = mi355_glu_bf16_f32((int)ggml_get_glu_op(node), &ta, &tb, &td, ((const int32_t *)node->op_params)[1], abf, ctx->stream);
        if (rc == MI355_E_UNSUPPORTED) return false;
        MI_CHECK(rc);
        ctx->act_remember(node, 1);
        return true;
    }
    if (next && next->op == GGML_OP_MUL_MAT && next->src[1] == node && !tensor_is_split(next->src[0]) && rows > 1 && K % 256 == 0 &&
        mi355_mul_mat_path(next->src[0], node) == MM_SKINNY_K && !MI355_ENV_SET("GGML_MI355_NO_NORM_Q8K")) {
        const mi355_act_layout L = mi355_act_planes(MM_SKINNY_K, K, rows);
        char * ws = (char *)ctx->get_scratch(0, L.total());
        const int rc = mi355_glu_q8k_f32((int)ggml_get_glu_op(node), &ta, b ? &tb : nullptr, &td, ((const int32_t *)node->op_params)[1],
                                         (int8_t *)ws, (float *)(ws + L.b0), (int16_t *)(ws + L.b0 + L.b1), ctx->stream);
        if (rc == MI355_OK) { ctx->act_remember(node, 2); return true; }
        if (rc != MI355_E_UNSUPPORTED) MI_CHECK(rc);
    }
    if (rows > 8 || K % 32 != 0) return false;
    const size_t qb = align_up((size_t)mi355_act_q_bytes(K, rows), 256), db = align_up((size_t)mi355_act_ds_bytes(K, rows), 256);
    char * ws = (char *)ctx->get_scratch(0, qb + db);
    const int rc = mi355_glu_quant_f32((int)ggml_get_glu_op(node), &ta, b ? &tb : nullptr, &td, ((const int32_t *)node->op_params)[1],
                                       (int8_t *)ws, (float *)(ws + qb), ctx->stream);
    if (rc == MI355_E_UNSUPPORTED) return false;
    MI_CHECK(rc);
    ctx->act_remember(node);
    return true;
}

// ------------------------------------------------------------------------------------------------
// Grouped decode mat-vecs.  wq/wk/wv (and ffn gate/up) read the same activation; llama.cpp's graph has them a few
// nodes apart (the rope of Q sits between wq and wk).  When MUL_MAT node i is a decode mat-vec, later MUL_MATs with
// the same src1, weight type and shapes are HOISTED into its launch (mi355_mul_mat_vec_q_group) -- legal when
// nothing between i and j reads or writes the bytes node j's result will occupy (ggml-alloc reuses freed blocks, so
// this is checked on addresses) and nothing in between writes src1.  Hoisted nodes are marked done and skipped.
// ------------------------------------------------------------------------------------------------
static bool ranges_overlap(const void * a, size_t an, const void * b, size_t bn) {
    const char * a0 = (const char *)a, * b0 = (const char *)b;
    return a0 < b0 + bn && b0 < a0 + an;
}

static bool is_view_op(const ggml_tensor * t) {
    return t->op == GGML_OP_NONE || t->op == GGML_OP_RESHAPE || t->op == GGML_OP_VIEW || t->op == GGML_OP_PERMUTE || t->op == GGML_OP_TRANSPOSE;
}

static const ggml_tensor * view_root(const ggml_tensor * t) { return t->view_src ? t->view_src : t; }

// Row-split weights (-sm row) in the fused decode launches: every fused kernel takes a row range, and the 128-row rounding of the
// split keeps rope pairs, GLU pairs and q8_1 blocks inside a slice -- so each device that owns rows runs the SAME launch on its slice
// (weights from its own memory, activations and results in the main device's over xGMI) between one event fork and one join
// (mi355_split_fan_out).  GGML_MI355_NO_SPLIT_FUSE=1: split MUL_MATs run one by one through mi355_mul_mat as before.
static bool split_fusable(const ggml_tensor * w) {
    static const bool off = MI355_ENV_SET("GGML_MI355_NO_SPLIT_FUSE");
    return !off && tensor_is_split(w) && w->extra && w->ne[2] == 1 && w->ne[3] == 1;
}

static bool matvec_groupable(const ggml_tensor * node) {
    if (node->op != GGML_OP_MUL_MAT) return false;
    const ggml_tensor * a = node->src[0], * b = node->src[1];
    if (!is_weight_type(a->type) || (tensor_is_split(a) && !split_fusable(a)) || a->op != GGML_OP_NONE || a->view_src) return false;
    if (a->ne[2] != 1 || a->ne[3] != 1 || b->ne[2] != 1 || b->ne[3] != 1 || b->ne[1] < 1 || b->ne[1] > 8) return false;
    if (mi355_mul_mat_path(a, b) != MM_VEC) return false;      // e.g. 4..8 columns of a K-quant: the skinny MFMA kernel is faster
    return ggml_is_contiguous(node) && a->ne[1] > 0;
}

static int mi355_collect_matvec_group(ggml_cgraph * cgraph, int i, const std::vector<char> & done, int * members) {
    const int n = ggml_graph_n_nodes(cgraph);
    const ggml_tensor * first = ggml_graph_node(cgraph, i);
    const ggml_tensor * src1 = first->src[1];
    int cnt = 0;
    members[cnt++] = i;
    const int horizon = i + 12 < n ? i + 12 : n;
    for (int j = i + 1; j < horizon && cnt < MI355_MMVQ_MAX_GROUP; j++) {
        const ggml_tensor * cand = ggml_graph_node(cgraph, j);
        if (done[j] || !matvec_groupable(cand) || cand->src[1] != src1 || cand->src[0]->type != first->src[0]->type ||
            cand->src[0]->ne[0] != first->src[0]->ne[0] || tensor_is_split(cand->src[0]) != tensor_is_split(first->src[0])) continue;
        const size_t dn = ggml_nbytes(cand);
        bool safe = !ranges_overlap(cand->data, dn, src1->data, ggml_nbytes(src1));
        for (int m = 0; m < cnt && safe; m++) {
            const ggml_tensor * g = ggml_graph_node(cgraph, members[m]);
            if (ranges_overlap(cand->data, dn, g->data, ggml_nbytes(g))) safe = false;
        }
        for (int t = i + 1; t < j && safe; t++) {
            const ggml_tensor * mid = ggml_graph_node(cgraph, t);
            if (done[t] || ggml_is_empty(mid)) continue;
            bool member = false;
            for (int m = 0; m < cnt; m++) if (members[m] == t) member = true;
            if (member) continue;
            if (!is_view_op(mid)) {
                if (ranges_overlap(cand->data, dn, mid->data, ggml_nbytes(mid))) safe = false;              // would clobber / be clobbered
                if (ranges_overlap(src1->data, ggml_nbytes(src1), mid->data, ggml_nbytes(mid))) safe = false; // src1 rewritten before j
            }
            for (int sidx = 0; sidx < GGML_MAX_SRC && safe; sidx++) {
                const ggml_tensor * ms = mid->src[sidx];
                if (ms && ms->data && ranges_overlap(cand->data, dn, ms->data, ggml_nbytes(ms))) safe = false; // reads the old contents
            }
        }
        if (safe) members[cnt++] = j;
    }
    return cnt;
}

// One fused decode op over row-split weights: `has[d]` says which devices own rows, `launch(d, stream)` issues device d's slice.  The
// helpers start behind everything the main stream has issued so far (one event), the main stream goes on behind all of them (one
// event each).  Under capture the helper streams join the capture through the first wait and leave it through the last.
// A slice whose device is the SAME physical GPU as the main device (virtual devices) is issued on the main stream itself: a second
// stream on one GPU buys no parallel hardware and every fork / join edge costs ~5 us on the GPU and in hipGraphLaunch (2 virtual
// devices, 8B Q4_K_M tg64: 313 t/s through helper streams, 100 before the split forms; unsplit 510).  GGML_MI355_SPLIT_STREAMS=1
// keeps the helper streams there too -- the tests run both, so that the fork / join path stays exercised on the one-GPU box.
template <class F>
static void mi355_split_fan_out(mi355_backend_ctx * ctx, const bool * has, F launch) {
    const int ndev = mi355_device_count();
    static const bool force_streams = MI355_ENV_SET("GGML_MI355_SPLIT_STREAMS");
    bool helper[MI355_MAX_DEVICES] = {};
    int n_helpers = 0;
    for (int d = 0; d < ndev; d++) {
        helper[d] = d != ctx->device && has[d] && (force_streams || mi355_phys(d) != mi355_phys(ctx->device));
        n_helpers += helper[d];
    }
    if (n_helpers == 0) {
        for (int d = 0; d < ndev; d++) if (has[d]) launch(d, ctx->stream);
        return;
    }
    if (!ctx->ev_ready) HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_ready, hipEventDisableTiming));
    HIP_CHECK(hipEventRecord(ctx->ev_ready, ctx->stream));
    for (int d = 0; d < ndev; d++) {
        if (!helper[d]) continue;
        mi355_backend_ctx::peer_t & p = ctx->peer(d);
        mi355_set_device(d);
        HIP_CHECK(hipStreamWaitEvent(p.stream, ctx->ev_ready, 0));
        launch(d, p.stream);
        HIP_CHECK(hipEventRecord(p.done, p.stream));
    }
    mi355_set_device(ctx->device);
    for (int d = 0; d < ndev; d++) if (has[d] && !helper[d]) launch(d, ctx->stream);      // the main device's slice (and same-GPU ones)
    for (int d = 0; d < ndev; d++) {
        if (!helper[d]) continue;
        HIP_CHECK(hipStreamWaitEvent(ctx->stream, ctx->peers[d].done, 0));
    }
}

static const mi355_split_extra * split_extra_of(const ggml_tensor * w) { return (const mi355_split_extra *)w->extra; }

// norm_x / norm_w / norm_eps (one column only): src1 = (norm_x * rsqrt(mean(norm_x^2) + eps)) * norm_w is computed, stored and quantized in the
// launch's prologue (mi355_mmvq_args.src_f32) instead of being read from a launch of its own
static void mi355_mul_mat_group(mi355_backend_ctx * ctx, ggml_cgraph * cgraph, const int * members, int cnt,
                                const float * residual = nullptr, float * sum = nullptr,
                                const float * norm_x = nullptr, const float * norm_w = nullptr, float norm_eps = 0.0f) {
    const ggml_tensor * first = ggml_graph_node(cgraph, members[0]);
    const ggml_tensor * src1 = first->src[1];
    const int64_t K = src1->ne[0], N = src1->ne[1];
    const size_t qb = align_up((size_t)mi355_act_q_bytes(K, N), 256);
    const size_t db = align_up((size_t)mi355_act_ds_bytes(K, N), 256);
    char * ws = (char *)ctx->get_scratch(0, qb + db);
    if (!norm_x && !ctx->act_cached(src1)) {
        MI_CHECK(mi355_quantize_act((const float *)src1->data, (int8_t *)ws, (float *)(ws + qb), K, N, 1, 1, src1->nb[1], src1->nb[2], src1->nb[3], ctx->stream));
        ctx->act_remember(src1);
    }
    mi355_mmvq_args a[MI355_MMVQ_MAX_GROUP];
    memset(a, 0, sizeof(a));
    for (int m = 0; m < cnt; m++) {
        const ggml_tensor * node = ggml_graph_node(cgraph, members[m]);
        const ggml_tensor * w = node->src[0];
        a[m].type = (int32_t)w->type;
        a[m].W = w->data; a[m].w_row_stride = (int64_t)w->nb[1];
        a[m].act_q = (const int8_t *)ws; a[m].act_ds = (const float *)(ws + qb); a[m].act_ch_stride = N;
        a[m].dst = (float *)node->data; a[m].dst_col_stride = (int64_t)(node->nb[1] / sizeof(float));
        a[m].m = w->ne[1]; a[m].k = K; a[m].ncols = (int32_t)N; a[m].nchannels = 1; a[m].ch_ratio = 1;
        if (norm_x) {
            a[m].act_q = nullptr; a[m].act_ds = nullptr;
            a[m].src_f32 = norm_x; a[m].norm_w = norm_w; a[m].norm_out = (float *)src1->data; a[m].norm_eps = norm_eps;
        }
    }
    a[0].residual = residual; a[0].sum = sum;      // mat-vec + ADD (cnt == 1)
    if (tensor_is_split(first->src[0])) {
        // every device launches the group over its row slices: the members it owns rows of, results (and the residual sum) at the
        // slice's place in the main device's tensors
        bool has[MI355_MAX_DEVICES] = {};
        const int ndev = mi355_device_count();
        for (int m = 0; m < cnt; m++) {
            const mi355_split_extra * ex = split_extra_of(ggml_graph_node(cgraph, members[m])->src[0]);
            for (int d = 0; d < ndev; d++) if (ex->data[d] && ex->row_high[d] > ex->row_low[d]) has[d] = true;
        }
        mi355_split_fan_out(ctx, has, [&](int d, hipStream_t st) {
            mi355_mmvq_args b[MI355_MMVQ_MAX_GROUP];
            int nb = 0;
            for (int m = 0; m < cnt; m++) {
                const mi355_split_extra * ex = split_extra_of(ggml_graph_node(cgraph, members[m])->src[0]);
                const int64_t lo = ex->row_low[d], rows = ex->row_high[d] - lo;
                if (!ex->data[d] || rows <= 0) continue;
                b[nb] = a[m];
                b[nb].W = ex->data[d]; b[nb].m = rows; b[nb].dst = a[m].dst + lo;
                if (a[m].residual) { b[nb].residual = a[m].residual + lo; b[nb].sum = a[m].sum + lo; }
                nb++;
            }
            if (nb) MI_CHECK(mi355_mul_mat_vec_q_group(b, nb, st));
        });
        return;
    }
    MI_CHECK(mi355_mul_mat_vec_q_group(a, cnt, ctx->stream));
}

// Decode mat-vec followed by the residual ADD (wo / ffn_down of every layer: cur = MUL_MAT(w, x); out = ADD(cur, inp)) in
// one launch: the lane that stores a result element also stores result + residual.  Per-element, so the ADD may be in
// place on either operand; both tensors are written unless the sum takes the result's place.  With the ADD gone from
// it, the norm that follows needs no residual input -- which is what lets it move into the next mat-vec's prologue
// (mi355_try_fuse_norm_glu_matvec): there every workgroup reads the row, so nothing may update it in that launch.
static bool mi355_try_fuse_matvec_add(mi355_backend_ctx * ctx, ggml_cgraph * cgraph, int i) {
    const int n = ggml_graph_n_nodes(cgraph);
    if (i + 1 >= n || MI355_ENV_SET("GGML_MI355_NO_MATVEC_ADD")) return false;
    ggml_tensor * mm = ggml_graph_node(cgraph, i), * add = ggml_graph_node(cgraph, i + 1);
    if (add->op != GGML_OP_ADD || add->type != GGML_TYPE_F32) return false;
    const ggml_tensor * r = add->src[0] == mm ? add->src[1] : (add->src[1] == mm ? add->src[0] : nullptr);
    if (!r || r == mm || r->type != GGML_TYPE_F32) return false;
    if (!ggml_are_same_shape(mm, add) || !ggml_are_same_shape(r, add) || !ggml_is_contiguous(add) || !ggml_is_contiguous(r)) return false;
    const size_t nb = ggml_nbytes(add);
    if (ranges_overlap(r->data, nb, mm->data, nb)) return false;
    if (add->data != mm->data && ranges_overlap(add->data, nb, mm->data, nb)) return false;
    if (add->data != r->data && ranges_overlap(add->data, nb, r->data, nb)) return false;
    // neither output may land on the activations this mat-vec is about to be given
    const ggml_tensor * src1 = mm->src[1];
    if (ranges_overlap(add->data, nb, src1->data, ggml_nbytes(src1))) return false;
    const int members[1] = {i};
    mi355_mul_mat_group(ctx, cgraph, members, 1, (const float *)r->data, (float *)add->data);
    ctx->act_invalidate_if_written(add);
    return true;
}

// ---- norm prologue of the fused decode kernels (mi355_glu_matvec / mi355_qkv_rope_matvec with norm_x set) ----------
// RMS_NORM at node i followed by MUL with the per-feature norm weight at i+1, one token, nothing else reading the bare
// norm (the checks of mi355_try_fuse_rms_norm_mul): returns the weight tensor, or NULL.
static const ggml_tensor * decode_norm_pattern(ggml_cgraph * cgraph, int i, const std::vector<char> & done) {
    const int n = ggml_graph_n_nodes(cgraph);
    if (i + 1 >= n || done[i] || done[i + 1]) return nullptr;
    const ggml_tensor * norm = ggml_graph_node(cgraph, i), * mul = ggml_graph_node(cgraph, i + 1);
    if (norm->op != GGML_OP_RMS_NORM || mul->op != GGML_OP_MUL) return nullptr;
    const ggml_tensor * w = mul->src[0] == norm ? mul->src[1] : (mul->src[1] == norm ? mul->src[0] : nullptr);
    if (!w || w == norm) return nullptr;
    if (w->type != GGML_TYPE_F32 || mul->type != GGML_TYPE_F32 || norm->type != GGML_TYPE_F32 || norm->src[0]->type != GGML_TYPE_F32) return nullptr;
    if (w->ne[0] != norm->ne[0] || ggml_nrows(w) != 1 || w->nb[0] != sizeof(float)) return nullptr;
    if (!ggml_are_same_shape(norm, mul) || ggml_nrows(mul) != 1 || !ggml_is_contiguous(mul) || (norm->flags & GGML_TENSOR_FLAG_OUTPUT)) return nullptr;
    if (norm->src[0]->nb[0] != sizeof(float)) return nullptr;
    for (int j = i + 2; j < n; j++) {           // any other reader of the un-multiplied norm?
        const ggml_tensor * t = ggml_graph_node(cgraph, j);
        for (int s2 = 0; s2 < GGML_MAX_SRC; s2++) if (t->src[s2] == norm || (t->src[s2] && t->src[s2]->view_src == norm)) return nullptr;
    }
    return w;
}

// In such a launch every workgroup reads the norm input x and the weight w while workgroup 0 stores the MUL result and
// all of them store their share of the other outputs: x, w and the outputs must be pairwise apart (outs[0] is the MUL
// result), K within the prologue's reach and the three rows 16-byte aligned.
static bool norm_prologue_ok(const ggml_tensor * x, const ggml_tensor * w, int64_t K, int64_t max_k, const ggml_tensor * const * outs,
                             const size_t * out_bytes, int n_outs) {
    if (K > max_k || K % 32 != 0) return false;
    if (((uintptr_t)x->data | (uintptr_t)w->data | (uintptr_t)outs[0]->data) & 15) return false;
    const size_t xb = (size_t)K * sizeof(float);
    for (int o = 0; o < n_outs; o++) {
        if (ranges_overlap(outs[o]->data, out_bytes[o], x->data, xb) || ranges_overlap(outs[o]->data, out_bytes[o], w->data, xb)) return false;
        for (int q = 0; q < o; q++) if (ranges_overlap(outs[o]->data, out_bytes[o], outs[q]->data, out_bytes[q])) return false;
    }
    return true;
}

// Feed-forward input of a decode step in one launch.  build_ffn (src/llama-graph.cpp:470-640) spells it as
//     tmp = MUL_MAT(ffn_up, x)   cur = MUL_MAT(ffn_gate, x)   cur = GLU(cur, tmp)            (split SWIGLU / GEGLU / REGLU)
// When the two mat-vecs form a group (members[0..1], consecutive nodes, one token) and the GLU is the very next node,
// mi355_glu_matvec computes all three tensors (bit-identical values) plus the q8_1 form of the GLU result, which
// ffn_down then finds in the activation cache.  Every tensor of the chain is still written, so no use analysis is
// needed -- except that a GLU result placed exactly on top of one of its inputs (ggml-alloc's in-place reuse) simply
// replaces it.  Grids beyond one round of workgroups (m > 16384) stay on the grouped mat-vec, which is faster there.
static bool mi355_try_fuse_glu_matvec(mi355_backend_ctx * ctx, ggml_cgraph * cgraph, const int * members, int norm_idx = -1) {
    const int n = ggml_graph_n_nodes(cgraph);
    const int i0 = members[0], i1 = members[1];
    if (i1 != i0 + 1 || i1 + 1 >= n || MI355_ENV_SET("GGML_MI355_NO_GLU_MATVEC")) return false;
    ggml_tensor * glu = ggml_graph_node(cgraph, i1 + 1);
    const ggml_tensor * m0 = ggml_graph_node(cgraph, i0), * m1 = ggml_graph_node(cgraph, i1);
    if (glu->op != GGML_OP_GLU || !glu->src[1] || glu->type != GGML_TYPE_F32) return false;
    const ggml_tensor * gate = glu->src[0], * up = glu->src[1];
    if (!((gate == m0 && up == m1) || (gate == m1 && up == m0))) return false;
    const int op = (int)ggml_get_glu_op(glu);
    if (op < 0 || op > 2) return false;
    const ggml_tensor * src1 = m0->src[1];
    const int64_t K = src1->ne[0], M = gate->ne[0];
    if (src1->ne[1] != 1 || up->ne[0] != M || M % 32 != 0 || M > 16384 || ggml_nrows(glu) != 1 || glu->ne[0] != M) return false;
    if (!ggml_is_contiguous(glu) || ((uintptr_t)glu->data & 15)) return false;
    if ((size_t)K + (size_t)K / 32 * 8 + 16 + 256 > 64 * 1024) return false;
    // the GLU result may coincide with one of its inputs (then that input is not stored), but not straddle them,
    // nor touch the activations the mat-vecs were given
    float * gate_out = (float *)gate->data, * up_out = (float *)up->data;
    if (glu->data == gate->data) gate_out = nullptr;
    else if (ranges_overlap(glu->data, ggml_nbytes(glu), gate->data, ggml_nbytes(gate))) return false;
    if (glu->data == up->data) up_out = nullptr;
    else if (ranges_overlap(glu->data, ggml_nbytes(glu), up->data, ggml_nbytes(up))) return false;
    // The GLU result on the bytes of src1 (ggml-alloc does that once the two mat-vecs, src1's last readers, are behind it -- seen with
    // row-split weights): the kernel never reads src1's f32 form (the activations arrive quantized, or are normalised from the norm's
    // input in every workgroup), so the only conflict is the prologue's store of the norm result, which then goes to a row nobody reads.
    const bool glu_on_src1 = ranges_overlap(glu->data, ggml_nbytes(glu), src1->data, ggml_nbytes(src1));

    const size_t qb = align_up((size_t)mi355_act_q_bytes(K, 1), 256), db = align_up((size_t)mi355_act_ds_bytes(K, 1), 256);
    const size_t oqb = align_up((size_t)mi355_act_q_bytes(M, 1), 256), odb = align_up((size_t)mi355_act_ds_bytes(M, 1), 256);
    mi355_glu_matvec_args a;
    memset(&a, 0, sizeof(a));
    char * ws = nullptr, * os;
    if (norm_idx >= 0) {
        // RMS_NORM (norm_idx) -> MUL by the norm weight (norm_idx + 1 == src1) in the prologue: every workgroup reads the
        // norm input, workgroup 0 stores src1 -- so src1, the norm input and the weight must be pairwise apart, and none
        // of the launch's other outputs may touch the two inputs
        const ggml_tensor * norm = ggml_graph_node(cgraph, norm_idx);
        const ggml_tensor * x = norm->src[0];
        const ggml_tensor * w = src1->src[0] == norm ? src1->src[1] : src1->src[0];
        // outputs that coincide (a GLU result in place on gate / up) were already reduced to one above
        ggml_tensor norm_dst = *src1;                      // where the prologue stores the norm result
        if (glu_on_src1) {
            norm_dst.data = ctx->get_scratch(6, (size_t)K * sizeof(float));
            if (ctx->scratch_size[6] < (size_t)K * sizeof(float)) return false;
        }
        const ggml_tensor * outs[4] = {&norm_dst, glu, nullptr, nullptr};
        size_t ob[4] = {ggml_nbytes(src1), ggml_nbytes(glu), 0, 0};
        int no = 2;
        if (gate_out) { outs[no] = gate; ob[no++] = ggml_nbytes(gate); }
        if (up_out)   { outs[no] = up;   ob[no++] = ggml_nbytes(up); }
        if (!norm_prologue_ok(x, w, K, 8192, outs, ob, no)) return false;
        a.norm_x = (const float *)x->data; a.norm_w = (const float *)w->data; a.norm_out = (float *)norm_dst.data;
        memcpy(&a.norm_eps, norm->op_params, sizeof(float));
        os = (char *)ctx->get_scratch(0, oqb + odb);      // scratch[0] is not read by this launch: the result goes straight in
    } else {
        ws = (char *)ctx->get_scratch(0, qb + db);
        if (!ctx->act_cached(src1)) {
            MI_CHECK(mi355_quantize_act((const float *)src1->data, (int8_t *)ws, (float *)(ws + qb), K, 1, 1, 1, src1->nb[1], src1->nb[2], src1->nb[3], ctx->stream));
            ctx->act_remember(src1);
        }
        os = (char *)ctx->get_scratch(2, oqb + odb);
    }
    a.type = (int32_t)gate->src[0]->type; a.glu_op = op;
    a.w_gate = gate->src[0]->data; a.w_up = up->src[0]->data;
    a.gate_row_stride = (int64_t)gate->src[0]->nb[1]; a.up_row_stride = (int64_t)up->src[0]->nb[1];
    a.m = M; a.k = K;
    if (ws) { a.act_q = (const int8_t *)ws; a.act_ds = (const float *)(ws + qb); }
    a.gate_out = gate_out; a.up_out = up_out; a.dst = (float *)glu->data;
    a.out_q = (int8_t *)os; a.out_ds = (float *)(os + oqb);
    if (tensor_is_split(gate->src[0])) {
        // row-split gate / up: both matrices are cut at the same rows (same shape, same split), a device computes rows [lo, hi) of
        // gate, up, the GLU result and its q8_1 form (128-row rounding: whole q8_1 blocks)
        const mi355_split_extra * eg = split_extra_of(gate->src[0]), * eu = split_extra_of(up->src[0]);
        if (!tensor_is_split(up->src[0])) return false;
        bool has[MI355_MAX_DEVICES] = {};
        const int ndev = mi355_device_count();
        for (int d = 0; d < ndev; d++) {
            if (eg->row_low[d] != eu->row_low[d] || eg->row_high[d] != eu->row_high[d] || (eg->row_low[d] % 32)) return false;
            const int64_t rows = eg->row_high[d] - eg->row_low[d];
            has[d] = rows > 0 && eg->data[d] && eu->data[d];
            if (has[d] && rows % 32) return false;
        }
        mi355_split_fan_out(ctx, has, [&](int d, hipStream_t st) {
            mi355_glu_matvec_args b = a;
            const int64_t lo = eg->row_low[d];
            b.w_gate = eg->data[d]; b.w_up = eu->data[d]; b.m = eg->row_high[d] - lo;
            if (a.gate_out) b.gate_out = a.gate_out + lo;
            if (a.up_out)   b.up_out   = a.up_out + lo;
            b.dst = a.dst + lo; b.out_q = a.out_q + lo; b.out_ds = a.out_ds + lo / 32 * 2;
            MI_CHECK(mi355_glu_matvec(&b, st));
        });
    } else {
        const int rc = mi355_glu_matvec(&a, ctx->stream);
        if (rc == MI355_E_UNSUPPORTED || rc == MI355_E_ALIGN) return false;
        MI_CHECK(rc);
    }
    if (norm_idx < 0) ctx->swap_act_scratch();            // scratch[0] now holds the q8_1 form of the GLU result
    ctx->act_remember(glu);
    return true;
}

// bf16 rows of a prefill MUL_MAT's activations in scratch[0] (converted unless the producer already left them there)
static uint16_t * prefill_act_bf16(mi355_backend_ctx * ctx, const ggml_tensor * src1) {
    const int64_t K = src1->ne[0], N = src1->ne[1];
    uint16_t * act = (uint16_t *)ctx->get_scratch(0, align_up((size_t)N * K * 2, 256));
    if (!ctx->act_cached(src1, 1)) {
        MI_CHECK(mi355_act_bf16((const float *)src1->data, act, K, N, 1, 1, src1->nb[1], src1->nb[2], src1->nb[3], ctx->stream));
        ctx->act_remember(src1, 1);
    }
    return act;
}

static bool prefill_mm_ok(const ggml_tensor * t);

// MUL_MAT (wo, ffn_down) -> ADD (residual) -> RMS_NORM -> MUL (norm weight) at nodes i .. i + 3 of a prefill batch: the GEMM leaves its
// K-slice planes and the norm launch adds them (see mi355_try_fuse_rms_norm_mul).  The product itself is never written, so the ADD must
// be its only reader.
static bool mi355_try_fuse_gemm_add_norm(mi355_backend_ctx * ctx, ggml_cgraph * cgraph, int i, const std::vector<char> & done) {
    const int n = ggml_graph_n_nodes(cgraph);
    if (i + 3 >= n || done[i + 1] || done[i + 2] || done[i + 3] || MI355_ENV_SET("GGML_MI355_NO_GEMM_NORM")) return false;
    const ggml_tensor * mm = ggml_graph_node(cgraph, i);
    ggml_tensor * add = ggml_graph_node(cgraph, i + 1);
    const ggml_tensor * norm = ggml_graph_node(cgraph, i + 2);
    if (!prefill_mm_ok(mm) || mm->src[1]->ne[1] <= MI355_BF16_ROWS_ABOVE || mm->src[1]->ne[2] != 1 || mm->src[1]->ne[3] != 1) return false;
    if (add->op != GGML_OP_ADD || norm->op != GGML_OP_RMS_NORM || norm->src[0] != add || ggml_graph_node(cgraph, i + 3)->op != GGML_OP_MUL) return false;
    if ((add->src[0] == mm) == (add->src[1] == mm)) return false;             // exactly one addend is the product
    if (!ggml_is_contiguous(mm) || (mm->flags & GGML_TENSOR_FLAG_OUTPUT)) return false;
    for (int j = 0; j < n; j++) {
        if (j == i + 1) continue;
        const ggml_tensor * t = ggml_graph_node(cgraph, j);
        for (int s = 0; s < GGML_MAX_SRC; s++) if (t->src[s] && (t->src[s] == mm || t->src[s]->view_src == mm)) return false;
    }
    return mi355_try_fuse_rms_norm_mul(ctx, cgraph, i + 2, add, mm);
}

// Feed-forward input of a prefill batch in one launch: MUL_MAT(ffn_gate) MUL_MAT(ffn_up) GLU at nodes i, i + 1, i + 2 (either order of
// the two products) -> mi355_glu_mul_mat_dq, whose tiles hold 64 rows of both matrices.  The two f32 products are NOT written, so they
// must have no other reader in this graph and must not be graph outputs (a scheduler eval callback that wants to see them gets them
// as one-node graphs, where this pattern cannot match).  The kernel library offers the fused form only where it has the bits of the
// three launches; everywhere else it answers MI355_E_UNSUPPORTED and the nodes run one by one.
static bool mi355_try_fuse_glu_gemm(mi355_backend_ctx * ctx, ggml_cgraph * cgraph, int i, const std::vector<char> & done) {
    const int n = ggml_graph_n_nodes(cgraph);
    if (i + 2 >= n || done[i + 1] || done[i + 2] || MI355_ENV_SET("GGML_MI355_NO_GLU_GEMM") || MI355_ENV_SET("GGML_MI355_MMQ_INT8")) return false;
    const ggml_tensor * m0 = ggml_graph_node(cgraph, i), * m1 = ggml_graph_node(cgraph, i + 1);
    ggml_tensor * glu = ggml_graph_node(cgraph, i + 2);
    if (m1->op != GGML_OP_MUL_MAT || glu->op != GGML_OP_GLU || !glu->src[1] || glu->type != GGML_TYPE_F32) return false;
    const ggml_tensor * ga = glu->src[0], * gb = glu->src[1];       // act(ga) * gb; `swapped` only applies to the single-tensor form
    if (!((ga == m0 && gb == m1) || (ga == m1 && gb == m0))) return false;
    const int op = (int)ggml_get_glu_op(glu);
    if (op < 0 || op > 2) return false;
    const ggml_tensor * src1 = m0->src[1], * wa = ga->src[0], * wb = gb->src[0];
    if (m1->src[1] != src1 || src1->type != GGML_TYPE_F32 || src1->ne[2] != 1 || src1->ne[3] != 1) return false;
    if (wa->type != wb->type || !ggml_are_same_shape(wa, wb) || wa->ne[2] != 1 || wa->ne[3] != 1 || tensor_is_split(wa) || tensor_is_split(wb)) return false;
    if (mi355_mul_mat_path(wa, src1) != MM_DQ || mi355_mul_mat_path(wb, src1) != MM_DQ) return false;
    const int64_t K = wa->ne[0], M = wa->ne[1], N = src1->ne[1];
    if (glu->ne[0] != M || ggml_nrows(glu) != N || !ggml_is_contiguous(glu) || ((uintptr_t)glu->data & 3)) return false;
    if ((m0->flags | m1->flags) & GGML_TENSOR_FLAG_OUTPUT) return false;
    for (int j = 0; j < n; j++) {                 // any other reader of the two products?
        if (j == i + 2) continue;
        const ggml_tensor * t = ggml_graph_node(cgraph, j);
        for (int s = 0; s < GGML_MAX_SRC; s++) {
            const ggml_tensor * r = t->src[s];
            if (r && (r == m0 || r == m1 || r->view_src == m0 || r->view_src == m1)) return false;
        }
    }
    if (ranges_overlap(glu->data, ggml_nbytes(glu), wa->data, ggml_nbytes(wa)) || ranges_overlap(glu->data, ggml_nbytes(glu), wb->data, ggml_nbytes(wb))) return false;

    const uint16_t * act = prefill_act_bf16(ctx, src1);
    uint16_t * hbf = (uint16_t *)ctx->get_scratch(2, align_up((size_t)N * M * 2, 256));
    const int rc = mi355_glu_mul_mat_dq((int)wa->type, op, wa->data, (int64_t)wa->nb[1], wb->data, (int64_t)wb->nb[1], act, K,
                                        (float *)glu->data, M, hbf, M, K, N, ctx->stream);
    if (rc == MI355_E_UNSUPPORTED || rc == MI355_E_ALIGN) return false;    // the activations stay cached for the separate launches
    MI_CHECK(rc);
    ctx->swap_act_scratch();                      // scratch[0] now holds the bf16 rows of the GLU result, for ffn_down
    ctx->act_remember(glu, 1);
    return true;
}

// RMS_NORM -> MUL(norm weight) -> MUL_MAT(ffn_up) MUL_MAT(ffn_gate) -> GLU of a decode token: five nodes, one launch
// (the norm runs in the prologue of mi355_glu_matvec).  The norm checks are those of mi355_try_fuse_rms_norm_mul.
static int mi355_collect_matvec_group(ggml_cgraph * cgraph, int i, const std::vector<char> & done, int * members);
static bool matvec_groupable(const ggml_tensor * node);
static bool mi355_try_fuse_norm_glu_matvec(mi355_backend_ctx * ctx, ggml_cgraph * cgraph, int i, const std::vector<char> & done) {
    const int n = ggml_graph_n_nodes(cgraph);
    if (i + 4 >= n || MI355_ENV_SET("GGML_MI355_NO_NORM_GLU")) return false;
    for (int j = i; j <= i + 4; j++) if (done[j]) return false;
    if (!decode_norm_pattern(cgraph, i, done)) return false;
    const ggml_tensor * mul = ggml_graph_node(cgraph, i + 1);
    const ggml_tensor * m0 = ggml_graph_node(cgraph, i + 2);
    if (!matvec_groupable(m0) || m0->src[1] != mul) return false;
    int members[MI355_MMVQ_MAX_GROUP];
    if (mi355_collect_matvec_group(cgraph, i + 2, done, members) != 2 || members[1] != i + 3) return false;
    return mi355_try_fuse_glu_matvec(ctx, cgraph, members, i);
}

// RMS_NORM -> MUL(weight) -> the group of decode mat-vecs that read it (ffn gate + up where the one-launch GLU form does not apply:
// Llama-3-70B's 28672 rows), the norm in the grouped launch's prologue.  *consumed = nodes taken after the norm (the MUL and the mat-vecs).
static bool mi355_try_fuse_norm_group(mi355_backend_ctx * ctx, ggml_cgraph * cgraph, int i, const std::vector<char> & done, int * consumed) {
    const int n = ggml_graph_n_nodes(cgraph);
    if (i + 2 >= n || MI355_ENV_SET("GGML_MI355_NO_NORM_GROUP")) return false;
    if (!decode_norm_pattern(cgraph, i, done)) return false;
    const ggml_tensor * norm = ggml_graph_node(cgraph, i);
    const ggml_tensor * mul = ggml_graph_node(cgraph, i + 1);
    const ggml_tensor * m0 = ggml_graph_node(cgraph, i + 2);
    if (done[i + 2] || !matvec_groupable(m0) || m0->src[1] != mul || mul->ne[1] != 1) return false;
    int members[MI355_MMVQ_MAX_GROUP];
    const int cnt = mi355_collect_matvec_group(cgraph, i + 2, done, members);
    if (cnt < 2) return false;                       // a single mat-vec: the fused norm + quantize launch and the residual forms serve it
    for (int m = 0; m < cnt; m++) if (members[m] != i + 2 + m) return false;
    const ggml_tensor * x = norm->src[0];
    const ggml_tensor * w = mul->src[0] == norm ? mul->src[1] : mul->src[0];
    const int64_t K = mul->ne[0];
    const ggml_tensor * outs[1 + MI355_MMVQ_MAX_GROUP] = {mul};
    size_t ob[1 + MI355_MMVQ_MAX_GROUP] = {ggml_nbytes(mul)};
    for (int m = 0; m < cnt; m++) { outs[1 + m] = ggml_graph_node(cgraph, members[m]); ob[1 + m] = ggml_nbytes(outs[1 + m]); }
    if (!norm_prologue_ok(x, w, K, 16384, outs, ob, 1 + cnt)) return false;
    float eps;
    memcpy(&eps, norm->op_params, sizeof(float));
    mi355_mul_mat_group(ctx, cgraph, members, cnt, nullptr, nullptr, (const float *)x->data, (const float *)w->data, eps);
    ctx->act_invalidate_if_written(mul);
    *consumed = 1 + cnt;
    return true;
}

// Expert feed-forward input of one decode token in one launch (build_moe_ffn, src/llama-graph.cpp:660-760):
//     up = MUL_MAT_ID(ffn_up_exps, x, ids)   gate = MUL_MAT_ID(ffn_gate_exps, x, ids)   cur = GLU(gate, up)
// three consecutive nodes, one token, x broadcast over the slots -> mi355_glu_matvec with ids: all three tensors are
// written (bit-identical values) plus the q8_1 form of the GLU result, which MUL_MAT_ID(ffn_down_exps) finds in the
// activation cache.  A GLU result exactly on top of one of its inputs replaces it.
static bool mi355_try_fuse_moe_glu_matvec(mi355_backend_ctx * ctx, ggml_cgraph * cgraph, int i, const std::vector<char> & done) {
    const int n = ggml_graph_n_nodes(cgraph);
    if (i + 2 >= n || done[i + 1] || done[i + 2] || MI355_ENV_SET("GGML_MI355_NO_GLU_MATVEC")) return false;
    const ggml_tensor * m0 = ggml_graph_node(cgraph, i), * m1 = ggml_graph_node(cgraph, i + 1);
    ggml_tensor * glu = ggml_graph_node(cgraph, i + 2);
    if (m1->op != GGML_OP_MUL_MAT_ID || glu->op != GGML_OP_GLU || !glu->src[1] || glu->type != GGML_TYPE_F32) return false;
    const ggml_tensor * gate = glu->src[0], * up = glu->src[1];
    if (!((gate == m0 && up == m1) || (gate == m1 && up == m0))) return false;
    const int op = (int)ggml_get_glu_op(glu);
    if (op < 0 || op > 2) return false;
    const ggml_tensor * b = m0->src[1], * ids = m0->src[2];
    if (m1->src[1] != b || m1->src[2] != ids) return false;
    const ggml_tensor * wg = gate->src[0], * wu = up->src[0];
    if (!is_weight_type(wg->type) || wg->type != wu->type || tensor_is_split(wg) || tensor_is_split(wu)) return false;
    const int64_t K = wg->ne[0], M = wg->ne[1], n_expert = wg->ne[2], n_used = ids->ne[0];
    if (wu->ne[0] != K || wu->ne[1] != M || wu->ne[2] != n_expert || wg->ne[3] != 1 || wu->ne[3] != 1) return false;
    if (b->ne[1] != 1 || b->ne[2] != 1 || b->ne[3] != 1 || ids->ne[1] != 1 || ids->type != GGML_TYPE_I32 || ids->nb[0] != sizeof(int32_t)) return false;
    if (n_used < 1 || n_used > 64 || M % 32 != 0 || K % ggml_blck_size(wg->type) != 0 || b->type != GGML_TYPE_F32 || b->nb[0] != sizeof(float)) return false;
    if ((size_t)K + (size_t)K / 32 * 8 + 16 + 256 > 64 * 1024) return false;
    const ggml_tensor * outs[3] = {gate, up, glu};
    for (const ggml_tensor * o : outs) {
        if (o->type != GGML_TYPE_F32 || o->ne[0] != M || o->ne[1] != n_used || ggml_nrows(o) != n_used || !ggml_is_contiguous(o)) return false;
        if (ranges_overlap(o->data, ggml_nbytes(o), b->data, ggml_nbytes(b)) || ranges_overlap(o->data, ggml_nbytes(o), ids->data, ggml_nbytes(ids))) return false;
    }
    if (ranges_overlap(gate->data, ggml_nbytes(gate), up->data, ggml_nbytes(up)) || ((uintptr_t)glu->data & 15)) return false;
    float * gate_out = (float *)gate->data, * up_out = (float *)up->data;
    if (glu->data == gate->data) gate_out = nullptr;
    else if (ranges_overlap(glu->data, ggml_nbytes(glu), gate->data, ggml_nbytes(gate))) return false;
    if (glu->data == up->data) up_out = nullptr;
    else if (ranges_overlap(glu->data, ggml_nbytes(glu), up->data, ggml_nbytes(up))) return false;

    const size_t qb = align_up((size_t)mi355_act_q_bytes(K, 1), 256), db = align_up((size_t)mi355_act_ds_bytes(K, 1), 256);
    const size_t oqb = align_up((size_t)mi355_act_q_bytes(M, n_used), 256), odb = align_up((size_t)mi355_act_ds_bytes(M, n_used), 256);
    char * ws = (char *)ctx->get_scratch(0, qb + db);
    if (!ctx->act_cached(b)) {
        MI_CHECK(mi355_quantize_act((const float *)b->data, (int8_t *)ws, (float *)(ws + qb), K, 1, 1, 1, b->nb[1], b->nb[2], b->nb[3], ctx->stream));
        ctx->act_remember(b);
    }
    char * os = (char *)ctx->get_scratch(2, oqb + odb);
    mi355_glu_matvec_args a;
    memset(&a, 0, sizeof(a));
    a.type = (int32_t)wg->type; a.glu_op = op;
    a.w_gate = wg->data; a.w_up = wu->data; a.gate_row_stride = (int64_t)wg->nb[1]; a.up_row_stride = (int64_t)wu->nb[1];
    a.m = M; a.k = K; a.act_q = (const int8_t *)ws; a.act_ds = (const float *)(ws + qb);
    a.gate_out = gate_out; a.up_out = up_out; a.dst = (float *)glu->data;
    a.out_q = (int8_t *)os; a.out_ds = (float *)(os + oqb);
    a.ids = (const int32_t *)ids->data; a.n_slots = (int32_t)n_used; a.n_expert = (int32_t)n_expert;
    a.gate_expert_stride = (int64_t)wg->nb[2]; a.up_expert_stride = (int64_t)wu->nb[2];
    const int rc = mi355_glu_matvec(&a, ctx->stream);
    if (rc == MI355_E_UNSUPPORTED || rc == MI355_E_ALIGN) return false;
    MI_CHECK(rc);
    ctx->swap_act_scratch();            // scratch[0] now holds the q8_1 form of the GLU result ([n_used][M], the layout of a [M, n_used] src1)
    ctx->act_remember(glu);
    return true;
}

// Expert feed-forward output of one decode token in one launch (build_moe_ffn, src/llama-graph.cpp:760-795):
//     y = MUL_MAT_ID(ffn_down_exps, h, ids)  [router nodes, views]  y = MUL(y, weights)  out = ADD(view(y, 0), view(y, 1))
//     [l_out = ADD(out, residual)]
// for two used experts (Mixtral).  The nodes after the MUL_MAT_ID must be the next ones still to run (the router's
// GET_ROWS / SUM_ROWS / DIV that llama.cpp places in between were already computed by mi355_try_fuse_moe_router, so the
// weights exist); every tensor of the chain is written with bit-identical values.  Outputs may coincide exactly where
// the graph computes in place (MUL on y, l_out on out or on the residual) and must be apart otherwise.
static int next_active_node(ggml_cgraph * cgraph, int j, const std::vector<char> & done) {
    const int n = ggml_graph_n_nodes(cgraph);
    for (j++; j < n; j++) {
        const ggml_tensor * t = ggml_graph_node(cgraph, j);
        if (done[j] || ggml_is_empty(t) || is_view_op(t)) continue;
        return j;
    }
    return -1;
}

static bool same_or_apart(const ggml_tensor * a, const ggml_tensor * b) {
    return a->data == b->data ? ggml_nbytes(a) == ggml_nbytes(b) : !ranges_overlap(a->data, ggml_nbytes(a), b->data, ggml_nbytes(b));
}

static bool mi355_try_fuse_moe_down(mi355_backend_ctx * ctx, ggml_cgraph * cgraph, int i, std::vector<char> & done) {
    if (MI355_ENV_SET("GGML_MI355_NO_MOE_DOWN")) return false;
    const ggml_tensor * down = ggml_graph_node(cgraph, i);
    const ggml_tensor * as = down->src[0], * b = down->src[1], * ids = down->src[2];
    const int64_t K = as->ne[0], M = as->ne[1], n_expert = as->ne[2], n_used = ids->ne[0];
    if (n_used != 2 || !is_weight_type(as->type) || tensor_is_split(as) || as->ne[3] != 1) return false;
    if (ids->ne[1] != 1 || ids->type != GGML_TYPE_I32 || ids->nb[0] != sizeof(int32_t)) return false;
    if (b->type != GGML_TYPE_F32 || b->ne[0] != K || b->ne[1] != n_used || b->ne[2] != 1 || b->ne[3] != 1 || !ggml_is_contiguous(b)) return false;
    if (down->type != GGML_TYPE_F32 || down->ne[0] != M || down->ne[1] != n_used || ggml_nrows(down) != n_used || !ggml_is_contiguous(down)) return false;
    if (K % ggml_blck_size(as->type) != 0 || K % 32 != 0) return false;
    const int jm = next_active_node(cgraph, i, done);
    if (jm < 0) return false;
    const ggml_tensor * mul = ggml_graph_node(cgraph, jm);
    if (mul->op != GGML_OP_MUL || mul->src[0] != down || mul->type != GGML_TYPE_F32 || !ggml_are_same_shape(mul, down) || !ggml_is_contiguous(mul)) return false;
    const ggml_tensor * w = mul->src[1];
    if (w->type != GGML_TYPE_F32 || w->ne[0] != 1 || w->ne[1] != n_used || ggml_nelements(w) != n_used || w->nb[1] != sizeof(float)) return false;
    const int ja = next_active_node(cgraph, jm, done);
    if (ja < 0) return false;
    const ggml_tensor * add = ggml_graph_node(cgraph, ja);
    if (add->op != GGML_OP_ADD || add->type != GGML_TYPE_F32 || add->ne[0] != M || ggml_nrows(add) != 1 || !ggml_is_contiguous(add)) return false;
    for (int s2 = 0; s2 < 2; s2++) {
        const ggml_tensor * v = add->src[s2];
        if (!v || view_root(v) != mul || v->type != GGML_TYPE_F32 || v->ne[0] != M || ggml_nrows(v) != 1 || v->nb[0] != sizeof(float)) return false;
        if (v->data != (const char *)mul->data + s2 * mul->nb[1]) return false;
    }
    // optional residual ADD right behind
    const ggml_tensor * radd = nullptr, * r = nullptr;
    const int jr = next_active_node(cgraph, ja, done);
    if (jr >= 0) {
        const ggml_tensor * t = ggml_graph_node(cgraph, jr);
        if (t->op == GGML_OP_ADD && t->type == GGML_TYPE_F32 && (t->src[0] == add || t->src[1] == add)) {
            const ggml_tensor * o = t->src[0] == add ? t->src[1] : t->src[0];
            if (o != add && o->type == GGML_TYPE_F32 && ggml_are_same_shape(o, add) && ggml_are_same_shape(t, add) && ggml_is_contiguous(o) &&
                ggml_is_contiguous(t)) { radd = t; r = o; }
        }
    }
    // placement
    if (!same_or_apart(mul, down)) return false;
    const ggml_tensor * big[2] = {down, mul};
    const ggml_tensor * small[3] = {add, radd, r};
    for (const ggml_tensor * sm : small) {
        if (!sm) continue;
        for (const ggml_tensor * bg : big) if (ranges_overlap(sm->data, ggml_nbytes(sm), bg->data, ggml_nbytes(bg))) return false;
    }
    if (radd && (!same_or_apart(radd, add) || !same_or_apart(radd, r) || !same_or_apart(add, r))) return false;
    if (radd && add->data == r->data) return false;      // the sum would overwrite the residual before it is read back
    const ggml_tensor * outs[4] = {down, mul, add, radd};
    for (const ggml_tensor * o : outs) {
        if (!o) continue;
        if (ranges_overlap(o->data, ggml_nbytes(o), ids->data, ggml_nbytes(ids)) || ranges_overlap(o->data, ggml_nbytes(o), w->data, ggml_nbytes(w)) ||
            ranges_overlap(o->data, ggml_nbytes(o), b->data, ggml_nbytes(b))) return false;
    }
    const size_t qb = align_up((size_t)mi355_act_q_bytes(K, n_used), 256), db = align_up((size_t)mi355_act_ds_bytes(K, n_used), 256);
    char * ws = (char *)ctx->get_scratch(0, qb + db);
    if (!ctx->act_cached(b)) {
        MI_CHECK(mi355_quantize_act((const float *)b->data, (int8_t *)ws, (float *)(ws + qb), K, n_used, 1, 1, b->nb[1], b->nb[2], b->nb[3], ctx->stream));
        ctx->act_remember(b);
    }
    mi355_moe_down_args a;
    memset(&a, 0, sizeof(a));
    a.type = (int32_t)as->type; a.n_slots = (int32_t)n_used; a.n_expert = (int32_t)n_expert;
    a.W = as->data; a.w_row_stride = (int64_t)as->nb[1]; a.w_expert_stride = (int64_t)as->nb[2];
    a.m = M; a.k = K; a.act_q = (const int8_t *)ws; a.act_ds = (const float *)(ws + qb);
    a.ids = (const int32_t *)ids->data; a.weights = (const float *)w->data;
    a.out = mul->data == down->data ? nullptr : (float *)down->data;
    a.weighted_out = (float *)mul->data; a.sum_out = (float *)add->data;
    if (radd) { a.residual = (const float *)r->data; a.residual_out = (float *)radd->data; }
    const int rc = mi355_moe_down_matvec(&a, ctx->stream);
    if (rc == MI355_E_UNSUPPORTED || rc == MI355_E_ALIGN) return false;
    MI_CHECK(rc);
    done[jm] = 1; done[ja] = 1;
    if (radd) done[jr] = 1;
    for (const ggml_tensor * o : outs) if (o) ctx->act_invalidate_if_written(o);
    return true;
}

// ------------------------------------------------------------------------------------------------
// Fused attention input of a decode step.  llm_build_llama (src/llama-model.cpp:4898-5052) spells it as
//     Qcur = MUL_MAT(wq, x)  RESHAPE  ROPE      Kcur = MUL_MAT(wk, x)  RESHAPE  ROPE      Vcur = MUL_MAT(wv, x)  RESHAPE
//     CPY(rope(Kcur) -> view of cache_k)        CPY(Vcur -> view of cache_v)
// (Kcur and Vcur even reuse Qcur's memory, so these mat-vecs cannot simply be grouped.)  When exactly this shape
// is found for a single token, with f16/f32 cache rows and no other reader of the intermediates, one launch of
// mi355_qkv_rope_matvec produces rope(Qcur) and both cache rows; Qcur, Kcur, Vcur and rope(Kcur) are never stored.
// ------------------------------------------------------------------------------------------------

// MoE router of a decode-sized batch in one launch: MUL_MAT(f32 gate_inp, x) SOFT_MAX ARGSORT(desc) [top-k view] GET_ROWS
// SUM_ROWS DIV (build_moe_ffn, src/llama-graph.cpp:660-720) -> mi355_moe_router_f32.  Every tensor of the chain is written,
// with bit-identical values, so no use analysis is needed; the outputs only have to be laid out so that one
// workgroup's (= one token's) stores cannot land in another token's rows of a different tensor.
static bool mi355_try_fuse_moe_router(mi355_backend_ctx * ctx, ggml_cgraph * cgraph, int i, std::vector<char> & done) {
    const int n = ggml_graph_n_nodes(cgraph);
    ggml_tensor * mm = ggml_graph_node(cgraph, i);
    const ggml_tensor * g = mm->src[0], * x = mm->src[1];
    if (g->type != GGML_TYPE_F32 || x->type != GGML_TYPE_F32 || tensor_is_split(g)) return false;
    const int64_t K = g->ne[0], E = g->ne[1], T = x->ne[1];
    if (g->ne[2] != 1 || g->ne[3] != 1 || x->ne[2] != 1 || x->ne[3] != 1 || E < 1 || E > 256 || T < 1 || T > 16) return false;
    if (g->nb[0] != sizeof(float) || x->nb[0] != sizeof(float) || !ggml_is_contiguous(mm)) return false;
    // The chain's nodes in graph order; llama.cpp's graph puts the three expert MUL_MAT_IDs between the top-k and the
    // weight gathering, so other nodes may sit in between -- they are remembered and checked below.
    static const ggml_op seq[5] = {GGML_OP_SOFT_MAX, GGML_OP_ARGSORT, GGML_OP_GET_ROWS, GGML_OP_SUM_ROWS, GGML_OP_DIV};
    ggml_tensor * t5[5];
    int idx[5], cnt = 0;
    std::vector<const ggml_tensor *> between;
    for (int j = i + 1; j < n && j < i + 40 && cnt < 5; j++) {
        ggml_tensor * t = ggml_graph_node(cgraph, j);
        if (done[j]) return false;
        if (is_view_op(t) || ggml_is_empty(t)) continue;
        const ggml_tensor * prev = cnt == 0 ? mm : t5[cnt - 1];
        const bool link = t->op == seq[cnt] && (cnt == 2 ? view_root(t->src[0]) == t5[0]                  // GET_ROWS reads the probabilities
                                               : cnt == 3 ? view_root(t->src[0]) == t5[2]                  // SUM_ROWS / DIV read the gathered weights
                                               : cnt == 4 ? view_root(t->src[0]) == t5[2] : t->src[0] == prev);
        if (link) { t5[cnt] = t; idx[cnt++] = j; }
        else {
            if (cnt < 2) return false;      // MUL_MAT, SOFT_MAX, ARGSORT are consecutive
            between.push_back(t);
        }
    }
    if (cnt < 5) return false;
    ggml_tensor * sm = t5[0], * as = t5[1], * gr = t5[2], * sr = t5[3], * dv = t5[4];
    float scale, max_bias;
    memcpy(&scale, (const float *)sm->op_params + 0, 4); memcpy(&max_bias, (const float *)sm->op_params + 1, 4);
    if (sm->src[0] != mm || sm->src[1] || scale != 1.0f || max_bias != 0.0f || !ggml_is_contiguous(sm) || sm->type != GGML_TYPE_F32) return false;
    if (as->src[0] != sm || ((const int32_t *)as->op_params)[0] != (int32_t)GGML_SORT_ORDER_DESC || !ggml_is_contiguous(as)) return false;
    const ggml_tensor * sel = gr->src[1];      // top-k: a view of the first n_used entries of every argsort row
    const int64_t U = sel->ne[0];
    if (view_root(gr->src[0]) != sm || view_root(sel) != as || sel->data != as->data || sel->ne[1] != T || sel->ne[2] != 1 || sel->ne[3] != 1 ||
        sel->nb[0] != sizeof(int32_t) || sel->nb[1] != (size_t)E * sizeof(int32_t) || U < 1 || U > E) return false;
    if (gr->src[0]->ne[0] != 1 || gr->src[0]->ne[1] != E || gr->src[0]->ne[2] != T || gr->src[0]->data != sm->data || !ggml_is_contiguous(gr->src[0])) return false;
    if (gr->type != GGML_TYPE_F32 || !ggml_is_contiguous(gr) || ggml_nelements(gr) != U * T) return false;
    if (view_root(sr->src[0]) != gr || sr->src[0]->data != gr->data || sr->src[0]->ne[0] != U || sr->src[0]->ne[1] != T || !ggml_is_contiguous(sr->src[0]) ||
        !ggml_is_contiguous(sr) || ggml_nelements(sr) != T) return false;
    if (view_root(dv->src[0]) != gr || dv->src[0]->data != gr->data || dv->src[1] != sr || dv->ne[0] != U || dv->ne[1] != T || !ggml_is_contiguous(dv) ||
        !ggml_is_contiguous(dv->src[0])) return false;
    // GET_ROWS, SUM_ROWS and DIV run early: no node in between may read or write the bytes they produce
    for (const ggml_tensor * t : between) {
        for (const ggml_tensor * o : {(const ggml_tensor *)gr, (const ggml_tensor *)sr, (const ggml_tensor *)dv}) {
            if (ranges_overlap(t->data, ggml_nbytes(t), o->data, ggml_nbytes(o))) return false;
            for (int k = 0; k < GGML_MAX_SRC; k++)
                if (t->src[k] && ranges_overlap(t->src[k]->data, ggml_nbytes(t->src[k]), o->data, ggml_nbytes(o))) return false;
        }
        // ... and none may overwrite the probabilities or the order before the (now earlier) gathering would have read them
        if (ranges_overlap(t->data, ggml_nbytes(t), sm->data, ggml_nbytes(sm)) || ranges_overlap(t->data, ggml_nbytes(t), as->data, ggml_nbytes(as))) return false;
    }
    // outputs: pairwise disjoint, or exactly the same rows (in-place reuse by ggml-alloc); inputs untouched
    const ggml_tensor * outs[6] = {mm, sm, as, gr, sr, dv};
    const size_t rowb[6] = {(size_t)E * 4, (size_t)E * 4, (size_t)E * 4, (size_t)U * 4, 4, (size_t)U * 4};
    for (int p = 0; p < 6; p++) {
        if (ranges_overlap(outs[p]->data, ggml_nbytes(outs[p]), x->data, ggml_nbytes(x)) ||
            ranges_overlap(outs[p]->data, ggml_nbytes(outs[p]), g->data, ggml_nbytes(g))) return false;
        for (int q = p + 1; q < 6; q++) {
            if (!ranges_overlap(outs[p]->data, ggml_nbytes(outs[p]), outs[q]->data, ggml_nbytes(outs[q]))) continue;
            if (T > 1 && (outs[p]->data != outs[q]->data || rowb[p] != rowb[q])) return false;
            // one token = one workgroup: logits, probs and order are written in phases separated by barriers, so a later
            // tensor may reuse an earlier one's memory freely; the last three are written together by one wave
            if (T == 1 && p >= 3 && !(p == 3 && q == 5 && outs[p]->data == outs[q]->data)) return false;
        }
    }
    MI_CHECK(mi355_moe_router_f32((const float *)x->data, (int64_t)x->nb[1], (const float *)g->data, (int64_t)g->nb[1], K, E, U, T,
                                  (float *)mm->data, (float *)sm->data, (int32_t *)as->data, (float *)gr->data, (float *)sr->data,
                                  (float *)dv->data, ctx->stream));
    for (int m = 0; m < 5; m++) done[idx[m]] = 1;
    for (const ggml_tensor * o : outs) ctx->act_invalidate_if_written(o);
    return true;
}

// The attention input of a llama-style layer as build_attn spells it (src/llama-graph.cpp, llama-kv-cache-unified.cpp:817-891):
//     q = MUL_MAT(wq, x)  q = ROPE(q)  k = MUL_MAT(wk, x)  k = ROPE(k)  v = MUL_MAT(wv, x)  CPY(k -> cache rows)  CPY(v -> cache rows)
// starting at node i, for N = x->ne[1] tokens whose cache rows lie one after the other.  Only q (rotated) and the two cache views
// are materialised by the fused launches, so nobody else may read the four tensors in between.
struct qkv_nodes {
    ggml_tensor * mq, * rq, * mk, * rk, * mv, * ck, * cv;
    int idx[7];
    bool kv_q8;
};

static bool qkv_match(ggml_cgraph * cgraph, int i, const std::vector<char> & done, bool (*mm_ok)(const ggml_tensor *), qkv_nodes & q) {
    const int n = ggml_graph_n_nodes(cgraph);
    ggml_tensor * mq = ggml_graph_node(cgraph, i);
    const ggml_tensor * x = mq->src[1];
    if (x->ne[2] != 1 || x->ne[3] != 1) return false;
    const int64_t N = x->ne[1];
    ggml_tensor * rq = nullptr, * mk = nullptr, * rk = nullptr, * mv = nullptr, * ck = nullptr, * cv = nullptr;
    int * idx = q.idx, last = -1;
    idx[0] = i;
    for (int m = 1; m < 7; m++) idx[m] = -1;
    const int horizon = i + 24 < n ? i + 24 : n;
    for (int j = i + 1; j < horizon; j++) {
        ggml_tensor * t = ggml_graph_node(cgraph, j);
        if (done[j]) return false;
        if (is_view_op(t) || ggml_is_empty(t)) continue;
        if (t->op == GGML_OP_ROPE) {
            const ggml_tensor * r = view_root(t->src[0]);
            if (r == mq && !rq)            { rq = t; idx[1] = j; }
            else if (mk && r == mk && !rk) { rk = t; idx[3] = j; }
            else return false;
        } else if (t->op == GGML_OP_MUL_MAT) {
            if (!mm_ok(t) || t->src[1] != x) return false;
            if (!mk)      { mk = t; idx[2] = j; }
            else if (!mv) { mv = t; idx[4] = j; }
            else return false;
        } else if (t->op == GGML_OP_CPY) {
            const ggml_tensor * r = view_root(t->src[0]);
            if (rk && r == rk && !ck)      { ck = t; idx[5] = j; }
            else if (mv && r == mv && !cv) { cv = t; idx[6] = j; }
            else return false;
        } else {
            return false;
        }
        if (rq && mk && rk && mv && ck && cv) { last = j; break; }
    }
    if (last < 0) return false;
    // shapes and types
    const ggml_tensor * wq = mq->src[0], * wk = mk->src[0], * wv = mv->src[0];
    if (wq->ne[0] != wk->ne[0] || wq->ne[0] != wv->ne[0]) return false;
    const int64_t Mq = wq->ne[1], Mk = wk->ne[1], Mv = wv->ne[1], hd = rq->ne[0];
    if (Mq % 8 || Mk % 8 || Mv % 8 || hd != rk->ne[0] || hd <= 0 || Mq % hd || Mk % hd) return false;
    for (ggml_tensor * r : {rq, rk}) {
        if (r->type != GGML_TYPE_F32 || !ggml_is_contiguous(r) || !ggml_is_contiguous(r->src[0]) || r->ne[2] != N || r->ne[3] != 1) return false;
        if (r->src[1] != rq->src[1] || r->src[2] != rq->src[2]) return false;
        if (memcmp(r->op_params, rq->op_params, 11 * sizeof(int32_t)) != 0) return false;
        if (r->src[1]->type != GGML_TYPE_I32 || (r->src[2] && r->src[2]->type != GGML_TYPE_F32)) return false;
    }
    if (rq->src[0]->data != mq->data || rk->src[0]->data != mk->data) return false;
    if (ggml_nelements(rq) != Mq * N || ggml_nelements(rk) != Mk * N) return false;
    const int32_t * rp = (const int32_t *)rq->op_params;
    if (rp[2] != 0 && rp[2] != 2) return false;
    q.kv_q8 = ck->type == GGML_TYPE_Q8_0 && cv->type == GGML_TYPE_Q8_0;
    for (ggml_tensor * c : {ck, cv}) {
        if ((c->type != GGML_TYPE_F16 && c->type != GGML_TYPE_F32 && !q.kv_q8) || !ggml_is_contiguous(c) || !ggml_is_contiguous(c->src[0])) return false;
        if (c->src[0]->type != GGML_TYPE_F32) return false;
    }
    if (q.kv_q8 && (Mk % 32 || Mv % 32)) return false;
    if (ck->src[0]->data != rk->data || cv->src[0]->data != mv->data || ggml_nelements(ck) != Mk * N || ggml_nelements(cv) != Mv * N) return false;
    // nobody else may read the tensors that will not be materialised
    const ggml_tensor * hidden[4] = {mq, mk, mv, rk};
    for (const ggml_tensor * h : hidden) if (h->flags & GGML_TENSOR_FLAG_OUTPUT) return false;
    for (int j = i + 1; j < n; j++) {
        const ggml_tensor * t = ggml_graph_node(cgraph, j);
        if (is_view_op(t)) { if (t->flags & GGML_TENSOR_FLAG_OUTPUT) for (const ggml_tensor * h : hidden) if (view_root(t) == h) return false; continue; }
        bool in_pattern = false;
        for (int m = 1; m < 7; m++) if (idx[m] == j) in_pattern = true;
        if (in_pattern) continue;
        for (int sidx = 0; sidx < GGML_MAX_SRC; sidx++) {
            if (!t->src[sidx]) continue;
            for (const ggml_tensor * h : hidden) if (view_root(t->src[sidx]) == h) return false;
        }
    }
    q.mq = mq; q.rq = rq; q.mk = mk; q.rk = rk; q.mv = mv; q.ck = ck; q.cv = cv;
    return true;
}

static void qkv_rope_params(const ggml_tensor * rq, int32_t * n_dims, int32_t * mode, int32_t * n_ctx_orig, float * f6) {
    const int32_t * rp = (const int32_t *)rq->op_params;
    *n_dims = rp[1]; *mode = rp[2]; *n_ctx_orig = rp[4];
    memcpy(f6, rp + 5, 6 * sizeof(float));       // freq_base, freq_scale, ext_factor, attn_factor, beta_fast, beta_slow
}

// Prefill form: three GEMMs that leave their K-slice planes in scratch (mi355_mul_mat_dq_parts) and ONE launch that adds the planes,
// rotates q and k and stores k and v into the cache rows (mi355_qkv_finish) -- 4 launches for the 7 nodes' 10 (3 GEMMs, up to 3 slice
// sums or memsets, 2 ROPE, 2 CPY), and q / k / v cross HBM once less.  Bit-identical to the nodes run one by one.
static bool prefill_mm_ok(const ggml_tensor * t) {
    const ggml_tensor * w = t->src[0], * x = t->src[1];
    return t->op == GGML_OP_MUL_MAT && t->type == GGML_TYPE_F32 && x->type == GGML_TYPE_F32 && w->ne[2] == 1 && w->ne[3] == 1 && !tensor_is_split(w) &&
           mi355_mul_mat_path(w, x) == MM_DQ;
}

static bool mi355_try_fuse_qkv_gemm(mi355_backend_ctx * ctx, ggml_cgraph * cgraph, int i, std::vector<char> & done) {
    ggml_tensor * mq0 = ggml_graph_node(cgraph, i);
    if (!prefill_mm_ok(mq0) || mq0->src[1]->ne[1] <= 1 || MI355_ENV_SET("GGML_MI355_NO_QKV_GEMM") || MI355_ENV_SET("GGML_MI355_MMQ_INT8")) return false;
    qkv_nodes q;
    if (!qkv_match(cgraph, i, done, prefill_mm_ok, q)) return false;
    const ggml_tensor * x = q.mq->src[1];
    const ggml_tensor * w[3] = {q.mq->src[0], q.mk->src[0], q.mv->src[0]};
    const int64_t K = x->ne[0], N = x->ne[1], hd = q.rq->ne[0];
    if (w[2]->ne[1] % 4) return false;
    // the three plane sets
    int64_t pb[3], off[3], total = 0;
    for (int m = 0; m < 3; m++) {
        pb[m] = mi355_mul_mat_dq_parts_bytes(w[m]->ne[1], w[m]->ne[1], K, N);
        if (pb[m] <= 0) return false;
        off[m] = total; total += (int64_t)align_up((size_t)pb[m], 256);
    }
    char * parts = (char *)ctx->get_scratch(3, (size_t)total);
    if (ctx->scratch_size[3] < (size_t)total) return false;
    // outputs must not alias one another or the positions (they are written by one launch, per element)
    const ggml_tensor * outs[3] = {q.rq, q.ck, q.cv};
    for (int a = 0; a < 3; a++) {
        for (int b = 0; b < a; b++) if (ranges_overlap(outs[a]->data, ggml_nbytes(outs[a]), outs[b]->data, ggml_nbytes(outs[b]))) return false;
        if (ranges_overlap(outs[a]->data, ggml_nbytes(outs[a]), q.rq->src[1]->data, ggml_nbytes(q.rq->src[1]))) return false;
    }
    // a q8_0 cache: k and v leave the finish launch as f32 rows in scratch and one more launch quantizes both into the cache rows
    // (mi355_kv_store_q8_0, the arithmetic of the CPY f32 -> q8_0 it replaces), as on the decode path
    const int64_t Mk = w[1]->ne[1], Mv = w[2]->ne[1];
    float * kv_tmp = nullptr;
    if (q.kv_q8) {
        const size_t need = (size_t)N * (Mk + Mv) * sizeof(float);
        kv_tmp = (float *)ctx->get_scratch(4, need);
        if (ctx->scratch_size[4] < need || N * (Mk > Mv ? Mk : Mv) > 0x7fffffff) return false;
    }
    const uint16_t * act = prefill_act_bf16(ctx, x);
    mi355_qkv_finish_args f;
    memset(&f, 0, sizeof(f));
    int32_t kz[3] = {0, 0, 0};
    bool paired = false;
    for (int m = 0; m < 3; m++) {
        if (m == 2 && paired) break;
        if (m == 1 && Mk == Mv && pb[1] == pb[2] && !MI355_ENV_SET("GGML_MI355_NO_QKV_PAIR")) {
            // wk and wv (M = 1024 each in a GQA model) in one launch: each alone fills half the chip
            int32_t kzp = 0;
            const int rcp = mi355_mul_mat_dq_parts_pair((int)w[1]->type, w[1]->data, (int64_t)w[1]->nb[1], (float *)(parts + off[1]),
                                                        (int)w[2]->type, w[2]->data, (int64_t)w[2]->nb[1], (float *)(parts + off[2]), pb[1],
                                                        act, K, Mk, K, N, &kzp, ctx->stream);
            if (rcp == MI355_OK) { kz[1] = kz[2] = kzp; paired = true; continue; }
            if (rcp != MI355_E_UNSUPPORTED && rcp != MI355_E_ALIGN) MI_CHECK(rcp);
        }
        const int rc = mi355_mul_mat_dq_parts((int)w[m]->type, w[m]->data, (int64_t)w[m]->nb[1], act, K, (float *)(parts + off[m]), pb[m], w[m]->ne[1], w[m]->ne[1], K, N,
                                              &kz[m], ctx->stream);
        // nothing of the pattern has been written yet when the first product is refused; a later refusal cannot happen for a type
        // and K the first one took
        if (m == 0 && (rc == MI355_E_UNSUPPORTED || rc == MI355_E_ALIGN)) return false;
        MI_CHECK(rc);
    }
    f.q_parts = (const float *)(parts + off[0]); f.k_parts = (const float *)(parts + off[1]); f.v_parts = (const float *)(parts + off[2]);
    f.kz_q = kz[0]; f.kz_k = kz[1]; f.kz_v = kz[2];
    f.m_q = w[0]->ne[1]; f.m_k = w[1]->ne[1]; f.m_v = w[2]->ne[1]; f.n = N;
    f.q_out = (float *)q.rq->data; f.k_out = q.ck->data; f.v_out = q.cv->data;
    f.k_out_type = (int32_t)q.ck->type; f.v_out_type = (int32_t)q.cv->type;
    if (kv_tmp) { f.k_out = kv_tmp; f.v_out = kv_tmp + N * Mk; f.k_out_type = f.v_out_type = (int32_t)GGML_TYPE_F32; }
    f.pos = (const int32_t *)q.rq->src[1]->data; f.freq_factors = q.rq->src[2] ? (const float *)q.rq->src[2]->data : nullptr;
    f.head_dim = (int32_t)hd;
    float f6[6];
    qkv_rope_params(q.rq, &f.n_dims, &f.mode, &f.n_ctx_orig, f6);
    f.freq_base = f6[0]; f.freq_scale = f6[1]; f.ext_factor = f6[2]; f.attn_factor = f6[3]; f.beta_fast = f6[4]; f.beta_slow = f6[5];
    MI_CHECK(mi355_qkv_finish(&f, ctx->stream));
    if (kv_tmp) MI_CHECK(mi355_kv_store_q8_0(kv_tmp, N * Mk, q.ck->data, kv_tmp + N * Mk, N * Mv, q.cv->data, nullptr, 0, ctx->stream));
    for (int m = 1; m < 7; m++) done[q.idx[m]] = 1;
    ctx->act_invalidate_if_written(q.rq);
    return true;
}

// Small-batch form (2..32 tokens on the skinny kernels): the three products go to scratch as one "plane" each and the same finish launch
// rotates q and k and stores k and v -- 4 launches for the nodes' 7 (3 MUL_MAT, 2 ROPE, 2 CPY).  The products are the bits of the
// separate MUL_MATs (same kernels, same activations), the finish launch adds nothing to a single plane: bit-identical.
static bool small_mm_ok(const ggml_tensor * t) {
    const ggml_tensor * w = t->src[0], * x = t->src[1];
    if (!(t->op == GGML_OP_MUL_MAT && t->type == GGML_TYPE_F32 && x->type == GGML_TYPE_F32 && w->ne[2] == 1 && w->ne[3] == 1 && !tensor_is_split(w) &&
          is_weight_type(w->type) && ggml_is_contiguous(t))) return false;
    const int path = mi355_mul_mat_path(w, x);
    return path == MM_SKINNY_K || path == MM_SKINNY;
}

static bool mi355_try_fuse_qkv_small(mi355_backend_ctx * ctx, ggml_cgraph * cgraph, int i, std::vector<char> & done) {
    ggml_tensor * mq0 = ggml_graph_node(cgraph, i);
    if (mq0->src[1]->ne[1] <= 1 || !small_mm_ok(mq0) || MI355_ENV_SET("GGML_MI355_NO_QKV_SMALL")) return false;
    qkv_nodes q;
    if (!qkv_match(cgraph, i, done, small_mm_ok, q)) return false;
    const ggml_tensor * x = q.mq->src[1];
    ggml_tensor * mm[3] = {q.mq, q.mk, q.mv};
    const int64_t N = x->ne[1], hd = q.rq->ne[0];
    const int64_t Mq = mm[0]->ne[0], Mk = mm[1]->ne[0], Mv = mm[2]->ne[0];
    if (Mv % 4 || (N * Mv) % 4) return false;
    size_t off[3], total = 0;
    for (int m = 0; m < 3; m++) { off[m] = total; total += align_up((size_t)N * mm[m]->ne[0] * sizeof(float), 256); }
    const ggml_tensor * outs[3] = {q.rq, q.ck, q.cv};
    for (int a = 0; a < 3; a++) {
        for (int b = 0; b < a; b++) if (ranges_overlap(outs[a]->data, ggml_nbytes(outs[a]), outs[b]->data, ggml_nbytes(outs[b]))) return false;
        if (ranges_overlap(outs[a]->data, ggml_nbytes(outs[a]), q.rq->src[1]->data, ggml_nbytes(q.rq->src[1]))) return false;
        if (ranges_overlap(outs[a]->data, ggml_nbytes(outs[a]), x->data, ggml_nbytes(x))) return false;
    }
    char * parts = (char *)ctx->get_scratch(3, total);
    if (ctx->scratch_size[3] < total) return false;
    float * kv_tmp = nullptr;
    if (q.kv_q8) {
        const size_t need = (size_t)N * (Mk + Mv) * sizeof(float);
        kv_tmp = (float *)ctx->get_scratch(4, need);
        if (ctx->scratch_size[4] < need) return false;
    }
    mi355_qkv_finish_args f;
    memset(&f, 0, sizeof(f));
    f.q_parts = (const float *)(parts + off[0]); f.k_parts = (const float *)(parts + off[1]); f.v_parts = (const float *)(parts + off[2]);
    f.kz_q = f.kz_k = f.kz_v = 1;
    f.m_q = Mq; f.m_k = Mk; f.m_v = Mv; f.n = N;
    f.q_out = (float *)q.rq->data; f.k_out = q.ck->data; f.v_out = q.cv->data;
    f.k_out_type = (int32_t)q.ck->type; f.v_out_type = (int32_t)q.cv->type;
    if (kv_tmp) { f.k_out = kv_tmp; f.v_out = kv_tmp + N * Mk; f.k_out_type = f.v_out_type = (int32_t)GGML_TYPE_F32; }
    f.pos = (const int32_t *)q.rq->src[1]->data; f.freq_factors = q.rq->src[2] ? (const float *)q.rq->src[2]->data : nullptr;
    f.head_dim = (int32_t)hd;
    float f6[6];
    qkv_rope_params(q.rq, &f.n_dims, &f.mode, &f.n_ctx_orig, f6);
    f.freq_base = f6[0]; f.freq_scale = f6[1]; f.ext_factor = f6[2]; f.attn_factor = f6[3]; f.beta_fast = f6[4]; f.beta_slow = f6[5];
    // everything the finish launch could refuse, before anything is launched (mi355_qkv_finish's own checks)
    if (hd <= 0 || (hd & 1) || f.n_dims > hd || (f.n_dims & 1) || Mq % hd || Mk % hd || (f.mode != 0 && f.mode != 2)) return false;
    if ((f.k_out_type != (int32_t)GGML_TYPE_F16 && f.k_out_type != (int32_t)GGML_TYPE_F32) || (f.v_out_type != (int32_t)GGML_TYPE_F16 && f.v_out_type != (int32_t)GGML_TYPE_F32)) return false;
    if (((uintptr_t)f.v_out & 15) || ((uintptr_t)f.q_out & 3) || ((uintptr_t)f.k_out & 3)) return false;
    // The K-quant products of wq's type leave as ONE launch over all their row tiles (Q4_K_M: wq + wk, and wv where it is q4_K too):
    // 128 + 32 + 32 tiles of an 8B layer are three launches that each cut K over two workgroups behind a memset, or one launch of
    // 192 workgroups with neither.
    bool grouped[3] = {false, false, false};
    const int64_t K = x->ne[0];
    if (N <= 32 && x->ne[2] == 1 && x->ne[3] == 1 && !MI355_ENV_SET("GGML_MI355_NO_SKINNY_GROUP") &&
        mi355_mul_mat_path(mm[0]->src[0], x, mm[0]) == MM_SKINNY_K) {
        const void * Wg[3]; int64_t rs[3], dcs[3], mg[3]; float * dg[3];
        int cnt = 0, who[3];
        for (int m = 0; m < 3; m++) {
            const ggml_tensor * w = mm[m]->src[0];
            if (w->type != mm[0]->src[0]->type || mi355_mul_mat_path(w, x, mm[m]) != MM_SKINNY_K) continue;
            Wg[cnt] = w->data; rs[cnt] = (int64_t)w->nb[1]; dg[cnt] = (float *)(parts + off[m]); dcs[cnt] = mg[cnt] = w->ne[1];
            who[cnt++] = m;
        }
        if (cnt >= 2) {
            const mi355_act_layout L = mi355_act_planes(MM_SKINNY_K, K, N);
            char * ws = (char *)ctx->get_scratch(0, L.total());
            if (!ctx->act_cached(x, 2)) {
                MI_CHECK(mi355_quantize_act_q8k((const float *)x->data, (int8_t *)ws, (float *)(ws + L.b0), (int16_t *)(ws + L.b0 + L.b1), K, N, 1, 1,
                                                x->nb[1], x->nb[2], x->nb[3], ctx->stream));
                ctx->act_remember(x, 2);
            }
            const int rc = mi355_mul_mat_skinny_k_group((int)mm[0]->src[0]->type, cnt, Wg, rs, dg, dcs, mg, (const int8_t *)ws, (const float *)(ws + L.b0),
                                                        (const int16_t *)(ws + L.b0 + L.b1), K, N, ctx->stream);
            if (rc == MI355_OK) for (int c = 0; c < cnt; c++) grouped[who[c]] = true;
            else if (rc != MI355_E_UNSUPPORTED && rc != MI355_E_ALIGN && rc != MI355_E_SHAPE) MI_CHECK(rc);
        }
    }
    for (int m = 0; m < 3; m++) {
        if (grouped[m]) continue;
        ggml_tensor t = *mm[m];              // the product of node m, into its plane instead of the node's own bytes
        t.data = parts + off[m];
        mi355_mul_mat(ctx, &t);
    }
    MI_CHECK(mi355_qkv_finish(&f, ctx->stream));
    if (kv_tmp) MI_CHECK(mi355_kv_store_q8_0(kv_tmp, N * Mk, q.ck->data, kv_tmp + N * Mk, N * Mv, q.cv->data, nullptr, 0, ctx->stream));
    for (int m = 1; m < 7; m++) done[q.idx[m]] = 1;
    ctx->act_invalidate_if_written(q.rq);
    return true;
}

// Consecutive small-batch MUL_MATs of one K-quant type on the same activations (ffn_up + ffn_gate of a step of 5..32 tokens,
// llama-graph.cpp build_ffn) as one launch over all their row tiles.
static bool mi355_try_group_small(mi355_backend_ctx * ctx, ggml_cgraph * cgraph, int i, std::vector<char> & done) {
    ggml_tensor * first = ggml_graph_node(cgraph, i);
    const ggml_tensor * x = first->src[1];
    if (!small_mm_ok(first) || x->ne[1] > 32 || x->ne[2] != 1 || x->ne[3] != 1 || MI355_ENV_SET("GGML_MI355_NO_SKINNY_GROUP")) return false;
    if (mi355_mul_mat_path(first->src[0], x, first) != MM_SKINNY_K) return false;
    const void * Wg[4]; int64_t rs[4], dcs[4], mg[4]; float * dg[4];
    int cnt = 0;
    const int n = ggml_graph_n_nodes(cgraph);
    for (int j = i; j < n && cnt < 4; j++) {
        ggml_tensor * t = ggml_graph_node(cgraph, j);
        if (done[j] || !small_mm_ok(t) || t->src[1] != x || t->src[0]->type != first->src[0]->type ||
            mi355_mul_mat_path(t->src[0], x, t) != MM_SKINNY_K) break;
        bool clash = ranges_overlap(t->data, ggml_nbytes(t), x->data, ggml_nbytes(x));
        for (int c = 0; c < cnt; c++) clash = clash || ranges_overlap(t->data, ggml_nbytes(t), dg[c], (size_t)dcs[c] * x->ne[1] * sizeof(float));
        if (clash) break;
        Wg[cnt] = t->src[0]->data; rs[cnt] = (int64_t)t->src[0]->nb[1]; dg[cnt] = (float *)t->data;
        dcs[cnt] = (int64_t)(t->nb[1] / sizeof(float)); mg[cnt] = t->src[0]->ne[1];
        cnt++;
    }
    if (cnt < 2) return false;
    const int64_t K = x->ne[0], N = x->ne[1];
    const mi355_act_layout L = mi355_act_planes(MM_SKINNY_K, K, N);
    char * ws = (char *)ctx->get_scratch(0, L.total());
    if (!ctx->act_cached(x, 2)) {
        MI_CHECK(mi355_quantize_act_q8k((const float *)x->data, (int8_t *)ws, (float *)(ws + L.b0), (int16_t *)(ws + L.b0 + L.b1), K, N, 1, 1,
                                        x->nb[1], x->nb[2], x->nb[3], ctx->stream));
        ctx->act_remember(x, 2);
    }
    const int rc = mi355_mul_mat_skinny_k_group((int)first->src[0]->type, cnt, Wg, rs, dg, dcs, mg, (const int8_t *)ws, (const float *)(ws + L.b0),
                                                (const int16_t *)(ws + L.b0 + L.b1), K, N, ctx->stream);
    if (rc == MI355_E_UNSUPPORTED || rc == MI355_E_ALIGN || rc == MI355_E_SHAPE) return false;
    MI_CHECK(rc);
    for (int c = 1; c < cnt; c++) done[i + c] = 1;
    for (int c = 0; c < cnt; c++) ctx->act_invalidate_if_written(ggml_graph_node(cgraph, i + c));
    return true;
}

// Decode form.  norm_idx >= 0: nodes norm_idx (RMS_NORM) and norm_idx + 1 (MUL by the norm weight, == x) come right before node i and
// run in the launch's prologue (attn_norm of every layer): 9 nodes, one launch.
static bool mi355_try_fuse_qkv_rope(mi355_backend_ctx * ctx, ggml_cgraph * cgraph, int i, std::vector<char> & done, int norm_idx = -1) {
    ggml_tensor * mq0 = ggml_graph_node(cgraph, i);
    if (!matvec_groupable(mq0) || mq0->src[1]->ne[1] != 1) return false;
    qkv_nodes qn;
    if (!qkv_match(cgraph, i, done, matvec_groupable, qn)) return false;
    ggml_tensor * mq = qn.mq, * rq = qn.rq, * mk = qn.mk, * mv = qn.mv, * ck = qn.ck, * cv = qn.cv;
    const int * idx = qn.idx;
    const bool kv_q8 = qn.kv_q8;
    const ggml_tensor * x = mq->src[1];
    const ggml_tensor * wq = mq->src[0], * wk = mk->src[0], * wv = mv->src[0];
    const int64_t K = wq->ne[0], Mq = wq->ne[1], Mk = wk->ne[1], Mv = wv->ne[1], hd = rq->ne[0];
    const int32_t * rp = (const int32_t *)rq->op_params;
    // activations of the token, q8_1
    const size_t qb = align_up((size_t)mi355_act_q_bytes(K, 1), 256), db = align_up((size_t)mi355_act_ds_bytes(K, 1), 256);
    char * ws = (char *)ctx->get_scratch(0, qb + db);
    mi355_qkv_rope_args a;
    memset(&a, 0, sizeof(a));
    if (norm_idx >= 0) {
        const ggml_tensor * norm = ggml_graph_node(cgraph, norm_idx);
        const ggml_tensor * w = x->src[0] == norm ? x->src[1] : x->src[0];
        const ggml_tensor * outs[4] = {x, rq, ck, cv};
        const size_t ob[4] = {ggml_nbytes(x), ggml_nbytes(rq), ggml_nbytes(ck), ggml_nbytes(cv)};
        if (!norm_prologue_ok(norm->src[0], w, K, 4096, outs, ob, 4)) return false;
        a.norm_x = (const float *)norm->src[0]->data; a.norm_w = (const float *)w->data; a.norm_out = (float *)x->data;
        memcpy(&a.norm_eps, norm->op_params, sizeof(float));
    }
    a.type_qk = (int32_t)wq->type; a.type_v = (int32_t)wv->type;
    a.type_k = wk->type != wq->type ? (int32_t)wk->type : 0;
    a.wq = wq->data; a.wk = wk->data; a.wv = wv->data;
    a.wq_row_stride = (int64_t)wq->nb[1]; a.wk_row_stride = (int64_t)wk->nb[1]; a.wv_row_stride = (int64_t)wv->nb[1];
    a.m_q = Mq; a.m_k = Mk; a.m_v = Mv; a.k = K;
    a.act_q = (const int8_t *)ws; a.act_ds = (const float *)(ws + qb);
    a.q_out = (float *)rq->data; a.k_out = ck->data; a.v_out = cv->data;
    void * const * kv_table = nullptr;
    int kv_slot = 0;
    if (ctx->gc.capturing) {
        if (ctx->gc.n_slots >= MI355_KV_SLOTS) { ctx->gc.capture_failed = true; }
        else { kv_table = ctx->gc.kv_dev; kv_slot = ctx->gc.n_slots; }
    }
    a.k_out_type = (int32_t)ck->type; a.v_out_type = (int32_t)cv->type;
    float * kv_tmp = nullptr;
    if (kv_q8) {
        kv_tmp = (float *)ctx->get_scratch(4, (size_t)(Mk + Mv) * sizeof(float));
        if (ctx->scratch_size[4] < (size_t)(Mk + Mv) * sizeof(float)) return false;       // under capture the buffer could not grow
        a.k_out = kv_tmp; a.v_out = kv_tmp + Mk;
        a.k_out_type = a.v_out_type = (int32_t)GGML_TYPE_F32;
    } else {
        a.kv_table = kv_table; a.kv_slot = kv_slot;      // the fused launch stores into the cache itself
    }
    a.pos = (const int32_t *)rq->src[1]->data; a.freq_factors = rq->src[2] ? (const float *)rq->src[2]->data : nullptr;
    a.head_dim = (int32_t)hd; a.n_dims = rp[1]; a.mode = rp[2]; a.n_ctx_orig = rp[4];
    memcpy(&a.freq_base, rp + 5, 4); memcpy(&a.freq_scale, rp + 6, 4); memcpy(&a.ext_factor, rp + 7, 4);
    memcpy(&a.attn_factor, rp + 8, 4); memcpy(&a.beta_fast, rp + 9, 4); memcpy(&a.beta_slow, rp + 10, 4);
    // dry validation first (argument errors are reported before anything is launched): an unsupported type pair or
    // shape leaves the graph to the unfused path
    const bool need_quant = norm_idx < 0 && !ctx->act_cached(x);
    if (need_quant) {
        MI_CHECK(mi355_quantize_act((const float *)x->data, (int8_t *)ws, (float *)(ws + qb), K, 1, 1, 1, x->nb[1], x->nb[2], x->nb[3], ctx->stream));
        ctx->act_remember(x);
    }
    if (tensor_is_split(wq)) {
        // row-split wq / wk / wv: a device rotates and stores the heads whose rows it owns (the split's 128-row rounding keeps a head's
        // rows together for head_dim <= 128); K and V land at the slice's place in the cache row -- under capture through a table
        // slot of the device's own, whose entries carry the slice's byte offset
        const mi355_split_extra * eq = split_extra_of(wq), * ek = split_extra_of(wk), * ev = split_extra_of(wv);
        if (kv_q8 || !tensor_is_split(wk) || !tensor_is_split(wv)) return false;
        bool has[MI355_MAX_DEVICES] = {};
        const int ndev = mi355_device_count();
        int n_has = 0;
        for (int d = 0; d < ndev; d++) {
            const int64_t rq_ = eq->row_high[d] - eq->row_low[d], rk_ = ek->row_high[d] - ek->row_low[d], rv_ = ev->row_high[d] - ev->row_low[d];
            if (rq_ <= 0 && rk_ <= 0 && rv_ <= 0) continue;
            // the kernel wants rows of all three; whole heads; 16-byte aligned slice starts in the f16 cache rows
            if (rq_ <= 0 || rk_ <= 0 || rv_ <= 0 || rq_ % hd || rk_ % hd || eq->row_low[d] % hd || ek->row_low[d] % hd || rv_ % 8 || ev->row_low[d] % 8) return false;
            has[d] = true; n_has++;
        }
        if (ctx->gc.capturing && ctx->gc.n_slots + n_has > MI355_KV_SLOTS) { ctx->gc.capture_failed = true; kv_table = nullptr; }
        const size_t ksz = ggml_type_size(ck->type), vsz = ggml_type_size(cv->type);
        int slot_of[MI355_MAX_DEVICES] = {};
        if (kv_table) {
            for (int d = 0; d < ndev; d++) {
                if (!has[d]) continue;
                slot_of[d] = ctx->gc.n_slots++;
                ctx->gc.kv_push(idx[5], (size_t)ek->row_low[d] * ksz);
                ctx->gc.kv_push(idx[6], (size_t)ev->row_low[d] * vsz);
            }
        }
        mi355_split_fan_out(ctx, has, [&](int d, hipStream_t st) {
            mi355_qkv_rope_args b = a;
            b.wq = eq->data[d]; b.wk = ek->data[d]; b.wv = ev->data[d];
            b.m_q = eq->row_high[d] - eq->row_low[d]; b.m_k = ek->row_high[d] - ek->row_low[d]; b.m_v = ev->row_high[d] - ev->row_low[d];
            b.q_out = a.q_out + eq->row_low[d];
            b.k_out = (char *)ck->data + (size_t)ek->row_low[d] * ksz;
            b.v_out = (char *)cv->data + (size_t)ev->row_low[d] * vsz;
            b.kv_table = kv_table; b.kv_slot = slot_of[d];
            MI_CHECK(mi355_qkv_rope_matvec(&b, st));
        });
        for (int m = 1; m < 7; m++) done[idx[m]] = 1;
        ctx->act_invalidate_if_written(rq);
        if (norm_idx >= 0) ctx->act_invalidate_if_written(x);
        return true;
    }
    const int rc = mi355_qkv_rope_matvec(&a, ctx->stream);
    if (rc == MI355_E_UNSUPPORTED || rc == MI355_E_SHAPE || rc == MI355_E_ALIGN) return false;
    MI_CHECK(rc);
    if (kv_q8) MI_CHECK(mi355_kv_store_q8_0(kv_tmp, Mk, ck->data, kv_tmp + Mk, Mv, cv->data, kv_table, kv_slot, ctx->stream));
    for (int m = 1; m < 7; m++) done[idx[m]] = 1;
    if (kv_table) { ctx->gc.kv_push(idx[5], 0); ctx->gc.kv_push(idx[6], 0); ctx->gc.n_slots++; }
    ctx->act_invalidate_if_written(rq);
    if (norm_idx >= 0) ctx->act_invalidate_if_written(x);
    return true;
}

// ---- a decode token's layer body in ONE persistent launch (mi355_decode_layer, kernels/mmvq_layer.hip) ------------------------
//     i: MUL_MAT(wo)  ADD | RMS_NORM  MUL  MUL_MAT(gate | up)  MUL_MAT(up | gate)  GLU | MUL_MAT(down)  ADD |
//     [the next layer's RMS_NORM  MUL  MUL_MAT(wq) .. CPY CPY: the pattern of mi355_try_fuse_qkv_rope]
// i.e. the launches of mi355_try_fuse_matvec_add, mi355_try_fuse_norm_glu_matvec, mi355_try_fuse_matvec_add and
// mi355_try_fuse_qkv_rope (with its norm prologue) as one, bit-identical to them (tests/test_decode_layer_gpu.py,
// test_backend_plugin_gpu.py::test_decode_layer_launch_is_bit_exact_and_runs).  OPT-IN (GGML_MI355_DECODE_LAYER=1): measured end to end
// it is worth +0.9 % on Llama-3-70B-shaped Q4_K_M (tg64 96.4 -> 97.3 t/s; the layer micro-benchmark's 119 -> 111 us per layer does
// not carry over: the plugin's 70B path already groups gate + up behind the norm) and costs 5 % on 8B (tg128 509 -> 484), and a
// launch that needs every workgroup resident at once is not something to run by default on a GPU that may be shared.
// Every tensor of the run is still written.  The launch orders its phases internally, so ggml-alloc's reuse of dead tensors' bytes
// by later nodes of the run is honoured (mi355_kernels.h); what is checked here is what the separate launches check: the ADDs are
// in place or apart, the GLU result is on one of its inputs or apart, the tensors every workgroup reads during a phase (the norm
// inputs and weights) are apart from that phase's outputs.
#define MI355_DL_MAX 256            // sync blocks (= layers) per graph
static bool mi355_try_fuse_decode_layer(mi355_backend_ctx * ctx, ggml_cgraph * cgraph, int i, std::vector<char> & done) {
    static const int mode = [] { const char * e = getenv("GGML_MI355_DECODE_LAYER"); return e ? atoi(e) : 0; }();
    if (mode != 1) return false;
    const int n = ggml_graph_n_nodes(cgraph);
    if (i + 8 >= n) return false;
    ggml_tensor * nd[9];
    for (int j = 0; j < 9; j++) { nd[j] = ggml_graph_node(cgraph, i + j); if (done[i + j]) return false; }
    ggml_tensor * mm0 = nd[0], * add0 = nd[1], * norm1 = nd[2], * mul1 = nd[3], * glu = nd[6], * mm2 = nd[7], * add2 = nd[8];
    if (!matvec_groupable(mm0) || tensor_is_split(mm0->src[0]) || mm0->src[1]->ne[1] != 1 || mm0->src[0]->ne[1] < 2048) return false;
    auto add_of = [](const ggml_tensor * add, const ggml_tensor * mm) -> const ggml_tensor * {     // the residual operand (mi355_try_fuse_matvec_add's checks)
        if (add->op != GGML_OP_ADD || add->type != GGML_TYPE_F32) return nullptr;
        const ggml_tensor * r = add->src[0] == mm ? add->src[1] : (add->src[1] == mm ? add->src[0] : nullptr);
        if (!r || r == mm || r->type != GGML_TYPE_F32) return nullptr;
        if (!ggml_are_same_shape(mm, add) || !ggml_are_same_shape(r, add) || !ggml_is_contiguous(add) || !ggml_is_contiguous(r)) return nullptr;
        const size_t nb = ggml_nbytes(add);
        if (ranges_overlap(r->data, nb, mm->data, nb)) return nullptr;
        if (add->data != mm->data && ranges_overlap(add->data, nb, mm->data, nb)) return nullptr;
        if (ranges_overlap(add->data, nb, r->data, nb)) return nullptr;       // (in place on the residual: left to the separate launches)
        return r;
    };
    const ggml_tensor * r0 = add_of(add0, mm0);
    if (!r0) return false;
    const ggml_tensor * w1 = decode_norm_pattern(cgraph, i + 2, done);
    if (!w1 || norm1->src[0] != add0) return false;
    const ggml_tensor * ma = nd[4], * mb = nd[5];
    if (!matvec_groupable(ma) || !matvec_groupable(mb) || ma->src[1] != mul1 || mb->src[1] != mul1) return false;
    if (glu->op != GGML_OP_GLU || !glu->src[1] || glu->type != GGML_TYPE_F32) return false;
    const ggml_tensor * gate = glu->src[0], * up = glu->src[1];
    if (!((gate == ma && up == mb) || (gate == mb && up == ma))) return false;
    const int op = (int)ggml_get_glu_op(glu);
    if (op < 0 || op > 2) return false;
    if (!matvec_groupable(mm2) || mm2->src[1] != glu) return false;
    const ggml_tensor * r2 = add_of(add2, mm2);
    if (r2 != add0) return false;
    const ggml_tensor * wo = mm0->src[0], * wg = gate->src[0], * wu = up->src[0], * wd = mm2->src[0];
    const int64_t n_embd = wo->ne[1], k_o = wo->ne[0], n_ff = wg->ne[1];
    if (wg->type != wu->type || !ggml_are_same_shape(wg, wu) || wg->nb[1] != wu->nb[1] || wg->ne[0] != n_embd || wd->ne[0] != n_ff || wd->ne[1] != n_embd) return false;
    if (ggml_nrows(glu) != 1 || glu->ne[0] != n_ff || !ggml_is_contiguous(glu) || mul1->ne[0] != n_embd) return false;
    // the GLU result: on one of its inputs (then that input is not stored) or apart from both, as in mi355_try_fuse_glu_matvec
    float * gate_out = (float *)gate->data, * up_out = (float *)up->data;
    if (glu->data == gate->data) gate_out = nullptr;
    else if (ranges_overlap(glu->data, ggml_nbytes(glu), gate->data, ggml_nbytes(gate))) return false;
    if (glu->data == up->data) up_out = nullptr;
    else if (ranges_overlap(glu->data, ggml_nbytes(glu), up->data, ggml_nbytes(up))) return false;
    if (ranges_overlap(gate->data, ggml_nbytes(gate), up->data, ggml_nbytes(up))) return false;
    {   // what all workgroups read while others store (per phase): ffn_inp + w1 against ffn_norm / gate / up / hidden
        const ggml_tensor * outs[4] = {mul1, glu, nullptr, nullptr};
        size_t ob[4] = {ggml_nbytes(mul1), ggml_nbytes(glu), 0, 0};
        int no = 2;
        if (gate_out) { outs[no] = gate; ob[no++] = ggml_nbytes(gate); }
        if (up_out)   { outs[no] = up;   ob[no++] = ggml_nbytes(up); }
        if (!norm_prologue_ok(add0, w1, n_embd, 8192, outs, ob, no)) return false;
    }
    // hidden is read by everybody during the down phase, ffn_inp by the stores of l_out
    if (ranges_overlap(glu->data, ggml_nbytes(glu), add2->data, ggml_nbytes(add2)) || ranges_overlap(glu->data, ggml_nbytes(glu), mm2->data, ggml_nbytes(mm2))) return false;
    if (ranges_overlap(glu->data, ggml_nbytes(glu), add0->data, ggml_nbytes(add0))) return false;

    mi355_decode_layer_args a;
    memset(&a, 0, sizeof(a));
    // ---- the tail: the next layer's attn_norm + Q/K/V + rope + KV store
    qkv_nodes qn;
    bool tail = false, kv_q8 = false;
    const ggml_tensor * x3 = nullptr;
    void * const * kv_table = nullptr;
    int kv_slot = 0;
    if (i + 11 < n && !MI355_ENV_SET("GGML_MI355_NO_NORM_QKV")) {
        const ggml_tensor * norm3 = ggml_graph_node(cgraph, i + 9);
        const ggml_tensor * w3 = decode_norm_pattern(cgraph, i + 9, done);
        ggml_tensor * mq = ggml_graph_node(cgraph, i + 11);
        x3 = ggml_graph_node(cgraph, i + 10);
        if (w3 && norm3->src[0] == add2 && mq->op == GGML_OP_MUL_MAT && mq->src[1] == x3 && matvec_groupable(mq) && qkv_match(cgraph, i + 11, done, matvec_groupable, qn) && !qn.kv_q8) {
            const ggml_tensor * wq = qn.mq->src[0], * wk = qn.mk->src[0], * wv = qn.mv->src[0];
            const ggml_tensor * outs[4] = {x3, qn.rq, qn.ck, qn.cv};
            const size_t ob[4] = {ggml_nbytes(x3), ggml_nbytes(qn.rq), ggml_nbytes(qn.ck), ggml_nbytes(qn.cv)};
            if (wq->ne[0] == n_embd && norm_prologue_ok(add2, w3, n_embd, 8192, outs, ob, 4)) {
                mi355_qkv_rope_args & q = a.qkv;
                q.norm_x = (const float *)add2->data; q.norm_w = (const float *)w3->data; q.norm_out = (float *)x3->data;
                memcpy(&q.norm_eps, norm3->op_params, sizeof(float));
                q.type_qk = (int32_t)wq->type; q.type_v = (int32_t)wv->type; q.type_k = wk->type != wq->type ? (int32_t)wk->type : 0;
                q.wq = wq->data; q.wk = wk->data; q.wv = wv->data;
                q.wq_row_stride = (int64_t)wq->nb[1]; q.wk_row_stride = (int64_t)wk->nb[1]; q.wv_row_stride = (int64_t)wv->nb[1];
                q.m_q = wq->ne[1]; q.m_k = wk->ne[1]; q.m_v = wv->ne[1]; q.k = n_embd;
                q.q_out = (float *)qn.rq->data; q.k_out = qn.ck->data; q.v_out = qn.cv->data;
                q.k_out_type = (int32_t)qn.ck->type; q.v_out_type = (int32_t)qn.cv->type;
                if (ctx->gc.capturing) {
                    if (ctx->gc.n_slots >= MI355_KV_SLOTS) ctx->gc.capture_failed = true;
                    else { kv_table = ctx->gc.kv_dev; kv_slot = ctx->gc.n_slots; }
                }
                q.kv_table = kv_table; q.kv_slot = kv_slot;
                q.pos = (const int32_t *)qn.rq->src[1]->data; q.freq_factors = qn.rq->src[2] ? (const float *)qn.rq->src[2]->data : nullptr;
                q.head_dim = (int32_t)qn.rq->ne[0];
                float f6[6];
                qkv_rope_params(qn.rq, &q.n_dims, &q.mode, &q.n_ctx_orig, f6);
                q.freq_base = f6[0]; q.freq_scale = f6[1]; q.ext_factor = f6[2]; q.attn_factor = f6[3]; q.beta_fast = f6[4]; q.beta_slow = f6[5];
                tail = true; kv_q8 = qn.kv_q8;
            }
        }
    }
    (void)kv_q8;
    a.type_o = (int32_t)wo->type; a.type_gu = (int32_t)wg->type; a.type_d = (int32_t)wd->type; a.glu_op = op;
    a.tail = tail ? MI355_TAIL_QKV : MI355_TAIL_NONE;
    a.n_embd = n_embd; a.k_o = k_o; a.n_ff = n_ff;
    a.wo = wo->data; a.wo_row_stride = (int64_t)wo->nb[1];
    const ggml_tensor * src1 = mm0->src[1];
    const size_t qb = align_up((size_t)mi355_act_q_bytes(k_o, 1), 256);
    if (ctx->act_cached(src1)) { a.attn_q = (const int8_t *)ctx->scratch[0]; a.attn_ds = (const float *)((const char *)ctx->scratch[0] + qb); }
    else {
        if (src1->type != GGML_TYPE_F32 || !ggml_is_contiguous(src1) || ((uintptr_t)src1->data & 15)) return false;
        a.attn_f32 = (const float *)src1->data;
    }
    a.residual = (const float *)r0->data; a.attn_out = (float *)mm0->data; a.ffn_inp = (float *)add0->data;
    a.ffn_norm_w = (const float *)w1->data; a.ffn_norm_out = (float *)mul1->data;
    memcpy(&a.norm_eps, norm1->op_params, sizeof(float));
    a.w_gate = wg->data; a.w_up = wu->data; a.gu_row_stride = (int64_t)wg->nb[1];
    a.gate_out = gate_out; a.up_out = up_out; a.hidden = (float *)glu->data;
    a.w_down = wd->data; a.down_row_stride = (int64_t)wd->nb[1];
    a.down_out = (float *)mm2->data; a.l_out = (float *)add2->data;
    // a sync block of this graph's memset
    const size_t sb = (size_t)mi355_decode_layer_sync_bytes();
    if (ctx->dl_used >= MI355_DL_MAX) return false;
    char * sync = (char *)ctx->get_scratch(5, sb * MI355_DL_MAX);
    if (ctx->scratch_size[5] < sb * MI355_DL_MAX) return false;          // under capture the buffer could not grow
    if (ctx->dl_used == 0) HIP_CHECK(hipMemsetAsync(sync, 0, sb * MI355_DL_MAX, ctx->stream));
    int rc = mi355_decode_layer(&a, sync + sb * ctx->dl_used, ctx->stream);
    if (rc == MI355_E_UNSUPPORTED && tail) {                             // e.g. a Q/K/V shape the fused tail does not deal: the body alone
        tail = false; a.tail = MI355_TAIL_NONE;
        rc = mi355_decode_layer(&a, sync + sb * ctx->dl_used, ctx->stream);
    }
    if (rc == MI355_E_UNSUPPORTED || rc == MI355_E_SHAPE || rc == MI355_E_ALIGN) return false;
    MI_CHECK(rc);
    ctx->dl_used++;
    for (int j = 1; j < 9; j++) done[i + j] = 1;
    if (tail) {
        done[i + 9] = 1; done[i + 10] = 1;
        for (int m = 0; m < 7; m++) done[qn.idx[m]] = 1;
        if (kv_table) { ctx->gc.kv_push(qn.idx[5], 0); ctx->gc.kv_push(qn.idx[6], 0); ctx->gc.n_slots++; }
    }
    ctx->act_cache.valid = false;        // nothing of what scratch[0] held describes a tensor the following nodes read
    return true;
}

// issue the launches of a graph on ctx->stream (eagerly, or into the stream capture in progress)
static enum ggml_status mi355_issue_graph(mi355_backend_ctx * ctx, ggml_cgraph * cgraph) {
    ctx->act_cache.valid = false; // tensors may have been rewritten through set_tensor / copies between calls
    ctx->dl_used = 0;
    const int n = ggml_graph_n_nodes(cgraph);
    const bool fuse = !MI355_ENV_SET("GGML_MI355_NO_FUSE");
    if (MI355_ENV_SET("GGML_MI355_DUMP_GRAPH")) {   // diagnostics: the node list this backend was handed
        for (int i = 0; i < n; i++) {
            const ggml_tensor * t = ggml_graph_node(cgraph, i);
            fprintf(stderr, "[mi355] %4d %-14s %-28s [%lld,%lld,%lld,%lld] %p", i, ggml_op_name(t->op), t->name, (long long)t->ne[0],
                    (long long)t->ne[1], (long long)t->ne[2], (long long)t->ne[3], t->data);
            for (int s2 = 0; s2 < GGML_MAX_SRC; s2++) if (t->src[s2]) fprintf(stderr, "  <- %s(%p)", t->src[s2]->name, t->src[s2]->data);
            fprintf(stderr, "\n");
        }
    }
    std::vector<char> done(n, 0);    // nodes already computed by a hoisting fusion
    for (int i = 0; i < n; i++) {
        ggml_tensor * node = ggml_graph_node(cgraph, i);
        if (done[i]) continue;
        // empty tensors and pure view ops compute nothing (ggml-cuda.cu:2589)
        if (ggml_is_empty(node) || node->op == GGML_OP_NONE || node->op == GGML_OP_RESHAPE || node->op == GGML_OP_VIEW ||
            node->op == GGML_OP_PERMUTE || node->op == GGML_OP_TRANSPOSE) {
            continue;
        }
        // fusion: residual ADD -> RMS_NORM -> MUL(weight) in one launch (the add result stays a graph tensor)
        if (node->op == GGML_OP_ADD && i + 2 < n && ggml_graph_node(cgraph, i + 1)->op == GGML_OP_RMS_NORM &&
            ggml_graph_node(cgraph, i + 1)->src[0] == node && mi355_try_fuse_rms_norm_mul(ctx, cgraph, i + 1, node)) {
            i += 2;
            continue;
        }
        ctx->act_invalidate_if_written(node);
        // fusion: Q/K/V projections + rope + KV-cache store of one token in one launch
        if (fuse && node->op == GGML_OP_MUL_MAT && node->src[0]->type == GGML_TYPE_F32 && mi355_try_fuse_moe_router(ctx, cgraph, i, done)) continue;
        if (fuse && node->op == GGML_OP_MUL_MAT && mi355_try_fuse_decode_layer(ctx, cgraph, i, done)) continue;
        if (fuse && node->op == GGML_OP_MUL_MAT && mi355_try_fuse_qkv_rope(ctx, cgraph, i, done)) continue;
        if (fuse && node->op == GGML_OP_MUL_MAT && mi355_try_fuse_qkv_gemm(ctx, cgraph, i, done)) continue;
        if (fuse && node->op == GGML_OP_MUL_MAT && mi355_try_fuse_qkv_small(ctx, cgraph, i, done)) continue;
        if (fuse && node->op == GGML_OP_MUL_MAT && mi355_try_group_small(ctx, cgraph, i, done)) continue;
        if (fuse && node->op == GGML_OP_MUL_MAT && mi355_try_fuse_gemm_add_norm(ctx, cgraph, i, done)) {
            i += 3;
            continue;
        }
        if (fuse && node->op == GGML_OP_MUL_MAT && mi355_try_fuse_glu_gemm(ctx, cgraph, i, done)) {
            done[i + 1] = 1;
            done[i + 2] = 1;
            continue;
        }
        if (fuse && node->op == GGML_OP_MUL_MAT_ID && mi355_try_fuse_moe_glu_matvec(ctx, cgraph, i, done)) {
            done[i + 1] = 1;
            done[i + 2] = 1;
            continue;
        }
        if (fuse && node->op == GGML_OP_MUL_MAT_ID && mi355_try_fuse_moe_down(ctx, cgraph, i, done)) continue;
        // fusion: decode mat-vecs that share their activation run as one grouped launch
        if (fuse && matvec_groupable(node)) {
            int members[MI355_MMVQ_MAX_GROUP];
            const int cnt = mi355_collect_matvec_group(cgraph, i, done, members);
            if (cnt == 2 && mi355_try_fuse_glu_matvec(ctx, cgraph, members)) {
                done[members[1]] = 1;
                done[members[1] + 1] = 1;
                continue;
            }
            if (cnt > 1) {
                mi355_mul_mat_group(ctx, cgraph, members, cnt);
                for (int m = 1; m < cnt; m++) done[members[m]] = 1;
                continue;
            }
            if (mi355_try_fuse_matvec_add(ctx, cgraph, i)) {
                done[i + 1] = 1;
                continue;
            }
        }
        // fusion: RMS_NORM followed by MUL with the per-feature norm weight (every llama layer), when the
        // norm result has no other reader in this graph
        if (fuse && node->op == GGML_OP_RMS_NORM && mi355_try_fuse_norm_glu_matvec(ctx, cgraph, i, done)) {
            i += 4;
            continue;
        }
        if (fuse && node->op == GGML_OP_RMS_NORM && i + 2 < n && !MI355_ENV_SET("GGML_MI355_NO_NORM_QKV") && decode_norm_pattern(cgraph, i, done) &&
            ggml_graph_node(cgraph, i + 2)->op == GGML_OP_MUL_MAT && ggml_graph_node(cgraph, i + 2)->src[1] == ggml_graph_node(cgraph, i + 1) &&
            mi355_try_fuse_qkv_rope(ctx, cgraph, i + 2, done, i)) {
            i += 2;      // the norm, its MUL and wq; the other six nodes are marked done
            continue;
        }
        {
            int consumed = 0;
            if (fuse && node->op == GGML_OP_RMS_NORM && mi355_try_fuse_norm_group(ctx, cgraph, i, done, &consumed)) {
                i += consumed;
                continue;
            }
        }
        if (node->op == GGML_OP_RMS_NORM && i + 1 < n && mi355_try_fuse_rms_norm_mul(ctx, cgraph, i)) {
            i++;
            continue;
        }
        if (!mi355_compute_forward(ctx, node, i + 1 < n ? ggml_graph_node(cgraph, i + 1) : nullptr)) {
            MI_LOG_ERROR("%s: op not supported %s (%s)\n", __func__, node->name, ggml_op_name(node->op));
            return GGML_STATUS_FAILED;
        }
    }
    return GGML_STATUS_SUCCESS;
}

// ---- decode graphs: signature, capture, replay -------------------------------------------------------------------
static bool node_computes(const ggml_tensor * t) {
    return !(ggml_is_empty(t) || t->op == GGML_OP_NONE || t->op == GGML_OP_RESHAPE || t->op == GGML_OP_VIEW || t->op == GGML_OP_PERMUTE ||
             t->op == GGML_OP_TRANSPOSE);
}

// A graph is a candidate when it is a single-token step on this device alone: every MUL_MAT / MUL_MAT_ID has one
// activation column (prefill graphs change shape from call to call and amortise their launches anyway) and no tensor
// lives in a row-split buffer (helper streams and cross-device events are not captured; ggml-cuda.cu:2593-2598 likewise).
static bool mi355_split_graphs_ok() {
    static const bool ok = [] {
        if (MI355_ENV_SET("GGML_MI355_NO_SPLIT_GRAPH") || MI355_ENV_SET("GGML_MI355_NO_SPLIT_FUSE") || mi355_split_gather_rccl()) return false;
        if (MI355_ENV_SET("GGML_MI355_SPLIT_GRAPH")) return true;
        for (int d = 1; d < mi355_device_count(); d++) if (mi355_phys(d) != mi355_phys(0)) return false;
        return true;
    }();
    return ok;
}

static bool mi355_graph_capturable(ggml_cgraph * cgraph) {
    const int n = ggml_graph_n_nodes(cgraph);
    if (n < 8) return false;
    bool any = false;
    for (int i = 0; i < n; i++) {
        const ggml_tensor * t = ggml_graph_node((ggml_cgraph *)cgraph, i);
        if (t->op == GGML_OP_MUL_MAT || t->op == GGML_OP_MUL_MAT_ID) {
            // GGML_MI355_GRAPH_COLS=n (lab switch): also steps of up to n tokens (batched / speculative decoding: the same small graph
            // step after step) -- default 1
            static const int64_t max_cols = [] { const char * e = getenv("GGML_MI355_GRAPH_COLS"); return e ? atoll(e) : 1; }();
            const int64_t cols = ggml_nelements(t->src[1]) / (t->src[1]->ne[0] * (t->op == GGML_OP_MUL_MAT_ID ? t->src[1]->ne[1] : 1));
            if (cols > max_cols) return false;
            if (t->op == GGML_OP_MUL_MAT && ggml_is_quantized(t->src[0]->type)) any = true;
        }
        // Row-split weights: the token's launches fork onto helper streams and join again (mi355_split_fan_out), which a capture
        // records like any other dependency.  Captured when every ggml device is the same physical GPU (virtual devices: measured
        // and tested) or on request (GGML_MI355_SPLIT_GRAPH=1): a graph whose kernel nodes sit on several physical GPUs could not
        // be tried on the one-GPU development box.  The reference never captures split graphs (ggml-cuda.cu:2593-2598).
        for (int k = 0; k < GGML_MAX_SRC; k++) if (t->src[k] && tensor_is_split(t->src[k]) && !mi355_split_graphs_ok()) return false;
    }
    return any;
}

static void mi355_node_signature(const ggml_tensor * t, mi355_backend_ctx::node_sig & s) {
    s.op = (int32_t)t->op; s.type = (int32_t)t->type; s.flags = t->flags; s.data = t->data; s.kv_dst = false;
    for (int k = 0; k < 4; k++) { s.ne[k] = t->ne[k]; s.nb[k] = t->nb[k]; }
    for (int k = 0; k < GGML_MAX_SRC; k++) s.src[k] = t->src[k] ? t->src[k]->data : nullptr;
    memcpy(s.params, t->op_params, sizeof(s.params));
}

static void mi355_graph_signature(ggml_cgraph * cgraph, std::vector<mi355_backend_ctx::node_sig> & sig) {
    const int n = ggml_graph_n_nodes(cgraph);
    sig.resize(n);
    for (int i = 0; i < n; i++) mi355_node_signature(ggml_graph_node((ggml_cgraph *)cgraph, i), sig[i]);
}

// same graph as the cached one?  View nodes compute nothing (their addresses count through their consumers' sources);
// a CPY whose destination goes through the table may move.
// `captured`: the cached signature belongs to an instantiated graph (only the CPYs it routed through the table may move);
// before that every CPY destination may move -- whether the launches can follow it is decided at capture time.
// (the graph's signature is taken node by node on the stack: the replay path -- every decode token -- allocates nothing)
static bool mi355_graph_same(const std::vector<mi355_backend_ctx::node_sig> & cached, ggml_cgraph * cgraph, bool captured) {
    static const bool dbg = getenv("GGML_MI355_GRAPH_DEBUG") != nullptr;
    const size_t n = (size_t)ggml_graph_n_nodes(cgraph);
    if (n != cached.size()) { if (dbg) fprintf(stderr, "[mi355] graph: %zu nodes, cached %zu\n", n, cached.size()); return false; }
    for (size_t i = 0; i < n; i++) {
        const ggml_tensor * t = ggml_graph_node(cgraph, (int)i);
        mi355_backend_ctx::node_sig x;
        mi355_node_signature(t, x);
        const mi355_backend_ctx::node_sig & c = cached[i];
        bool ok = x.op == c.op && x.type == c.type && !memcmp(x.ne, c.ne, sizeof(x.ne)) && !memcmp(x.nb, c.nb, sizeof(x.nb));
        // view nodes compute nothing: where they point (their op_params hold the offset) counts through their consumers' sources
        if (ok && node_computes(t)) ok = x.flags == c.flags && !memcmp(x.params, c.params, sizeof(x.params));
        if (ok && node_computes(t)) {
            if (captured ? c.kv_dst : x.op == (int32_t)GGML_OP_CPY) ok = x.src[0] == c.src[0];   // destination (data, src[1]) free to move
            else ok = x.data == c.data && !memcmp(x.src, c.src, sizeof(x.src));
        }
        if (!ok) {
            if (dbg) fprintf(stderr, "[mi355] graph differs at node %zu %s (%s): data %p / %p, ne0 %lld / %lld\n", i, ggml_op_name(t->op), t->name,
                             x.data, c.data, (long long)x.ne[0], (long long)c.ne[0]);
            return false;
        }
    }
    return true;
}

static void mi355_graph_refresh_kv_table(mi355_backend_ctx * ctx, ggml_cgraph * cgraph) {
    mi355_backend_ctx::graph_cache_t & gc = ctx->gc;
    if (gc.kv_nodes.empty()) return;
    void ** h = gc.kv_host + (size_t)gc.ring * MI355_KV_SLOTS * 2;
    hipEvent_t & ev = gc.ring_ev[gc.ring];
    // a caller may enqueue more than MI355_KV_RING decode graphs without a host sync (pipelined decode): the copy that read
    // this slot must have run before the slot is rewritten
    if (ev) HIP_CHECK(hipEventSynchronize(ev));
    else    HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    gc.ring = (gc.ring + 1) % MI355_KV_RING;
    for (size_t j = 0; j < gc.kv_nodes.size(); j++) h[j] = (char *)ggml_graph_node(cgraph, gc.kv_nodes[j])->data + gc.kv_off[j];
    HIP_CHECK(hipMemcpyAsync(gc.kv_dev, h, gc.kv_nodes.size() * sizeof(void *), hipMemcpyHostToDevice, ctx->stream));
    HIP_CHECK(hipEventRecord(ev, ctx->stream));
}

static enum ggml_status mi355_backend_graph_compute(ggml_backend_t backend, ggml_cgraph * cgraph) {
    mi355_backend_ctx * ctx = (mi355_backend_ctx *)backend->context;
    mi355_set_device(ctx->device);
    const int n = ggml_graph_n_nodes(cgraph);
    static const bool time_graph = getenv("GGML_MI355_TIME_GRAPH") != nullptr;   // diagnostics: host time spent issuing a graph
    struct host_timer {
        bool on; int n; std::chrono::steady_clock::time_point t0;
        ~host_timer() {
            if (!on) return;
            static double tot = 0, between = 0; static long calls = 0;
            static std::chrono::steady_clock::time_point last_end;
            const auto now = std::chrono::steady_clock::now();
            tot += std::chrono::duration<double, std::micro>(now - t0).count();
            if (calls) between += std::chrono::duration<double, std::micro>(t0 - last_end).count();
            last_end = now; calls++;
            if (calls % 64 == 0) {
                fprintf(stderr, "[mi355] graph_compute host time: %.1f us avg over 64 calls (%d nodes), %.1f us between calls\n", tot / 64, n, between / 64);
                fprintf(stderr, "[mi355]   per graph: %.1f set_tensor calls %.1f us %.0f bytes, %.1f get_tensor calls %.1f us %.0f bytes\n",
                        g_copy_stats.n[0] / 64.0, g_copy_stats.us[0] / 64, g_copy_stats.bytes[0] / 64.0, g_copy_stats.n[1] / 64.0, g_copy_stats.us[1] / 64, g_copy_stats.bytes[1] / 64.0);
                g_copy_stats = mi355_copy_stats();
                tot = between = 0;
            }
        }
    } timer{time_graph, n, std::chrono::steady_clock::now()};
    mi355_backend_ctx::graph_cache_t & gc = ctx->gc;
    // GGML_MI355_KERNEL_TIMES: every launch carries an event pair (kernels/timing.h), which a captured graph would not replay
    static const bool graphs_off = getenv("GGML_MI355_NO_GRAPHS") != nullptr || MI355_ENV_SET("GGML_MI355_KERNEL_TIMES");
    if (graphs_off || gc.disabled || !mi355_graph_capturable(cgraph)) return mi355_issue_graph(ctx, cgraph);

    const bool same = mi355_graph_same(gc.sig, cgraph, gc.exec != nullptr);
    if (same && gc.exec) {                         // replay
        gc.recaptures = 0; gc.n_replay++;
        mi355_graph_refresh_kv_table(ctx, cgraph);
        HIP_CHECK(hipGraphLaunch(gc.exec, ctx->stream));
        return GGML_STATUS_SUCCESS;
    }
    if (!same) {
        // first sighting of this graph: run it eagerly (this also sizes the scratch buffers); a second identical one is captured
        if (gc.exec) { HIP_CHECK(hipGraphExecDestroy(gc.exec)); gc.exec = nullptr; }
        mi355_graph_signature(cgraph, gc.sig);
        gc.kv_clear();
        if (++gc.recaptures > 8) {                 // a graph that never settles (ggml-cuda.cu:2674-2690 gives up the same way)
            gc.disabled = true;
            MI_LOG_INFO("%s: decode graph keeps changing, hipGraph replay disabled for this backend\n", __func__);
        }
        gc.n_eager++;
        return mi355_issue_graph(ctx, cgraph);
    }
    // second identical graph: capture
    std::vector<mi355_backend_ctx::node_sig> sig;
    mi355_graph_signature(cgraph, sig);
    if (!gc.kv_dev) {
        HIP_CHECK(hipMalloc((void **)&gc.kv_dev, MI355_KV_SLOTS * 2 * sizeof(void *)));
        HIP_CHECK(hipHostMalloc((void **)&gc.kv_host, (size_t)MI355_KV_RING * MI355_KV_SLOTS * 2 * sizeof(void *), hipHostMallocDefault));
    }
    gc.kv_clear(); gc.n_slots = 0; gc.capture_failed = false;
    HIP_CHECK(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
    gc.capturing = true;
    const enum ggml_status st = mi355_issue_graph(ctx, cgraph);
    gc.capturing = false;
    hipGraph_t graph = nullptr;
    HIP_CHECK(hipStreamEndCapture(ctx->stream, &graph));
    if (st != GGML_STATUS_SUCCESS || gc.capture_failed || !graph) {
        if (graph) HIP_CHECK(hipGraphDestroy(graph));
        gc.kv_clear();
        if (st != GGML_STATUS_SUCCESS) return st;
        if (++gc.recaptures > 8) gc.disabled = true;
        gc.n_eager++;
        return mi355_issue_graph(ctx, cgraph);     // e.g. a scratch buffer had to grow: eager now, capture next time
    }
    HIP_CHECK(hipGraphInstantiate(&gc.exec, graph, nullptr, nullptr, 0));
    HIP_CHECK(hipGraphDestroy(graph));
    gc.n_capture++;
    // the signature kept is the CAPTURED graph's (this one), not the first sighting's: a CPY that did not go through the
    // table has this graph's destination baked in, and only an identical destination may replay it
    gc.sig.swap(sig);
    for (int j : gc.kv_nodes) gc.sig[j].kv_dst = true;
    mi355_graph_refresh_kv_table(ctx, cgraph);
    HIP_CHECK(hipGraphLaunch(gc.exec, ctx->stream));
    return GGML_STATUS_SUCCESS;
}

static void mi355_backend_event_record(ggml_backend_t backend, ggml_backend_event_t event) {
    mi355_backend_ctx * ctx = (mi355_backend_ctx *)backend->context;
    mi355_set_device(ctx->device);
    HIP_CHECK(hipEventRecord((hipEvent_t)event->context, ctx->stream));
}

static void mi355_backend_event_wait(ggml_backend_t backend, ggml_backend_event_t event) {
    mi355_backend_ctx * ctx = (mi355_backend_ctx *)backend->context;
    mi355_set_device(ctx->device);
    HIP_CHECK(hipStreamWaitEvent(ctx->stream, (hipEvent_t)event->context, 0));
}

static const ggml_backend_i mi355_backend_iface = {
    /* .get_name           = */ mi355_backend_get_name,
    /* .free               = */ mi355_backend_free,
    /* .set_tensor_async   = */ mi355_backend_set_tensor_async,
    /* .get_tensor_async   = */ mi355_backend_get_tensor_async,
    /* .cpy_tensor_async   = */ mi355_backend_cpy_tensor_async,
    /* .synchronize        = */ mi355_backend_synchronize,
    /* .graph_plan_create  = */ NULL,
    /* .graph_plan_free    = */ NULL,
    /* .graph_plan_update  = */ NULL,
    /* .graph_plan_compute = */ NULL,
    /* .graph_compute      = */ mi355_backend_graph_compute,
    /* .event_record       = */ mi355_backend_event_record,
    /* .event_wait         = */ mi355_backend_event_wait,
};

// ------------------------------------------------------------------------------------------------
// device interface
// ------------------------------------------------------------------------------------------------
static const char * mi355_dev_get_name(ggml_backend_dev_t dev) { return ((mi355_device_ctx *)dev->context)->name.c_str(); }
static const char * mi355_dev_get_description(ggml_backend_dev_t dev) { return ((mi355_device_ctx *)dev->context)->description.c_str(); }

static void mi355_dev_get_memory(ggml_backend_dev_t dev, size_t * free, size_t * total) {
    mi355_set_device(((mi355_device_ctx *)dev->context)->device);
    HIP_CHECK(hipMemGetInfo(free, total));
}

static enum ggml_backend_dev_type mi355_dev_get_type(ggml_backend_dev_t) { return GGML_BACKEND_DEVICE_TYPE_GPU; }

static void mi355_dev_get_props(ggml_backend_dev_t dev, ggml_backend_dev_props * props) {
    props->name        = mi355_dev_get_name(dev);
    props->description = mi355_dev_get_description(dev);
    props->type        = mi355_dev_get_type(dev);
    mi355_dev_get_memory(dev, &props->memory_free, &props->memory_total);
    props->caps = {
        /* .async                = */ true,
        /* .host_buffer          = */ getenv("GGML_MI355_NO_PINNED") == nullptr,
        /* .buffer_from_host_ptr = */ false,
        /* .events               = */ true,
    };
}

static ggml_backend_t mi355_dev_init_backend(ggml_backend_dev_t dev, const char *) {
    mi355_device_ctx * dctx = (mi355_device_ctx *)dev->context;
    mi355_set_device(dctx->device);
    mi355_backend_ctx * ctx = new mi355_backend_ctx;
    ctx->device = dctx->device;
    ctx->name   = dctx->name;
    HIP_CHECK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    if (MI355_ENV_SET("GGML_MI355_KERNEL_TIMES")) mi355_timing_enable(1);
    ggml_backend_t backend = new ggml_backend{
        /* .guid    = */ mi355_guid(),
        /* .iface   = */ mi355_backend_iface,
        /* .device  = */ dev,
        /* .context = */ ctx,
    };
    return backend;
}

static ggml_backend_buffer_type_t mi355_dev_get_buffer_type(ggml_backend_dev_t dev) {
    return mi355_buffer_type(((mi355_device_ctx *)dev->context)->device);
}

static ggml_backend_buffer_type_t mi355_dev_get_host_buffer_type(ggml_backend_dev_t) { return mi355_host_buffer_type(); }

static bool mi355_dev_supports_op(ggml_backend_dev_t dev, const ggml_tensor * op) {
    const int device = ((mi355_device_ctx *)dev->context)->device;
    // every source must live on this device (ggml-cuda.cu:3088-3096); buffers may be absent while
    // the scheduler is still assigning
    for (int i = 0; i < GGML_MAX_SRC; i++) {
        const ggml_tensor * s = op->src[i];
        if (!s) continue;
        ggml_backend_buffer_t buf = s->view_src ? s->view_src->buffer : s->buffer;
        if (buf && buft_is_mi355(buf->buft) && ((mi355_buffer_ctx *)buf->context) &&
            buf->size > 0 && ((mi355_buffer_ctx *)buf->context)->device != device) {
            return false;
        }
    }
    // a row-split tensor can only be the weight of a MUL_MAT (ggml-cuda.cu:3079-3086)
    for (int i = 0; i < GGML_MAX_SRC; i++) {
        if (tensor_is_split(op->src[i]) && !(op->op == GGML_OP_MUL_MAT && i == 0)) return false;
    }
    switch (op->op) {
        case GGML_OP_NONE:
        case GGML_OP_RESHAPE:
        case GGML_OP_VIEW:
        case GGML_OP_PERMUTE:
        case GGML_OP_TRANSPOSE:
            return true;
        case GGML_OP_MUL_MAT: {
            const ggml_tensor * a = op->src[0];
            const ggml_tensor * b = op->src[1];
            if (tensor_is_split(a)) {
                // 2-D weights only, and this device must own rows (ggml-cuda.cu:3137-3150)
                if (a->ne[2] > 1 || a->ne[3] > 1) return false;
                int64_t lo, hi;
                mi355_row_split(((mi355_split_buft_ctx *)a->buffer->buft->context)->split, ggml_nrows(a), device, &lo, &hi);
                if (lo == hi) return false;
            }
            if (is_float_weight_type(a->type)) {
                // float weights: the MoE router (small f32), an unquantized model, K / V without flash attention (mmf.hip,
                // mmdq.hip); rows contiguous, broadcast over dims 2 / 3, src1 f32 or f16
                if (tensor_is_split(a) || op->type != GGML_TYPE_F32 || (b->type != GGML_TYPE_F32 && b->type != GGML_TYPE_F16)) return false;
                if (a->nb[0] != ggml_type_size(a->type) || b->nb[0] != ggml_type_size(b->type) || op->nb[0] != sizeof(float)) return false;
                if (b->ne[2] % a->ne[2] != 0 || b->ne[3] % a->ne[3] != 0 || b->ne[2] * b->ne[3] > 65535) return false;
                if ((a->nb[1] | a->nb[2] | a->nb[3]) & (ggml_type_size(a->type) - 1)) return false;
                if ((b->nb[1] | b->nb[2] | b->nb[3]) & (ggml_type_size(b->type) - 1)) return false;
                if ((op->nb[1] | op->nb[2] | op->nb[3]) & 3) return false;
                return true;
            }
            if (!is_block_weight_type(a->type) || b->type != GGML_TYPE_F32 || op->type != GGML_TYPE_F32) return false;
            if (a->nb[0] != ggml_type_size(a->type) || b->nb[0] != sizeof(float)) return false; // rows contiguous
            if (ggml_is_transposed(a) || !ggml_is_contiguous(op)) return false;
            if (b->ne[2] % a->ne[2] != 0 || b->ne[3] % a->ne[3] != 0) return false;
            if (b->ne[2] > 65535) return false;
            // alignment the 128-bit loads rely on
            if ((b->nb[1] | b->nb[2] | b->nb[3]) & 3) return false;
            if ((a->type == GGML_TYPE_Q4_K || a->type == GGML_TYPE_Q5_K) && ((a->nb[1] | a->nb[2] | a->nb[3]) & 15)) return false;
            return true;
        }
        case GGML_OP_MUL_MAT_ID: {
            const ggml_tensor * a = op->src[0];
            const ggml_tensor * b = op->src[1];
            const ggml_tensor * ids = op->src[2];
            if (!is_block_weight_type(a->type) || b->type != GGML_TYPE_F32 || op->type != GGML_TYPE_F32 || ids->type != GGML_TYPE_I32) return false;
            if (a->nb[0] != ggml_type_size(a->type) || b->nb[0] != sizeof(float) || ids->nb[0] != sizeof(int32_t)) return false;
            if (!ggml_is_contiguous(op) || a->ne[3] != 1 || b->ne[3] != 1) return false;
            if (b->ne[1] != 1 && b->ne[1] != ids->ne[0]) return false;
            if (ids->ne[0] > 65535) return false;
            if ((a->type == GGML_TYPE_Q4_K || a->type == GGML_TYPE_Q5_K) && ((a->nb[1] | a->nb[2]) & 15)) return false;
            return true;
        }
        case GGML_OP_FLASH_ATTN_EXT:
            return mi355_flash_attn_supported(op);
        default:
            return mi355_ops_supported(op);
    }
}

static bool mi355_dev_supports_buft(ggml_backend_dev_t dev, ggml_backend_buffer_type_t buft) {
    if (buft_is_mi355_split(buft)) return buft->device == dev;   // ggml-cuda.cu:3433
    if (!buft_is_mi355(buft)) return false;
    return ((mi355_buft_ctx *)buft->context)->device == ((mi355_device_ctx *)dev->context)->device;
}

static int64_t op_batch_size(const ggml_tensor * op) {
    switch (op->op) {
        case GGML_OP_GET_ROWS:   return 0;
        case GGML_OP_MUL_MAT:    return op->ne[1];
        case GGML_OP_MUL_MAT_ID:
        case GGML_OP_ROPE:       return op->ne[2];
        default:                 return ggml_nrows(op);
    }
}

// weights in host memory: worth copying to the GPU only for large batches (ggml-cuda.cu:3436-3457)
static bool mi355_dev_offload_op(ggml_backend_dev_t, const ggml_tensor * op) { return op_batch_size(op) >= 32; }

static ggml_backend_event_t mi355_dev_event_new(ggml_backend_dev_t dev) {
    mi355_set_device(((mi355_device_ctx *)dev->context)->device);
    hipEvent_t ev;
    HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    return new ggml_backend_event{dev, ev};
}

static void mi355_dev_event_free(ggml_backend_dev_t, ggml_backend_event_t event) {
    HIP_CHECK(hipEventDestroy((hipEvent_t)event->context));
    delete event;
}

static void mi355_dev_event_synchronize(ggml_backend_dev_t, ggml_backend_event_t event) {
    HIP_CHECK(hipEventSynchronize((hipEvent_t)event->context));
}

static const ggml_backend_device_i mi355_device_iface = {
    /* .get_name             = */ mi355_dev_get_name,
    /* .get_description      = */ mi355_dev_get_description,
    /* .get_memory           = */ mi355_dev_get_memory,
    /* .get_type             = */ mi355_dev_get_type,
    /* .get_props            = */ mi355_dev_get_props,
    /* .init_backend         = */ mi355_dev_init_backend,
    /* .get_buffer_type      = */ mi355_dev_get_buffer_type,
    /* .get_host_buffer_type = */ mi355_dev_get_host_buffer_type,
    /* .buffer_from_host_ptr = */ NULL,
    /* .supports_op          = */ mi355_dev_supports_op,
    /* .supports_buft        = */ mi355_dev_supports_buft,
    /* .offload_op           = */ mi355_dev_offload_op,
    /* .event_new            = */ mi355_dev_event_new,
    /* .event_free           = */ mi355_dev_event_free,
    /* .event_synchronize    = */ mi355_dev_event_synchronize,
};

// ------------------------------------------------------------------------------------------------
// registry
// ------------------------------------------------------------------------------------------------
struct mi355_reg_ctx {
    std::vector<ggml_backend_dev_t> devices;
};

static const char * mi355_reg_get_name(ggml_backend_reg_t) { return MI355_NAME; }
static size_t mi355_reg_get_device_count(ggml_backend_reg_t reg) { return ((mi355_reg_ctx *)reg->context)->devices.size(); }

static ggml_backend_dev_t mi355_reg_get_device(ggml_backend_reg_t reg, size_t index) {
    mi355_reg_ctx * ctx = (mi355_reg_ctx *)reg->context;
    GGML_ASSERT(index < ctx->devices.size());
    return ctx->devices[index];
}

static ggml_backend_feature * mi355_get_features(ggml_backend_reg_t) {
    static ggml_backend_feature features[] = {
        {"ARCH", "gfx950"}, {"WAVE", "64"}, {"DOT4_I8", "1"}, {"MFMA_I8", "1"}, {"NO_CUDA_COMPAT", "1"}, {nullptr, nullptr},
    };
    return features;
}

static void * mi355_reg_get_proc_address(ggml_backend_reg_t, const char * name) {
    if (strcmp(name, "ggml_backend_get_features") == 0) return (void *)mi355_get_features;
    if (strcmp(name, "ggml_backend_split_buffer_type") == 0) return (void *)mi355_split_buffer_type;
    // "ggml_backend_set_n_threads", host-buffer registration: not provided
    return nullptr;
}

static const ggml_backend_reg_i mi355_reg_iface = {
    /* .get_name         = */ mi355_reg_get_name,
    /* .get_device_count = */ mi355_reg_get_device_count,
    /* .get_device       = */ mi355_reg_get_device,
    /* .get_proc_address = */ mi355_reg_get_proc_address,
};

static ggml_backend_reg_t ggml_backend_mi355_reg(void) {
    static ggml_backend_reg reg;
    static bool initialized = false;
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    if (!initialized) {
        mi355_reg_ctx * ctx = new mi355_reg_ctx;
        for (int i = 0; i < mi355_device_count(); i++) {
            mi355_device_ctx * dctx = new mi355_device_ctx;
            dctx->device = i;
            dctx->name   = std::string(MI355_DEV_PREFIX) + std::to_string(i);
            hipDeviceProp_t prop;
            HIP_CHECK(hipGetDeviceProperties(&prop, mi355_phys(i)));
            dctx->description = std::string(prop.name) + " (" + prop.gcnArchName + ")";
            ctx->devices.push_back(new ggml_backend_device{mi355_device_iface, &reg, dctx});
        }
        reg.api_version = GGML_BACKEND_API_VERSION;
        reg.iface       = mi355_reg_iface;
        reg.context     = ctx;
        initialized = true;
    }
    return &reg;
}

// exported entry points: what ggml_backend_load() looks up with dlsym (ggml-backend-reg.cpp:234-250)
extern "C" {
// the kernel library next to this plugin must be the one this file was compiled against: its argument structs are
// plain C and a stale library would read their tails as garbage
static bool mi355_kernels_abi_ok() {
    const bool ok = mi355_abi_version() == MI355_ABI_VERSION && mi355_args_sizeof(0) == (int64_t)sizeof(mi355_mmvq_args) &&
                    mi355_args_sizeof(1) == (int64_t)sizeof(mi355_qkv_rope_args) && mi355_args_sizeof(2) == (int64_t)sizeof(mi355_glu_matvec_args) &&
                    mi355_args_sizeof(3) == (int64_t)sizeof(mi355_moe_down_args) && mi355_args_sizeof(4) == (int64_t)sizeof(mi355_fattn_args) &&
                    mi355_args_sizeof(5) == (int64_t)sizeof(mi355_tensor) && mi355_args_sizeof(6) == (int64_t)sizeof(mi355_qkv_finish_args);
    if (!ok) MI_LOG_ERROR("ggml-mi355: libmi355_kernels.so has ABI version %d, the plugin was built for %d: rebuild both\n", mi355_abi_version(), MI355_ABI_VERSION);
    return ok;
}
__attribute__((visibility("default"))) ggml_backend_reg_t ggml_backend_init(void) { return mi355_kernels_abi_ok() ? ggml_backend_mi355_reg() : nullptr; }
__attribute__((visibility("default"))) int ggml_backend_score(void) { return mi355_kernels_abi_ok() && mi355_device_count() > 0 ? 100 : 0; }
}
