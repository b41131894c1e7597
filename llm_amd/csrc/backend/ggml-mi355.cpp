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
#include "mi355-backend.h"

static ggml_backend_reg_t ggml_backend_mi355_reg(void);

mi355_rccl & mi355_rccl_state() { static mi355_rccl r; return r; }
mi355_copy_stats g_copy_stats;

// ------------------------------------------------------------------------------------------------
// device info
// ------------------------------------------------------------------------------------------------

// Physical gfx950 devices.  GGML_MI355_VIRTUAL_DEVICES=n (diagnostics / tests) presents n ggml devices
// that all map round-robin onto the physical ones, so the multi-device code paths (row split, peer
// copies, cross-device events) can be exercised on a single-GPU machine.
int mi355_physical_count() {
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

int mi355_device_count() {
    static int count = [] {
        const int phys = mi355_physical_count();
        const char * v = getenv("GGML_MI355_VIRTUAL_DEVICES");
        if (phys > 0 && v && atoi(v) > 0) return atoi(v) < MI355_MAX_DEVICES ? atoi(v) : MI355_MAX_DEVICES;
        return phys;
    }();
    return count;
}

int mi355_phys(int device) { return device % mi355_physical_count(); }

void mi355_set_device(int device) {
    int cur = -1;
    HIP_CHECK(hipGetDevice(&cur));
    if (cur != mi355_phys(device)) HIP_CHECK(hipSetDevice(mi355_phys(device)));
}

// device-to-device copy between two ggml devices (which may share a physical GPU)
void mi355_copy_d2d_async(void * dst, int dst_dev, const void * src, int src_dev, size_t n, hipStream_t stream) {
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

bool mi355_split_gather_rccl() {
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

struct mi355_split_buft_ctx {
    int         main_device;
    float       split[MI355_MAX_DEVICES];   // cumulative, normalised: device d starts at nrows * split[d]
    std::string name;
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

bool tensor_is_split(const ggml_tensor * t) { return t && t->buffer && buft_is_mi355_split(t->buffer->buft); }

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


static ggml_guid_t mi355_guid() {
    static ggml_guid guid = {0x4d, 0x49, 0x33, 0x35, 0x35, 0x58, 0x67, 0x66, 0x78, 0x39, 0x35, 0x30, 0x67, 0x67, 0x6d, 0x6c};
    return &guid;
}

static const char * mi355_backend_get_name(ggml_backend_t backend) { return ((mi355_backend_ctx *)backend->context)->name.c_str(); }

static void mi355_backend_free(ggml_backend_t backend) {
    mi355_backend_ctx * ctx = (mi355_backend_ctx *)backend->context;
    mi355_set_device(ctx->device);
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 9; i++) if (ctx->scratch[i]) HIP_CHECK(hipFree(ctx->scratch[i]));
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

