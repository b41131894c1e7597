// rope.h -- rotary-embedding arithmetic shared by rope_kernel (elementwise.hip) and the fused
// QKV + rope + KV-store mat-vec (mmvq.hip), so both produce bit-identical rotations.
// Behavioural spec: ggml_compute_forward_rope_f32 / rope_yarn (ggml/src/ggml-cpu/ops.cpp:5862-6110).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

struct RopeP {
    int n_dims, mode, ne0;
    float theta_scale, freq_scale, ext_factor, attn_factor, corr0, corr1;
};

static inline RopeP rope_params(int n_dims, int mode, int ne0, int n_ctx_orig, float freq_base, float freq_scale, float ext_factor,
                                float attn_factor, float beta_fast, float beta_slow) {
    RopeP p;
    p.n_dims = n_dims; p.mode = mode; p.ne0 = ne0;
    p.theta_scale = powf(freq_base, -2.0f / n_dims);
    p.freq_scale = freq_scale; p.ext_factor = ext_factor; p.attn_factor = attn_factor;
    // ggml_rope_yarn_corr_dims, ggml.c:4002-4014
    auto corr_dim = [&](float n_rot) { return n_dims * logf(n_ctx_orig / (n_rot * 2 * 3.14159265358979323846f)) / (2 * logf(freq_base)); };
    const float start = floorf(corr_dim(beta_fast)), end = ceilf(corr_dim(beta_slow));
    p.corr0 = start > 0 ? start : 0; p.corr1 = end < n_dims - 1 ? end : n_dims - 1;
    return p;
}

static __device__ __forceinline__ void rope_yarn(float theta_extrap, const RopeP & p, int i0, float & c, float & s) {
    // rope_yarn, ggml-cpu/ops.cpp:5862-5877
    float theta_interp = p.freq_scale * theta_extrap;
    float theta = theta_interp;
    float mscale = p.attn_factor;
    if (p.ext_factor != 0.0f) {
        const float y = (i0 / 2 - p.corr0) / fmaxf(0.001f, p.corr1 - p.corr0);
        const float ramp_mix = (1.0f - fminf(1.0f, fmaxf(0.0f, y))) * p.ext_factor;
        theta = theta_interp * (1 - ramp_mix) + theta_extrap * ramp_mix;
        mscale *= 1.0f + 0.1f * logf(1.0f / p.freq_scale);
    }
    c = cosf(theta) * mscale;
    s = sinf(theta) * mscale;
}

// cos / sin of the rotation of pair `ic` (elements 2 ic, 2 ic + 1 in NORMAL mode; ic, ic + n_dims/2 in NEOX) at
// position `pos`: theta = pos * theta_scale^ic / freq_factor[ic]
static __device__ __forceinline__ void rope_cos_sin(const RopeP & p, int pos, const float * ff, int ic, float & c, float & s) {
    const float theta_base = (float)pos * powf(p.theta_scale, (float)ic);
    rope_yarn(theta_base / (ff ? ff[ic] : 1.0f), p, 2 * ic, c, s);
}
