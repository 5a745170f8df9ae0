// GAN / L1 losses and their gradients.  Reference: keras BinaryCrossentropy(from_logits=True) and
// tf.reduce_mean(tf.abs(...)) at pix2pix_model.py:19,44-56, the tanh head at networks.py:75-78.
#include "p2p_common.hpp"

__device__ __forceinline__ float bce_logit(float x, float z) {
    return fmaxf(x, 0.f) - x * z + log1pf(expf(-fabsf(x)));
}
__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

// partials[k * P2P_LOSS_BLOCKS + b] = workgroup b's share of k = 0: BCE(1, real), 1: BCE(0, fake), 2: BCE(1, fake)
// (scaled by inv_count); p2p_loss_partials_sum adds them in workgroup order
// PAD8: the gradient views hold 8-channel pixels whose channels 1..7 are padding (the operand layout of the few-channel
// convolution kernels): the whole pixel [g 0 0 0 0 0 0 0] is stored -- a 2-byte store into a 16-byte pixel is a partial sector
template <typename T, bool PAD8>
__global__ __launch_bounds__(1024) void bce_logits_kernel(int N2, int n_real, PixDec dec, TView logits, float inv_count, TView dld,
                                                         TView dlg, float* __restrict__ partials) {
    typedef __attribute__((__vector_size__(8 * sizeof(T)))) T vec8_t;
    auto put = [](T* p, T v) {
        if (PAD8) {
            vec8_t q;
#pragma unroll
            for (int k = 0; k < 8; ++k) q[k] = from_f32<T>(0.f);
            q[0] = v;
            *(vec8_t*)p = q;
        } else {
            *p = v;
        }
    };
    __shared__ float red[16];
    const unsigned total = (unsigned)N2 * dec.H * dec.W;
    float l[3] = {0.f, 0.f, 0.f};
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        int n, y, x;
        dec(i, n, y, x);
        float v = to_f32(((const T*)logits.ptr)[logits.off(n, y, x)]);
        float s = sigmoidf(v);
        if (n < n_real) {
            l[0] += bce_logit(v, 1.f);
            put((T*)dld.ptr + dld.off(n, y, x), from_f32<T>((s - 1.f) * inv_count));
        } else {
            l[1] += bce_logit(v, 0.f);
            l[2] += bce_logit(v, 1.f);
            put((T*)dld.ptr + dld.off(n, y, x), from_f32<T>(s * inv_count));
            if (dlg.ptr) put((T*)dlg.ptr + dlg.off(n - n_real, y, x), from_f32<T>((s - 1.f) * inv_count));
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        l[k] = block_sum(l[k], red);
        if (threadIdx.x == 0) partials[k * P2P_LOSS_BLOCKS + blockIdx.x] = l[k] * inv_count;
    }
}

// fake = tanh(z), L1 partial.  VEC: 4 channels per lane as one 8/16-byte access per view (RGBA head).
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void tanh_l1_fwd_kernel(int N, PixDec dec, int C, TView z, TView real, TView fake, float inv_count,
                                                          float* __restrict__ partials, float* __restrict__ fake_f32) {
    __shared__ float red[16];
    typedef __attribute__((__vector_size__(4 * sizeof(T)))) T vec4_t;
    const unsigned npix = (unsigned)N * dec.H * dec.W;
    float acc = 0.f;
    for (unsigned p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += gridDim.x * blockDim.x) {
        int n, y, x;
        dec(p, n, y, x);
        const T* zp = (const T*)z.ptr + z.off(n, y, x);
        const T* rp = (const T*)real.ptr + real.off(n, y, x);
        T* fp = (T*)fake.ptr + fake.off(n, y, x);
        if (VEC) {
            const vec4_t zv = *(const vec4_t*)zp, rv = *(const vec4_t*)rp;
            vec4_t fv;
            f32x4 ff;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float t = tanhf(to_f32((T)zv[c]));
                ff[c] = t;
                T fq = from_f32<T>(t);
                fv[c] = fq;
                acc += fabsf(to_f32((T)rv[c]) - to_f32(fq));
            }
            *(vec4_t*)fp = fv;
            if (fake_f32) *(f32x4*)(fake_f32 + (long long)p * 4) = ff;      // unrounded copy for the histogram loss
        } else {
            for (int c = 0; c < C; ++c) {
                const float t = tanhf(to_f32(zp[c]));
                T fq = from_f32<T>(t);
                fp[c] = fq;
                acc += fabsf(to_f32(rp[c]) - to_f32(fq));
                if (fake_f32) fake_f32[(long long)p * C + c] = t;
            }
        }
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc * inv_count;
}

// out[k] = sum_b partials[k * P2P_LOSS_BLOCKS + b], fixed order
__global__ __launch_bounds__(P2P_LOSS_BLOCKS) void loss_partials_sum_kernel(const float* __restrict__ partials, float* __restrict__ out) {
    __shared__ float red[16];
    float s = block_sum(partials[blockIdx.x * P2P_LOSS_BLOCKS + threadIdx.x], red);
    if (threadIdx.x == 0) out[blockIdx.x] = s;
}

// four channels of a gradient source at pixel p (VEC path: ld % 4 == 0, coff % 4 == 0, 8/16-byte aligned base)
template <typename T>
__device__ __forceinline__ void gsrc_load4(const GSrc& g, long long pix, float* v) {
    typedef __attribute__((__vector_size__(4 * sizeof(T)))) T vec4_t;
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = 0.f;
    if (g.kind == 0) return;
    const long long e = pix * g.ld + g.coff;
    if (g.kind == 1) {
        const vec4_t q = *(const vec4_t*)((const T*)g.ptr + e);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = to_f32((T)q[k]);
        return;
    }
    const float* p = (const float*)g.ptr + e;
    for (int sI = 0; sI < g.nslabs; ++sI) {
        const f32x4 q = *(const f32x4*)(p + (long long)sI * g.slab);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] += q[k];
    }
}

template <typename T, bool VEC, bool PAD8 = false>
__global__ void tanh_l1_bwd_kernel(int N, PixDec dec, int C, TView fake, TView real, GSrc gd, GSrc gx,
                                   float l1_scale, TView dz) {
    typedef __attribute__((__vector_size__(4 * sizeof(T)))) T vec4_t;
    typedef __attribute__((__vector_size__(8 * sizeof(T)))) T vec8_t;
    const unsigned npix = (unsigned)N * dec.H * dec.W;
    for (unsigned p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += gridDim.x * blockDim.x) {
        int n, y, x;
        dec(p, n, y, x);
        const T* fp = (const T*)fake.ptr + fake.off(n, y, x);
        const T* rp = (const T*)real.ptr + real.off(n, y, x);
        T* dp = (T*)dz.ptr + dz.off(n, y, x);
        if (VEC) {      // RGBA head: one 8/16-byte access per view
            const vec4_t fv = *(const vec4_t*)fp, rv = *(const vec4_t*)rp;
            float g1[4], g2[4];
            gsrc_load4<T>(gd, p, g1);
            gsrc_load4<T>(gx, p, g2);
            vec4_t out;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float f = to_f32((T)fv[c]);
                const float d = f - to_f32((T)rv[c]);
                const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
                out[c] = from_f32<T>((g1[c] + g2[c] + l1_scale * sgn) * (1.f - f * f));
            }
            if (PAD8) {       // dz holds 8-channel pixels [dz | 4 padding channels]: whole-pixel store
                vec8_t o8;
#pragma unroll
                for (int c = 0; c < 4; ++c) { o8[c] = out[c]; o8[4 + c] = from_f32<T>(0.f); }
                *(vec8_t*)dp = o8;
            } else {
                *(vec4_t*)dp = out;
            }
            continue;
        }
        for (int c = 0; c < C; ++c) {
            float f = to_f32(fp[c]);
            float d = f - to_f32(rp[c]);
            float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);     // tf.abs gradient = sign(x), 0 at 0
            float g = gsrc_load<T>(gd, p, c) + gsrc_load<T>(gx, p, c) + l1_scale * sgn;
            dp[c] = from_f32<T>(g * (1.f - f * f));
        }
    }
}

// The loss kernels always launch P2P_LOSS_BLOCKS workgroups (grid-stride loops) and write one partial per workgroup and
// loss term; p2p_loss_partials_sum turns K consecutive partial rows into K sums (one launch for all loss terms of a step).
extern "C" int p2p_loss_partials_sum(const float* partials, int K, float* out, void* stream) {
    P2P_REQUIRE(partials && out && K >= 1 && K <= 64, "p2p_loss_partials_sum: bad args");
    loss_partials_sum_kernel<<<dim3(K), P2P_LOSS_BLOCKS, 0, (hipStream_t)stream>>>(partials, out);
    return p2p_check_launch("p2p_loss_partials_sum");
}

static int bce_logits_impl(int dtype, int N2, int n_real, int H, int W, const p2p_tensor* logits, float inv_count,
                           const p2p_tensor* dlogits_d, const p2p_tensor* dlogits_g, float* partials, bool pad8, void* stream) {
    P2P_REQUIRE(N2 > 0 && n_real >= 0 && n_real <= N2 && H > 0 && W > 0, "p2p_bce_logits: bad shape");
    P2P_REQUIRE(logits && logits->ptr && dlogits_d && dlogits_d->ptr && partials, "p2p_bce_logits: null pointer");
    P2P_REQUIRE((long long)N2 * H * W < (1LL << 31), "p2p_bce_logits: too many pixels");
    hipStream_t st = (hipStream_t)stream;
    TView g;
    if (dlogits_g && dlogits_g->ptr) g = make_view(dlogits_g);
    else { g.ptr = nullptr; g.img = 0; g.row = 0; g.ld = 0; }
    if (pad8) {
        const int esz = dtype == P2P_BF16 ? 2 : 4;
        auto ok = [&](const p2p_tensor* t) { return !t || !t->ptr || (t->ld % 8 == 0 && ((uintptr_t)t->ptr % (8 * esz)) == 0); };
        P2P_REQUIRE(ok(dlogits_d) && ok(dlogits_g), "p2p_bce_logits_pad8: the gradient views must be whole 8-channel pixels");
        P2P_DISPATCH_DTYPE(dtype, (bce_logits_kernel<T, true><<<dim3(P2P_LOSS_BLOCKS), 1024, 0, st>>>(
                                      N2, n_real, PixDec::make(H, W), make_view(logits), inv_count, make_view(dlogits_d), g, partials)));
    } else {
        P2P_DISPATCH_DTYPE(dtype, (bce_logits_kernel<T, false><<<dim3(P2P_LOSS_BLOCKS), 1024, 0, st>>>(
                                      N2, n_real, PixDec::make(H, W), make_view(logits), inv_count, make_view(dlogits_d), g, partials)));
    }
    return p2p_check_launch("p2p_bce_logits");
}

extern "C" int p2p_bce_logits(int dtype, int N2, int n_real, int H, int W, const p2p_tensor* logits, float inv_count,
                              const p2p_tensor* dlogits_d, const p2p_tensor* dlogits_g, float* partials, void* stream) {
    return bce_logits_impl(dtype, N2, n_real, H, W, logits, inv_count, dlogits_d, dlogits_g, partials, false, stream);
}

// As p2p_bce_logits for gradient views that are 8-channel pixels [g | 7 padding channels]: whole pixels are stored (the padding
// is written as zeros, which it holds anyway).
extern "C" int p2p_bce_logits_pad8(int dtype, int N2, int n_real, int H, int W, const p2p_tensor* logits, float inv_count,
                                   const p2p_tensor* dlogits_d, const p2p_tensor* dlogits_g, float* partials, void* stream) {
    return bce_logits_impl(dtype, N2, n_real, H, W, logits, inv_count, dlogits_d, dlogits_g, partials, true, stream);
}

extern "C" int p2p_tanh_l1_fwd(int dtype, int N, int H, int W, int C, const p2p_tensor* z, const p2p_tensor* real,
                               const p2p_tensor* fake, float inv_count, float* partials, float* fake_f32, void* stream) {
    P2P_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0, "p2p_tanh_l1_fwd: bad shape");
    P2P_REQUIRE(z && z->ptr && real && real->ptr && fake && fake->ptr && partials, "p2p_tanh_l1_fwd: null pointer");
    P2P_REQUIRE((long long)N * H * W < (1LL << 31), "p2p_tanh_l1_fwd: too many pixels");
    hipStream_t st = (hipStream_t)stream;
    const int esz = dtype == P2P_BF16 ? 2 : 4;
    auto al = [&](const p2p_tensor* t) { return t->ld % 4 == 0 && ((uintptr_t)t->ptr % (4 * esz)) == 0; };
    const bool vec = C == 4 && al(z) && al(real) && al(fake);
    const dim3 grid(P2P_LOSS_BLOCKS);
    if (vec) {
        P2P_DISPATCH_DTYPE(dtype, (tanh_l1_fwd_kernel<T, true><<<grid, 256, 0, st>>>(N, PixDec::make(H, W), C, make_view(z), make_view(real),
                                                                                    make_view(fake), inv_count, partials, fake_f32)));
    } else {
        P2P_DISPATCH_DTYPE(dtype, (tanh_l1_fwd_kernel<T, false><<<grid, 256, 0, st>>>(N, PixDec::make(H, W), C, make_view(z), make_view(real),
                                                                                     make_view(fake), inv_count, partials, fake_f32)));
    }
    return p2p_check_launch("p2p_tanh_l1_fwd");
}

// The train step's form for 4-channel images whose discriminator inputs are 8-channel pixels [image | source]
// (networks.py:45): `real_pair` = [target | source] is read as whole pixels and the WHOLE fake pixel [tanh(z) | source] is
// written -- written as two halves by two kernels, the 8 of every 16 bytes were partial sector writes in both.
template <typename T>
__global__ __launch_bounds__(1024) void tanh_l1_fwd_pair_kernel(int N, PixDec dec, TView z, TView real, TView fake, float inv_count,
                                                                float* __restrict__ partials, float* __restrict__ fake_f32) {
    __shared__ float red[16];
    typedef __attribute__((__vector_size__(4 * sizeof(T)))) T vec4_t;
    typedef __attribute__((__vector_size__(8 * sizeof(T)))) T vec8_t;
    const unsigned npix = (unsigned)N * dec.H * dec.W;
    float acc = 0.f;
    for (unsigned p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += gridDim.x * blockDim.x) {
        int n, y, x;
        dec(p, n, y, x);
        const vec4_t zv = *(const vec4_t*)((const T*)z.ptr + z.off(n, y, x));
        const vec8_t rv = *(const vec8_t*)((const T*)real.ptr + real.off(n, y, x));
        vec8_t fv;
        f32x4 ff;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float t = tanhf(to_f32((T)zv[c]));
            ff[c] = t;
            T fq = from_f32<T>(t);
            fv[c] = fq;
            fv[4 + c] = rv[4 + c];
            acc += fabsf(to_f32((T)rv[c]) - to_f32(fq));
        }
        *(vec8_t*)((T*)fake.ptr + fake.off(n, y, x)) = fv;
        if (fake_f32) *(f32x4*)(fake_f32 + (long long)p * 4) = ff;      // unrounded copy for the histogram loss
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc * inv_count;
}

extern "C" int p2p_tanh_l1_fwd_pair(int dtype, int N, int H, int W, const p2p_tensor* z, const p2p_tensor* real_pair,
                                    const p2p_tensor* fake_pair, float inv_count, float* partials, float* fake_f32, void* stream) {
    P2P_REQUIRE(N > 0 && H > 0 && W > 0, "p2p_tanh_l1_fwd_pair: bad shape");
    P2P_REQUIRE(z && z->ptr && real_pair && real_pair->ptr && fake_pair && fake_pair->ptr && partials, "p2p_tanh_l1_fwd_pair: null pointer");
    P2P_REQUIRE((long long)N * H * W < (1LL << 31), "p2p_tanh_l1_fwd_pair: too many pixels");
    const int esz = dtype == P2P_BF16 ? 2 : 4;
    P2P_REQUIRE(z->ld % 4 == 0 && ((uintptr_t)z->ptr % (4 * esz)) == 0, "p2p_tanh_l1_fwd_pair: z must be 4-channel aligned");
    P2P_REQUIRE(real_pair->ld % 8 == 0 && fake_pair->ld % 8 == 0 && ((uintptr_t)real_pair->ptr % (8 * esz)) == 0 &&
                    ((uintptr_t)fake_pair->ptr % (8 * esz)) == 0,
                "p2p_tanh_l1_fwd_pair: the pair views must be whole 8-channel pixels");
    P2P_DISPATCH_DTYPE(dtype, (tanh_l1_fwd_pair_kernel<T><<<dim3(P2P_LOSS_BLOCKS), 1024, 0, (hipStream_t)stream>>>(
                                  N, PixDec::make(H, W), make_view(z), make_view(real_pair), make_view(fake_pair), inv_count, partials,
                                  fake_f32)));
    return p2p_check_launch("p2p_tanh_l1_fwd_pair");
}

static int tanh_l1_bwd_impl(int dtype, int N, int H, int W, int C, const p2p_tensor* fake, const p2p_tensor* real,
                            const p2p_gsrc* g_d, const p2p_gsrc* g_extra, float l1_scale, const p2p_tensor* dz, bool pad8,
                            void* stream) {
    P2P_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0, "p2p_tanh_l1_bwd: bad shape");
    P2P_REQUIRE(fake && fake->ptr && real && real->ptr && dz && dz->ptr, "p2p_tanh_l1_bwd: null pointer");
    P2P_REQUIRE((long long)N * H * W < (1LL << 31), "p2p_tanh_l1_bwd: too many pixels");
    long long blocks = ((long long)N * H * W + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    const int esz = dtype == P2P_BF16 ? 2 : 4;
    auto al = [&](const p2p_tensor* t) { return t->ld % 4 == 0 && ((uintptr_t)t->ptr % (4 * esz)) == 0; };
    auto gal = [&](const p2p_gsrc* g) {
        if (!g || !g->ptr || g->kind == 0) return true;
        if (g->ld % 4 || g->coff % 4) return false;
        return g->kind == 1 ? ((uintptr_t)g->ptr % (4 * esz)) == 0 : (((uintptr_t)g->ptr % 16) == 0 && g->slab_stride % 4 == 0);
    };
    const bool vec = C == 4 && al(fake) && al(real) && al(dz) && gal(g_d) && gal(g_extra);
    if (pad8) {
        P2P_REQUIRE(vec && dz->ld % 8 == 0 && ((uintptr_t)dz->ptr % (8 * esz)) == 0,
                    "p2p_tanh_l1_bwd_pad8: 4-channel images, dz in whole 8-channel pixels, aligned views");
        P2P_DISPATCH_DTYPE(dtype, (tanh_l1_bwd_kernel<T, true, true><<<dim3((unsigned)blocks), 256, 0, (hipStream_t)stream>>>(
                                      N, PixDec::make(H, W), C, make_view(fake), make_view(real), make_gsrc(g_d), make_gsrc(g_extra),
                                      l1_scale, make_view(dz))));
    } else if (vec) {
        P2P_DISPATCH_DTYPE(dtype, (tanh_l1_bwd_kernel<T, true><<<dim3((unsigned)blocks), 256, 0, (hipStream_t)stream>>>(
                                      N, PixDec::make(H, W), C, make_view(fake), make_view(real), make_gsrc(g_d), make_gsrc(g_extra),
                                      l1_scale, make_view(dz))));
    } else {
        P2P_DISPATCH_DTYPE(dtype, (tanh_l1_bwd_kernel<T, false><<<dim3((unsigned)blocks), 256, 0, (hipStream_t)stream>>>(
                                      N, PixDec::make(H, W), C, make_view(fake), make_view(real), make_gsrc(g_d), make_gsrc(g_extra),
                                      l1_scale, make_view(dz))));
    }
    return p2p_check_launch("p2p_tanh_l1_bwd");
}

extern "C" int p2p_tanh_l1_bwd(int dtype, int N, int H, int W, int C, const p2p_tensor* fake, const p2p_tensor* real,
                               const p2p_gsrc* g_d, const p2p_gsrc* g_extra, float l1_scale, const p2p_tensor* dz,
                               void* stream) {
    return tanh_l1_bwd_impl(dtype, N, H, W, C, fake, real, g_d, g_extra, l1_scale, dz, false, stream);
}

// As p2p_tanh_l1_bwd (C = 4) for a dz view of 8-channel pixels [dz | 4 padding channels] (the operand layout of the head's
// weight- and data-gradient kernels): whole pixels are stored, the padding as the zeros it holds anyway.
extern "C" int p2p_tanh_l1_bwd_pad8(int dtype, int N, int H, int W, const p2p_tensor* fake, const p2p_tensor* real,
                                    const p2p_gsrc* g_d, const p2p_gsrc* g_extra, float l1_scale, const p2p_tensor* dz,
                                    void* stream) {
    return tanh_l1_bwd_impl(dtype, N, H, W, 4, fake, real, g_d, g_extra, l1_scale, dz, true, stream);
}
