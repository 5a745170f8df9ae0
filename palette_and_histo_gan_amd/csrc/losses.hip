// GAN / L1 losses and their gradients.  Reference: keras BinaryCrossentropy(from_logits=True) and
// tf.reduce_mean(tf.abs(...)) at pix2pix_model.py:19,44-56, the tanh head at networks.py:75-78.
#include "p2p_common.hpp"

__device__ __forceinline__ float bce_logit(float x, float z) {
    return fmaxf(x, 0.f) - x * z + log1pf(expf(-fabsf(x)));
}
__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

// loss_out[0] += BCE(1, real) ; [1] += BCE(0, fake) ; [2] += BCE(1, fake)   (all scaled by inv_count)
template <typename T>
__global__ void bce_logits_kernel(int N2, int n_real, int H, int W, TView logits, float inv_count, TView dld,
                                  TView dlg, float* __restrict__ loss_out) {
    __shared__ float red[16];
    long long total = (long long)N2 * H * W;
    float l0 = 0.f, l1 = 0.f, l2 = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int x = (int)(i % W);
        int y = (int)((i / W) % H);
        int n = (int)(i / ((long long)W * H));
        float v = to_f32(((const T*)logits.ptr)[logits.off(n, y, x)]);
        float s = sigmoidf(v);
        if (n < n_real) {
            l0 += bce_logit(v, 1.f);
            ((T*)dld.ptr)[dld.off(n, y, x)] = from_f32<T>((s - 1.f) * inv_count);
        } else {
            l1 += bce_logit(v, 0.f);
            l2 += bce_logit(v, 1.f);
            ((T*)dld.ptr)[dld.off(n, y, x)] = from_f32<T>(s * inv_count);
            if (dlg.ptr) ((T*)dlg.ptr)[dlg.off(n - n_real, y, x)] = from_f32<T>((s - 1.f) * inv_count);
        }
    }
    l0 = block_sum(l0, red);
    l1 = block_sum(l1, red);
    l2 = block_sum(l2, red);
    if (threadIdx.x == 0) {
        atomicAdd(loss_out + 0, l0 * inv_count);
        atomicAdd(loss_out + 1, l1 * inv_count);
        atomicAdd(loss_out + 2, l2 * inv_count);
    }
}

template <typename T>
__global__ void tanh_l1_fwd_kernel(int N, int H, int W, int C, TView z, TView real, TView fake, float inv_count,
                                   float* __restrict__ l1_out) {
    __shared__ float red[16];
    long long total = (long long)N * H * W * C;
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int c = (int)(i % C);
        long long p = i / C;
        int x = (int)(p % W);
        int y = (int)((p / W) % H);
        int n = (int)(p / ((long long)W * H));
        float f = tanhf(to_f32(((const T*)z.ptr)[z.off(n, y, x) + c]));
        T fq = from_f32<T>(f);
        ((T*)fake.ptr)[fake.off(n, y, x) + c] = fq;
        acc += fabsf(to_f32(((const T*)real.ptr)[real.off(n, y, x) + c]) - to_f32(fq));
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) atomicAdd(l1_out, acc * inv_count);
}

template <typename T>
__global__ void tanh_l1_bwd_kernel(int N, int H, int W, int C, TView fake, TView real, GSrc gd, GSrc gx,
                                   float l1_scale, TView dz) {
    long long total = (long long)N * H * W * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int c = (int)(i % C);
        long long p = i / C;
        int x = (int)(p % W);
        int y = (int)((p / W) % H);
        int n = (int)(p / ((long long)W * H));
        float f = to_f32(((const T*)fake.ptr)[fake.off(n, y, x) + c]);
        float r = to_f32(((const T*)real.ptr)[real.off(n, y, x) + c]);
        float d = f - r;
        float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);     // tf.abs gradient = sign(x), 0 at 0
        float g = gsrc_load<T>(gd, p, c) + gsrc_load<T>(gx, p, c) + l1_scale * sgn;
        ((T*)dz.ptr)[dz.off(n, y, x) + c] = from_f32<T>(g * (1.f - f * f));
    }
}

// few workgroups: every one ends in atomics on the same 1-3 loss words
static inline unsigned grid_for(long long total) {
    long long b = (total + 1023) / 1024;
    return (unsigned)(b < 256 ? (b < 1 ? 1 : b) : 256);
}

extern "C" int p2p_bce_logits(int dtype, int N2, int n_real, int H, int W, const p2p_tensor* logits, float inv_count,
                              const p2p_tensor* dlogits_d, const p2p_tensor* dlogits_g, float* loss_out, void* stream) {
    P2P_REQUIRE(N2 > 0 && n_real >= 0 && n_real <= N2 && H > 0 && W > 0, "p2p_bce_logits: bad shape");
    P2P_REQUIRE(logits && logits->ptr && dlogits_d && dlogits_d->ptr && loss_out, "p2p_bce_logits: null pointer");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(loss_out, 0, 3 * sizeof(float), st);
    if (e != hipSuccess) { p2p_set_error("p2p_bce_logits memset: %s", hipGetErrorString(e)); return (int)e; }
    TView g;
    if (dlogits_g && dlogits_g->ptr) g = make_view(dlogits_g);
    else { g.ptr = nullptr; g.img = 0; g.row = 0; g.ld = 0; }
    P2P_DISPATCH_DTYPE(dtype, (bce_logits_kernel<T><<<dim3(grid_for((long long)N2 * H * W)), 256, 0, st>>>(
                                  N2, n_real, H, W, make_view(logits), inv_count, make_view(dlogits_d), g, loss_out)));
    return p2p_check_launch("p2p_bce_logits");
}

extern "C" int p2p_tanh_l1_fwd(int dtype, int N, int H, int W, int C, const p2p_tensor* z, const p2p_tensor* real,
                               const p2p_tensor* fake, float inv_count, float* l1_out, void* stream) {
    P2P_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0, "p2p_tanh_l1_fwd: bad shape");
    P2P_REQUIRE(z && z->ptr && real && real->ptr && fake && fake->ptr && l1_out, "p2p_tanh_l1_fwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(l1_out, 0, sizeof(float), st);
    if (e != hipSuccess) { p2p_set_error("p2p_tanh_l1_fwd memset: %s", hipGetErrorString(e)); return (int)e; }
    long long tb = ((long long)N * H * W * C + 1023) / 1024;      // 4 elements per thread, one atomic per workgroup
    if (tb > 1024) tb = 1024;
    P2P_DISPATCH_DTYPE(dtype, (tanh_l1_fwd_kernel<T><<<dim3((unsigned)(tb < 1 ? 1 : tb)), 256, 0, st>>>(
                                  N, H, W, C, make_view(z), make_view(real), make_view(fake), inv_count, l1_out)));
    return p2p_check_launch("p2p_tanh_l1_fwd");
}

extern "C" int p2p_tanh_l1_bwd(int dtype, int N, int H, int W, int C, const p2p_tensor* fake, const p2p_tensor* real,
                               const p2p_gsrc* g_d, const p2p_gsrc* g_extra, float l1_scale, const p2p_tensor* dz,
                               void* stream) {
    P2P_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0, "p2p_tanh_l1_bwd: bad shape");
    P2P_REQUIRE(fake && fake->ptr && real && real->ptr && dz && dz->ptr, "p2p_tanh_l1_bwd: null pointer");
    long long blocks = ((long long)N * H * W * C + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    P2P_DISPATCH_DTYPE(dtype, (tanh_l1_bwd_kernel<T><<<dim3((unsigned)blocks), 256, 0, (hipStream_t)stream>>>(
                                  N, H, W, C, make_view(fake), make_view(real), make_gsrc(g_d), make_gsrc(g_extra),
                                  l1_scale, make_view(dz))));
    return p2p_check_launch("p2p_tanh_l1_bwd");
}
