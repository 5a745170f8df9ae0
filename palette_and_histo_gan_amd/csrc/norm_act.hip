// Fused InstanceNorm + Dropout + LeakyReLU/ReLU, forward and backward, writing straight into a channel
// slice of the (haloed) concat buffer.  Reference: tfa InstanceNormalization / Dropout / LeakyReLU / ReLU /
// Concatenate at networks.py:18-19,29-34,94; backward = the tape gradient of that chain, closed form in
// SURVEY.md 8a A13 (checked against autograd in tests/test_oracle.py).
#include "p2p_common.hpp"

// one workgroup = one image x CB consecutive channels; threads = (pixel lane) x (channel)
template <typename T>
__device__ __forceinline__ float raw_load(const void* raw, int raw_kind, int nslabs, long long slab, long long e) {
    if (raw_kind == 1) return to_f32(((const T*)raw)[e]);
    float s = 0.f;
    const float* p = (const float*)raw + e;
    for (int k = 0; k < nslabs; ++k) s += p[(long long)k * slab];
    return to_f32(from_f32<T>(s));   // round through the storage type so fwd and bwd see the same x
}

template <typename T>
__global__ void norm_act_fwd_kernel(int H, int W, int C, int CB, const void* __restrict__ raw, int raw_kind,
                                    int nslabs, long long slab, const float* __restrict__ gamma,
                                    const float* __restrict__ beta, float eps, int act, float alpha,
                                    const unsigned char* __restrict__ mask, TView out, T* __restrict__ raw_out,
                                    float* __restrict__ stats) {
    extern __shared__ float sm[];   // [PL][CB] scratch
    const int n = blockIdx.x;
    const int c0 = blockIdx.y * CB;
    const int cb = min(CB, C - c0);
    const int PL = blockDim.x / CB;           // pixel lanes
    const int c = threadIdx.x % CB;
    const int pl = threadIdx.x / CB;
    const int HW = H * W;
    const bool live = c < cb && pl < PL;
    const long long base = (long long)n * HW * C + c0 + c;

    float mu = 0.f, r = 1.f, ga = 1.f, be = 0.f;
    if (gamma) {
        float s = 0.f;
        if (live)
            for (int p = pl; p < HW; p += PL) s += raw_load<T>(raw, raw_kind, nslabs, slab, base + (long long)p * C);
        sm[pl * CB + c] = live ? s : 0.f;
        __syncthreads();
        float tot = 0.f;
        for (int i = 0; i < PL; ++i) tot += sm[i * CB + c];
        mu = tot / (float)HW;
        __syncthreads();
        float q = 0.f;
        if (live)
            for (int p = pl; p < HW; p += PL) {
                float d = raw_load<T>(raw, raw_kind, nslabs, slab, base + (long long)p * C) - mu;
                q += d * d;
            }
        sm[pl * CB + c] = live ? q : 0.f;
        __syncthreads();
        float var = 0.f;
        for (int i = 0; i < PL; ++i) var += sm[i * CB + c];
        var /= (float)HW;
        r = rsqrtf(var + eps);
        if (c < cb) {
            ga = gamma[c0 + c];
            be = beta[c0 + c];
            if (pl == 0) {
                stats[((long long)n * C + c0 + c) * 2 + 0] = mu;
                stats[((long long)n * C + c0 + c) * 2 + 1] = r;
            }
        }
    }
    if (!live) return;
    for (int p = pl; p < HW; p += PL) {
        long long e = base + (long long)p * C;
        float x = raw_load<T>(raw, raw_kind, nslabs, slab, e);
        if (raw_out) raw_out[e] = from_f32<T>(x);
        float y = gamma ? (x - mu) * r * ga + be : x;
        if (mask) y = mask[e] ? y * 2.f : 0.f;
        if (act == P2P_ACT_LEAKY) y = y > 0.f ? y : alpha * y;
        else if (act == P2P_ACT_RELU) y = y > 0.f ? y : 0.f;
        int yy = p / W, xx = p - yy * W;
        ((T*)out.ptr)[out.off(n, yy, xx) + c0 + c] = from_f32<T>(y);
    }
}

template <typename T>
__global__ void norm_act_bwd_kernel(int H, int W, int C, int CB, const T* __restrict__ raw,
                                    const float* __restrict__ stats, const float* __restrict__ gamma,
                                    const float* __restrict__ beta, int act, float alpha,
                                    const unsigned char* __restrict__ mask, GSrc g1, GSrc g2, TView draw,
                                    float* __restrict__ dgamma_part, float* __restrict__ dbeta_part) {
    extern __shared__ float sm[];   // 2 x [PL][CB]
    const int n = blockIdx.x;
    const int c0 = blockIdx.y * CB;
    const int cb = min(CB, C - c0);
    const int PL = blockDim.x / CB;
    const int c = threadIdx.x % CB;
    const int pl = threadIdx.x / CB;
    const int HW = H * W;
    const bool live = c < cb && pl < PL;
    const long long pix0 = (long long)n * HW;
    const long long base = pix0 * C + c0 + c;

    float mu = 0.f, r = 1.f, ga = 1.f, be = 0.f;
    if (gamma && c < cb) {
        mu = stats[((long long)n * C + c0 + c) * 2 + 0];
        r = stats[((long long)n * C + c0 + c) * 2 + 1];
        ga = gamma[c0 + c];
        be = beta[c0 + c];
    }
    // d(yhat) at pixel p (yhat = the normalised value before dropout/activation)
    auto dyhat = [&](int p, float& xh) -> float {
        long long e = base + (long long)p * C;
        float x = to_f32(raw[e]);
        xh = gamma ? (x - mu) * r : x;
        float a = gamma ? xh * ga + be : x;
        float keep = 1.f;
        if (mask) { keep = mask[e] ? 2.f : 0.f; a *= keep; }
        float da = gsrc_load<T>(g1, pix0 + p, c0 + c) + gsrc_load<T>(g2, pix0 + p, c0 + c);
        float slope = 1.f;
        if (act == P2P_ACT_LEAKY) slope = a > 0.f ? 1.f : alpha;
        else if (act == P2P_ACT_RELU) slope = a > 0.f ? 1.f : 0.f;
        return da * slope * keep;
    };

    float m1 = 0.f, m2 = 0.f;
    if (gamma) {
        float s1 = 0.f, s2 = 0.f;
        if (live)
            for (int p = pl; p < HW; p += PL) {
                float xh;
                float d = dyhat(p, xh);
                s1 += d;
                s2 += d * xh;
            }
        float* sa = sm;
        float* sb = sm + PL * CB;
        sa[pl * CB + c] = live ? s1 : 0.f;
        sb[pl * CB + c] = live ? s2 : 0.f;
        __syncthreads();
        float t1 = 0.f, t2 = 0.f;
        for (int i = 0; i < PL; ++i) { t1 += sa[i * CB + c]; t2 += sb[i * CB + c]; }
        if (c < cb && pl == 0) {
            dbeta_part[(long long)n * C + c0 + c] = t1;
            dgamma_part[(long long)n * C + c0 + c] = t2;
        }
        m1 = t1 / (float)HW;
        m2 = t2 / (float)HW;
    }
    if (!live) return;
    for (int p = pl; p < HW; p += PL) {
        float xh;
        float d = dyhat(p, xh);
        float dx = gamma ? ga * r * (d - m1 - xh * m2) : d;
        int yy = p / W, xx = p - yy * W;
        ((T*)draw.ptr)[draw.off(n, yy, xx) + c0 + c] = from_f32<T>(dx);
    }
}

// Backward of a bare activation whose OUTPUT is stored (LeakyReLU fused into the conv epilogue of down1 / D.down,
// networks.py:19,46,58): sign(out) == sign(pre-activation), so d(pre) = (g1 + g2) * (out > 0 ? 1 : alpha).
template <typename T>
__global__ void act_bwd_kernel(int N, int H, int W, int C, TView actv, GSrc g1, GSrc g2, float alpha, TView draw) {
    long long total = (long long)N * H * W * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int c = (int)(i % C);
        long long p = i / C;
        int x = (int)(p % W);
        int y = (int)((p / W) % H);
        int n = (int)(p / ((long long)W * H));
        float a = to_f32(((const T*)actv.ptr)[actv.off(n, y, x) + c]);
        float g = gsrc_load<T>(g1, p, c) + gsrc_load<T>(g2, p, c);
        ((T*)draw.ptr)[draw.off(n, y, x) + c] = from_f32<T>(a > 0.f ? g : alpha * g);
    }
}

extern "C" int p2p_act_bwd(int dtype, int N, int H, int W, int C, const p2p_tensor* act_out, const p2p_gsrc* g1,
                           const p2p_gsrc* g2, float alpha, const p2p_tensor* draw, void* stream) {
    P2P_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && act_out && act_out->ptr && g1 && draw && draw->ptr, "p2p_act_bwd: bad args");
    long long total = (long long)N * H * W * C;
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    P2P_DISPATCH_DTYPE(dtype, (act_bwd_kernel<T><<<dim3((unsigned)blocks), 256, 0, (hipStream_t)stream>>>(
                                  N, H, W, C, make_view(act_out), make_gsrc(g1), make_gsrc(g2), alpha, make_view(draw))));
    return p2p_check_launch("p2p_act_bwd");
}

__global__ void colsum_kernel(const float* __restrict__ part, int rows, int cols, float scale, float* __restrict__ out) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    float s = 0.f;
    for (int r = 0; r < rows; ++r) s += part[(long long)r * cols + c];
    out[c] = s * scale;
}

static inline int pick_cb(int C) { return C >= 64 ? 64 : (C >= 32 ? 32 : (C >= 16 ? 16 : (C >= 8 ? 8 : (C >= 4 ? 4 : (C >= 2 ? 2 : 1))))); }

extern "C" int p2p_norm_act_fwd(int dtype, int N, int H, int W, int C, const void* raw, int raw_kind, int nslabs,
                                long long slab_stride, const float* gamma, const float* beta, float eps, int act,
                                float alpha, const unsigned char* mask, const p2p_tensor* out, void* raw_out,
                                float* stats, void* stream) {
    P2P_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0, "p2p_norm_act_fwd: bad shape");
    P2P_REQUIRE(raw && out && out->ptr, "p2p_norm_act_fwd: null pointer");
    P2P_REQUIRE(raw_kind == 1 || (raw_kind == 2 && nslabs >= 1), "p2p_norm_act_fwd: bad raw_kind/nslabs");
    P2P_REQUIRE((gamma == nullptr) == (beta == nullptr), "p2p_norm_act_fwd: gamma/beta must both be set or null");
    P2P_REQUIRE(!gamma || stats, "p2p_norm_act_fwd: stats required with normalisation");
    int CB = pick_cb(C);
    dim3 grid(N, (C + CB - 1) / CB);
    size_t shm = sizeof(float) * 256;
    P2P_DISPATCH_DTYPE(dtype, (norm_act_fwd_kernel<T><<<grid, 256, shm, (hipStream_t)stream>>>(
                                  H, W, C, CB, raw, raw_kind, nslabs, slab_stride, gamma, beta, eps, act, alpha, mask,
                                  make_view(out), (T*)raw_out, stats)));
    return p2p_check_launch("p2p_norm_act_fwd");
}

extern "C" int p2p_norm_act_bwd(int dtype, int N, int H, int W, int C, const void* raw, const float* stats,
                                const float* gamma, const float* beta, int act, float alpha,
                                const unsigned char* mask, const p2p_gsrc* g1, const p2p_gsrc* g2,
                                const p2p_tensor* draw, float* dgamma_part, float* dbeta_part, void* stream) {
    P2P_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0, "p2p_norm_act_bwd: bad shape");
    P2P_REQUIRE(raw && draw && draw->ptr && g1, "p2p_norm_act_bwd: null pointer");
    P2P_REQUIRE(!gamma || (stats && beta && dgamma_part && dbeta_part), "p2p_norm_act_bwd: norm needs stats/beta/partials");
    int CB = pick_cb(C);
    dim3 grid(N, (C + CB - 1) / CB);
    size_t shm = sizeof(float) * 512;
    P2P_DISPATCH_DTYPE(dtype, (norm_act_bwd_kernel<T><<<grid, 256, shm, (hipStream_t)stream>>>(
                                  H, W, C, CB, (const T*)raw, stats, gamma, beta, act, alpha, mask, make_gsrc(g1),
                                  make_gsrc(g2), make_view(draw), dgamma_part, dbeta_part)));
    return p2p_check_launch("p2p_norm_act_bwd");
}

extern "C" int p2p_colsum(const float* part, int rows, int cols, float scale, float* out, void* stream) {
    P2P_REQUIRE(part && out && rows > 0 && cols > 0, "p2p_colsum: bad args");
    colsum_kernel<<<dim3((cols + 255) / 256), 256, 0, (hipStream_t)stream>>>(part, rows, cols, scale, out);
    return p2p_check_launch("p2p_colsum");
}
