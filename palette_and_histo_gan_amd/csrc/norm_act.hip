// Fused InstanceNorm + Dropout + LeakyReLU/ReLU, forward and backward, writing straight into a channel
// slice of the (haloed) concat buffer.  Reference: tfa InstanceNormalization / Dropout / LeakyReLU / ReLU /
// Concatenate at networks.py:18-19,29-34,94; backward = the tape gradient of that chain, closed form in
// SURVEY.md 8a A13 (checked against autograd in tests/test_oracle.py).
#include "p2p_common.hpp"
#include <stdlib.h>

typedef __attribute__((__vector_size__(2 * sizeof(float)))) float f32x2;

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

// ---------------------------------------------------------------------------------------------------
// Vectorised variants (C % (16/sizeof(T)) == 0, 16-byte aligned views): one workgroup = one image x one
// group of <= 64 channels; every lane moves 16 bytes (8 bf16 / 4 f32 channels of one pixel) per access, so a
// wave reads whole 128-byte pixel segments.  Pass 1 reduces per channel over the image (shifted sums: the
// first pixel's value is subtracted before squaring, so E[x^2]-E[x]^2 does not cancel), pass 2 re-reads the
// image (L2-resident: <= 512 KB per workgroup) and writes the result.
template <typename T> struct VecOf;
template <> struct VecOf<bf16_t> { static constexpr int N = 8; typedef bf16x8 type; };
template <> struct VecOf<float> { static constexpr int N = 4; typedef f32x4 type; };

template <typename T>
__device__ __forceinline__ void vload(const T* p, float* v) {
    typename VecOf<T>::type r = *(const typename VecOf<T>::type*)p;
#pragma unroll
    for (int k = 0; k < VecOf<T>::N; ++k) v[k] = to_f32((T)r[k]);
}
template <typename T>
__device__ __forceinline__ void vstore(T* p, const float* v) {
    typename VecOf<T>::type r;
#pragma unroll
    for (int k = 0; k < VecOf<T>::N; ++k) r[k] = from_f32<T>(v[k]);
    *(typename VecOf<T>::type*)p = r;
}

// SLABS: 1 = f32 split-K slabs summed one by one (the two-pass kernels: the unrolled form below costs them registers -- the
// 64x64x32 backward launch of c2 went 68 -> 96 us with it), 4 = four slabs per trip, 0 = dense input only (the slab code and
// its registers are compiled out: the host picks the instantiation by what the launch reads).
template <typename T, int SLABS = 1>
__device__ __forceinline__ void raw_vload(const void* raw, int raw_kind, int nslabs, long long slab, long long e, float* v) {
    constexpr int VN = VecOf<T>::N;
    if (SLABS == 0 || raw_kind == 1) { vload<T>((const T*)raw + e, v); return; }
#pragma unroll
    for (int k = 0; k < VN; ++k) v[k] = 0.f;
    const float* p = (const float*)raw + e;
    // four slabs per trip: their loads are issued together, the sums keep the slab order (a small batch reads 4-16 slabs per
    // element and the one-slab loop was a chain of dependent memory latencies: c1 14 us for 32 K elements)
    int sIdx = 0;
    for (; SLABS == 4 && sIdx + 4 <= nslabs; sIdx += 4) {
        f32x4 r[4][VN / 4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < VN; k += 4) r[u][k / 4] = *(const f32x4*)(p + (long long)(sIdx + u) * slab + k);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < VN; k += 4) {
                v[k] += r[u][k / 4][0]; v[k + 1] += r[u][k / 4][1]; v[k + 2] += r[u][k / 4][2]; v[k + 3] += r[u][k / 4][3];
            }
    }
    for (; sIdx < nslabs; ++sIdx) {
#pragma unroll
        for (int k = 0; k < VN; k += 4) {
            f32x4 r = *(const f32x4*)(p + (long long)sIdx * slab + k);
            v[k] += r[0]; v[k + 1] += r[1]; v[k + 2] += r[2]; v[k + 3] += r[3];
        }
    }
#pragma unroll
    for (int k = 0; k < VN; ++k) v[k] = to_f32(from_f32<T>(v[k]));
}

template <typename T, int SLABS = 1>
__device__ __forceinline__ void gsrc_vload(const GSrc& g, long long pix, int c, float* v) {
    constexpr int VN = VecOf<T>::N;
    if (g.kind == 0) {
#pragma unroll
        for (int k = 0; k < VN; ++k) v[k] = 0.f;
        return;
    }
    long long e = pix * g.ld + g.coff + c;
    if (SLABS == 0 || g.kind == 1) { vload<T>((const T*)g.ptr + e, v); return; }
#pragma unroll
    for (int k = 0; k < VN; ++k) v[k] = 0.f;
    const float* p = (const float*)g.ptr + e;
    int sIdx = 0;
    if (SLABS == 4 && (g.slab & 3) == 0 && (e & 3) == 0 && ((uintptr_t)g.ptr & 15) == 0) {         // whole 16-byte loads, four slabs per trip, sums in slab order (see raw_vload)
        for (; sIdx + 4 <= g.nslabs; sIdx += 4) {
            f32x4 r[4][VN / 4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < VN; k += 4) r[u][k / 4] = *(const f32x4*)(p + (long long)(sIdx + u) * g.slab + k);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < VN; k += 4) {
                    v[k] += r[u][k / 4][0]; v[k + 1] += r[u][k / 4][1]; v[k + 2] += r[u][k / 4][2]; v[k + 3] += r[u][k / 4][3];
                }
        }
    }
    for (; sIdx < g.nslabs; ++sIdx)
#pragma unroll
        for (int k = 0; k < VN; ++k) v[k] += p[(long long)sIdx * g.slab + k];
}

__device__ __forceinline__ void mask_vload8(const unsigned char* m, float* keep, int VN) {
    if (VN == 8) {
        unsigned long long r = *(const unsigned long long*)m;
#pragma unroll
        for (int k = 0; k < 8; ++k) keep[k] = ((r >> (8 * k)) & 0xff) ? 2.f : 0.f;
    } else {
        unsigned r = *(const unsigned*)m;
#pragma unroll
        for (int k = 0; k < 4; ++k) keep[k] = ((r >> (8 * k)) & 0xff) ? 2.f : 0.f;
    }
}

// Column totals of two per-thread partial vectors over the PR pixel rows of a 256-thread workgroup (fixed order, so
// deterministic).  Three short LDS stages instead of every thread re-adding all PR rows: (1) partials to red[.][pr][CG],
// (2) thread (q, ch) adds rows q, q+Q, ... (Q = 256 / CG, i.e. VN rows each), (3) the first CG threads add the Q
// part-sums and publish them at red[.][256 + ch].
template <int VN>
__device__ __forceinline__ void col_reduce2(float (*red)[2048], int CG, int PR, int vid, int pr, const float* s1,
                                            const float* s2, float* t1, float* t2) {
#pragma unroll
    for (int k = 0; k < VN; k += 4) {
        *(f32x4*)&red[0][pr * CG + vid * VN + k] = f32x4{s1[k], s1[k + 1], s1[k + 2], s1[k + 3]};
        *(f32x4*)&red[1][pr * CG + vid * VN + k] = f32x4{s2[k], s2[k + 1], s2[k + 2], s2[k + 3]};
    }
    __syncthreads();
    const int Q = 256 / CG;
    const int ch = threadIdx.x % CG, q = threadIdx.x / CG;
    float a = 0.f, b = 0.f;
    for (int i = q; i < PR; i += Q) { a += red[0][i * CG + ch]; b += red[1][i * CG + ch]; }
    __syncthreads();
    red[0][q * CG + ch] = a;
    red[1][q * CG + ch] = b;
    __syncthreads();
    if (threadIdx.x < CG) {
        float ta = 0.f, tb = 0.f;
        for (int i = 0; i < Q; ++i) { ta += red[0][i * CG + threadIdx.x]; tb += red[1][i * CG + threadIdx.x]; }
        red[0][256 + threadIdx.x] = ta;
        red[1][256 + threadIdx.x] = tb;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < VN; ++k) { t1[k] = red[0][256 + vid * VN + k]; t2[k] = red[1][256 + vid * VN + k]; }
}

// MODE 0: one launch per layer (statistics + apply, grid.z = 1 when normalising).
// MODE 1 / 2: the pixel range of every image is split over grid.z workgroups; launch 1 writes per-split partial
// sums to ws[N][SP][C][2], launch 2 adds them up (fixed order: deterministic) and applies.  Used when one workgroup
// per (image, 64 channels) would leave most of the chip idle (64x64 maps with 32-64 channels).
template <typename T, int MODE>
__global__ __launch_bounds__(256) void norm_act_fwd_vec(int H, int W, int C, int CG, const void* __restrict__ raw,
                                                        int raw_kind, int nslabs, long long slab,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float eps, int act, float alpha,
                                                        const unsigned char* __restrict__ mask, TView out,
                                                        T* __restrict__ raw_out, float* __restrict__ stats,
                                                        float* __restrict__ ws, int nslots, TView tail, int tail_vecs) {
    constexpr int VN = VecOf<T>::N;
    __shared__ __attribute__((aligned(16))) float red[2][2048];      // [2][PR][CG], PR * CG = 256 * VN <= 2048
    const int n = blockIdx.x;
    const int cg0 = blockIdx.y * CG;
    const int SP = gridDim.z, sp = blockIdx.z;
    const int VPP = CG / VN, PR = 256 / VPP;
    const int vid = threadIdx.x % VPP, pr = threadIdx.x / VPP;
    const int c = cg0 + vid * VN;
    const int HW = H * W;
    const int chunk = (HW + SP - 1) / SP;
    const int pb = sp * chunk, pe = min(HW, pb + chunk);
    const long long base = (long long)n * HW * C + c;
    float mu[VN], rs[VN], ga[VN], be[VN];
    if (gamma) {
        float sh[VN], t1[VN], t2[VN];
        if (MODE != 3) raw_vload<T>(raw, raw_kind, nslabs, slab, base, sh);        // image-wide shift: the first pixel's value
        if (MODE < 2) {
            float s1[VN], s2[VN];
#pragma unroll
            for (int k = 0; k < VN; ++k) { s1[k] = 0.f; s2[k] = 0.f; }
            for (int p = pb + pr; p < pe; p += PR) {
                float x[VN];
                raw_vload<T>(raw, raw_kind, nslabs, slab, base + (long long)p * C, x);
#pragma unroll
                for (int k = 0; k < VN; ++k) { float d = x[k] - sh[k]; s1[k] += d; s2[k] += d * d; }
            }
            col_reduce2<VN>(red, CG, PR, vid, pr, s1, s2, t1, t2);
            if (MODE == 1) {
                if (pr == 0)
#pragma unroll
                    for (int k = 0; k < VN; ++k) {
                        ws[(((long long)n * SP + sp) * C + c + k) * 2 + 0] = t1[k];
                        ws[(((long long)n * SP + sp) * C + c + k) * 2 + 1] = t2[k];
                    }
                return;
            }
        } else if (MODE == 2) {
            // add the SP split partials once per workgroup (one thread per channel, fixed order), share through LDS
            if (threadIdx.x < CG) {
                const int cc = cg0 + threadIdx.x;
                float a1 = 0.f, a2 = 0.f;
                for (int i = 0; i < SP; ++i) {
                    const f32x2 v = *(const f32x2*)(ws + (((long long)n * SP + i) * C + cc) * 2);
                    a1 += v[0]; a2 += v[1];
                }
                red[0][threadIdx.x] = a1;
                red[1][threadIdx.x] = a2;
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < VN; ++k) { t1[k] = red[0][vid * VN + k]; t2[k] = red[1][vid * VN + k]; }
        } else {
            // MODE 3: `ws` holds the conv epilogue's per-slot (mean, centred sum of squares) of equal-sized pixel groups
            // (p2p_igemm stat_part, nslots groups per image): pooled mean and variance by the parallel-variance rule,
            // computed once per workgroup (one thread per channel) and shared through LDS.
            if (threadIdx.x < CG) {
                const int cc = cg0 + threadIdx.x;
                float ms = 0.f;
                for (int i = 0; i < nslots; ++i) ms += ws[(((long long)n * nslots + i) * C + cc) * 2 + 0];
                const float mall = ms / (float)nslots;
                const float cnt = (float)HW / (float)nslots;
                float m2 = 0.f;
                for (int i = 0; i < nslots; ++i) {
                    float mi = ws[(((long long)n * nslots + i) * C + cc) * 2 + 0];
                    m2 += ws[(((long long)n * nslots + i) * C + cc) * 2 + 1] + cnt * (mi - mall) * (mi - mall);
                }
                red[0][threadIdx.x] = mall;
                red[1][threadIdx.x] = m2;
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < VN; ++k) {
                sh[k] = red[0][vid * VN + k];     // shift := pooled mean, so t1 = 0 and t2 = pooled centred sum of squares
                t1[k] = 0.f;
                t2[k] = red[1][vid * VN + k];
            }
        }
#pragma unroll
        for (int k = 0; k < VN; ++k) {
            float m = t1[k] / (float)HW;
            float var = fmaxf(t2[k] / (float)HW - m * m, 0.f);
            mu[k] = sh[k] + m;
            rs[k] = rsqrtf(var + eps);
            ga[k] = gamma[c + k];
            be[k] = beta[c + k];
            if (pr == 0 && sp == 0) {
                stats[((long long)n * C + c + k) * 2 + 0] = mu[k];
                stats[((long long)n * C + c + k) * 2 + 1] = rs[k];
            }
        }
    }
    if (raw_kind == 1 && !raw_out) {
        // Dense input in the activation type (every launch of a bf16 step): U pixels per trip, every load of the trip issued
        // before its first store and kept PACKED until used.  The output view may alias the inputs as far as the compiler
        // knows, so the one-pixel loop below is "load, wait, store" with one 16-byte load in flight per lane.
        typedef typename VecOf<T>::type vec_t;
        constexpr int U = 4;
        const bool do_tail = tail_vecs && c + VN == C;
        for (int p0 = pb + pr; p0 < pe; p0 += PR * U) {
            vec_t xr[U], tr[U];
            unsigned long long mk[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int p = p0 + u * PR;
                if (p < pe) {
                    const long long e = base + (long long)p * C;
                    xr[u] = *(const vec_t*)((const T*)raw + e);
                    if (mask) mk[u] = VN == 8 ? *(const unsigned long long*)(mask + e) : (unsigned long long)*(const unsigned*)(mask + e);
                    if (do_tail) {
                        const int yy = p / W, xx = p - yy * W;
                        tr[u] = *(const vec_t*)((const T*)tail.ptr + tail.off(n, yy, xx));
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int p = p0 + u * PR;
                if (p >= pe) continue;
                vec_t o;
#pragma unroll
                for (int k = 0; k < VN; ++k) {
                    const float xv = to_f32((T)xr[u][k]);
                    float y = gamma ? (xv - mu[k]) * rs[k] * ga[k] + be[k] : xv;
                    if (mask) y *= ((mk[u] >> (8 * k)) & 0xff) ? 2.f : 0.f;
                    if (act == P2P_ACT_LEAKY) y = y > 0.f ? y : alpha * y;
                    else if (act == P2P_ACT_RELU) y = y > 0.f ? y : 0.f;
                    o[k] = from_f32<T>(y);
                }
                const int yy = p / W, xx = p - yy * W;
                T* op = (T*)out.ptr + out.off(n, yy, xx);
                *(vec_t*)(op + c) = o;
                // tail: the channels that follow this layer's slice in the concat buffer are copied by the lane that wrote the
                // slice's last vector, so the pixel leaves the wave complete (no partially written 32-byte sectors in HBM)
                if (do_tail) {
                    *(vec_t*)(op + C) = tr[u];
                    for (int t = 1; t < tail_vecs; ++t)
                        *(vec_t*)(op + C + t * VN) = *(const vec_t*)((const T*)tail.ptr + tail.off(n, yy, xx) + t * VN);
                }
            }
        }
        return;
    }
    for (int p = pb + pr; p < pe; p += PR) {
        long long e = base + (long long)p * C;
        float x[VN], keep[VN];
        raw_vload<T>(raw, raw_kind, nslabs, slab, e, x);
        if (raw_out) vstore<T>(raw_out + e, x);
        if (mask) mask_vload8(mask + e, keep, VN);
#pragma unroll
        for (int k = 0; k < VN; ++k) {
            float y = gamma ? (x[k] - mu[k]) * rs[k] * ga[k] + be[k] : x[k];
            if (mask) y *= keep[k];
            if (act == P2P_ACT_LEAKY) y = y > 0.f ? y : alpha * y;
            else if (act == P2P_ACT_RELU) y = y > 0.f ? y : 0.f;
            x[k] = y;
        }
        int yy = p / W, xx = p - yy * W;
        vstore<T>((T*)out.ptr + out.off(n, yy, xx) + c, x);
        // tail: the channels that follow this layer's slice in the concat buffer are copied by the lane that wrote the
        // slice's last vector, so the pixel leaves the wave complete (no partially written 32-byte sectors in HBM)
        if (tail_vecs && c + VN == C) {
            const long long to = tail.off(n, yy, xx), oo = out.off(n, yy, xx) + C;
            for (int t = 0; t < tail_vecs; ++t) {
                float tv[VN];
                vload<T>((const T*)tail.ptr + to + t * VN, tv);
                vstore<T>((T*)out.ptr + oo + t * VN, tv);
            }
        }
    }
}

template <typename T, int MODE>
__global__ __launch_bounds__(256) void norm_act_bwd_vec(int H, int W, int C, int CG, const T* __restrict__ raw,
                                                        const float* __restrict__ stats, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int act, float alpha,
                                                        const unsigned char* __restrict__ mask, GSrc g1, GSrc g2,
                                                        TView draw, float* __restrict__ dgamma_part,
                                                        float* __restrict__ dbeta_part, float* __restrict__ ws) {
    constexpr int VN = VecOf<T>::N;
    __shared__ __attribute__((aligned(16))) float red[2][2048];
    const int n = blockIdx.x;
    const int cg0 = blockIdx.y * CG;
    const int SP = gridDim.z, sp = blockIdx.z;
    const int VPP = CG / VN, PR = 256 / VPP;
    const int vid = threadIdx.x % VPP, pr = threadIdx.x / VPP;
    const int c = cg0 + vid * VN;
    const int HW = H * W;
    const int chunk = (HW + SP - 1) / SP;
    const int pb = sp * chunk, pe = min(HW, pb + chunk);
    const long long pix0 = (long long)n * HW;
    const long long base = pix0 * C + c;
    float mu[VN], rs[VN], ga[VN], be[VN];
#pragma unroll
    for (int k = 0; k < VN; ++k) {
        mu[k] = stats[((long long)n * C + c + k) * 2 + 0];
        rs[k] = stats[((long long)n * C + c + k) * 2 + 1];
        ga[k] = gamma[c + k];
        be[k] = beta[c + k];
    }
    // d(yhat) and xhat of one pixel
    auto dyhat = [&](int p, float* d, float* xh) {
        long long e = base + (long long)p * C;
        float x[VN], a1[VN], a2[VN], keep[VN];
        vload<T>(raw + e, x);
        gsrc_vload<T>(g1, pix0 + p, c, a1);
        gsrc_vload<T>(g2, pix0 + p, c, a2);
        if (mask) mask_vload8(mask + e, keep, VN);
#pragma unroll
        for (int k = 0; k < VN; ++k) {
            xh[k] = (x[k] - mu[k]) * rs[k];
            float a = xh[k] * ga[k] + be[k];
            float kp = mask ? keep[k] : 1.f;
            a *= kp;
            float slope = 1.f;
            if (act == P2P_ACT_LEAKY) slope = a > 0.f ? 1.f : alpha;
            else if (act == P2P_ACT_RELU) slope = a > 0.f ? 1.f : 0.f;
            d[k] = (a1[k] + a2[k]) * slope * kp;
        }
    };
    float t1[VN], t2[VN];
    if (MODE != 2) {
        float s1[VN], s2[VN];
#pragma unroll
        for (int k = 0; k < VN; ++k) { s1[k] = 0.f; s2[k] = 0.f; }
        for (int p = pb + pr; p < pe; p += PR) {
            float d[VN], xh[VN];
            dyhat(p, d, xh);
#pragma unroll
            for (int k = 0; k < VN; ++k) { s1[k] += d[k]; s2[k] += d[k] * xh[k]; }
        }
        col_reduce2<VN>(red, CG, PR, vid, pr, s1, s2, t1, t2);
        if (MODE == 1) {
            if (pr == 0)
#pragma unroll
                for (int k = 0; k < VN; ++k) {
                    ws[(((long long)n * SP + sp) * C + c + k) * 2 + 0] = t1[k];
                    ws[(((long long)n * SP + sp) * C + c + k) * 2 + 1] = t2[k];
                }
            return;
        }
    } else {
        if (threadIdx.x < CG) {
            const int cc = cg0 + threadIdx.x;
            float a1 = 0.f, a2 = 0.f;
            for (int i = 0; i < SP; ++i) {
                const f32x2 v = *(const f32x2*)(ws + (((long long)n * SP + i) * C + cc) * 2);
                a1 += v[0]; a2 += v[1];
            }
            red[0][threadIdx.x] = a1;
            red[1][threadIdx.x] = a2;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < VN; ++k) { t1[k] = red[0][vid * VN + k]; t2[k] = red[1][vid * VN + k]; }
    }
    float m1[VN], m2[VN];
#pragma unroll
    for (int k = 0; k < VN; ++k) {
        if (pr == 0 && sp == 0) {
            dbeta_part[(long long)n * C + c + k] = t1[k];
            dgamma_part[(long long)n * C + c + k] = t2[k];
        }
        m1[k] = t1[k] / (float)HW;
        m2[k] = t2[k] / (float)HW;
    }
    for (int p = pb + pr; p < pe; p += PR) {
        float d[VN], xh[VN];
        dyhat(p, d, xh);
#pragma unroll
        for (int k = 0; k < VN; ++k) d[k] = ga[k] * rs[k] * (d[k] - m1[k] - xh[k] * m2[k]);
        int yy = p / W, xx = p - yy * W;
        vstore<T>((T*)draw.ptr + draw.off(n, yy, xx) + c, d);
    }
}

// ---------------------------------------------------------------------------------------------------
// Register-resident form (r05): one launch, every operand read ONCE.  A workgroup owns (image, CG channels) like MODE 0 above,
// but a thread keeps its <= PPL pixels (x packed in the storage type, d(yhat) in f32) in registers between the reduction and
// the apply pass instead of reading them again, and all of its loads are issued before the first use.  Geometry (host side,
// norm_bwd_reg_geom): the widest channel group whose pixel rows fit PPL <= 8 and that still yields
// >= 512 workgroups, else the narrowest one -- at batch 4 a 16x16x128 tensor becomes 64 workgroups of one pixel per thread
// instead of 16 workgroups looping twice over 16 pixels (c1: 29 us for two launches -> one launch-sized kernel).  The channel
// groups of one image sit on ONE XCD, next to each other in dispatch order (workgroup id = ((n / 8) * groups + group) * 8 + n % 8):
// with groups narrower than a 128-byte line the line's other readers find it in that XCD's L2.
template <typename T, int PPL, int SLABS>
__global__ __launch_bounds__(256) void norm_act_bwd_reg(int N, int HW, int W, int C, int CG, const T* __restrict__ raw,
                                                        const float* __restrict__ stats, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int act, float alpha,
                                                        const unsigned char* __restrict__ mask, GSrc g1, GSrc g2, TView draw,
                                                        float* __restrict__ dgamma_part, float* __restrict__ dbeta_part) {
    constexpr int VN = VecOf<T>::N;
    typedef typename VecOf<T>::type vec_t;
    __shared__ __attribute__((aligned(16))) float red[2][2048];
    const int ncg = C / CG;
    const int lin = blockIdx.x;
    const int n = ((lin >> 3) / ncg) * 8 + (lin & 7), cg0 = ((lin >> 3) % ncg) * CG;
    if (n >= N) return;                                  // (whole workgroup: the grid is padded to 8 images)
    const int VPP = CG / VN, PR = 256 / VPP;
    const int vid = threadIdx.x % VPP, pr = threadIdx.x / VPP;
    const int c = cg0 + vid * VN;
    const long long pix0 = (long long)n * HW;
    const long long base = pix0 * C + c;
    float mu[VN], rs[VN], ga[VN], be[VN];
#pragma unroll
    for (int k = 0; k < VN; ++k) {
        mu[k] = stats[((long long)n * C + c + k) * 2 + 0];
        rs[k] = stats[((long long)n * C + c + k) * 2 + 1];
        ga[k] = gamma[c + k];
        be[k] = beta[c + k];
    }
    vec_t xr[PPL];
    float d[PPL][VN];
    float s1[VN], s2[VN];
#pragma unroll
    for (int k = 0; k < VN; ++k) { s1[k] = 0.f; s2[k] = 0.f; }
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
        const int p = pr + i * PR;
        if (p < HW) {
            const long long e = base + (long long)p * C;
            float a1[VN], a2[VN], keep[VN];
            xr[i] = *(const vec_t*)(raw + e);
            gsrc_vload<T, SLABS>(g1, pix0 + p, c, a1);
            gsrc_vload<T, SLABS>(g2, pix0 + p, c, a2);
            if (mask) mask_vload8(mask + e, keep, VN);
#pragma unroll
            for (int k = 0; k < VN; ++k) {
                const float xh = (to_f32((T)xr[i][k]) - mu[k]) * rs[k];
                float a = xh * ga[k] + be[k];
                const float kp = mask ? keep[k] : 1.f;
                a *= kp;
                float slope = 1.f;
                if (act == P2P_ACT_LEAKY) slope = a > 0.f ? 1.f : alpha;
                else if (act == P2P_ACT_RELU) slope = a > 0.f ? 1.f : 0.f;
                d[i][k] = (a1[k] + a2[k]) * slope * kp;
                s1[k] += d[i][k];
                s2[k] += d[i][k] * xh;
            }
        } else {
#pragma unroll
            for (int k = 0; k < VN; ++k) d[i][k] = 0.f;
        }
    }
    float t1[VN], t2[VN];
    col_reduce2<VN>(red, CG, PR, vid, pr, s1, s2, t1, t2);
    float m1[VN], m2[VN];
#pragma unroll
    for (int k = 0; k < VN; ++k) {
        if (pr == 0) {
            dbeta_part[(long long)n * C + c + k] = t1[k];
            dgamma_part[(long long)n * C + c + k] = t2[k];
        }
        m1[k] = t1[k] / (float)HW;
        m2[k] = t2[k] / (float)HW;
    }
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
        const int p = pr + i * PR;
        if (p >= HW) continue;
        float r[VN];
#pragma unroll
        for (int k = 0; k < VN; ++k) {
            const float xh = (to_f32((T)xr[i][k]) - mu[k]) * rs[k];
            r[k] = ga[k] * rs[k] * (d[i][k] - m1[k] - xh * m2[k]);
        }
        const int yy = p / W, xx = p - yy * W;
        vstore<T>((T*)draw.ptr + draw.off(n, yy, xx) + c, r);
    }
}

// Geometry of the register-resident form: channel group CG and pixels per thread (1, 2, 4, 8), or CG = 0 when the map
// does not fit.
static inline int norm_bwd_reg_geom(int Neff, int HW, int C, int vn, int& ppl_out, bool fwd = false) {
    static int max_hw[2] = {-1, -1};          // largest map served, 0 = off (P2P_NORM_BWD_REG / P2P_NORM_FWD_REG: A/B runs)
    if (max_hw[fwd] < 0) { const char* e = getenv(fwd ? "P2P_NORM_FWD_REG" : "P2P_NORM_BWD_REG"); max_hw[fwd] = e ? atoi(e) : 4096; }
    static int min_wgs = -1;                  // workgroups wanted before a wider channel group is accepted (P2P_NORM_REG_WGS: sweeps)
    if (min_wgs < 0) { const char* e = getenv("P2P_NORM_REG_WGS"); min_wgs = e ? atoi(e) : 512; }
    ppl_out = 0;
    if (HW > max_hw[fwd]) return 0;
    int pick = 0, pick_ppl = 0;
    for (int CG = 64; CG >= vn; CG >>= 1) {
        if (CG > C || C % CG) continue;
        const int pr = 256 / (CG / vn);
        int ppl = (HW + pr - 1) / pr;
        ppl = ppl <= 1 ? 1 : (ppl <= 2 ? 2 : (ppl <= 4 ? 4 : 8));
        if ((long long)ppl * pr < HW) continue;        // more than 8 pixels per thread (c1: 16 of them on 16 workgroups were slower than the split form)
        pick = CG; pick_ppl = ppl;
        if ((long long)Neff * (C / CG) >= min_wgs) break;
    }
    ppl_out = pick_ppl;
    return pick;
}

// Register-resident forward (r05), the counterpart of norm_act_bwd_reg above: one launch, the input (dense, or f32 split-K slabs
// summed in slab order and rounded through the storage type like everywhere else) is read ONCE, a thread keeps its <= 8 pixels in
// registers between the statistics and the apply pass.  Same statistics as the two-pass form: sums shifted by the image's first
// pixel.  At batch 4 this is what makes "split-K convolution + normalisation" cheaper than the fused block on two workgroups
// (norm over 16 slabs: 36 -> ~10 us per layer).
template <typename T, int PPL, int SLABS>
__global__ __launch_bounds__(256) void norm_act_fwd_reg(int N, int HW, int W, int C, int CG, const void* __restrict__ raw,
                                                        int raw_kind, int nslabs, long long slab,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float eps, int act, float alpha,
                                                        const unsigned char* __restrict__ mask, TView out,
                                                        T* __restrict__ raw_out, float* __restrict__ stats) {
    constexpr int VN = VecOf<T>::N;
    __shared__ __attribute__((aligned(16))) float red[2][2048];
    const int ncg = C / CG;
    const int lin = blockIdx.x;
    const int n = ((lin >> 3) / ncg) * 8 + (lin & 7), cg0 = ((lin >> 3) % ncg) * CG;
    if (n >= N) return;
    const int VPP = CG / VN, PR = 256 / VPP;
    const int vid = threadIdx.x % VPP, pr = threadIdx.x / VPP;
    const int c = cg0 + vid * VN;
    const long long base = (long long)n * HW * C + c;
    float x[PPL][VN], sh[VN], s1[VN], s2[VN];
    unsigned long long mk[PPL];
    raw_vload<T, SLABS>(raw, raw_kind, nslabs, slab, base, sh);
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
        const int p = pr + i * PR;
        mk[i] = 0;
        if (p < HW) {
            const long long e = base + (long long)p * C;
            raw_vload<T, SLABS>(raw, raw_kind, nslabs, slab, e, x[i]);
            if (mask) mk[i] = VN == 8 ? *(const unsigned long long*)(mask + e) : (unsigned long long)*(const unsigned*)(mask + e);
        } else {
#pragma unroll
            for (int k = 0; k < VN; ++k) x[i][k] = 0.f;
        }
    }
#pragma unroll
    for (int k = 0; k < VN; ++k) { s1[k] = 0.f; s2[k] = 0.f; }
#pragma unroll
    for (int i = 0; i < PPL; ++i)
        if (pr + i * PR < HW)
#pragma unroll
            for (int k = 0; k < VN; ++k) { const float d = x[i][k] - sh[k]; s1[k] += d; s2[k] += d * d; }
    float t1[VN], t2[VN], mu[VN], rs[VN], ga[VN], be[VN];
    col_reduce2<VN>(red, CG, PR, vid, pr, s1, s2, t1, t2);
#pragma unroll
    for (int k = 0; k < VN; ++k) {
        const float m = t1[k] / (float)HW;
        const float var = fmaxf(t2[k] / (float)HW - m * m, 0.f);
        mu[k] = sh[k] + m;
        rs[k] = rsqrtf(var + eps);
        ga[k] = gamma[c + k];
        be[k] = beta[c + k];
        if (pr == 0) {
            stats[((long long)n * C + c + k) * 2 + 0] = mu[k];
            stats[((long long)n * C + c + k) * 2 + 1] = rs[k];
        }
    }
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
        const int p = pr + i * PR;
        if (p >= HW) continue;
        if (raw_out) vstore<T>(raw_out + base + (long long)p * C, x[i]);
        float y[VN];
#pragma unroll
        for (int k = 0; k < VN; ++k) {
            float v = (x[i][k] - mu[k]) * rs[k] * ga[k] + be[k];
            if (mask) v *= ((mk[i] >> (8 * k)) & 0xff) ? 2.f : 0.f;
            if (act == P2P_ACT_LEAKY) v = v > 0.f ? v : alpha * v;
            else if (act == P2P_ACT_RELU) v = v > 0.f ? v : 0.f;
            y[k] = v;
        }
        const int yy = p / W, xx = p - yy * W;
        vstore<T>((T*)out.ptr + out.off(n, yy, xx) + c, y);
    }
}

// ---------------------------------------------------------------------------------------------------
// Small maps (dispatched for H*W <= 16: the 1x1 ... 4x4 layers of the U-Net bottom; 8x8 measured faster on the workgroup form).  One workgroup per (image, 64 channels) leaves
// most lanes idle there and spends its time in barriers.  Here G = min(16, H*W) lanes own one (image, VN-channel
// vector): each lane keeps its <= 4 pixels in registers, the statistics are exact two-pass sums combined with wave
// shuffles inside the lane group (no LDS, no barrier), and the result is written in the same pass.
template <typename T, int PPL, int SLABS>
__global__ __launch_bounds__(256) void norm_act_fwd_small(int HW, int W, int C, int lgG, long long items, const void* __restrict__ raw,
                                                          int raw_kind, int nslabs, long long slab, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps, int act, float alpha,
                                                          const unsigned char* __restrict__ mask, TView out,
                                                          T* __restrict__ raw_out, float* __restrict__ stats) {
    constexpr int VN = VecOf<T>::N;
    const int G = 1 << lgG;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long item = gid >> lgG;
    const int g = (int)(gid & (G - 1));
    if (item >= items) return;                       // whole lane groups leave together (G divides 64)
    const int cvn = C / VN;
    const int n = (int)(item / cvn), c = (int)(item - (long long)n * cvn) * VN;
    const long long base = (long long)n * HW * C + c;
    float x[PPL][VN];
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
        const int p = g + i * G;
        if (p < HW) raw_vload<T, SLABS>(raw, raw_kind, nslabs, slab, base + (long long)p * C, x[i]);
        else {
#pragma unroll
            for (int k = 0; k < VN; ++k) x[i][k] = 0.f;
        }
    }
    float mu[VN], rs[VN];
    if (gamma) {
#pragma unroll
        for (int k = 0; k < VN; ++k) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < PPL; ++i) s += x[i][k];               // absent pixels hold 0
            for (int o = G >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            mu[k] = s / (float)HW;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < PPL; ++i) {
                const float d = (g + i * G < HW) ? x[i][k] - mu[k] : 0.f;
                q += d * d;
            }
            for (int o = G >> 1; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
            rs[k] = rsqrtf(q / (float)HW + eps);
            if (g == 0) {
                stats[((long long)n * C + c + k) * 2 + 0] = mu[k];
                stats[((long long)n * C + c + k) * 2 + 1] = rs[k];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
        const int p = g + i * G;
        if (p >= HW) continue;
        const long long e = base + (long long)p * C;
        float keep[VN], y[VN];
        if (raw_out) vstore<T>(raw_out + e, x[i]);
        if (mask) mask_vload8(mask + e, keep, VN);
#pragma unroll
        for (int k = 0; k < VN; ++k) {
            float v = gamma ? (x[i][k] - mu[k]) * rs[k] * gamma[c + k] + beta[c + k] : x[i][k];
            if (mask) v *= keep[k];
            if (act == P2P_ACT_LEAKY) v = v > 0.f ? v : alpha * v;
            else if (act == P2P_ACT_RELU) v = v > 0.f ? v : 0.f;
            y[k] = v;
        }
        const int yy = p / W, xx = p - yy * W;
        vstore<T>((T*)out.ptr + out.off(n, yy, xx) + c, y);
    }
}

template <typename T, int PPL, int SLABS>
__global__ __launch_bounds__(256) void norm_act_bwd_small(int HW, int W, int C, int lgG, long long items, const T* __restrict__ raw,
                                                          const float* __restrict__ stats, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, int act, float alpha,
                                                          const unsigned char* __restrict__ mask, GSrc g1, GSrc g2, TView draw,
                                                          float* __restrict__ dgamma_part, float* __restrict__ dbeta_part) {
    constexpr int VN = VecOf<T>::N;
    const int G = 1 << lgG;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long item = gid >> lgG;
    const int g = (int)(gid & (G - 1));
    if (item >= items) return;
    const int cvn = C / VN;
    const int n = (int)(item / cvn), c = (int)(item - (long long)n * cvn) * VN;
    const long long pix0 = (long long)n * HW, base = pix0 * C + c;
    float mu[VN], rs[VN], ga[VN], be[VN];
#pragma unroll
    for (int k = 0; k < VN; ++k) {
        mu[k] = stats[((long long)n * C + c + k) * 2 + 0];
        rs[k] = stats[((long long)n * C + c + k) * 2 + 1];
        ga[k] = gamma[c + k];
        be[k] = beta[c + k];
    }
    float d[PPL][VN], xh[PPL][VN];
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
        const int p = g + i * G;
        if (p < HW) {
            const long long e = base + (long long)p * C;
            float x[VN], a1[VN], a2[VN], keep[VN];
            vload<T>(raw + e, x);
            gsrc_vload<T, SLABS>(g1, pix0 + p, c, a1);
            gsrc_vload<T, SLABS>(g2, pix0 + p, c, a2);
            if (mask) mask_vload8(mask + e, keep, VN);
#pragma unroll
            for (int k = 0; k < VN; ++k) {
                xh[i][k] = (x[k] - mu[k]) * rs[k];
                float a = xh[i][k] * ga[k] + be[k];
                const float kp = mask ? keep[k] : 1.f;
                a *= kp;
                float slope = 1.f;
                if (act == P2P_ACT_LEAKY) slope = a > 0.f ? 1.f : alpha;
                else if (act == P2P_ACT_RELU) slope = a > 0.f ? 1.f : 0.f;
                d[i][k] = (a1[k] + a2[k]) * slope * kp;
            }
        } else {
#pragma unroll
            for (int k = 0; k < VN; ++k) { d[i][k] = 0.f; xh[i][k] = 0.f; }
        }
    }
    float m1[VN], m2[VN];
#pragma unroll
    for (int k = 0; k < VN; ++k) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int i = 0; i < PPL; ++i) { t1 += d[i][k]; t2 += d[i][k] * xh[i][k]; }
        for (int o = G >> 1; o > 0; o >>= 1) { t1 += __shfl_xor(t1, o, 64); t2 += __shfl_xor(t2, o, 64); }
        if (g == 0) {
            dbeta_part[(long long)n * C + c + k] = t1;
            dgamma_part[(long long)n * C + c + k] = t2;
        }
        m1[k] = t1 / (float)HW;
        m2[k] = t2 / (float)HW;
    }
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
        const int p = g + i * G;
        if (p >= HW) continue;
        float r[VN];
#pragma unroll
        for (int k = 0; k < VN; ++k) r[k] = ga[k] * rs[k] * (d[i][k] - m1[k] - xh[i][k] * m2[k]);
        const int yy = p / W, xx = p - yy * W;
        vstore<T>((T*)draw.ptr + draw.off(n, yy, xx) + c, r);
    }
}

// lane-group geometry of the small-map kernels: G = 2^lgG lanes per item, PPL pixels per lane
static inline void small_geom(int HW, int& lgG, int& ppl) {
    lgG = 0;
    while ((1 << lgG) < HW && lgG < 4) ++lgG;
    const int G = 1 << lgG;
    const int need = (HW + G - 1) / G;
    ppl = need <= 1 ? 1 : (need <= 2 ? 2 : 4);
}

// Batched column sums: task t reduces part[off_t .. off_t + rows*cols) (dense [rows][cols]) over rows into
// out[out_off_t .. + cols).  table = int32[ntasks][4] = {part_off, rows, cols, out_off} on the device.
// One launch replaces the per-layer dgamma/dbeta reductions of a whole backward pass.
__global__ __launch_bounds__(256) void colsum_batched_kernel(const float* __restrict__ part, const int* __restrict__ table,
                                                             float* __restrict__ out) {
    // 16 column quads (64 columns) x 16 row groups per workgroup; a lane adds rows rg, rg+16, ... of its 4 columns with
    // 16-byte loads (cols % 4 == 0: channel counts), the 16 groups are combined through LDS in a fixed order
    __shared__ f32x4 red[16][16];
    const int* t = table + 4 * blockIdx.y;
    const int poff = t[0], rows = t[1], cols = t[2], ooff = t[3];
    const int cq = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + cq * 4;
    if (blockIdx.x * 64 >= cols) return;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (c < cols) {
        const float* p = part + (long long)poff + c;
        if ((cols & 3) == 0 && (poff & 3) == 0) {
#pragma unroll 4
            for (int r = rg; r < rows; r += 16) s += *(const f32x4*)(p + (long long)r * cols);
        } else {
            for (int r = rg; r < rows; r += 16)
                for (int k = 0; k < 4; ++k)
                    if (c + k < cols) s[k] += p[(long long)r * cols + k];
        }
    }
    red[rg][cq] = s;
    __syncthreads();
    if (rg == 0 && c < cols) {
        f32x4 tsum = red[0][cq];
#pragma unroll
        for (int i = 1; i < 16; ++i) tsum += red[i][cq];
        for (int k = 0; k < 4; ++k)
            if (c + k < cols) out[ooff + c + k] = tsum[k];
    }
}

extern "C" int p2p_colsum_batched(const float* part, const int* table, int ntasks, int max_cols, float* out, void* stream) {
    P2P_REQUIRE(part && table && out && ntasks > 0 && max_cols > 0, "p2p_colsum_batched: bad args");
    P2P_REQUIRE(((uintptr_t)part % 16) == 0, "p2p_colsum_batched: partials must be 16-byte aligned");
    colsum_batched_kernel<<<dim3((max_cols + 63) / 64, ntasks), 256, 0, (hipStream_t)stream>>>(part, table, out);
    return p2p_check_launch("p2p_colsum_batched");
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

template <typename T>
__global__ void act_bwd_vec_kernel(int N, int H, int W, int C, TView actv, GSrc g1, GSrc g2, float alpha, TView draw) {
    constexpr int VN = VecOf<T>::N;
    const int cv = C / VN;
    long long total = (long long)N * H * W * cv;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int c = (int)(i % cv) * VN;
        long long p = i / cv;
        int x = (int)(p % W);
        int y = (int)((p / W) % H);
        int n = (int)(p / ((long long)W * H));
        float a[VN], ga[VN], gb[VN];
        vload<T>((const T*)actv.ptr + actv.off(n, y, x) + c, a);
        gsrc_vload<T>(g1, p, c, ga);
        gsrc_vload<T>(g2, p, c, gb);
#pragma unroll
        for (int k = 0; k < VN; ++k) { float g = ga[k] + gb[k]; a[k] = a[k] > 0.f ? g : alpha * g; }
        vstore<T>((T*)draw.ptr + draw.off(n, y, x) + c, a);
    }
}

extern "C" int p2p_act_bwd(int dtype, int N, int H, int W, int C, const p2p_tensor* act_out, const p2p_gsrc* g1,
                           const p2p_gsrc* g2, float alpha, const p2p_tensor* draw, void* stream) {
    P2P_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && act_out && act_out->ptr && g1 && draw && draw->ptr, "p2p_act_bwd: bad args");
    {
        const int esz = dtype == P2P_BF16 ? 2 : 4, vn = 16 / esz;
        auto gs_ok = [&](const p2p_gsrc* g) {
            if (!g || !g->ptr || g->kind == 0) return true;
            return g->kind != 1 || (g->ld % vn == 0 && g->coff % vn == 0 && ((uintptr_t)g->ptr % 16) == 0);
        };
        if (C % 8 == 0 && act_out->ld % vn == 0 && draw->ld % vn == 0 && ((uintptr_t)act_out->ptr % 16) == 0 &&
            ((uintptr_t)draw->ptr % 16) == 0 && gs_ok(g1) && gs_ok(g2)) {
            long long tv = (long long)N * H * W * (C / vn);
            long long bl = (tv + 255) / 256;
            if (bl > 8192) bl = 8192;
            P2P_DISPATCH_DTYPE(dtype, P2P_LAUNCH_LAST((act_bwd_vec_kernel<T>), dim3((unsigned)bl), dim3(256), 0, (hipStream_t)stream,
                                                      N, H, W, C, make_view(act_out), make_gsrc(g1), make_gsrc(g2), alpha, make_view(draw)));
            return p2p_check_launch("p2p_act_bwd");
        }
    }
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

static int norm_act_fwd_impl(int dtype, int N, int H, int W, int C, const void* raw, int raw_kind, int nslabs,
                             long long slab_stride, const float* gamma, const float* beta, float eps, int act,
                             float alpha, const unsigned char* mask, const p2p_tensor* out, void* raw_out,
                             float* stats, float* ws, long long ws_bytes, int nsplit, const p2p_tensor* tail, int tail_ch,
                             void* stream) {
    P2P_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0, "p2p_norm_act_fwd: bad shape");
    P2P_REQUIRE(raw && out && out->ptr, "p2p_norm_act_fwd: null pointer");
    P2P_REQUIRE(raw_kind == 1 || (raw_kind == 2 && nslabs >= 1), "p2p_norm_act_fwd: bad raw_kind/nslabs");
    P2P_REQUIRE((gamma == nullptr) == (beta == nullptr), "p2p_norm_act_fwd: gamma/beta must both be set or null");
    P2P_REQUIRE(!gamma || stats, "p2p_norm_act_fwd: stats required with normalisation");
    const int esz = dtype == P2P_BF16 ? 2 : 4, vn = 16 / esz;
    const bool vec = C % vn == 0 && out->ld % vn == 0 && ((uintptr_t)out->ptr % 16) == 0 && ((uintptr_t)raw % 16) == 0 &&
                     (!raw_out || ((uintptr_t)raw_out % 16) == 0) && (raw_kind == 1 || slab_stride % 4 == 0) && C % 8 == 0;
    const int tvecs = tail ? tail_ch / vn : 0;
    TView tv = tail ? make_view(tail) : TView{};
    if (tail) {
        // the tail copy lives in the workgroup form only (the caller asks for it on wide maps: the last skip connection)
        P2P_REQUIRE(tail->ptr && tail_ch > 0 && tail_ch % vn == 0 && tail->ld % vn == 0 && ((uintptr_t)tail->ptr % 16) == 0,
                    "p2p_norm_act_fwd_tail: the tail must be whole 16-byte vectors");
        int cgt = C > 64 ? 64 : C;
        while (C % cgt) cgt -= vn;
        P2P_REQUIRE(vec && H * W > 16 && 256 % (cgt / vn) == 0, "p2p_norm_act_fwd_tail: shape not served by the vector form");
    }
    if (vec && H * W <= 16) {
        // small maps: lane groups with register-resident pixels (also when the conv epilogue produced statistics:
        // recomputing them from <= 64 pixels is cheaper than pooling the slots)
        int lgG, ppl;
        small_geom(H * W, lgG, ppl);
        const long long items = (long long)N * (C / vn);
        const long long threads = items << lgG;
        const dim3 grid((unsigned)((threads + 255) / 256));
        hipStream_t st = (hipStream_t)stream;
#define NF_SMALL2(P_, S_) norm_act_fwd_small<T, P_, S_><<<grid, 256, 0, st>>>(H * W, W, C, lgG, items, raw, raw_kind, nslabs, slab_stride, gamma, \
                                                                   beta, eps, act, alpha, mask, make_view(out), (T*)raw_out, stats)
#define NF_SMALL(P_) do { if (raw_kind == 2) NF_SMALL2(P_, 4); else NF_SMALL2(P_, 0); } while (0)
        if (ppl == 1) { P2P_DISPATCH_DTYPE(dtype, NF_SMALL(1)); }
        else if (ppl == 2) { P2P_DISPATCH_DTYPE(dtype, NF_SMALL(2)); }
        else { P2P_DISPATCH_DTYPE(dtype, NF_SMALL(4)); }
#undef NF_SMALL
#undef NF_SMALL2
        return p2p_check_launch("p2p_norm_act_fwd");
    }
    // nsplit | 0x100 (the engine's f32 parity mode): the two-pass workgroup forms only -- their order of sums does not depend on N,
    // and the parity gate (tests/test_train_step_gpu.py, 1e-4 on every gradient tensor at batch 2) holds only while no ReLU gate
    // flips against the f64 oracle: a forward pass that rounds differently by a few ulp flips one (a 1e-2 error on a weight gradient)
    const bool legacy = nsplit > 0 && (nsplit & 0x100);
    const int n_geom = (nsplit > 0 && (nsplit & 0x200)) ? 256 : N;      // | 0x200: the register-resident geometry of batch 256 (tests)
    if (nsplit > 0) nsplit &= 0xff;
    if (vec && gamma && !tail && nsplit >= 0 && !legacy) {
        int ppl = 0;
        const int rcg = norm_bwd_reg_geom(n_geom, H * W, C, vn, ppl, true);
        if (rcg) {
            hipStream_t st = (hipStream_t)stream;
            const dim3 grid((unsigned)(((N + 7) / 8) * 8 * (C / rcg)));
#define NF_REG2(P_, S_) norm_act_fwd_reg<T, P_, S_><<<grid, 256, 0, st>>>(N, H * W, W, C, rcg, raw, raw_kind, nslabs, slab_stride, gamma, beta, eps, act, alpha, \
                                                                mask, make_view(out), (T*)raw_out, stats)
#define NF_REG(P_) do { if (raw_kind == 2) NF_REG2(P_, 4); else NF_REG2(P_, 0); } while (0)
            if (ppl == 1) { P2P_DISPATCH_DTYPE(dtype, NF_REG(1)); }
            else if (ppl == 2) { P2P_DISPATCH_DTYPE(dtype, NF_REG(2)); }
            else if (ppl == 4) { P2P_DISPATCH_DTYPE(dtype, NF_REG(4)); }
            else { P2P_DISPATCH_DTYPE(dtype, NF_REG(8)); }
#undef NF_REG
#undef NF_REG2
            return p2p_check_launch("p2p_norm_act_fwd");
        }
    }
    if (vec) {
        int CG = C > 64 ? 64 : C;
        while (C % CG) CG -= vn;
        int vpp = CG / vn;
        if (256 % vpp == 0) {
            hipStream_t st = (hipStream_t)stream;
            const int prr = 256 / vpp;
            int sp = nsplit < 1 ? 1 : nsplit;
            while (sp > 1 && (H * W + sp - 1) / sp < prr) sp >>= 1;       // every split keeps all pixel lanes busy
            if (nsplit < 0 && gamma && ws) {
                // statistics were produced by the conv epilogue (-nsplit slots per image in ws): apply only, and the
                // pixel range can be split freely for parallelism
                const int nslots = -nsplit;
                int sp2 = 1;
                while ((long long)N * (C / CG) * sp2 < 2048 && (H * W) / (sp2 * 2) >= prr && sp2 < 64) sp2 *= 2;
                dim3 grid3(N, C / CG, sp2);
                P2P_DISPATCH_DTYPE(dtype, (norm_act_fwd_vec<T, 3><<<grid3, 256, 0, st>>>(
                                              H, W, C, CG, raw, raw_kind, nslabs, slab_stride, gamma, beta, eps, act, alpha, mask,
                                              make_view(out), (T*)raw_out, stats, ws, nslots, tv, tvecs)));
                return p2p_check_launch("p2p_norm_act_fwd");
            }
            if (!ws || (long long)N * sp * C * 2 * 4 > ws_bytes) sp = gamma ? 1 : sp;
            dim3 grid(N, C / CG, sp);
            if (sp == 1 || !gamma) {
                P2P_DISPATCH_DTYPE(dtype, (norm_act_fwd_vec<T, 0><<<grid, 256, 0, st>>>(
                                              H, W, C, CG, raw, raw_kind, nslabs, slab_stride, gamma, beta, eps, act, alpha, mask,
                                              make_view(out), (T*)raw_out, stats, ws, 0, tv, tvecs)));
            } else {
                P2P_DISPATCH_DTYPE(dtype, (norm_act_fwd_vec<T, 1><<<grid, 256, 0, st>>>(
                                              H, W, C, CG, raw, raw_kind, nslabs, slab_stride, gamma, beta, eps, act, alpha, mask,
                                              make_view(out), (T*)raw_out, stats, ws, 0, tv, tvecs)));
                P2P_DISPATCH_DTYPE(dtype, (norm_act_fwd_vec<T, 2><<<grid, 256, 0, st>>>(
                                              H, W, C, CG, raw, raw_kind, nslabs, slab_stride, gamma, beta, eps, act, alpha, mask,
                                              make_view(out), (T*)raw_out, stats, ws, 0, tv, tvecs)));
            }
            return p2p_check_launch("p2p_norm_act_fwd");
        }
    }
    int CB = pick_cb(C);
    dim3 grid(N, (C + CB - 1) / CB);
    size_t shm = sizeof(float) * 256;
    P2P_DISPATCH_DTYPE(dtype, (norm_act_fwd_kernel<T><<<grid, 256, shm, (hipStream_t)stream>>>(
                                  H, W, C, CB, raw, raw_kind, nslabs, slab_stride, gamma, beta, eps, act, alpha, mask,
                                  make_view(out), (T*)raw_out, stats)));
    return p2p_check_launch("p2p_norm_act_fwd");
}

extern "C" int p2p_norm_act_fwd(int dtype, int N, int H, int W, int C, const void* raw, int raw_kind, int nslabs,
                                long long slab_stride, const float* gamma, const float* beta, float eps, int act,
                                float alpha, const unsigned char* mask, const p2p_tensor* out, void* raw_out,
                                float* stats, float* ws, long long ws_bytes, int nsplit, void* stream) {
    return norm_act_fwd_impl(dtype, N, H, W, C, raw, raw_kind, nslabs, slab_stride, gamma, beta, eps, act, alpha, mask, out,
                             raw_out, stats, ws, ws_bytes, nsplit, nullptr, 0, stream);
}

// As p2p_norm_act_fwd, and additionally copies `tail_ch` channels per pixel from `tail` (same N, H, W) into the channels that
// FOLLOW this layer's C channels in `out` (networks.py:92-94: the last concat is [up6 | input image]) -- the whole pixel of the
// concat buffer is written by one wave.  Written apart (the copy in p2p_pack_pair), the 16 of every 80 bytes were partial
// sector writes: 41 us of a 71 us launch with cold caches at B = 256 (tools/ubench/pack_abl.py).
extern "C" int p2p_norm_act_fwd_tail(int dtype, int N, int H, int W, int C, const void* raw, int raw_kind, int nslabs,
                                     long long slab_stride, const float* gamma, const float* beta, float eps, int act,
                                     float alpha, const unsigned char* mask, const p2p_tensor* out, void* raw_out,
                                     float* stats, float* ws, long long ws_bytes, int nsplit, const p2p_tensor* tail,
                                     int tail_ch, void* stream) {
    P2P_REQUIRE(tail, "p2p_norm_act_fwd_tail: null tail");
    return norm_act_fwd_impl(dtype, N, H, W, C, raw, raw_kind, nslabs, slab_stride, gamma, beta, eps, act, alpha, mask, out,
                             raw_out, stats, ws, ws_bytes, nsplit, tail, tail_ch, stream);
}

extern "C" int p2p_norm_act_bwd(int dtype, int N, int H, int W, int C, const void* raw, const float* stats,
                                const float* gamma, const float* beta, int act, float alpha,
                                const unsigned char* mask, const p2p_gsrc* g1, const p2p_gsrc* g2,
                                const p2p_tensor* draw, float* dgamma_part, float* dbeta_part, float* ws,
                                long long ws_bytes, int nsplit, void* stream) {
    P2P_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0, "p2p_norm_act_bwd: bad shape");
    P2P_REQUIRE(raw && draw && draw->ptr && g1, "p2p_norm_act_bwd: null pointer");
    P2P_REQUIRE(!gamma || (stats && beta && dgamma_part && dbeta_part), "p2p_norm_act_bwd: norm needs stats/beta/partials");
    const int esz = dtype == P2P_BF16 ? 2 : 4, vn = 16 / esz;
    auto gs_ok = [&](const p2p_gsrc* g) {
        if (!g || !g->ptr || g->kind == 0) return true;
        if (g->kind == 1) return g->ld % vn == 0 && g->coff % vn == 0 && ((uintptr_t)g->ptr % 16) == 0;
        return true;    // f32 slabs are read element-wise
    };
    const bool vec = gamma && C % 8 == 0 && draw->ld % vn == 0 && ((uintptr_t)draw->ptr % 16) == 0 &&
                     ((uintptr_t)raw % 16) == 0 && gs_ok(g1) && gs_ok(g2);
    const bool g_slabs = (g1 && g1->kind == 2) || (g2 && g2->kind == 2);      // picks the instantiation with the f32-slab loader
    if (vec && H * W <= 16) {
        int lgG, ppl;
        small_geom(H * W, lgG, ppl);
        const long long items = (long long)N * (C / vn);
        const long long threads = items << lgG;
        const dim3 grid((unsigned)((threads + 255) / 256));
        hipStream_t st = (hipStream_t)stream;
#define NB_SMALL2(P_, S_) P2P_LAUNCH_LAST((norm_act_bwd_small<T, P_, S_>), grid, dim3(256), 0, st, H * W, W, C, lgG, items, (const T*)raw, stats, gamma, beta, act, \
                                     alpha, mask, make_gsrc(g1), make_gsrc(g2), make_view(draw), dgamma_part, dbeta_part)
#define NB_SMALL(P_) do { if (g_slabs) NB_SMALL2(P_, 4); else NB_SMALL2(P_, 0); } while (0)
        if (ppl == 1) { P2P_DISPATCH_DTYPE(dtype, NB_SMALL(1)); }
        else if (ppl == 2) { P2P_DISPATCH_DTYPE(dtype, NB_SMALL(2)); }
        else { P2P_DISPATCH_DTYPE(dtype, NB_SMALL(4)); }
#undef NB_SMALL
#undef NB_SMALL2
        return p2p_check_launch("p2p_norm_act_bwd");
    }
    // nsplit | 0x100 (the engine's f32 parity mode): the two-pass workgroup forms only (see norm_act_fwd_impl)
    const bool legacy = nsplit > 0 && (nsplit & 0x100);
    const int n_geom = (nsplit > 0 && (nsplit & 0x200)) ? 256 : N;      // | 0x200: the register-resident geometry of batch 256 (tests)
    if (nsplit > 0) nsplit &= 0xff;
    if (vec && !legacy) {
        int ppl = 0;
        const int rcg = norm_bwd_reg_geom(n_geom, H * W, C, vn, ppl);
        if (rcg) {
            hipStream_t st = (hipStream_t)stream;
            const dim3 grid((unsigned)(((N + 7) / 8) * 8 * (C / rcg)));
#define NB_REG2(P_, S_) P2P_LAUNCH_LAST((norm_act_bwd_reg<T, P_, S_>), grid, dim3(256), 0, st, N, H * W, W, C, rcg, (const T*)raw, stats, gamma, beta, act, \
                                   alpha, mask, make_gsrc(g1), make_gsrc(g2), make_view(draw), dgamma_part, dbeta_part)
#define NB_REG(P_) do { if (g_slabs) NB_REG2(P_, 4); else NB_REG2(P_, 0); } while (0)
            if (ppl == 1) { P2P_DISPATCH_DTYPE(dtype, NB_REG(1)); }
            else if (ppl == 2) { P2P_DISPATCH_DTYPE(dtype, NB_REG(2)); }
            else if (ppl == 4) { P2P_DISPATCH_DTYPE(dtype, NB_REG(4)); }
            else { P2P_DISPATCH_DTYPE(dtype, NB_REG(8)); }
#undef NB_REG
#undef NB_REG2
            return p2p_check_launch("p2p_norm_act_bwd");
        }
    }
    if (vec) {
        int CG = C > 64 ? 64 : C;
        while (C % CG) CG -= vn;
        // a pixel split was asked for, but 32-channel groups alone already give >= 1024 workgroups: take those and keep
        // the one-launch form (no second read of x and the gradient from HBM)
        const bool narrow = nsplit > 1 && CG == 64 && C % 32 == 0 && (long long)N * (C / 32) >= 1024 && H * W >= 64;
        if (narrow) CG = 32;
        int vpp = CG / vn;
        if (256 % vpp == 0) {
            hipStream_t st = (hipStream_t)stream;
            const int prr = 256 / vpp;
            int sp = (nsplit < 1 || narrow) ? 1 : nsplit;
            while (sp > 1 && (H * W + sp - 1) / sp < prr) sp >>= 1;
            if (!ws || (long long)N * sp * C * 2 * 4 > ws_bytes) sp = 1;
            dim3 grid(N, C / CG, sp);
            if (sp == 1) {
                P2P_DISPATCH_DTYPE(dtype, P2P_LAUNCH_LAST((norm_act_bwd_vec<T, 0>), grid, dim3(256), 0, st,
                                                          H, W, C, CG, (const T*)raw, stats, gamma, beta, act, alpha, mask, make_gsrc(g1),
                                                          make_gsrc(g2), make_view(draw), dgamma_part, dbeta_part, ws));
            } else {
                P2P_DISPATCH_DTYPE(dtype, (norm_act_bwd_vec<T, 1><<<grid, 256, 0, st>>>(
                                              H, W, C, CG, (const T*)raw, stats, gamma, beta, act, alpha, mask, make_gsrc(g1),
                                              make_gsrc(g2), make_view(draw), dgamma_part, dbeta_part, ws)));
                P2P_DISPATCH_DTYPE(dtype, P2P_LAUNCH_LAST((norm_act_bwd_vec<T, 2>), grid, dim3(256), 0, st,
                                                          H, W, C, CG, (const T*)raw, stats, gamma, beta, act, alpha, mask, make_gsrc(g1),
                                                          make_gsrc(g2), make_view(draw), dgamma_part, dbeta_part, ws));
            }
            return p2p_check_launch("p2p_norm_act_bwd");
        }
    }
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
