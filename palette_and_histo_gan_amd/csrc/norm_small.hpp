// Pieces of the InstanceNorm + dropout + activation forward (norm_act.hip) that the implicit-GEMM kernel shares: on the small maps
// of the U-Net bottom (output maps of <= 16 pixels) the convolution's LAST workgroup normalises its images in the same launch
// (igemm.hip, p2p_igemm_norm_small) with exactly the code of p2p_norm_act_fwd's lane-group form -- same arithmetic, same order.
// Reference: tfa InstanceNormalization / Dropout / LeakyReLU / ReLU at networks.py:18-19,29-34.
#pragma once
#include "p2p_common.hpp"

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

template <typename T>
__device__ __forceinline__ void raw_vload(const void* raw, int raw_kind, int nslabs, long long slab, long long e, float* v) {
    constexpr int VN = VecOf<T>::N;
    if (raw_kind == 1) { vload<T>((const T*)raw + e, v); return; }
#pragma unroll
    for (int k = 0; k < VN; ++k) v[k] = 0.f;
    const float* p = (const float*)raw + e;
    for (int sIdx = 0; sIdx < nslabs; ++sIdx) {
#pragma unroll
        for (int k = 0; k < VN; k += 4) {
            f32x4 r = *(const f32x4*)(p + (long long)sIdx * slab + k);
            v[k] += r[0]; v[k + 1] += r[1]; v[k + 2] += r[2]; v[k + 3] += r[3];
        }
    }
#pragma unroll
    for (int k = 0; k < VN; ++k) v[k] = to_f32(from_f32<T>(v[k]));
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

// lane-group geometry of the small-map kernels: G = 2^lgG lanes per item, PPL pixels per lane
static inline void small_geom(int HW, int& lgG, int& ppl) {
    lgG = 0;
    while ((1 << lgG) < HW && lgG < 4) ++lgG;
    const int G = 1 << lgG;
    const int need = (HW + G - 1) / G;
    ppl = need <= 1 ? 1 : (need <= 2 ? 2 : 4);
}

// One lane group's share of the small-map forward: G = 2^lgG = min(16, HW) lanes own one (image n, VN-channel vector at c); lane g
// keeps its <= PPL pixels in registers, the statistics are exact two-pass sums combined with wave shuffles inside the lane group
// (no LDS, no barrier -- every lane of a group must call this together), and the result is written in the same pass:
// y = act(drop(gamma (x - mu) rsqrt(var + eps) + beta)) into the (haloed, channel-sliced) view `out`, (mu, rstd) into
// stats[n][C][2], the summed raw tensor into raw_out (if given).
template <typename T, int PPL>
__device__ __forceinline__ void norm_fwd_small_item(int n, int c, int g, int HW, int W, int C, int lgG, const void* __restrict__ raw,
                                                    int raw_kind, int nslabs, long long slab, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, float eps, int act, float alpha,
                                                    const unsigned char* __restrict__ mask, const TView& out, T* __restrict__ raw_out,
                                                    float* __restrict__ stats) {
    constexpr int VN = VecOf<T>::N;
    const int G = 1 << lgG;
    const long long base = (long long)n * HW * C + c;
    float x[PPL][VN];
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
        const int p = g + i * G;
        if (p < HW) raw_vload<T>(raw, raw_kind, nslabs, slab, base + (long long)p * C, x[i]);
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
