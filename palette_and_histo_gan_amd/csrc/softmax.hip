// Palette-index head of Pix2PixIndexedModel (pix2pix_model.py:261-325): softmax over the 256 palette slots
// (networks.py:75-78 with last_activation="softmax"), tf.argmax(..., output_type=int32) (:286,292; ties -> lowest
// index), CategoricalCrossentropy against one_hot(real) (:265,274,300-301) and its gradient.  The one-hot tensor and
// the (B,S,S,256) probabilities are never materialised unless asked for: one wave owns one pixel, 4 logits per lane.
#include "p2p_common.hpp"

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Per-workgroup partials (deterministic: no float atomics; softmax_loss_sum_kernel adds them in workgroup order):
//   part[0][b] = inv_count * sum over the workgroup's pixels of -log softmax(z)[target] = log(sum exp(z - max)) - (z_t - max)
//                (segmentation loss, mean over pixels; the log-sum-exp form of Keras' logits path stays finite when
//                p_t underflows)
//   part[1][b] = inv_count/C * sum_pixels sum_c |onehot - p|    (the reported, zero-weighted L1 term, :263,273-278)
template <typename T>
__global__ __launch_bounds__(256) void softmax_cce_argmax_kernel(int N, int H, int W, int C, TView z, TView target,
                                                                TView fake_idx, float grad_scale, float inv_count,
                                                                TView dz, float* __restrict__ probs_out,
                                                                float* __restrict__ loss_part) {
    __shared__ float red[16];
    const int lane = threadIdx.x & 63;
    const long long wave_id = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    const long long M = (long long)N * H * W;
    const int per = (C + 63) / 64;          // logits per lane (4 for C = 256)
    float seg = 0.f, l1 = 0.f;
    for (long long m = wave_id; m < M; m += nwaves) {
        int x = (int)(m % W);
        int y = (int)((m / W) % H);
        int n = (int)(m / ((long long)W * H));
        const T* zp = (const T*)z.ptr + z.off(n, y, x);
        float v[8];
        float mx = -INFINITY;
        if (C == 256) {       // 4 logits per lane: one 8-byte (bf16) / 16-byte (f32) load
            typedef __attribute__((__vector_size__(4 * sizeof(T)))) T vec4_t;
            vec4_t r = *(const vec4_t*)(zp + lane * 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[k] = to_f32((T)r[k]); mx = fmaxf(mx, v[k]); }
        } else {
            for (int k = 0; k < per; ++k) {
                int c = lane * per + k;
                v[k] = c < C ? to_f32(zp[c]) : -INFINITY;
                mx = fmaxf(mx, v[k]);
            }
        }
        mx = wave_max(mx);
        const int t = (int)to_f32(((const T*)target.ptr)[target.off(n, y, x)]);
        float ztm = 0.f;            // z_t - max, taken before the exponentials
        for (int k = 0; k < per; ++k)
            if (lane * per + k == t) ztm = v[k] - mx;
        ztm = wave_sum(ztm);        // exactly one lane holds it
        float s = 0.f;
        for (int k = 0; k < per; ++k) { v[k] = expf(v[k] - mx); s += v[k]; }
        s = wave_sum(s);
        // probabilities exactly as tf.nn.softmax forms them: exp(z - max) / sum
        float best = -1.f;
        int besti = 0;
        for (int k = 0; k < per; ++k) {
            v[k] = v[k] / s;
            int c = lane * per + k;
            if (c < C && v[k] > best) { best = v[k]; besti = c; }     // strict '>' keeps the lowest index within the lane
        }
        // argmax across lanes, ties -> lowest index
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            float ob = __shfl_xor(best, o, 64);
            int oi = __shfl_xor(besti, o, 64);
            if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
        }
        float pt = 0.f;
        for (int k = 0; k < per; ++k)
            if (lane * per + k == t) pt = v[k];
        pt = wave_sum(pt);          // exactly one lane holds it
        if (lane == 0) {
            ((T*)fake_idx.ptr)[fake_idx.off(n, y, x)] = from_f32<T>((float)besti);
            seg += logf(s) - ztm;
            l1 += 2.f * (1.f - pt);          // sum_c |onehot_c - p_c| = (1 - p_t) + sum_{c != t} p_c
        }
        if (dz.ptr) {
            T* dp = (T*)dz.ptr + dz.off(n, y, x);
            if (C == 256) {
                typedef __attribute__((__vector_size__(4 * sizeof(T)))) T vec4_t;
                vec4_t r;
#pragma unroll
                for (int k = 0; k < 4; ++k) r[k] = from_f32<T>((v[k] - (lane * 4 + k == t ? 1.f : 0.f)) * grad_scale);
                *(vec4_t*)(dp + lane * 4) = r;
            } else {
                for (int k = 0; k < per; ++k) {
                    int c = lane * per + k;
                    if (c < C) dp[c] = from_f32<T>((v[k] - (c == t ? 1.f : 0.f)) * grad_scale);
                }
            }
        }
        if (probs_out)
            for (int k = 0; k < per; ++k) {
                int c = lane * per + k;
                if (c < C) probs_out[m * C + c] = v[k];
            }
    }
    seg = block_sum(seg, red);
    l1 = block_sum(l1, red);
    if (threadIdx.x == 0) {
        loss_part[blockIdx.x] = seg * inv_count;
        loss_part[gridDim.x + blockIdx.x] = l1 * inv_count / (float)C;
    }
}

// loss_out[k] = sum_b part[k][b], b in workgroup order (fixed order -> bit-reproducible)
__global__ __launch_bounds__(256) void softmax_loss_sum_kernel(const float* __restrict__ part, int nb, float* __restrict__ loss_out) {
    __shared__ float red[16];
    for (int k = 0; k < 2; ++k) {
        float s = 0.f;
        for (int b = threadIdx.x; b < nb; b += 256) s += part[k * nb + b];
        s = block_sum(s, red);
        if (threadIdx.x == 0) loss_out[k] = s;
        __syncthreads();
    }
}

// 256-way fast path: 16 lanes own one pixel (16 consecutive logits each, 16-byte accesses), a wave works on 4 pixels at
// once, every cross-lane step is a 4-level butterfly inside the 16-lane group (17 shuffles per group instead of 30 per
// wave and pixel), pixel decode in 32-bit arithmetic.  Same arithmetic as the generic kernel: exp(z - max) / sum, argmax
// with the lowest index on ties, -log p_t, 2 (1 - p_t).
template <typename T>
__global__ __launch_bounds__(256) void softmax256_kernel(int N, PixDec dec, TView z, TView target, TView fake_idx, float grad_scale,
                                                         float inv_count, TView dz, float* __restrict__ probs_out,
                                                         float* __restrict__ loss_part) {
    constexpr int VN = 16 / sizeof(T);               // elements per 16-byte access
    typedef __attribute__((__vector_size__(16))) T vec_t;
    __shared__ float red[16];
    const int lane = threadIdx.x & 63, sub = lane & 15;
    const unsigned M = (unsigned)N * dec.H * dec.W;
    const unsigned gid = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, ngroups = (gridDim.x * blockDim.x) >> 4;
    float seg = 0.f, l1 = 0.f;
    for (unsigned m = gid; m < M; m += ngroups) {
        int n, y, x;
        dec(m, n, y, x);
        const T* zp = (const T*)z.ptr + z.off(n, y, x) + sub * 16;
        float v[16];
#pragma unroll
        for (int q = 0; q < 16 / VN; ++q) {
            const vec_t r = *(const vec_t*)(zp + q * VN);
#pragma unroll
            for (int k = 0; k < VN; ++k) v[q * VN + k] = to_f32((T)r[k]);
        }
        float mx = v[0];
#pragma unroll
        for (int k = 1; k < 16; ++k) mx = fmaxf(mx, v[k]);
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        const int t = (int)to_f32(((const T*)target.ptr)[target.off(n, y, x)]);
        float zmine = 0.f;          // z_t - max of the lane that owns logit t, taken before the exponentials
#pragma unroll
        for (int k = 0; k < 16; ++k) zmine = (k == (t & 15)) ? v[k] - mx : zmine;
        const float ztm = __shfl(zmine, (lane & 48) + ((t >> 4) & 15), 64);
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) { v[k] = expf(v[k] - mx); sum += v[k]; }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        float best = -1.f;
        int besti = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            v[k] = v[k] / sum;
            if (v[k] > best) { best = v[k]; besti = sub * 16 + k; }      // strict '>': lowest index inside the lane
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(besti, o, 64);
            if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
        }
        float mine = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) mine = (k == (t & 15)) ? v[k] : mine;
        const float pt = __shfl(mine, (lane & 48) + ((t >> 4) & 15), 64);
        if (sub == 0) {
            ((T*)fake_idx.ptr)[fake_idx.off(n, y, x)] = from_f32<T>((float)besti);
            seg += logf(sum) - ztm;
            l1 += 2.f * (1.f - pt);
        }
        if (dz.ptr) {
            T* dp = (T*)dz.ptr + dz.off(n, y, x) + sub * 16;
#pragma unroll
            for (int q = 0; q < 16 / VN; ++q) {
                vec_t r;
#pragma unroll
                for (int k = 0; k < VN; ++k) {
                    const int c = sub * 16 + q * VN + k;
                    r[k] = from_f32<T>((v[q * VN + k] - (c == t ? 1.f : 0.f)) * grad_scale);
                }
                *(vec_t*)(dp + q * VN) = r;
            }
        }
        if (probs_out) {
            float* pp = probs_out + (long long)m * 256 + sub * 16;
#pragma unroll
            for (int q = 0; q < 4; ++q) *(f32x4*)(pp + 4 * q) = f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
        }
    }
    seg = block_sum(seg, red);
    l1 = block_sum(l1, red);
    if (threadIdx.x == 0) {
        loss_part[blockIdx.x] = seg * inv_count;
        loss_part[gridDim.x + blockIdx.x] = l1 * inv_count / 256.f;
    }
}

// argmax over the last dimension of given probabilities (pix2pix_model.py:286): int32, ties -> lowest index
__global__ void argmax_lastdim_kernel(const float* __restrict__ p, long long M, int C, int* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long long wave_id = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long m = wave_id; m < M; m += nwaves) {
        float best = -INFINITY;
        int besti = 0x7fffffff;
        for (int c = lane; c < C; c += 64) {
            float v = p[m * C + c];
            if (v > best) { best = v; besti = c; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            float ob = __shfl_xor(best, o, 64);
            int oi = __shfl_xor(besti, o, 64);
            if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
        }
        if (lane == 0) out[m] = besti;
    }
}

extern "C" int p2p_softmax_cce_argmax(int dtype, int N, int H, int W, int C, const p2p_tensor* z, const p2p_tensor* target,
                                      const p2p_tensor* fake_idx, float grad_scale, float inv_count, const p2p_tensor* dz,
                                      float* probs_out, float* loss_part, float* loss_out, void* stream) {
    P2P_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C <= 512, "p2p_softmax_cce_argmax: bad shape (C <= 512)");
    P2P_REQUIRE(z && z->ptr && target && target->ptr && fake_idx && fake_idx->ptr && loss_out && loss_part, "p2p_softmax_cce_argmax: null pointer");
    hipStream_t st = (hipStream_t)stream;
    TView d;
    if (dz && dz->ptr) d = make_view(dz);
    else { d.ptr = nullptr; d.img = 0; d.row = 0; d.ld = 0; }
    long long M = (long long)N * H * W;
    long long blocks = (M + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    const int vn = dtype == P2P_BF16 ? 8 : 4;
    if (C == 256 && M < (1LL << 31) && z->ld % vn == 0 && ((uintptr_t)z->ptr % 16) == 0 &&
        (!dz || !dz->ptr || (dz->ld % vn == 0 && ((uintptr_t)dz->ptr % 16) == 0)) && (!probs_out || ((uintptr_t)probs_out % 16) == 0)) {
        long long b16 = (M + 15) / 16;               // 16 pixel groups per 256-thread workgroup
        if (b16 > P2P_SOFTMAX_MAX_BLOCKS) b16 = P2P_SOFTMAX_MAX_BLOCKS;
        P2P_DISPATCH_DTYPE(dtype, (softmax256_kernel<T><<<dim3((unsigned)b16), 256, 0, st>>>(
                                      N, PixDec::make(H, W), make_view(z), make_view(target), make_view(fake_idx), grad_scale, inv_count,
                                      d, probs_out, loss_part)));
        softmax_loss_sum_kernel<<<1, 256, 0, st>>>(loss_part, (int)b16, loss_out);
        return p2p_check_launch("p2p_softmax_cce_argmax");
    }
    if (C == 256)
        P2P_REQUIRE(z->ld % 4 == 0 && ((uintptr_t)z->ptr % 16) == 0 && (!dz || !dz->ptr || (dz->ld % 4 == 0 && ((uintptr_t)dz->ptr % 16) == 0)),
                    "p2p_softmax_cce_argmax: 256-way views must be 4-channel aligned");
    P2P_DISPATCH_DTYPE(dtype, (softmax_cce_argmax_kernel<T><<<dim3((unsigned)blocks), 256, 0, st>>>(
                                  N, H, W, C, make_view(z), make_view(target), make_view(fake_idx), grad_scale, inv_count, d,
                                  probs_out, loss_part)));
    softmax_loss_sum_kernel<<<1, 256, 0, st>>>(loss_part, (int)blocks, loss_out);
    return p2p_check_launch("p2p_softmax_cce_argmax");
}

extern "C" int p2p_argmax_lastdim(const float* probs, long long M, int C, int* out, void* stream) {
    P2P_REQUIRE(probs && out && M > 0 && C > 0, "p2p_argmax_lastdim: bad args");
    long long blocks = (M + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    argmax_lastdim_kernel<<<dim3((unsigned)blocks), 256, 0, (hipStream_t)stream>>>(probs, M, C, out);
    return p2p_check_launch("p2p_argmax_lastdim");
}
