// Differentiable RGB-uv histogram + Hellinger loss, forward and backward (reference: histogram.py:4-89, used by
// Pix2PixHistogramModel.generator_loss, pix2pix_model.py:242-250).
//
// The reference materialises three (B, HW, 64) tensors per colour component and image (18.9 MB per image at
// 64x64).  Here one workgroup owns one (image, component) pair and never leaves the CU: per batch of pixels it
// evaluates the two inverse-quadratic kernel rows k(u_p - d_i), k(v_p - d_j) into LDS and contracts them over the
// pixels with the exact-f32 MFMA (v_mfma_f32_32x32x2_f32; the 1/(x+1e-6) factors of the gradient rule out bf16):
//     H_c[i][j] = sum_p Iy[p] * ku[p][i] * kv[p][j]                                   (histogram.py:29-30)
// Backward is the closed form of SURVEY.md 8a A11 (checked against autograd in tests/test_oracle.py):
//     A[i][p] = sum_j GH[i][j] kv[p][j],  Bm[j][p] = sum_i GH[i][j] ku[p][i]          (two MFMA contractions)
//     dIy[p] = sum_i A ku,  du[p] = sum_i Iy A g(u_p-d_i) ku^2,  dv likewise,  g(t) = -2 t / sigma^2
// followed by the chain rule through u, v = log-chroma and Iy.  Each component's pixel gradient goes to its own
// f32 slab [3][N*HW][4]; the consumer (p2p_tanh_l1_bwd) sums the slabs on load.
#include "p2p_common.hpp"
#include <stdlib.h>
#include <type_traits>

#define HB 64                 // histogram size (histogram.py:36)
#define HIST_EPS 1e-6f        // histogram.py:53
#define INV_SIGMA2 2500.0f    // 1 / 0.02^2  (histogram.py:36,54)

__device__ __forceinline__ float hist_center(int i) { return -3.0f + (float)i * (6.0f / 63.0f); }   // linspace(-3,3,64)
// inverse-quadratic kernel 1/(1 + t^2/sigma^2) (histogram.py:26-27).  v_rcp_f32 (1 ulp) instead of the IEEE division
// sequence: the kernel evaluations are half of the forward kernel's instruction stream.
__device__ __forceinline__ float iq_kernel(float t) { return __builtin_amdgcn_rcpf(fmaf(t * t, INV_SIGMA2, 1.0f)); }

// log-chroma coordinates of pixel p for component c: (comp, p1, p2) = (R,G,B), (G,R,B), (B,R,G)   (histogram.py:72-74)
__device__ __forceinline__ void comp_order(int c, int& a, int& p1, int& p2) {
    a = c; p1 = c == 0 ? 1 : 0; p2 = c == 2 ? 1 : 2;
}

template <typename T>
__device__ __forceinline__ void load_rgb01(const TView& img, int n, int p, int W, float* x) {
    int yy = p / W, xx = p - yy * W;
    const T* q = (const T*)img.ptr + img.off(n, yy, xx);
#pragma unroll
    for (int k = 0; k < 3; ++k) x[k] = to_f32(q[k]) * 0.5f + 0.5f;      // histogram.py:58,61
}

// ---- forward: raw (unnormalised) histogram [N][3][64][64] -----------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void rgbuv_hist_fwd_kernel(int H, int W, TView img, float* __restrict__ hist) {
    constexpr int PB = 96;           // 48 KB of operands: three workgroups per CU = all N x 3 workgroups of B=256 resident at once
    __shared__ float As[PB][HB];     // Iy * ku   [pixel][i]
    __shared__ float Bs[PB][HB];     // kv        [pixel][j]
    __shared__ float su[PB], sv[PB], siy[PB];
    const int n = blockIdx.x, c = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ti = wave >> 1, tj = wave & 1;
    const int HW = H * W;
    int ca, cp1, cp2;
    comp_order(c, ca, cp1, cp2);
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int p0 = 0; p0 < HW; p0 += PB) {
        if (tid < PB) {
            int p = p0 + tid;
            float u = 0.f, v = 0.f, iy = 0.f;
            if (p < HW) {
                float x[3];
                load_rgb01<T>(img, n, p, W, x);
                iy = sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + HIST_EPS);     // histogram.py:65-66
                float la = logf(x[ca] + HIST_EPS);
                u = la - logf(x[cp1] + HIST_EPS);                                    // histogram.py:13
                v = la - logf(x[cp2] + HIST_EPS);                                    // histogram.py:16
            }
            su[tid] = u; sv[tid] = v; siy[tid] = iy;       // iy = 0 for the tail: contributes nothing
        }
        __syncthreads();
        for (int idx = tid; idx < PB * HB; idx += 256) {
            int p = idx >> 6, i = idx & 63;
            float d = hist_center(i);
            As[p][i] = siy[p] * iq_kernel(su[p] - d);
            Bs[p][i] = iq_kernel(sv[p] - d);
        }
        __syncthreads();
#pragma unroll 8
        for (int kk = 0; kk < PB / 2; ++kk) {
            int row = 2 * kk + (lane >> 5);
            float a = As[row][ti * 32 + (lane & 31)];
            float b = Bs[row][tj * 32 + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    float* out = hist + ((long long)n * 3 + c) * HB * HB;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        int i = ti * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        int j = tj * 32 + (lane & 31);
        out[i * HB + j] = acc[e];
    }
}

// ---- colour points: the distinct colours of an image with their pixel counts ----------------------------------------------
// Iy, u and v of a pixel depend on its RGB value alone, so  H_c[i][j] = sum over DISTINCT colours of count * Iy * ku[i] * kv[j].
// A palette sprite (the REAL image of every step: 10-54 colours in the reference's dataset, hue rotation maps colour to colour,
// translation moves pixels) has two orders of magnitude fewer colours than pixels.  One workgroup per image walks the image in
// tiles of 1024 pixels; inside a tile equal colours (bitwise equal f32 RGB) meet in an LDS hash table, every pixel learns
// whether it is the FIRST pixel of its colour in the tile, and the first pixels append (r, g, b, count) to the image's list in
// pixel order.  Integer LDS atomics only: the list (content and order) does not depend on scheduling.  Tiles do not merge
// with each other (the contributions are additive).  An image whose list would exceed `cap` entries is marked dense
// (npoints = -1): the histogram kernel then reads its pixels.
#define PT_TILE 1024
#define PT_SLOTS 2048

template <typename T>
__global__ __launch_bounds__(256) void rgbuv_points_kernel(int H, int W, TView img, int cap, f32x4* __restrict__ points,
                                                          int* __restrict__ npoints) {
    __shared__ float px[PT_TILE][3];
    __shared__ int owner[PT_SLOTS];          // pixel (tile-local) that claimed the slot, -1 = empty
    __shared__ int cnt[PT_SLOTS], rep[PT_SLOTS];
    __shared__ int wsum[4];
    __shared__ int base_s, dense_s;
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int HW = H * W;
    f32x4* out = points + (long long)n * cap;
    if (tid == 0) { base_s = 0; dense_s = 0; }
    for (int t0 = 0; t0 < HW; t0 += PT_TILE) {
        for (int s = tid; s < PT_SLOTS; s += 256) { owner[s] = -1; cnt[s] = 0; rep[s] = 0x7fffffff; }
        for (int q = tid; q < PT_TILE; q += 256) {
            const int p = t0 + q;
            float x[3] = {0.f, 0.f, 0.f};
            if (p < HW) {
                const int yy = p / W, xx = p - yy * W;
                const T* g = (const T*)img.ptr + img.off(n, yy, xx);
                x[0] = to_f32(g[0]); x[1] = to_f32(g[1]); x[2] = to_f32(g[2]);
            }
            px[q][0] = x[0]; px[q][1] = x[1]; px[q][2] = x[2];
        }
        __syncthreads();
        int slot[PT_TILE / 256];
#pragma unroll
        for (int k = 0; k < PT_TILE / 256; ++k) {
            const int q = tid + 256 * k;       // pixels of one thread are 256 apart: the prefix sum below runs per k
            slot[k] = -1;
            if (t0 + q >= HW) continue;
            const unsigned b0 = __float_as_uint(px[q][0]), b1 = __float_as_uint(px[q][1]), b2 = __float_as_uint(px[q][2]);
            unsigned h = (b0 * 0x9E3779B1u) ^ (b1 * 0x85EBCA77u) ^ (b2 * 0xC2B2AE3Du);
            h ^= h >> 15;
            int sl = (int)(h & (PT_SLOTS - 1));
            for (int probe = 0; probe < PT_SLOTS; ++probe) {
                int o = atomicCAS(&owner[sl], -1, q);
                if (o == -1) o = q;
                if (__float_as_uint(px[o][0]) == b0 && __float_as_uint(px[o][1]) == b1 && __float_as_uint(px[o][2]) == b2) break;
                sl = (sl + 1) & (PT_SLOTS - 1);
            }
            slot[k] = sl;
            atomicAdd(&cnt[sl], 1);
            atomicMin(&rep[sl], q);
        }
        __syncthreads();
        // first pixels in pixel order: q = tid + 256 k, so order = (k, tid): one block-wide exclusive scan per k
        int base = base_s;
#pragma unroll
        for (int k = 0; k < PT_TILE / 256; ++k) {
            const int q = tid + 256 * k;
            const int first = (slot[k] >= 0 && rep[slot[k]] == q) ? 1 : 0;
            int incl = first;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                int v = __shfl_up(incl, o, 64);
                if (lane >= o) incl += v;
            }
            if (lane == 63) wsum[wave] = incl;
            __syncthreads();
            int before = 0, total = 0;
            for (int w = 0; w < 4; ++w) { if (w < wave) before += wsum[w]; total += wsum[w]; }
            const int pos = base + before + incl - first;
            if (first && pos < cap) {
                f32x4 e = {px[q][0], px[q][1], px[q][2], (float)cnt[slot[k]]};
                out[pos] = e;
            }
            base += total;
            __syncthreads();
        }
        if (tid == 0) { base_s = base; if (base > cap) dense_s = 1; }
        __syncthreads();
    }
    if (tid == 0) npoints[n] = dense_s ? -1 : base_s;
}

// ---- forward, all three components of an image in one workgroup ------------------------------------------------------------
// The three (comp, p1, p2) orders of histogram.py:72-74 use only three log-chroma differences, a = lR - lG, b = lR - lB,
// c = lG - lB:  (u, v) = (a, b), (-a, c), (-b, -c); and the bin centres are symmetric (linspace(-3, 3, 64): d[63-i] = -d[i]), so
// k(-a - d[i]) = k(a - d[63-i]).  Three kernel rows per pixel serve all six:
//     H_R[i][j] = sum Iy ka[i]    kb[j]        H_G[i][j] = sum Iy ka[63-i] kc[j]        H_B[i][j] = sum Iy kb[63-i] kc[63-j]
// grid = (image, pixel range): each workgroup contracts its range of pixels -- or of the image's colour points (weights =
// pixel counts) -- and writes one partial [3][64][64]; rgbuv_hist_fold_kernel adds the ranges in order.
// Round 4: the contraction runs on the bf16 matrix pipe by three-way operand splitting (see rgbuv_hist_bwd3_kernel: x = x1 + x2 + x3
// in bf16 parts, six partial products, f32 accumulate) instead of v_mfma_f32_32x32x2_f32 (f32 vector rate).  Twelve waves =
// (component, row tile, column tile), one 32x32 accumulator each.  The weight Iy enters as sqrt(Iy) on BOTH factors -- Iy ka[i] kb[j]
// = (s ka[i]) (s kb[j]), s = sqrt(Iy): three operand arrays s ka, s kb, s kc serve all three products (with Iy on one factor the
// three products need four arrays: one row kind both weighted and unweighted); each term differs from the reference's
// (Iy ka) kb by two more f32 roundings.  The arrays live in LDS as [part][row][bin][pixel] bf16 -- the contraction index (pixel)
// contiguous, 16-byte chunks XOR-swizzled by the bin -- so a mirrored bin index (63 - i) is just another row address.
#ifndef P2P_HIST_ABL
#define P2P_HIST_ABL 0        // tools/ubench: 1 no matrix products, 2 no kernel-row evaluation, 3 a quarter of the pairing sums, 4 no single-wave work
#endif
#define H3_PB 64
#define H3_PS 3
#define H3_NT 768
#define H3_ARR (HB * H3_PB * 2)               // bytes of one part of one array
#define H3_KIMG (3 * 3 * H3_ARR)              // one operand image: [3 parts][3 rows][64 bins][64 pixels] bf16
#define H3_SHM (2 * H3_KIMG + 2 * 4 * H3_PB * 4)

__device__ __forceinline__ void split3(float x, bf16_t& p1, bf16_t& p2, bf16_t& p3) {
    p1 = (bf16_t)x;
    const float r1 = x - (float)p1;
    p2 = (bf16_t)r1;
    p3 = (bf16_t)(r1 - (float)p2);
}

// The same split for two values at a time, on packed instructions (v_cvt_pk_bf16_f32, v_pk_add_f32): the kernel-row evaluation is
// vector-ALU work of the same order as the matrix work it feeds, so its instruction count is the kernels' running time.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16x2(f32x2 x) { return __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2v)); }
__device__ __forceinline__ f32x2 unpack_bf16x2(unsigned u) {
    f32x2 r = {__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)};
    return r;
}
#ifndef P2P_HIST_PK
#define P2P_HIST_PK 1         // 1: two-value arithmetic on v_pk_*_f32; 0: plain instructions, value by value.  Measured equal within 2 % (a packed
#endif                        // f32 instruction takes the issue time of two plain ones: tools/ubench/clock_probe.hip); 1 is the smaller code
#if P2P_HIST_PK
__device__ __forceinline__ f32x2 sub2(f32x2 a, f32x2 b) { return a - b; }
__device__ __forceinline__ f32x2 add2(f32x2 a, f32x2 b) { return a + b; }
__device__ __forceinline__ f32x2 mul2(f32x2 a, f32x2 b) { return a * b; }
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
#else
// value by value; the empty asm keeps the two lanes' instructions from being re-packed
__device__ __forceinline__ float keep1(float x) { asm("" : "+v"(x)); return x; }
__device__ __forceinline__ f32x2 sub2(f32x2 a, f32x2 b) { f32x2 r = {keep1(a.x - b.x), keep1(a.y - b.y)}; return r; }
__device__ __forceinline__ f32x2 add2(f32x2 a, f32x2 b) { f32x2 r = {keep1(a.x + b.x), keep1(a.y + b.y)}; return r; }
__device__ __forceinline__ f32x2 mul2(f32x2 a, f32x2 b) { f32x2 r = {keep1(a.x * b.x), keep1(a.y * b.y)}; return r; }
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { f32x2 r = {keep1(fmaf(a.x, b.x, c.x)), keep1(fmaf(a.y, b.y, c.y))}; return r; }
#endif
__device__ __forceinline__ void split3x2(f32x2 x, unsigned& p1, unsigned& p2, unsigned& p3) {
    p1 = pack_bf16x2(x);
    const f32x2 r1 = sub2(x, unpack_bf16x2(p1));
    p2 = pack_bf16x2(r1);
    p3 = pack_bf16x2(sub2(r1, unpack_bf16x2(p2)));
}
// inverse-quadratic kernel of two coordinates (iq_kernel, same operation order)
__device__ __forceinline__ f32x2 iq_kernel2(f32x2 t) {
    const f32x2 inv = {INV_SIGMA2, INV_SIGMA2}, one = {1.0f, 1.0f};
    const f32x2 q = fma2(mul2(t, t), inv, one);
    f32x2 k = {__builtin_amdgcn_rcpf(q.x), __builtin_amdgcn_rcpf(q.y)};
    return k;
}

// One barrier per batch of 64 pixels (see rgbuv_hist_bwd3_kernel for the measurement behind it): the operand image is double-
// buffered, batch b's matrix products share an instruction stream with batch b+1's kernel-row evaluation, and waves 8-10 take the
// per-pixel logs of batch b+2 (one row a, b, c each) along the way.
template <typename T>
__global__ __launch_bounds__(H3_NT) void rgbuv_hist_fwd3_kernel(int H, int W, TView img, const f32x4* __restrict__ points,
                                                               const int* __restrict__ npoints, int cap, float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) char h3_smem[];
    char* const Kimg = h3_smem;                             // 2 x [3 parts][3 rows: s ka, s kb, s kc; s = sqrt(Iy)][64 bins][64 pixels] bf16
    float* const slots = (float*)(h3_smem + 2 * H3_KIMG);   // 2 x {a, b, c, s}[64]: batch b lives in slot b & 1
    const int n = blockIdx.x, ps = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = wave >> 2, ti = (wave >> 1) & 1, tj = wave & 1;
    const int HW = H * W;
    const int np = (points && npoints) ? npoints[n] : -1;
    const bool listed = np >= 0;
    const int total = listed ? np : HW;
    int chunk = (total + H3_PS - 1) / H3_PS;
    chunk = (chunk + H3_PB - 1) / H3_PB * H3_PB;
    const int q0 = ps * chunk, q1 = min(total, q0 + chunk);
    const int nb = (q1 - q0 + H3_PB - 1) / H3_PB;            // <= 0: this range is empty, the partial is zero
    // byte offset of (array, bin, 16-byte chunk of 8 pixels) inside one part
    auto koff = [](int arr, int bin, int ch) { return arr * H3_ARR + bin * (H3_PB * 2) + ((ch ^ ((bin >> 1) & 7)) << 4); };
    // component c: H_R = (s ka)[i] (s kb)[j],  H_G = (s ka)[63-i] (s kc)[j],  H_B = (s kb)[63-i] (s kc)[63-j]
    const int li = lane & 31, hk = lane >> 5;
    const int ci = ti * 32 + li, cj = tj * 32 + li;
    const int arow = c == 2 ? 1 : 0, abin = c == 0 ? ci : 63 - ci;
    const int brow = c == 0 ? 1 : 2, bbin = c == 2 ? 63 - cj : cj;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;

    // per-pixel values of batch b: wave 8 -> a = lR - lG, wave 9 -> b = lR - lB, wave 10 -> c = lG - lB and the weight's square root
    auto prep = [&](int b) {
        const int k = wave - 8, p = q0 + b * H3_PB + lane;
        float co = 0.f, s = 0.f;                             // s = 0 beyond the range: contributes nothing
        if (p < q1) {
            float x[3], wgt = 1.f;
            if (listed) {
                const f32x4 e = points[(long long)n * cap + p];
                x[0] = e[0] * 0.5f + 0.5f; x[1] = e[1] * 0.5f + 0.5f; x[2] = e[2] * 0.5f + 0.5f;
                wgt = e[3];
            } else {
                load_rgb01<T>(img, n, p, W, x);
            }
            const float xi = k == 2 ? x[1] : x[0], xj = k == 0 ? x[1] : x[2];
            co = logf(xi + HIST_EPS) - logf(xj + HIST_EPS);
            s = sqrtf(wgt * sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + HIST_EPS));
        }
        float* sl = slots + (b & 1) * (4 * H3_PB);
        sl[k * H3_PB + lane] = co;
        if (k == 2) sl[3 * H3_PB + lane] = s;
    };
    // kernel-row evaluation: item = (row a / b / c, bin, chunk of 8 pixels) = tid + 768 it, it = 0, 1; the chunk is the thread's own
    const int ebin = (tid >> 3) & 63;
    auto eval_load = [&](const float* sl, int it, f32x2 (&cs)[4], f32x2 (&ws)[4], float& d) {
        const int rty = (tid + H3_NT * it) >> 9, bin = (ebin + 32 * it) & 63;
        d = hist_center(bin);
        const f32x4* co = (const f32x4*)(sl + rty * H3_PB + (tid & 7) * 8);
        const f32x4* sw = (const f32x4*)(sl + 3 * H3_PB + (tid & 7) * 8);
        const f32x4 c0 = co[0], c1 = co[1], w0 = sw[0], w1 = sw[1];
        cs[0].x = c0[0]; cs[0].y = c0[1]; cs[1].x = c0[2]; cs[1].y = c0[3]; cs[2].x = c1[0]; cs[2].y = c1[1]; cs[3].x = c1[2]; cs[3].y = c1[3];
        ws[0].x = w0[0]; ws[0].y = w0[1]; ws[1].x = w0[2]; ws[1].y = w0[3]; ws[2].x = w1[0]; ws[2].y = w1[1]; ws[3].x = w1[2]; ws[3].y = w1[3];
    };
    auto eval_pairs = [&](const f32x2 (&cs)[4], const f32x2 (&ws)[4], float d, int half, u32x4& k1, u32x4& k2, u32x4& k3) {
        const f32x2 d2 = {d, d};
#pragma unroll
        for (int e2 = 2 * half; e2 < 2 * half + 2; ++e2) {
            unsigned q1b, q2b, q3b;
            split3x2(mul2(ws[e2], iq_kernel2(sub2(cs[e2], d2))), q1b, q2b, q3b);
            k1[e2] = q1b; k2[e2] = q2b; k3[e2] = q3b;
        }
    };
    auto eval_store = [&](char* Kd, int it, const u32x4& k1, const u32x4& k2, const u32x4& k3) {
        const int rty = (tid + H3_NT * it) >> 9, bin = (ebin + 32 * it) & 63;
        const int o = koff(rty, bin, tid & 7);
        *(u32x4*)(Kd + o) = k1; *(u32x4*)(Kd + 3 * H3_ARR + o) = k2; *(u32x4*)(Kd + 6 * H3_ARR + o) = k3;
    };
    auto batch = [&](int b, auto more_tag) {
        constexpr bool MORE = decltype(more_tag)::value;
        const char* const Kc = Kimg + (b & 1) * H3_KIMG;
        char* const Kn = Kimg + ((b + 1) & 1) * H3_KIMG;
        const float* const sln = slots + ((b + 1) & 1) * (4 * H3_PB);
        if (wave >= 8 && wave < 11 && b + 2 < nb) prep(b + 2);
        f32x2 cs[4], ws[4];
        float d = 0.f;
        u32x4 k1, k2, k3;
#pragma unroll
        for (int s4 = 0; s4 < H3_PB / 16; ++s4) {
            bf16x8 fa[3], fb[3];
#pragma unroll
            for (int part = 0; part < 3; ++part) {
                fa[part] = *(const bf16x8*)(Kc + part * 3 * H3_ARR + koff(arow, abin, 2 * s4 + hk));
                fb[part] = *(const bf16x8*)(Kc + part * 3 * H3_ARR + koff(brow, bbin, 2 * s4 + hk));
            }
            if (MORE && P2P_HIST_ABL != 2) {
                if ((s4 & 1) == 0) eval_load(sln, s4 >> 1, cs, ws, d);
                eval_pairs(cs, ws, d, s4 & 1, k1, k2, k3);
            }
#if P2P_HIST_ABL == 1
#pragma unroll
            for (int part = 0; part < 3; ++part)
                acc[0] += __uint_as_float((__builtin_bit_cast(u32x4, fa[part])[0] ^ __builtin_bit_cast(u32x4, fb[part])[3]) & 0x7fffffu);
            if (MORE && (s4 & 1) == 1) eval_store(Kn, s4 >> 1, k1, k2, k3);
            continue;
#endif
            // smallest partial products first
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2], fb[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[0], acc, 0, 0, 0);
            if (MORE && P2P_HIST_ABL != 2 && (s4 & 1) == 1) eval_store(Kn, s4 >> 1, k1, k2, k3);
        }
        __syncthreads();
    };
    if (nb > 0) {
        // prologue: per-pixel values of batches 0 and 1, operand image of batch 0
        if (wave >= 8 && wave < 11) { prep(0); if (nb > 1) prep(1); }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            f32x2 cs[4], ws[4];
            float d;
            u32x4 k1, k2, k3;
            eval_load(slots, it, cs, ws, d);
            eval_pairs(cs, ws, d, 0, k1, k2, k3);
            eval_pairs(cs, ws, d, 1, k1, k2, k3);
            eval_store(Kimg, it, k1, k2, k3);
        }
        __syncthreads();
        for (int b = 0; b < nb - 1; ++b) batch(b, std::true_type{});
        batch(nb - 1, std::false_type{});
    }
    float* out = part + ((long long)n * H3_PS + ps) * 3 * HB * HB;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int i = ti * 32 + (e & 3) + 8 * (e >> 2) + 4 * hk;
        const int j = tj * 32 + li;
        out[(c * HB + i) * HB + j] = acc[e];
    }
}

__global__ __launch_bounds__(256) void rgbuv_hist_fold_kernel(const float* __restrict__ part, float* __restrict__ hist) {
    const int n = blockIdx.x;
    const int E = 3 * HB * HB;
    const f32x4* src = (const f32x4*)(part + (long long)n * H3_PS * E);
    f32x4* dst = (f32x4*)(hist + (long long)n * E);
    for (int i = threadIdx.x; i < E / 4; i += 256) {
        f32x4 s = src[i];
#pragma unroll
        for (int k = 1; k < H3_PS; ++k) s += src[k * (E / 4) + i];
        dst[i] = s;
    }
}

// ---- per-image totals, Hellinger partial sum and dL/d(raw histogram) ---------------------------------------
// totals[n] = sum_{c,i,j} raw[n]  (histogram.py:78);  sq_part[n] = sum (sqrt(p/Tp) - sqrt(q/Tq))^2 of image n (histogram.py:88-89);
// hellinger_sq_sum_kernel adds the images in index order (deterministic: no float atomics)
__global__ __launch_bounds__(256) void hellinger_fwd_kernel(const float* __restrict__ h_true, const float* __restrict__ h_pred,
                                                           float* __restrict__ tot_true, float* __restrict__ tot_pred,
                                                           float* __restrict__ sq_part) {
    __shared__ float red[16];
    const int n = blockIdx.x;
    const int E = 3 * HB * HB;
    const float* a = h_true + (long long)n * E;
    const float* b = h_pred + (long long)n * E;
    float sa = 0.f, sb = 0.f;
    for (int i = threadIdx.x; i < E; i += 256) { sa += a[i]; sb += b[i]; }
    sa = block_sum(sa, red);
    sb = block_sum(sb, red);
    float s = 0.f;
    for (int i = threadIdx.x; i < E; i += 256) {
        float d = sqrtf(b[i] / sb) - sqrtf(a[i] / sa);
        s += d * d;
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) { tot_true[n] = sa; tot_pred[n] = sb; sq_part[n] = s; }
}

__global__ __launch_bounds__(256) void hellinger_sq_sum_kernel(const float* __restrict__ sq_part, int n, float* __restrict__ sq) {
    __shared__ float red[16];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += sq_part[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) sq[0] = s;
}

// loss_out[0] = sqrt(sq_global) / (sqrt(2) * B_global)    (histogram.py:88-89)
__global__ void hellinger_finish_kernel(const float* __restrict__ sq, float inv_b, float* __restrict__ loss_out) {
    loss_out[0] = sqrtf(sq[0]) * 0.70710678118654752f * inv_b;
}

// GH = dL/d(raw pred histogram): G = D / (2 sqrt2 Bg sqrt(Sigma) sqrt(Hn)), GH = (G - sum(G Hn)) / T
__global__ __launch_bounds__(256) void hist_grad_prep_kernel(const float* __restrict__ h_true, const float* __restrict__ h_pred,
                                                            const float* __restrict__ tot_true, const float* __restrict__ tot_pred,
                                                            const float* __restrict__ sq, float coef, float* __restrict__ gh) {
    __shared__ float red[16];
    const int n = blockIdx.x;
    const int E = 3 * HB * HB;
    const float* a = h_true + (long long)n * E;
    const float* b = h_pred + (long long)n * E;
    const float ta = tot_true[n], tb = tot_pred[n];
    const float k = coef / sqrtf(sq[0]);          // coef = lambda / (2 sqrt2 B_global)
    float s = 0.f;
    for (int i = threadIdx.x; i < E; i += 256) {
        float hn = b[i] / tb;
        float shn = sqrtf(hn);
        float g = (shn - sqrtf(a[i] / ta)) * k / shn;
        s += g * hn;
    }
    s = block_sum(s, red);
    for (int i = threadIdx.x; i < E; i += 256) {
        float hn = b[i] / tb;
        float shn = sqrtf(hn);
        float g = (shn - sqrtf(a[i] / ta)) * k / shn;
        gh[(long long)n * E + i] = (g - s) / tb;
    }
}

// ---- backward: d loss / d fake image, one f32 slab per component ------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void rgbuv_hist_bwd_kernel(int H, int W, TView img, const float* __restrict__ gh,
                                                            float* __restrict__ dimg, long long slab) {
    constexpr int PB = 64;
    __shared__ float G[HB][HB + 1];    // GH[i][j]; the odd row stride makes the transposed operand read (product 0) conflict-free
    __shared__ float KuT[HB][PB];      // ku[i][pixel]
    __shared__ float KvT[HB][PB];      // kv[j][pixel]
    __shared__ float su[PB], sv[PB], siy[PB], sx[PB][3];
    __shared__ float r_diy[PB], r_du[PB], r_dv[PB];
    const int n = blockIdx.x, c = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int HW = H * W;
    int ca, cp1, cp2;
    comp_order(c, ca, cp1, cp2);
    const float* g = gh + ((long long)n * 3 + c) * HB * HB;
    for (int idx = tid; idx < HB * HB; idx += 256) {
        int i = idx >> 6, j = idx & 63;
        G[i][j] = g[idx];
    }
    const int prod = wave >> 1;        // 0: A = GH kv (rows i), 1: Bm = GH^T ku (rows j)
    const int pt = wave & 1;           // pixel tile of 32
    float* out = dimg + (long long)c * slab;
    for (int p0 = 0; p0 < HW; p0 += PB) {
        __syncthreads();
        if (tid < PB) {
            int p = p0 + tid;
            float u = 0.f, v = 0.f, iy = 1.f, x[3] = {1.f, 1.f, 1.f};
            if (p < HW) {
                load_rgb01<T>(img, n, p, W, x);
                iy = sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + HIST_EPS);
                float la = logf(x[ca] + HIST_EPS);
                u = la - logf(x[cp1] + HIST_EPS);
                v = la - logf(x[cp2] + HIST_EPS);
            }
            su[tid] = u; sv[tid] = v; siy[tid] = iy;
            sx[tid][0] = x[0]; sx[tid][1] = x[1]; sx[tid][2] = x[2];
        }
        __syncthreads();
        for (int idx = tid; idx < PB * HB; idx += 256) {
            int i = idx / PB, p = idx % PB;
            float d = hist_center(i);
            KuT[i][p] = iq_kernel(su[p] - d);
            KvT[i][p] = iq_kernel(sv[p] - d);
        }
        __syncthreads();
        // D[row][col = pixel] = sum_k Lhs[row][k] * Rhs[k][pixel];  lane: row/col = lane&31, k = lane>>5
        // product 0 wants Lhs[k = j][row = i] = GH[i][j]; product 1 wants Lhs[k = i][row = j] = GH[i][j]
        const float (*Rhs)[PB] = prod == 0 ? KvT : KuT;
        const float (*Kown)[PB] = prod == 0 ? KuT : KvT;   // the kernel row the result is paired with in the epilogue
        const float* coord = prod == 0 ? su : sv;
        float part0 = 0.f, part1 = 0.f;                    // sum_rows D*k  and  sum_rows D*g*k^2
        const int pcol = pt * 32 + (lane & 31);
        const float cval = coord[pcol];
        // both 32-row tiles at once: two independent accumulation chains per wave (one chain of v_mfma_f32_32x32x2_f32 leaves
        // the matrix pipe idle between dependent issues: r03 PMC 48 % busy, 50 % of the wave cycles issue-stalled), one read of
        // the right-hand operand for both
        f32x16 acc[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[rt][e] = 0.f;
#pragma unroll 8
        for (int kk = 0; kk < HB / 2; ++kk) {
            const int k = 2 * kk + (lane >> 5);
            const int r0 = lane & 31, r1 = 32 + (lane & 31);
            const float a0 = prod == 0 ? G[r0][k] : G[k][r0];
            const float a1 = prod == 0 ? G[r1][k] : G[k][r1];
            const float b = Rhs[k][pcol];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc[1], 0, 0, 0);
        }
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                int row = rt * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                float kq = Kown[row][pcol];
                float t = cval - hist_center(row);
                part0 += acc[rt][e] * kq;
                part1 += acc[rt][e] * (-2.0f * t * INV_SIGMA2) * kq * kq;
            }
        part0 += __shfl_xor(part0, 32, 64);
        part1 += __shfl_xor(part1, 32, 64);
        if ((lane >> 5) == 0) {
            if (prod == 0) { r_diy[pcol] = part0; r_du[pcol] = siy[pcol] * part1; }
            else { r_dv[pcol] = siy[pcol] * part1; }
        }
        __syncthreads();
        if (tid < PB && p0 + tid < HW) {
            // dIy[p] = sum_i A[p,i] ku[p,i]  (= sum_j Bm[p,j] kv[p,j]; taken once, from product 0)
            float dx[3] = {0.f, 0.f, 0.f};
            float du = r_du[tid], dv = r_dv[tid], diy = r_diy[tid];
            float x0 = sx[tid][0], x1 = sx[tid][1], x2 = sx[tid][2];
            float xs[3] = {x0, x1, x2};
            dx[ca] += (du + dv) / (xs[ca] + HIST_EPS);
            dx[cp1] -= du / (xs[cp1] + HIST_EPS);
            dx[cp2] -= dv / (xs[cp2] + HIST_EPS);
            float iy = siy[tid];
#pragma unroll
            for (int k = 0; k < 3; ++k) dx[k] += diy * xs[k] / iy;
            float* o = out + ((long long)n * HW + p0 + tid) * 4;
            o[0] = 0.5f * dx[0]; o[1] = 0.5f * dx[1]; o[2] = 0.5f * dx[2]; o[3] = 0.f;     // x = img*0.5+0.5; alpha has no gradient
        }
    }
}

// ---- backward, all three components of an image in one workgroup ---------------------------------------------------------
// Round 3 ran this contraction on v_mfma_f32_32x32x2_f32: that instruction runs at the f32 VECTOR rate (1/16 of the bf16 matrix
// rate) and, measured, never overlaps with the other waves' vector instructions on its SIMD -- kernel time = matrix cycles +
// vector cycles, two thirds of it matrix cycles (r03 PMC: MFMA-busy 48 %, VALU 45 %).  Round 4: the SAME f32 products on the bf16
// matrix pipe by three-way splitting.  Every f32 operand x is written as x1 + x2 + x3, each part the bf16 rounding of what the
// previous parts left (3 x 8 = 24 significant bits: the sum reproduces x to <= 2^-24 |x|); a product a b is taken as the six
// partial products a1 b1, a1 b2, a2 b1, a1 b3, a3 b1, a2 b2 (everything above 2^-24 of |a b|; each bf16 x bf16 product is exact in
// f32) accumulated in f32 by v_mfma_f32_32x32x16_bf16.  Six instructions at 16x the rate: 2.7x fewer matrix cycles for a result
// that differs from the f32 fma chain by rounding-level terms (tests: same bounds as before against the float64 oracle, and 1e-4
// against the per-component f32-MFMA kernel kept as the cross-check).
// Structure as in round 3: the three kernel rows ka, kb, kc of the log-chroma differences a = lR - lG, b = lR - lB, c = lG - lB
// serve every product of the closed form (component R has (u, v) = (a, b), G has (-a, c), B has (-b, -c); the bin grid is symmetric,
// so a mirrored coordinate is a flip of the GRADIENT matrix's index, applied once when it is loaded); twelve waves = (component,
// product, row tile): product 0 = GH kv (paired with ku: dIy and du), product 1 = GH^T ku (paired with kv: dv).  New: the wave's
// 32 x 64 slice of GH (constant over the image) lives in REGISTERS as MFMA A fragments (3 parts x 4 k-steps), so the loop reads only
// the kernel rows from LDS: [part][row][pixel][bin] bf16, 16-byte chunks XOR-swizzled by the pixel (conflict-free ds_read_b128).
#define B3_PB 64
#define B3_NT 768
#define B3_KBUF (3 * 3 * B3_PB * HB * 2)      // bytes of one split kernel-row image: [3 parts][3 rows][64 pixels][64 bins] bf16
#define B3_SLOT (6 * B3_PB)                   // floats of one batch's per-pixel values: log x [3][64], x [3][64]
#define B3_RACC (18 * B3_PB)                  // floats of one batch's partial sums: [3 comps][2 row tiles][diy | du | dv][pixel]
#define B3_SHM (2 * B3_KBUF + (4 * B3_SLOT + 2 * B3_RACC) * 4)

// Round 4, second half: ONE barrier per batch of 64 pixels.  Measured on the five-phase form (per-pixel logs by one wave | kernel rows
// | matrix products | pairing sums | per-pixel result by one wave, a barrier between each): the matrix pipe holds a third of the
// batch time, vector work a third, and the two single-wave phases with their barriers the rest.  Now the kernel-row image is double-
// buffered and batch b's matrix products run in the same instruction stream as batch b+1's kernel-row evaluation (no dependence:
// the vector instructions issue under the 8-pass MFMAs), the per-pixel logs of batch b+2 (waves 8-10, one colour channel each) and
// the per-pixel result of batch b-1 (wave 11) ride along on single waves, and the only barrier closes the batch.
template <typename T>
__global__ __launch_bounds__(B3_NT) void rgbuv_hist_bwd3_kernel(int H, int W, TView img, const float* __restrict__ gh,
                                                               float* __restrict__ dimg, int nsplit) {
    extern __shared__ __attribute__((aligned(16))) char b3_smem[];
    char* const Kimg = b3_smem;                                                      // 2 x [3 parts][3 rows a, b, c][64 pixels][64 bins] bf16
    float* const slots = (float*)(b3_smem + 2 * B3_KBUF);                             // 4 x {log x [3][64], x [3][64]}: batch b lives in slot b & 3
    float* const raccs = slots + 4 * B3_SLOT;                                         // 2 x [3 comps][2 row tiles][3][64]: batch b in b & 1
    const int n = blockIdx.x, ps = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int HW = H * W;
    int chunk = (HW + nsplit - 1) / nsplit;
    chunk = (chunk + B3_PB - 1) / B3_PB * B3_PB;
    const int q0 = ps * chunk, q1 = min(HW, q0 + chunk);
    const int nb = (q1 - q0 + B3_PB - 1) / B3_PB;
    if (nb <= 0) return;
    const int c = wave >> 2, prod = (wave >> 1) & 1, rt = wave & 1;
    // (u, v) of component c in terms of the shared rows: row index, mirrored?
    const int ua = c == 2 ? 1 : 0, va = c == 0 ? 1 : 2;
    const bool um = c != 0, vm = c == 2;
    const int ra = prod == 0 ? va : ua, oa = prod == 0 ? ua : va;          // contracted rows / the rows the result is paired with
    const bool om = prod == 0 ? um : vm;
    const float osign = __int_as_float(__builtin_amdgcn_readfirstlane(om ? 0xbf800000 : 0x3f800000));      // wave-uniform: a scalar register
    const int li = lane & 31, hk = lane >> 5;
    // row r = log x[ri] - log x[rj]:  a = lR - lG, b = lR - lB, c = lG - lB
    const int oi = oa == 2 ? 1 : 0, oj = oa == 0 ? 1 : 2;

    // ---- this wave's slice of the gradient matrix as A fragments: GH'[i'][j'] = GH_c[um ? 63 - i' : i'][vm ? 63 - j' : j'];
    // product 0: A[row][k] = GH'[rt*32 + row][k],  product 1: A[row][k] = GH'[k][rt*32 + row]   (lane: row = li, k = 16 s + 8 hk + e)
    bf16x8 ga[3][4];
    {
        const float* g = gh + ((long long)n * 3 + c) * HB * HB;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int kk = 16 * s4 + 8 * hk + e, rr = rt * 32 + li;
                const int ip = prod == 0 ? rr : kk, jp = prod == 0 ? kk : rr;
                const float v = g[(um ? 63 - ip : ip) * HB + (vm ? 63 - jp : jp)];
                bf16_t p1, p2, p3;
                split3(v, p1, p2, p3);
                ga[0][s4][e] = p1; ga[1][s4][e] = p2; ga[2][s4][e] = p3;
            }
    }
    const float cen0 = hist_center(rt * 32 + 4 * hk);       // centre of this lane's first result row; row e lies (e&3) + 8 (e>>2) bins on
    // byte offset of (row, pixel, 16-byte chunk) inside one part of the kernel-row image
    auto koff = [](int row, int p, int ch) { return (row * B3_PB + p) * (HB * 2) + ((ch ^ ((p >> 1) & 7)) << 4); };
    constexpr int KPART = 3 * B3_PB * HB * 2;
    // kernel-row evaluation: item = (row, pixel, chunk of 8 bins) = tid + 768 it, it = 0, 1; the chunk is the thread's own (768 = 0
    // mod 8): its eight bin centres stay in registers
    f32x2 cen2[4];
#pragma unroll
    for (int e2 = 0; e2 < 4; ++e2) { cen2[e2].x = hist_center((tid & 7) * 8 + 2 * e2); cen2[e2].y = hist_center((tid & 7) * 8 + 2 * e2 + 1); }
    const int ep = (tid >> 3) & 63;                                         // the items' pixel (768 / 8 = 96 = 32 mod 64: + 32 for it = 1)
    auto eval_coord = [&](const float* sl, int it) {
        const int r = (tid + B3_NT * it) >> 9, p = (ep + 32 * it) & 63;     // r: 0 0 0 0 0 0 0 0 1 1 1 1 | 1 1 1 1 2 2 2 2 2 2 2 2 by wave
        const int ri = r == 2 ? 1 : 0, rj = r == 0 ? 1 : 2;
        return sl[ri * B3_PB + p] - sl[rj * B3_PB + p];
    };
    auto eval_pairs = [&](float co, int half, u32x4& k1, u32x4& k2, u32x4& k3) {
        const f32x2 co2 = {co, co};
#pragma unroll
        for (int e2 = 2 * half; e2 < 2 * half + 2; ++e2) {
            unsigned q1b, q2b, q3b;
            split3x2(iq_kernel2(sub2(co2, cen2[e2])), q1b, q2b, q3b);
            k1[e2] = q1b; k2[e2] = q2b; k3[e2] = q3b;
        }
    };
    auto eval_store = [&](char* Kd, int it, const u32x4& k1, const u32x4& k2, const u32x4& k3) {
        const int r = (tid + B3_NT * it) >> 9, p = (ep + 32 * it) & 63;
        const int o = koff(r, p, tid & 7);
        *(u32x4*)(Kd + o) = k1;
        *(u32x4*)(Kd + KPART + o) = k2;
        *(u32x4*)(Kd + 2 * KPART + o) = k3;
    };
    // per-pixel logs of batch b: waves 8, 9, 10 take one colour channel each
    auto prep = [&](int b) {
        const int k = wave - 8, p = q0 + b * B3_PB + lane;
        float x[3] = {1.f, 1.f, 1.f};
        if (p < q1) load_rgb01<T>(img, n, p, W, x);
        const float xk = k == 0 ? x[0] : (k == 1 ? x[1] : x[2]);
        float* sl = slots + (b & 3) * B3_SLOT;
        sl[k * B3_PB + lane] = logf(xk + HIST_EPS);
        sl[(3 + k) * B3_PB + lane] = xk;
    };
    // per-pixel result of batch b (wave 11): the twelve waves' partial sums -> d img
    auto finish = [&](int b) {
        const int p = q0 + b * B3_PB + lane;
        if (p >= q1) return;
        const float* sl = slots + (b & 3) * B3_SLOT;
        const float (*racc)[2][3][B3_PB] = (const float (*)[2][3][B3_PB])(raccs + (b & 1) * B3_RACC);
        float diy = 0.f, du[3], dv[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            diy += racc[k][0][0][lane] + racc[k][1][0][lane];
            du[k] = racc[k][0][1][lane] + racc[k][1][1][lane];
            dv[k] = racc[k][0][2][lane] + racc[k][1][2][lane];
        }
        // u0 = lR - lG, v0 = lR - lB;  u1 = lG - lR, v1 = lG - lB;  u2 = lB - lR, v2 = lB - lG    (histogram.py:72-74)
        const float dl[3] = {(du[0] + dv[0]) - du[1] - du[2], (du[1] + dv[1]) - du[0] - dv[2], (du[2] + dv[2]) - dv[0] - dv[1]};
        const float x0 = sl[3 * B3_PB + lane], x1 = sl[4 * B3_PB + lane], x2 = sl[5 * B3_PB + lane];
        const float xs[3] = {x0, x1, x2};
        const float iy = sqrtf(x0 * x0 + x1 * x1 + x2 * x2 + HIST_EPS);
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 3; ++k) o[k] = 0.5f * (iy * dl[k] / (xs[k] + HIST_EPS) + diy * xs[k] / iy);      // x = img * 0.5 + 0.5; du, dv carry Iy
        o[3] = 0.f;                                                                                          // alpha has no gradient
        *(f32x4*)(dimg + ((long long)n * HW + p) * 4) = o;
    };

    // ---- one batch: [result of b-1 | logs of b+2] on single waves; products of b with the kernel rows of b+1 in their shadow;
    // pairing sums of b; barrier
    auto batch = [&](int b, auto more_tag) {
        constexpr bool MORE = decltype(more_tag)::value;
        char* const Kc = Kimg + (b & 1) * B3_KBUF;
        char* const Kn = Kimg + ((b + 1) & 1) * B3_KBUF;
        const float* const slc = slots + (b & 3) * B3_SLOT;
        const float* const sln = slots + ((b + 1) & 3) * B3_SLOT;
#if P2P_HIST_ABL != 4
        if (wave >= 8 && wave < 11 && b + 2 < nb) prep(b + 2);
#endif
        f32x16 acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        u32x4 k1, k2, k3;
        float co = 0.f;
        int kb_keep = 0; (void)kb_keep;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            bf16x8 kb[3][2];                                  // B fragments: [part][pixel tile]: pixel = t*32 + li, bins 16 s + 8 hk ..
#pragma unroll
            for (int part = 0; part < 3; ++part)
#pragma unroll
                for (int t = 0; t < 2; ++t) kb[part][t] = *(const bf16x8*)(Kc + part * KPART + koff(ra, t * 32 + li, 2 * s4 + hk));
            if (MORE && P2P_HIST_ABL != 2) {
                if ((s4 & 1) == 0) co = eval_coord(sln, s4 >> 1);
                eval_pairs(co, s4 & 1, k1, k2, k3);
            }
#if P2P_HIST_ABL == 1
#pragma unroll
            for (int part = 0; part < 3; ++part)
#pragma unroll
                for (int t = 0; t < 2; ++t) kb_keep ^= __builtin_bit_cast(u32x4, kb[part][t])[0] ^ __builtin_bit_cast(u32x4, kb[part][t])[3];
#endif
#pragma unroll
            for (int t = 0; t < 2 && P2P_HIST_ABL != 1; ++t) {
                // smallest partial products first
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[1][s4], kb[1][t], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[2][s4], kb[0][t], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[0][s4], kb[2][t], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[1][s4], kb[0][t], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[0][s4], kb[1][t], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[0][s4], kb[0][t], acc[t], 0, 0, 0);
            }
            if (MORE && P2P_HIST_ABL != 2 && (s4 & 1) == 1) eval_store(Kn, s4 >> 1, k1, k2, k3);
        }
#if P2P_HIST_ABL == 1
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t][0] = __int_as_float(kb_keep);
#endif
        // The result of batch b-1 goes out AFTER this batch's last MFMA and is waited for before the barrier: an MFMA's write-back is
        // not interlocked against a pending global store's read of its data registers (measured, tools/exp/hist_repro.py: with the
        // store ahead of the batch's first MFMA, whose result registers happened to be the store's data registers, one workgroup in
        // a thousand stored a wrong first dword for its last 16 lanes; vector-instruction and LDS-return writes ARE interlocked).
#if P2P_HIST_ABL != 4
        if (wave == 11 && b > 0) finish(b - 1);
#endif
        float (*racc)[2][3][B3_PB] = (float (*)[2][3][B3_PB])(raccs + (b & 1) * B3_RACC);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int pcol = j * 32 + li;
            // a, b or c at the lane's pixel minus the first row's centre (the mirror is the sign `osign`)
            const float cval = (slc[oi * B3_PB + pcol] - slc[oj * B3_PB + pcol]) - cen0;
            f32x2 t0 = {0.f, 0.f}, t1 = {0.f, 0.f};      // sum D k,  sum D k^2 (coord - d), two rows at a time (packed f32 instructions)
            const f32x2 cv2 = {cval, cval};
#pragma unroll
            for (int g4 = 0; g4 < (P2P_HIST_ABL == 3 ? 1 : 4); ++g4) {
                // rows rt*32 + 8 g4 + 4 hk + 0..3 = bins of chunk rt*4 + g4, second half for hk = 1
                const int o = koff(oa, pcol, rt * 4 + g4) + 8 * hk;
                const u32x2 v1 = *(const u32x2*)(Kc + o), v2 = *(const u32x2*)(Kc + KPART + o), v3 = *(const u32x2*)(Kc + 2 * KPART + o);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int e = 4 * g4 + 2 * h;
                    const f32x2 kq = add2(add2(unpack_bf16x2(v1[h]), unpack_bf16x2(v2[h])), unpack_bf16x2(v3[h]));
                    const f32x2 d2 = {acc[j][e], acc[j][e + 1]};
                    const f32x2 w = mul2(d2, kq);
                    const f32x2 off = {(float)((e & 3) + 8 * (e >> 2)) * (6.0f / 63.0f), (float)(((e + 1) & 3) + 8 * ((e + 1) >> 2)) * (6.0f / 63.0f)};
                    t0 = add2(t0, w);
                    t1 = fma2(mul2(w, kq), sub2(cv2, off), t1);
                }
            }
            float s0 = t0.x + t0.y, s1 = t1.x + t1.y;
            s0 += __shfl_xor(s0, 32, 64);
            s1 += __shfl_xor(s1, 32, 64);
            if (hk == 0) {
                // sum_rows D g(t) k^2 with g(t) = -2 (coord - d) / sigma^2, coord - d = osign (shared coordinate - mirrored centre);
                // the factor Iy of du, dv is applied once, to their sum, by finish()
                const float part1 = (-2.0f * INV_SIGMA2) * osign * s1;
                if (prod == 0) { racc[c][rt][0][pcol] = s0; racc[c][rt][1][pcol] = part1; }
                else { racc[c][rt][2][pcol] = part1; }
            }
        }
        if (wave == 11) __builtin_amdgcn_s_waitcnt(0x0f70);      // vmcnt(0): finish()'s store has left the registers the next batch's MFMAs write
        __syncthreads();
    };

    // ---- prologue: logs of batches 0 and 1, kernel rows of batch 0
    if (wave >= 8 && wave < 11) { prep(0); if (nb > 1) prep(1); }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        u32x4 k1, k2, k3;
        const float co = eval_coord(slots, it);
        eval_pairs(co, 0, k1, k2, k3);
        eval_pairs(co, 1, k1, k2, k3);
        eval_store(Kimg, it, k1, k2, k3);
    }
    __syncthreads();
    for (int b = 0; b < nb - 1; ++b) batch(b, std::true_type{});
    batch(nb - 1, std::false_type{});
    if (wave == 11) finish(nb - 1);
}

// hist_out[n][i][j][c] = raw[n][c][i][j] / sum(raw[n])  -- the reference's normalised (B,64,64,3) tensor (histogram.py:75-79)
__global__ __launch_bounds__(256) void hist_normalize_kernel(const float* __restrict__ raw, float* __restrict__ out) {
    __shared__ float red[16];
    const int n = blockIdx.x;
    const int E = 3 * HB * HB;
    const float* a = raw + (long long)n * E;
    float s = 0.f;
    for (int i = threadIdx.x; i < E; i += 256) s += a[i];
    s = block_sum(s, red);
    for (int i = threadIdx.x; i < E; i += 256) {
        int c = i / (HB * HB), ij = i % (HB * HB);
        out[(long long)n * E + ij * 3 + c] = a[i] / s;
    }
}

extern "C" int p2p_hist_normalize(const float* raw, int N, float* out, void* stream) {
    P2P_REQUIRE(raw && out && N > 0, "p2p_hist_normalize: bad args");
    hist_normalize_kernel<<<dim3(N), 256, 0, (hipStream_t)stream>>>(raw, out);
    return p2p_check_launch("p2p_hist_normalize");
}

// ---- forward with the reference's other arguments (histogram.py:36: size, method, sigma) --------------------------------------------
// No call site of the reference uses anything but (64, "inverse-quadratic", 0.02) -- the specialised kernels above and below serve
// that -- but the function's signature takes them, so the standalone call does too: one workgroup per (image, component), kernel
// rows of a batch of pixels in LDS, every thread owns size^2 / 256 bins and adds the pixels in pixel order (deterministic).  f32
// throughout, no matrix pipe: a few milliseconds per batch, for evaluation code.
// method: 0 = inverse-quadratic 1 / (1 + t^2 / sigma^2), 1 = RBF exp(-t^2 / sigma^2), 2 = anything else: the reference applies NO
// kernel function then and multiplies the scaled squared distances themselves (histogram.py:20-27 has no third branch).
template <typename T>
__global__ __launch_bounds__(256) void rgbuv_hist_general_kernel(int H, int W, TView img, int S, int method, float inv_sigma2,
                                                                 float* __restrict__ hist) {
    constexpr int PB = 32, MAXB = 64;               // pixels per batch; bins per thread (S <= 128)
    extern __shared__ float gsm[];                  // ku[PB][S] | kv[PB][S] | iy[PB]
    float* const ku = gsm;
    float* const kv = gsm + PB * S;
    float* const iyL = gsm + 2 * PB * S;
    const int n = blockIdx.x, c = blockIdx.y, tid = threadIdx.x;
    const int HW = H * W, nb = (S * S + 255) / 256;
    int ca, cp1, cp2;
    comp_order(c, ca, cp1, cp2);
    float acc[MAXB];
    int ij[MAXB];                                    // (i << 8) | j of the thread's k-th bin: registers (the loops below are unrolled)
#pragma unroll
    for (int k = 0; k < MAXB; ++k) {
        acc[k] = 0.f;
        const int bin = min(tid + 256 * k, S * S - 1);
        ij[k] = ((bin / S) << 8) | (bin % S);
    }
    const float step = 6.0f / (float)(S - 1);
    for (int p0 = 0; p0 < HW; p0 += PB) {
        const int np = min(PB, HW - p0);
        for (int idx = tid; idx < np * S; idx += 256) {
            const int p = idx / S, i = idx - p * S;
            float x[3];
            load_rgb01<T>(img, n, p0 + p, W, x);
            const float la = logf(x[ca] + HIST_EPS);
            const float u = la - logf(x[cp1] + HIST_EPS), v = la - logf(x[cp2] + HIST_EPS);
            const float d = -3.0f + (float)i * step;
            float tu = (u - d) * (u - d) * inv_sigma2, tv = (v - d) * (v - d) * inv_sigma2;
            if (method == 0) { tu = 1.0f / (1.0f + tu); tv = 1.0f / (1.0f + tv); }
            else if (method == 1) { tu = expf(-tu); tv = expf(-tv); }
            ku[idx] = tu;
            kv[idx] = tv;
            if (i == 0) iyL[p] = sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + HIST_EPS);
        }
        __syncthreads();
        for (int p = 0; p < np; ++p) {
            const float iy = iyL[p];
            const float* const kup = ku + p * S;
            const float* const kvp = kv + p * S;
#pragma unroll
            for (int k = 0; k < MAXB; ++k)
                if (k < nb) acc[k] += iy * kup[ij[k] >> 8] * kvp[ij[k] & 255];
        }
        __syncthreads();
    }
    float* out = hist + ((long long)n * 3 + c) * S * S;
#pragma unroll
    for (int k = 0; k < MAXB; ++k) {
        const int bin = tid + 256 * k;
        if (k < nb && bin < S * S) out[bin] = acc[k];
    }
}

extern "C" int p2p_rgbuv_hist_general(int dtype, int N, int H, int W, const p2p_tensor* img, int size, int method, float sigma,
                                      float* hist, void* stream) {
    P2P_REQUIRE(N > 0 && H > 0 && W > 0 && img && img->ptr && hist, "p2p_rgbuv_hist_general: bad args");
    P2P_REQUIRE(size >= 2 && size <= 128 && sigma > 0.f && method >= 0 && method <= 2, "p2p_rgbuv_hist_general: size in 2..128, sigma > 0, method 0..2");
    const size_t shm = (size_t)(2 * 32 * size + 32) * sizeof(float);
    const float inv_sigma2 = 1.0f / (sigma * sigma);
    P2P_DISPATCH_DTYPE(dtype, (rgbuv_hist_general_kernel<T><<<dim3(N, 3), 256, shm, (hipStream_t)stream>>>(H, W, make_view(img), size, method,
                                                                                                           inv_sigma2, hist)));
    return p2p_check_launch("p2p_rgbuv_hist_general");
}

extern "C" int p2p_rgbuv_hist_fwd(int dtype, int N, int H, int W, const p2p_tensor* img, float* hist, void* stream) {
    P2P_REQUIRE(N > 0 && H > 0 && W > 0 && img && img->ptr && hist, "p2p_rgbuv_hist_fwd: bad args");
    P2P_DISPATCH_DTYPE(dtype, (rgbuv_hist_fwd_kernel<T><<<dim3(N, 3), 256, 0, (hipStream_t)stream>>>(H, W, make_view(img), hist)));
    return p2p_check_launch("p2p_rgbuv_hist_fwd");
}

extern "C" long long p2p_rgbuv_hist_fwd3_workspace_bytes(int N) { return (long long)N * H3_PS * 3 * HB * HB * (long long)sizeof(float); }

extern "C" int p2p_rgbuv_points(int dtype, int N, int H, int W, const p2p_tensor* img, int cap, float* points, int* npoints, void* stream) {
    P2P_REQUIRE(N > 0 && H > 0 && W > 0 && img && img->ptr && points && npoints && cap > 0, "p2p_rgbuv_points: bad args");
    P2P_REQUIRE(((uintptr_t)points % 16) == 0, "p2p_rgbuv_points: the point list must be 16-byte aligned");
    P2P_DISPATCH_DTYPE(dtype, (rgbuv_points_kernel<T><<<dim3(N), 256, 0, (hipStream_t)stream>>>(H, W, make_view(img), cap, (f32x4*)points, npoints)));
    return p2p_check_launch("p2p_rgbuv_points");
}

extern "C" int p2p_rgbuv_hist_fwd3(int dtype, int N, int H, int W, const p2p_tensor* img, const float* points, const int* npoints,
                                   int cap, float* hist, float* workspace, void* stream) {
    P2P_REQUIRE(N > 0 && H > 0 && W > 0 && img && img->ptr && hist && workspace, "p2p_rgbuv_hist_fwd3: bad args");
    P2P_REQUIRE((points == nullptr) == (npoints == nullptr), "p2p_rgbuv_hist_fwd3: points and npoints come together");
    P2P_REQUIRE(((uintptr_t)workspace % 16) == 0 && ((uintptr_t)hist % 16) == 0, "p2p_rgbuv_hist_fwd3: alignment");
    hipStream_t st = (hipStream_t)stream;
    constexpr int SHM = H3_SHM;
    static bool attr = false;
    if (!attr)
        attr = (int)p2p_allow_lds((const void*)rgbuv_hist_fwd3_kernel<float>, SHM, "rgbuv_hist_fwd3_kernel<float>") &
               (int)p2p_allow_lds((const void*)rgbuv_hist_fwd3_kernel<bf16_t>, SHM, "rgbuv_hist_fwd3_kernel<bf16>");
    P2P_DISPATCH_DTYPE(dtype, (rgbuv_hist_fwd3_kernel<T><<<dim3(N, H3_PS), H3_NT, SHM, st>>>(H, W, make_view(img), (const f32x4*)points,
                                                                                            npoints, cap, workspace)));
    int rc = p2p_check_launch("p2p_rgbuv_hist_fwd3");
    if (rc) return rc;
    rgbuv_hist_fold_kernel<<<dim3(N), 256, 0, st>>>(workspace, hist);
    return p2p_check_launch("p2p_rgbuv_hist_fwd3 fold");
}

extern "C" int p2p_hellinger_fwd(const float* hist_true, const float* hist_pred, int N, float* tot_true, float* tot_pred,
                                 float* sq_part, float* sq_sum, void* stream) {
    P2P_REQUIRE(hist_true && hist_pred && N > 0 && tot_true && tot_pred && sq_part && sq_sum, "p2p_hellinger_fwd: bad args");
    hipStream_t st = (hipStream_t)stream;
    hellinger_fwd_kernel<<<dim3(N), 256, 0, st>>>(hist_true, hist_pred, tot_true, tot_pred, sq_part);
    hellinger_sq_sum_kernel<<<1, 256, 0, st>>>(sq_part, N, sq_sum);
    return p2p_check_launch("p2p_hellinger_fwd");
}

extern "C" int p2p_hellinger_finish(const float* sq_sum, float inv_global_batch, float* loss_out, void* stream) {
    P2P_REQUIRE(sq_sum && loss_out, "p2p_hellinger_finish: bad args");
    hellinger_finish_kernel<<<1, 1, 0, (hipStream_t)stream>>>(sq_sum, inv_global_batch, loss_out);
    return p2p_check_launch("p2p_hellinger_finish");
}

extern "C" int p2p_rgbuv_hist_hellinger_bwd3(int dtype, int N, int H, int W, const p2p_tensor* fake, const float* hist_true,
                                             const float* hist_pred, const float* tot_true, const float* tot_pred,
                                             const float* sq_sum, float coef, float* gh_ws, float* dimg, void* stream) {
    P2P_REQUIRE(N > 0 && H > 0 && W > 0 && fake && fake->ptr && hist_true && hist_pred && tot_true && tot_pred && sq_sum && gh_ws && dimg,
                "p2p_rgbuv_hist_hellinger_bwd3: bad args");
    hipStream_t st = (hipStream_t)stream;
    hist_grad_prep_kernel<<<dim3(N), 256, 0, st>>>(hist_true, hist_pred, tot_true, tot_pred, sq_sum, coef, gh_ws);
    int rc = p2p_check_launch("p2p_rgbuv_hist_hellinger_bwd3 prep");
    if (rc) return rc;
    int nsplit = 1;
    while (N * nsplit < 256 && (H * W) / (nsplit * 2) >= 8 * B3_PB) nsplit *= 2;       // one 12-wave workgroup per CU
    constexpr int SHM = B3_SHM;
    static_assert(SHM <= 160 * 1024, "rgbuv_hist_bwd3_kernel: LDS");
    static bool attr = false;
    if (!attr)
        attr = (int)p2p_allow_lds((const void*)rgbuv_hist_bwd3_kernel<float>, SHM, "rgbuv_hist_bwd3_kernel<float>") &
               (int)p2p_allow_lds((const void*)rgbuv_hist_bwd3_kernel<bf16_t>, SHM, "rgbuv_hist_bwd3_kernel<bf16>");
    P2P_DISPATCH_DTYPE(dtype, (rgbuv_hist_bwd3_kernel<T><<<dim3(N, nsplit), B3_NT, SHM, st>>>(H, W, make_view(fake), gh_ws, dimg, nsplit)));
    return p2p_check_launch("p2p_rgbuv_hist_hellinger_bwd3");
}

extern "C" int p2p_rgbuv_hist_hellinger_bwd(int dtype, int N, int H, int W, const p2p_tensor* fake, const float* hist_true,
                                            const float* hist_pred, const float* tot_true, const float* tot_pred,
                                            const float* sq_sum, float coef, float* gh_ws, float* dimg, void* stream) {
    P2P_REQUIRE(N > 0 && H > 0 && W > 0 && fake && fake->ptr && hist_true && hist_pred && tot_true && tot_pred && sq_sum && gh_ws && dimg,
                "p2p_rgbuv_hist_hellinger_bwd: bad args");
    hipStream_t st = (hipStream_t)stream;
    hist_grad_prep_kernel<<<dim3(N), 256, 0, st>>>(hist_true, hist_pred, tot_true, tot_pred, sq_sum, coef, gh_ws);
    int rc = p2p_check_launch("p2p_rgbuv_hist_hellinger_bwd prep");
    if (rc) return rc;
    long long slab = (long long)N * H * W * 4;
    P2P_DISPATCH_DTYPE(dtype, (rgbuv_hist_bwd_kernel<T><<<dim3(N, 3), 256, 0, st>>>(H, W, make_view(fake), gh_ws, dimg, slab)));
    return p2p_check_launch("p2p_rgbuv_hist_hellinger_bwd");
}
