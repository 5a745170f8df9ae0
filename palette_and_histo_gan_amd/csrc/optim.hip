// Keras Adam over the flat parameter buffer, weight-copy preparation and batch packing.
// Reference: tf.keras.optimizers.Adam(0.0002, beta_1=0.5) applied at pix2pix_model.py:28-29,81-83
// (OptimizerV2 formulation: eps added to sqrt(v) outside the bias correction);
// batch value contract dataset_utils.py:39-48,209-246.
#include "p2p_common.hpp"
#include <math.h>
#include <stdlib.h>

__global__ void adam_flat_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                 float* __restrict__ v, long long n, float lr_t, float b1, float b2, float eps,
                                 float gscale) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        float gi = g[i] * gscale;
        float mi = b1 * m[i] + (1.f - b1) * gi;
        float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] = p[i] - lr_t * mi / (sqrtf(vi) + eps);
    }
}

extern "C" int p2p_adam_flat(float* p, const float* g, float* m, float* v, long long n, int t, float lr, float beta1,
                             float beta2, float eps, float gscale, void* stream) {
    P2P_REQUIRE(p && g && m && v && n > 0 && t >= 1, "p2p_adam_flat: bad args");
    double lr_t = (double)lr * sqrt(1.0 - pow((double)beta2, (double)t)) / (1.0 - pow((double)beta1, (double)t));
    long long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    adam_flat_kernel<<<dim3((unsigned)blocks), 256, 0, (hipStream_t)stream>>>(p, g, m, v, n, (float)lr_t, beta1, beta2,
                                                                             eps, gscale);
    return p2p_check_launch("p2p_adam_flat");
}

// Device-resident step state, so that a whole train step can be captured once in a hipGraph and replayed: the Adam
// iteration count / bias-corrected step size and the dropout counter live in HBM and are advanced by these kernels.
__global__ void adam_tick_kernel(int* __restrict__ t, float* __restrict__ lr_t, float lr, float b1, float b2) {
    int tt = t[0] + 1;
    t[0] = tt;
    lr_t[0] = (float)((double)lr * sqrt(1.0 - pow((double)b2, (double)tt)) / (1.0 - pow((double)b1, (double)tt)));
}

__global__ void adam_flat_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                     float* __restrict__ v, long long n, const float* __restrict__ lr_t_dev, float b1,
                                     float b2, float eps, float gscale) {
    const float lr_t = lr_t_dev[0];
    long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    long long stride = (long long)gridDim.x * blockDim.x * 4;
    for (; i + 3 < n; i += stride) {
        f32x4 gi = *(const f32x4*)(g + i), mi = *(const f32x4*)(m + i), vi = *(const f32x4*)(v + i), pi = *(const f32x4*)(p + i);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float gg = gi[k] * gscale;
            mi[k] = b1 * mi[k] + (1.f - b1) * gg;
            vi[k] = b2 * vi[k] + (1.f - b2) * gg * gg;
            pi[k] = pi[k] - lr_t * mi[k] / (sqrtf(vi[k]) + eps);
        }
        *(f32x4*)(m + i) = mi; *(f32x4*)(v + i) = vi; *(f32x4*)(p + i) = pi;
    }
    // tail (n is padded to a multiple of 4 by the host layout, kept for safety)
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (long long j = n / 4 * 4; j < n; ++j) {
            float gg = g[j] * gscale;
            float mj = b1 * m[j] + (1.f - b1) * gg, vj = b2 * v[j] + (1.f - b2) * gg * gg;
            m[j] = mj; v[j] = vj; p[j] = p[j] - lr_t * mj / (sqrtf(vj) + eps);
        }
}

extern "C" int p2p_adam_tick(int* t_dev, float* lr_t_dev, float lr, float beta1, float beta2, void* stream) {
    P2P_REQUIRE(t_dev && lr_t_dev, "p2p_adam_tick: null pointer");
    adam_tick_kernel<<<1, 1, 0, (hipStream_t)stream>>>(t_dev, lr_t_dev, lr, beta1, beta2);
    return p2p_check_launch("p2p_adam_tick");
}

extern "C" int p2p_adam_flat_dev(float* p, const float* g, float* m, float* v, long long n, const float* lr_t_dev,
                                 float beta1, float beta2, float eps, float gscale, void* stream) {
    P2P_REQUIRE(p && g && m && v && n > 0 && lr_t_dev, "p2p_adam_flat_dev: bad args");
    P2P_REQUIRE(((uintptr_t)p % 16) == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)m % 16) == 0 && ((uintptr_t)v % 16) == 0,
                "p2p_adam_flat_dev: buffers must be 16-byte aligned");
    long long blocks = (n / 4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    adam_flat_dev_kernel<<<dim3((unsigned)blocks), 256, 0, (hipStream_t)stream>>>(p, g, m, v, n, lr_t_dev, beta1, beta2, eps, gscale);
    return p2p_check_launch("p2p_adam_flat_dev");
}

__global__ void counter_add_kernel(long long* c, long long inc) { c[0] += inc; }

extern "C" int p2p_counter_add(long long* counter_dev, long long inc, void* stream) {
    P2P_REQUIRE(counter_dev, "p2p_counter_add: null pointer");
    counter_add_kernel<<<1, 1, 0, (hipStream_t)stream>>>(counter_dev, inc);
    return p2p_check_launch("p2p_counter_add");
}

// wn[t][g][d] = T(w[t][g][d]) for g < wn_rows, d < wn_cols;  wt[t][d][g] = T(w[t][g][d]) for d < wt_rows,
// g < wt_cols; entries outside the real [Cg][Cd] block are written as zeros (channel padding of the edge
// layers).  32x32 LDS tile transpose per tap.
template <typename T>
__device__ __forceinline__ void weight_prep_tile(const float* __restrict__ w, int Cg, int Cd, T* __restrict__ wn, int wn_rows,
                                                 int wn_cols, T* __restrict__ wt, int wt_rows, int wt_cols, int t, int g0, int d0,
                                                 float (*tile)[33]) {
    const float* wp = w + (long long)t * Cg * Cd;
    int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        int g = g0 + r, d = d0 + tx;
        float val = (g < Cg && d < Cd) ? wp[(long long)g * Cd + d] : 0.f;
        tile[r][tx] = val;
        if (wn && g < wn_rows && d < wn_cols) wn[((long long)t * wn_rows + g) * wn_cols + d] = from_f32<T>(val);
    }
    __syncthreads();
    if (wt) {
        for (int r = ty; r < 32; r += 8) {
            int d = d0 + r, g = g0 + tx;
            if (d < wt_rows && g < wt_cols) wt[((long long)t * wt_rows + d) * wt_cols + g] = from_f32<T>(tile[tx][r]);
        }
    }
}

template <typename T>
__global__ void weight_prep_kernel(const float* __restrict__ w, int Cg, int Cd, T* __restrict__ wn, int wn_rows,
                                   int wn_cols, T* __restrict__ wt, int wt_rows, int wt_cols) {
    __shared__ float tile[32][33];
    weight_prep_tile<T>(w, Cg, Cd, wn, wn_rows, wn_cols, wt, wt_rows, wt_cols, blockIdx.z, blockIdx.y * 32, blockIdx.x * 32, tile);
}

// Batched form: one launch re-derives the copies of every layer after an optimizer step.  Workgroup b serves the
// task whose [first_block, first_block + 16 * tiles_g * tiles_d) range holds b (<= 64 tasks, scanned linearly).
// 64x64 tiles with 16-byte reads and 4-element stores for the unpadded layers whose channel counts are multiples of 64
// (every encoder/decoder block but up6): a quarter of the workgroups, whole 128/256-byte rows per wave access.
static __host__ __device__ inline bool prep_tile64(int Cg, int Cd, int wn_rows, int wn_cols, int wt_rows, int wt_cols,
                                                   bool have_wn, bool have_wt) {
    return Cg % 64 == 0 && Cd % 64 == 0 && (!have_wn || (wn_rows == Cg && wn_cols == Cd)) &&
           (!have_wt || (wt_rows == Cd && wt_cols == Cg));
}

template <typename T>
__device__ __forceinline__ void weight_prep_tile64(const float* __restrict__ w, int Cg, int Cd, T* __restrict__ wn,
                                                   T* __restrict__ wt, int t, int g0, int d0, float (*tile)[65]) {
    typedef __attribute__((__vector_size__(4 * sizeof(T)))) T vec4_t;
    const float* wp = w + (long long)t * Cg * Cd;
    const int c4 = (threadIdx.x & 15) * 4, r0 = threadIdx.x >> 4;         // 16 column quads x 16 rows per pass
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + 16 * i;
        const f32x4 v = *(const f32x4*)(wp + (long long)(g0 + r) * Cd + d0 + c4);
        tile[r][c4] = v[0]; tile[r][c4 + 1] = v[1]; tile[r][c4 + 2] = v[2]; tile[r][c4 + 3] = v[3];
        if (wn) {
            vec4_t q;
#pragma unroll
            for (int k = 0; k < 4; ++k) q[k] = from_f32<T>(v[k]);
            *(vec4_t*)(wn + ((long long)t * Cg + g0 + r) * Cd + d0 + c4) = q;
        }
    }
    __syncthreads();
    if (wt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int d = r0 + 16 * i;
            vec4_t q;
#pragma unroll
            for (int k = 0; k < 4; ++k) q[k] = from_f32<T>(tile[c4 + k][d]);
            *(vec4_t*)(wt + ((long long)t * Cd + d0 + d) * Cg + g0 + c4) = q;
        }
    }
}

template <typename T>
__global__ void weight_prep_batched_kernel(const p2p_prep_task* __restrict__ tasks, int ntasks) {
    __shared__ float tile[64][65];
    const long long b = blockIdx.x;
    int ti = 0;
    while (ti + 1 < ntasks && tasks[ti + 1].first_block <= b) ++ti;
    const p2p_prep_task k = tasks[ti];
    const int local = (int)(b - k.first_block);
    const int per_tap = k.tiles_g * k.tiles_d;
    const int t = local / per_tap, rem = local - t * per_tap;
    const int gy = rem / k.tiles_d, dx = rem - gy * k.tiles_d;
    if (prep_tile64(k.Cg, k.Cd, k.wn_rows, k.wn_cols, k.wt_rows, k.wt_cols, k.wn != nullptr, k.wt != nullptr)) {
        weight_prep_tile64<T>(k.w, k.Cg, k.Cd, (T*)k.wn, (T*)k.wt, t, gy * 64, dx * 64, tile);
        return;
    }
    weight_prep_tile<T>(k.w, k.Cg, k.Cd, (T*)k.wn, k.wn_rows, k.wn_cols, (T*)k.wt, k.wt_rows, k.wt_cols, t, gy * 32, dx * 32,
                        (float (*)[33])tile);
}

// ---- Adam + weight copies in one pass (SURVEY.md 2.3 K18 "also emits the bf16 weight copy") -------------------------------
// The same tiling as the batched weight prep, but the master tile is UPDATED on the way: theta, g, m, v are read, the Keras
// Adam step of adam_flat_dev_kernel is applied (same expressions: bit-identical results), theta, m, v are written back and the
// fresh value goes straight into the two operand copies.  Saves the second read of the 117 MB master.  Every master element
// lies in exactly one tile of its task; the caller lists every kernel ONCE.
struct AdamCtx { float* p0; const float* g0; float* m0; float* v0; const float* lr_t; float b1, b2, eps; };

#define ADAM_ONE(p, g, m, v)                                   \
    do {                                                       \
        m = c.b1 * (m) + (1.f - c.b1) * (g);                   \
        v = c.b2 * (v) + (1.f - c.b2) * (g) * (g);             \
        p = (p) - lr_t * (m) / (sqrtf(v) + c.eps);             \
    } while (0)

template <typename T>
__global__ void adam_prep_batched_kernel(const p2p_prep_task* __restrict__ tasks, int ntasks, AdamCtx c) {
    typedef __attribute__((__vector_size__(4 * sizeof(T)))) T vec4_t;
    __shared__ float tile[64][65];
    const long long b = blockIdx.x;
    int ti = 0;
    while (ti + 1 < ntasks && tasks[ti + 1].first_block <= b) ++ti;
    const p2p_prep_task k = tasks[ti];
    const int local = (int)(b - k.first_block);
    const int per_tap = k.tiles_g * k.tiles_d;
    const int t = local / per_tap, rem = local - t * per_tap;
    const int gy = rem / k.tiles_d, dx = rem - gy * k.tiles_d;
    const float lr_t = c.lr_t[0];
    const long long base = (k.w - c.p0) + (long long)t * k.Cg * k.Cd;        // element index of this tap in the flat buffers
    T* wn = (T*)k.wn;
    T* wt = (T*)k.wt;
    if (prep_tile64(k.Cg, k.Cd, k.wn_rows, k.wn_cols, k.wt_rows, k.wt_cols, wn != nullptr, wt != nullptr)) {
        const int g0 = gy * 64, d0 = dx * 64;
        const int c4 = (threadIdx.x & 15) * 4, r0 = threadIdx.x >> 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = r0 + 16 * i;
            const long long e = base + (long long)(g0 + r) * k.Cd + d0 + c4;
            f32x4 pv = *(const f32x4*)(c.p0 + e), mv = *(const f32x4*)(c.m0 + e), vv = *(const f32x4*)(c.v0 + e);
            const f32x4 gv = *(const f32x4*)(c.g0 + e);
#pragma unroll
            for (int q = 0; q < 4; ++q) ADAM_ONE(pv[q], gv[q], mv[q], vv[q]);
            *(f32x4*)(c.p0 + e) = pv; *(f32x4*)(c.m0 + e) = mv; *(f32x4*)(c.v0 + e) = vv;
            tile[r][c4] = pv[0]; tile[r][c4 + 1] = pv[1]; tile[r][c4 + 2] = pv[2]; tile[r][c4 + 3] = pv[3];
            if (wn) {
                vec4_t q4;
#pragma unroll
                for (int q = 0; q < 4; ++q) q4[q] = from_f32<T>(pv[q]);
                *(vec4_t*)(wn + ((long long)t * k.Cg + g0 + r) * k.Cd + d0 + c4) = q4;
            }
        }
        __syncthreads();
        if (wt) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int d = r0 + 16 * i;
                vec4_t q4;
#pragma unroll
                for (int q = 0; q < 4; ++q) q4[q] = from_f32<T>(tile[c4 + q][d]);
                *(vec4_t*)(wt + ((long long)t * k.Cd + d0 + d) * k.Cg + g0 + c4) = q4;
            }
        }
        return;
    }
    float (*t32)[33] = (float (*)[33])tile;
    const int g0 = gy * 32, d0 = dx * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int g = g0 + r, d = d0 + tx;
        float val = 0.f;
        if (g < k.Cg && d < k.Cd) {
            const long long e = base + (long long)g * k.Cd + d;
            float pv = c.p0[e], mv = c.m0[e], vv = c.v0[e];
            const float gv = c.g0[e];
            ADAM_ONE(pv, gv, mv, vv);
            val = pv;
            c.p0[e] = pv; c.m0[e] = mv; c.v0[e] = vv;
        }
        t32[r][tx] = val;
        if (wn && g < k.wn_rows && d < k.wn_cols) wn[((long long)t * k.wn_rows + g) * k.wn_cols + d] = from_f32<T>(val);
    }
    __syncthreads();
    if (wt) {
        for (int r = ty; r < 32; r += 8) {
            const int d = d0 + r, g = g0 + tx;
            if (d < k.wt_rows && g < k.wt_cols) wt[((long long)t * k.wt_rows + d) * k.wt_cols + g] = from_f32<T>(t32[tx][r]);
        }
    }
}

extern "C" int p2p_adam_prep_batched(int dtype, long long n_elems, const p2p_prep_task* tasks_dev, int ntasks, long long total_blocks,
                                     float* params, const float* grads, float* m, float* v, const float* lr_t_dev, float beta1,
                                     float beta2, float eps, void* stream) {
    P2P_REQUIRE(tasks_dev && ntasks >= 1 && ntasks <= 64 && total_blocks >= 1 && total_blocks < (1LL << 31) && n_elems > 0,
                "p2p_adam_prep_batched: bad args");
    P2P_REQUIRE(params && grads && m && v && lr_t_dev, "p2p_adam_prep_batched: null pointer");
    P2P_REQUIRE(((uintptr_t)params % 16) == 0 && ((uintptr_t)grads % 16) == 0 && ((uintptr_t)m % 16) == 0 && ((uintptr_t)v % 16) == 0,
                "p2p_adam_prep_batched: buffers must be 16-byte aligned");
    AdamCtx c = {params, grads, m, v, lr_t_dev, beta1, beta2, eps};
    P2P_DISPATCH_DTYPE(dtype, (adam_prep_batched_kernel<T><<<dim3((unsigned)total_blocks), 256, 0, (hipStream_t)stream>>>(tasks_dev, ntasks, c)));
    return p2p_check_launch("p2p_adam_prep_batched");
}

extern "C" long long p2p_weight_prep_task_blocks(int Cg, int Cd, int wn_rows, int wn_cols, int wt_rows, int wt_cols,
                                                 int have_wn, int have_wt, int* tiles_g, int* tiles_d) {
    int gmax = Cg, dmax = Cd;
    if (have_wn) { gmax = gmax > wn_rows ? gmax : wn_rows; dmax = dmax > wn_cols ? dmax : wn_cols; }
    if (have_wt) { gmax = gmax > wt_cols ? gmax : wt_cols; dmax = dmax > wt_rows ? dmax : wt_rows; }
    const int ts = prep_tile64(Cg, Cd, wn_rows, wn_cols, wt_rows, wt_cols, have_wn != 0, have_wt != 0) ? 64 : 32;
    const int tg = (gmax + ts - 1) / ts, td = (dmax + ts - 1) / ts;
    if (tiles_g) *tiles_g = tg;
    if (tiles_d) *tiles_d = td;
    return 16LL * tg * td;
}

extern "C" int p2p_weight_prep_batched(int dtype, const p2p_prep_task* tasks_dev, int ntasks, long long total_blocks, void* stream) {
    P2P_REQUIRE(tasks_dev && ntasks >= 1 && ntasks <= 64 && total_blocks >= 1 && total_blocks < (1LL << 31),
                "p2p_weight_prep_batched: bad args");
    P2P_DISPATCH_DTYPE(dtype, (weight_prep_batched_kernel<T><<<dim3((unsigned)total_blocks), 256, 0, (hipStream_t)stream>>>(tasks_dev, ntasks)));
    return p2p_check_launch("p2p_weight_prep_batched");
}

extern "C" int p2p_weight_prep_pad(int dtype, const float* w, int Cg, int Cd, void* wn, int wn_rows, int wn_cols,
                                   void* wt, int wt_rows, int wt_cols, void* stream) {
    P2P_REQUIRE(w && Cg > 0 && Cd > 0 && (wn || wt), "p2p_weight_prep: bad args");
    int gmax = Cg, dmax = Cd;
    if (wn) { gmax = gmax > wn_rows ? gmax : wn_rows; dmax = dmax > wn_cols ? dmax : wn_cols; }
    if (wt) { gmax = gmax > wt_cols ? gmax : wt_cols; dmax = dmax > wt_rows ? dmax : wt_rows; }
    dim3 grid((dmax + 31) / 32, (gmax + 31) / 32, 16);
    P2P_DISPATCH_DTYPE(dtype, (weight_prep_kernel<T><<<grid, 256, 0, (hipStream_t)stream>>>(
                                  w, Cg, Cd, (T*)wn, wn_rows, wn_cols, (T*)wt, wt_rows, wt_cols)));
    return p2p_check_launch("p2p_weight_prep");
}

extern "C" int p2p_weight_prep(int dtype, const float* w, int Cg, int Cd, void* wn, void* wt, void* stream) {
    return p2p_weight_prep_pad(dtype, w, Cg, Cd, wn, Cg, Cd, wt, Cd, Cg, stream);
}

template <typename T>
__global__ void pack_input_kernel(int N, PixDec dec, int C, const void* __restrict__ src, int src_is_int, TView dst) {
    const unsigned npix = (unsigned)N * dec.H * dec.W;
    for (unsigned p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += gridDim.x * blockDim.x) {
        int n, y, x;
        dec(p, n, y, x);
        T* d = (T*)dst.ptr + dst.off(n, y, x);
        const long long e = (long long)p * C;
        for (int c = 0; c < C; ++c) {
            float v = src_is_int ? (float)((const int*)src)[e + c] : ((const float*)src)[e + c];
            d[c] = from_f32<T>(v);
        }
    }
}

// One launch scatters a dense batch into up to 4 views (the source image feeds down1, the last skip connection and
// both discriminator inputs; networks.py:45,92-94): read once, write everywhere.
struct PackDst { TView v[4]; int n; };

template <typename T>
__global__ void pack_multi_kernel(int N, PixDec dec, int C, const void* __restrict__ src, int src_is_int, PackDst dst) {
    const unsigned npix = (unsigned)N * dec.H * dec.W;
    for (unsigned p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += gridDim.x * blockDim.x) {
        int n, y, x;
        dec(p, n, y, x);
        if (C == 4 && !src_is_int) {      // RGBA pixel: one 16-byte read, one 4-channel vector store per view
            typedef __attribute__((__vector_size__(4 * sizeof(T)))) T vec4_t;
            f32x4 v = *(const f32x4*)((const float*)src + (long long)p * 4);
            vec4_t q;
#pragma unroll
            for (int k = 0; k < 4; ++k) q[k] = from_f32<T>(v[k]);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < dst.n) *(vec4_t*)((T*)dst.v[k].ptr + dst.v[k].off(n, y, x)) = q;
            continue;
        }
        const long long e = (long long)p * C;
        for (int c = 0; c < C; ++c) {
            float v = src_is_int ? (float)((const int*)src)[e + c] : ((const float*)src)[e + c];
            T q = from_f32<T>(v);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < dst.n) ((T*)dst.v[k].ptr)[dst.v[k].off(n, y, x) + c] = q;
        }
    }
}

// The RGBA batch of a train step in ONE launch (dataset_utils.py:209-229 -> networks.py:45,92-94): per pixel one 16-byte read
// of the source and of the target, and whole 16-byte pixel stores wherever the destination pixel is complete:
//   v_src   [source | 0 0 0 0]            down1's input
//   v_c6    [source | 0 0 0 0]            channels 32..39 of the last concat buffer
//   v_dreal [target | source]             discriminator input, real half
//   v_dfake [  ...  | source]             discriminator input, fake half: only the source half (the generator writes the rest)
template <typename T>
__global__ void pack_pair_kernel(int N, PixDec dec, const float* __restrict__ source, const float* __restrict__ target, TView v_src,
                                 TView v_c6, TView v_dreal, TView v_dfake) {
    typedef __attribute__((__vector_size__(4 * sizeof(T)))) T vec4_t;
    typedef __attribute__((__vector_size__(8 * sizeof(T)))) T vec8_t;
    const unsigned npix = (unsigned)N * dec.H * dec.W;
    for (unsigned p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += gridDim.x * blockDim.x) {
        int n, y, x;
        dec(p, n, y, x);
        const f32x4 s4 = *(const f32x4*)(source + (long long)p * 4), t4 = *(const f32x4*)(target + (long long)p * 4);
        vec4_t sq;
        vec8_t sz, ts;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            sq[k] = from_f32<T>(s4[k]);
            sz[k] = sq[k]; sz[4 + k] = from_f32<T>(0.f);
            ts[k] = from_f32<T>(t4[k]); ts[4 + k] = sq[k];
        }
        *(vec8_t*)((T*)v_src.ptr + v_src.off(n, y, x)) = sz;
        if (v_c6.ptr) *(vec8_t*)((T*)v_c6.ptr + v_c6.off(n, y, x)) = sz;
        *(vec8_t*)((T*)v_dreal.ptr + v_dreal.off(n, y, x)) = ts;
        if (v_dfake.ptr) *(vec4_t*)((T*)v_dfake.ptr + v_dfake.off(n, y, x) + 4) = sq;
    }
}

extern "C" int p2p_pack_pair(int dtype, int N, int H, int W, const float* source, const float* target, const p2p_tensor* v_src,
                             const p2p_tensor* v_c6, const p2p_tensor* v_dreal, const p2p_tensor* v_dfake, void* stream) {
    P2P_REQUIRE(N > 0 && H > 0 && W > 0 && source && target && v_src && v_dreal, "p2p_pack_pair: bad args");
    P2P_REQUIRE((long long)N * H * W < (1LL << 31), "p2p_pack_pair: too many pixels");
    const int esz = dtype == P2P_BF16 ? 2 : 4;
    // v_c6 / v_dfake may be NULL: those two stores are PARTIAL pixels (16 of 80 bytes, 8 of 16), and the train step leaves them
    // to the kernels that write the rest of the pixel (p2p_norm_act_fwd_tail, p2p_tanh_l1_fwd_pair)
    const p2p_tensor* vs[4] = {v_src, v_c6, v_dreal, v_dfake};
    for (int k = 0; k < 4; ++k)
        P2P_REQUIRE(!vs[k] || (vs[k]->ptr && vs[k]->ld % 8 == 0 && ((uintptr_t)vs[k]->ptr % (8 * esz)) == 0),
                    "p2p_pack_pair: views must start on an 8-channel boundary (whole 16/32-byte pixels)");
    TView none;
    none.ptr = nullptr; none.img = 0; none.row = 0; none.ld = 0;
    const TView vc = v_c6 ? make_view(v_c6) : none, vd = v_dfake ? make_view(v_dfake) : none;
    P2P_REQUIRE(((uintptr_t)source % 16) == 0 && ((uintptr_t)target % 16) == 0, "p2p_pack_pair: batches must be 16-byte aligned");
    long long blocks = ((long long)N * H * W + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    P2P_DISPATCH_DTYPE(dtype, (pack_pair_kernel<T><<<dim3((unsigned)blocks), 256, 0, (hipStream_t)stream>>>(
                                  N, PixDec::make(H, W), source, target, make_view(v_src), vc, make_view(v_dreal), vd)));
    return p2p_check_launch("p2p_pack_pair");
}

// The palette-index batch of a train step (Pix2PixIndexedModel: one int32 index per pixel, dataset_utils.py:232-246) in one
// launch, WHOLE 8-channel pixels everywhere (a 2-byte store into a 16-byte pixel is a partial sector write):
//   v_src   [source 0 0 0 0 0 0 0]      down1's input
//   v_c6    the same pixel               channels 32..39 of the last concat buffer (NULL: left to p2p_norm_act_fwd_tail)
//   v_dreal [target source 0 ...]        discriminator input, real half
//   v_dfake [0      source 0 ...]        fake half: channel 0 is written later by the head (argmax)
template <typename T>
__global__ void pack_pair_idx_kernel(int N, PixDec dec, const int* __restrict__ source, const int* __restrict__ target, TView v_src,
                                     TView v_c6, TView v_dreal, TView v_dfake) {
    typedef __attribute__((__vector_size__(8 * sizeof(T)))) T vec8_t;
    const unsigned npix = (unsigned)N * dec.H * dec.W;
    for (unsigned p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += gridDim.x * blockDim.x) {
        int n, y, x;
        dec(p, n, y, x);
        const T sq = from_f32<T>((float)source[p]), tq = from_f32<T>((float)target[p]), z = from_f32<T>(0.f);
        vec8_t a, b, c;
#pragma unroll
        for (int k = 0; k < 8; ++k) { a[k] = z; b[k] = z; c[k] = z; }
        a[0] = sq;
        b[0] = tq; b[1] = sq;
        c[1] = sq;
        *(vec8_t*)((T*)v_src.ptr + v_src.off(n, y, x)) = a;
        if (v_c6.ptr) *(vec8_t*)((T*)v_c6.ptr + v_c6.off(n, y, x)) = a;
        *(vec8_t*)((T*)v_dreal.ptr + v_dreal.off(n, y, x)) = b;
        *(vec8_t*)((T*)v_dfake.ptr + v_dfake.off(n, y, x)) = c;
    }
}

extern "C" int p2p_pack_pair_idx(int dtype, int N, int H, int W, const int* source, const int* target, const p2p_tensor* v_src,
                                 const p2p_tensor* v_c6, const p2p_tensor* v_dreal, const p2p_tensor* v_dfake, void* stream) {
    P2P_REQUIRE(N > 0 && H > 0 && W > 0 && source && target && v_src && v_dreal && v_dfake, "p2p_pack_pair_idx: bad args");
    P2P_REQUIRE((long long)N * H * W < (1LL << 31), "p2p_pack_pair_idx: too many pixels");
    const int esz = dtype == P2P_BF16 ? 2 : 4;
    const p2p_tensor* vs[4] = {v_src, v_c6, v_dreal, v_dfake};
    for (int k = 0; k < 4; ++k)
        P2P_REQUIRE(!vs[k] || (vs[k]->ptr && vs[k]->ld % 8 == 0 && ((uintptr_t)vs[k]->ptr % (8 * esz)) == 0),
                    "p2p_pack_pair_idx: views must start on an 8-channel boundary (whole 16/32-byte pixels)");
    TView none;
    none.ptr = nullptr; none.img = 0; none.row = 0; none.ld = 0;
    long long blocks = ((long long)N * H * W + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    P2P_DISPATCH_DTYPE(dtype, (pack_pair_idx_kernel<T><<<dim3((unsigned)blocks), 256, 0, (hipStream_t)stream>>>(
                                  N, PixDec::make(H, W), source, target, make_view(v_src), v_c6 ? make_view(v_c6) : none,
                                  make_view(v_dreal), make_view(v_dfake))));
    return p2p_check_launch("p2p_pack_pair_idx");
}

extern "C" int p2p_pack_input_multi(int dtype, int N, int H, int W, int C, const void* src, int src_is_int,
                                    const p2p_tensor* dsts, int ndst, void* stream) {
    P2P_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && src && dsts && ndst >= 1 && ndst <= 4, "p2p_pack_input_multi: bad args");
    PackDst d;
    d.n = ndst;
    for (int k = 0; k < 4; ++k) d.v[k] = make_view(&dsts[k < ndst ? k : 0]);
    if (C == 4 && !src_is_int) {      // the vector path needs 4-channel-aligned views
        const int esz = dtype == P2P_BF16 ? 2 : 4;
        for (int k = 0; k < ndst; ++k)
            P2P_REQUIRE(((uintptr_t)dsts[k].ptr % (4 * esz)) == 0 && (dsts[k].ld % 4) == 0 && ((uintptr_t)src % 16) == 0,
                        "p2p_pack_input_multi: RGBA views must be aligned to 4 channels");
    }
    long long total = (long long)N * H * W;
    P2P_REQUIRE(total < (1LL << 31), "p2p_pack_input_multi: too many pixels");
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    P2P_DISPATCH_DTYPE(dtype, (pack_multi_kernel<T><<<dim3((unsigned)blocks), 256, 0, (hipStream_t)stream>>>(N, PixDec::make(H, W), C, src, src_is_int, d)));
    return p2p_check_launch("p2p_pack_input_multi");
}

// out[0..6] = [g_total, g_adv, g_l1, g_aux, d_total, d_real, d_fake] from the loss slots (pix2pix_model.py:44-56,
// 242-250,273-278): g_total = adv + lambda_l1 * l1 + lambda_aux * aux, d_total = real + fake.
__global__ void finish_losses_kernel(const float* __restrict__ l, int aux_slot, int l1_slot, float lambda_l1, float lambda_aux,
                                     float* __restrict__ out) {
    float adv = l[2], l1 = l[l1_slot], aux = aux_slot >= 0 ? l[aux_slot] : 0.f;
    out[0] = adv + lambda_l1 * l1 + lambda_aux * aux;
    out[1] = adv; out[2] = l1; out[3] = aux;
    out[4] = l[0] + l[1]; out[5] = l[0]; out[6] = l[1];
}

extern "C" int p2p_finish_losses(const float* slots, int aux_slot, int l1_slot, float lambda_l1, float lambda_aux, float* out,
                                 void* stream) {
    P2P_REQUIRE(slots && out && l1_slot >= 0, "p2p_finish_losses: bad args");
    finish_losses_kernel<<<1, 1, 0, (hipStream_t)stream>>>(slots, aux_slot, l1_slot, lambda_l1, lambda_aux, out);
    return p2p_check_launch("p2p_finish_losses");
}

template <typename T>
__global__ void unpack_kernel(int N, int H, int W, int C, TView src, float* __restrict__ dst) {
    long long total = (long long)N * H * W * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int c = (int)(i % C);
        long long p = i / C;
        int x = (int)(p % W);
        int y = (int)((p / W) % H);
        int n = (int)(p / ((long long)W * H));
        dst[i] = to_f32(((const T*)src.ptr)[src.off(n, y, x) + c]);
    }
}

static inline unsigned grid_for(long long total) {
    long long b = (total + 255) / 256;
    return (unsigned)(b < 4096 ? (b < 1 ? 1 : b) : 4096);
}

extern "C" int p2p_pack_input(int dtype, int N, int H, int W, int C, const void* src, int src_is_int,
                              const p2p_tensor* dst, void* stream) {
    P2P_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && src && dst && dst->ptr, "p2p_pack_input: bad args");
    P2P_REQUIRE((long long)N * H * W < (1LL << 31), "p2p_pack_input: too many pixels");
    P2P_DISPATCH_DTYPE(dtype, (pack_input_kernel<T><<<dim3(grid_for((long long)N * H * W)), 256, 0, (hipStream_t)stream>>>(
                                  N, PixDec::make(H, W), C, src, src_is_int, make_view(dst))));
    return p2p_check_launch("p2p_pack_input");
}

extern "C" int p2p_unpack(int dtype, int N, int H, int W, int C, const p2p_tensor* src, float* dst, void* stream) {
    P2P_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && src && src->ptr && dst, "p2p_unpack: bad args");
    P2P_DISPATCH_DTYPE(dtype, (unpack_kernel<T><<<dim3(grid_for((long long)N * H * W * C)), 256, 0, (hipStream_t)stream>>>(
                                  N, H, W, C, make_view(src), dst)));
    return p2p_check_launch("p2p_unpack");
}

// Bernoulli(0.5) keep mask of keras Dropout(0.5) (networks.py:31-32): counter-based (splitmix64 of
// seed, call counter, element index), one byte per element, 8 elements per hash.
__global__ void dropout_mask_kernel(unsigned char* __restrict__ mask, long long n, unsigned long long seed,
                                    unsigned long long counter, const long long* __restrict__ counter_dev, long long group0) {
    if (counter_dev) counter += (unsigned long long)counter_dev[0] * 16ull;      // device step counter (graph replay)
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long stride = (long long)gridDim.x * blockDim.x;
    for (; i * 8 < n; i += stride) {
        // the stream is keyed by the GLOBAL element index (group0 = 8-element groups in front of this shard): an N-rank
        // data-parallel step draws the masks of the single-process global batch
        unsigned long long z = seed * 0x9E3779B97F4A7C15ull + counter * 0xD1B54A32D192ED03ull + (unsigned long long)(i + group0);
        z += 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z = z ^ (z >> 31);
        for (int k = 0; k < 8 && i * 8 + k < n; ++k) mask[i * 8 + k] = (unsigned char)((z >> (8 * k + 3)) & 1ull);
    }
}

extern "C" int p2p_dropout_mask(unsigned char* mask, long long n, long long seed, long long counter, void* stream) {
    P2P_REQUIRE(mask && n > 0, "p2p_dropout_mask: bad args");
    long long blocks = (n / 8 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    dropout_mask_kernel<<<dim3((unsigned)blocks), 256, 0, (hipStream_t)stream>>>(mask, n, (unsigned long long)seed,
                                                                                (unsigned long long)counter, nullptr, 0);
    return p2p_check_launch("p2p_dropout_mask");
}

// same, with the call counter = counter_dev[0] * 16 + salt read on the device (one salt per dropout layer)
extern "C" int p2p_dropout_mask_dev(unsigned char* mask, long long n, long long seed, const long long* counter_dev,
                                    long long salt, long long elem_offset, void* stream) {
    P2P_REQUIRE(mask && n > 0 && counter_dev, "p2p_dropout_mask_dev: bad args");
    P2P_REQUIRE(elem_offset >= 0 && elem_offset % 8 == 0, "p2p_dropout_mask_dev: elem_offset must be a non-negative multiple of 8");
    long long blocks = (n / 8 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    dropout_mask_kernel<<<dim3((unsigned)blocks), 256, 0, (hipStream_t)stream>>>(mask, n, (unsigned long long)seed,
                                                                                (unsigned long long)salt, counter_dev, elem_offset / 8);
    return p2p_check_launch("p2p_dropout_mask_dev");
}
