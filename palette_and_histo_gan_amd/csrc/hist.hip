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
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll 8
            for (int kk = 0; kk < HB / 2; ++kk) {
                int k = 2 * kk + (lane >> 5);
                const int row = rt * 32 + (lane & 31);
                float a = prod == 0 ? G[row][k] : G[k][row];
                float b = Rhs[k][pcol];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                int row = rt * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                float kq = Kown[row][pcol];
                float t = cval - hist_center(row);
                part0 += acc[e] * kq;
                part1 += acc[e] * (-2.0f * t * INV_SIGMA2) * kq * kq;
            }
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

extern "C" int p2p_rgbuv_hist_fwd(int dtype, int N, int H, int W, const p2p_tensor* img, float* hist, void* stream) {
    P2P_REQUIRE(N > 0 && H > 0 && W > 0 && img && img->ptr && hist, "p2p_rgbuv_hist_fwd: bad args");
    P2P_DISPATCH_DTYPE(dtype, (rgbuv_hist_fwd_kernel<T><<<dim3(N, 3), 256, 0, (hipStream_t)stream>>>(H, W, make_view(img), hist)));
    return p2p_check_launch("p2p_rgbuv_hist_fwd");
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
