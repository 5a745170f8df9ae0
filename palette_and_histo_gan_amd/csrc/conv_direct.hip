// Direct 4x4 convolution (no MFMA): edge layers with tiny channel counts and the on-device cross-check
// of the implicit-GEMM kernels.  Reference ops: Conv2D / Conv2DTranspose and their tape gradients at
// networks.py:10-16,26-27,46-48,75-78.  Forms G / P / W are defined in include/p2pgan.h.
#include "p2p_common.hpp"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void p2p_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static thread_local char g_attr_err[256] = "";

bool p2p_allow_lds(const void* kernel, int bytes, const char* name) {
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) return true;
    (void)hipGetLastError();
    snprintf(g_attr_err, sizeof(g_attr_err), "%s: %d bytes of dynamic LDS refused (%s)", name, bytes, hipGetErrorString(e));
    return false;
}

int p2p_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (g_attr_err[0]) {        // the launch that followed a refused attribute: say why it failed
        p2p_set_error("%s: %s", what, g_attr_err);
        g_attr_err[0] = 0;
        return e != hipSuccess ? (int)e : -2;
    }
    if (e != hipSuccess) {
        p2p_set_error("%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

extern "C" const char* p2p_last_error(void) { return g_err; }
extern "C" int p2p_version(void) { return 2; }
extern "C" int p2p_view_halo_pixels(void) { return 2; }

// ---- op G: lo[m][d] = bias[d] + sum_{t,g} hi[pix(m,t)][g] * w[t][g][d] ------------------------------
template <typename T>
__global__ void conv_direct_G(int stride, int N, int LH, int LW, int Cg, int Cd, TView hi, TView lo,
                              const T* __restrict__ w, const float* __restrict__ bias) {
    long long total = (long long)N * LH * LW * Cd;
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    int d = (int)(idx % Cd);
    long long m = idx / Cd;
    int x = (int)(m % LW);
    int y = (int)((m / LW) % LH);
    int n = (int)(m / ((long long)LW * LH));
    int HH = stride * LH, HW = stride * LW;
    float acc = bias ? bias[d] : 0.f;
    for (int kh = 0; kh < 4; ++kh) {
        int ih = stride * y + kh - 1;
        if (ih < 0 || ih >= HH) continue;
        for (int kw = 0; kw < 4; ++kw) {
            int iw = stride * x + kw - 1;
            if (iw < 0 || iw >= HW) continue;
            const T* hp = (const T*)hi.ptr + hi.off(n, ih, iw);
            const T* wp = w + (long long)(kh * 4 + kw) * Cg * Cd + d;
            for (int g = 0; g < Cg; ++g) acc += to_f32(hp[g]) * to_f32(wp[(long long)g * Cd]);
        }
    }
    ((T*)lo.ptr)[lo.off(n, y, x) + d] = from_f32<T>(acc);
}

// ---- op P: hi[n,Y,X,g] = sum_{t: Y=s*y+kh-1, X=s*x+kw-1} sum_d lo[n,y,x,d] * w[t][g][d] --------------
template <typename T>
__global__ void conv_direct_P(int stride, int N, int LH, int LW, int Cg, int Cd, TView hi, TView lo,
                              const T* __restrict__ w) {
    int HH = stride * LH, HW = stride * LW;
    long long total = (long long)N * HH * HW * Cg;
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    int g = (int)(idx % Cg);
    long long p = idx / Cg;
    int X = (int)(p % HW);
    int Y = (int)((p / HW) % HH);
    int n = (int)(p / ((long long)HW * HH));
    float acc = 0.f;
    for (int kh = 0; kh < 4; ++kh) {
        int ty = Y + 1 - kh;
        if (ty < 0 || (ty % stride) != 0) continue;
        int y = ty / stride;
        if (y >= LH) continue;
        for (int kw = 0; kw < 4; ++kw) {
            int tx = X + 1 - kw;
            if (tx < 0 || (tx % stride) != 0) continue;
            int x = tx / stride;
            if (x >= LW) continue;
            const T* lp = (const T*)lo.ptr + lo.off(n, y, x);
            const T* wp = w + ((long long)(kh * 4 + kw) * Cg + g) * Cd;
            for (int d = 0; d < Cd; ++d) acc += to_f32(lp[d]) * to_f32(wp[d]);
        }
    }
    ((T*)hi.ptr)[hi.off(n, Y, X) + g] = from_f32<T>(acc);
}

// ---- op W: dw[t][g][d] = sum over ALL pixels in pixel order, one thread per output (cross-check path: slow at large
// batches, bit-reproducible, no float atomics) -----------------------------------------------------------------------
template <typename T>
__global__ void conv_direct_W(int stride, int N, int LH, int LW, int Cg, int Cd, TView hi, TView lo,
                              float* __restrict__ dw, int chunk) {
    int total = 16 * Cg * Cd;
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    int d = idx % Cd;
    int g = (idx / Cd) % Cg;
    int t = idx / (Cd * Cg);
    int kh = t >> 2, kw = t & 3;
    int HH = stride * LH, HW = stride * LW;
    long long M = (long long)N * LH * LW;
    long long m0 = (long long)blockIdx.y * chunk;
    long long m1 = m0 + chunk < M ? m0 + chunk : M;
    float acc = 0.f;
    for (long long m = m0; m < m1; ++m) {
        int x = (int)(m % LW);
        int y = (int)((m / LW) % LH);
        int n = (int)(m / ((long long)LW * LH));
        int ih = stride * y + kh - 1, iw = stride * x + kw - 1;
        if (ih < 0 || ih >= HH || iw < 0 || iw >= HW) continue;
        acc += to_f32(((const T*)hi.ptr)[hi.off(n, ih, iw) + g]) * to_f32(((const T*)lo.ptr)[lo.off(n, y, x) + d]);
    }
    dw[idx] = acc;
}

// sum over a chunk of pixels of lo[m][d]: threads = (pixel lane) x (channel), coalesced over channels, pixel lanes folded
// through LDS; each workgroup writes its partial row (`part`), or -- launched as ONE workgroup over all pixels -- the sum itself
template <typename T>
__global__ void view_colsum(int N, int LH, int LW, int Cd, TView lo, float* __restrict__ out, int chunk, float* __restrict__ part = nullptr) {
    __shared__ float red[256];
    const int M = N * LH * LW;                     // < 2^31 pixels
    const int m0 = blockIdx.x * chunk;
    const int m1 = m0 + chunk < M ? m0 + chunk : M;
    const int cpt = Cd < 256 ? Cd : 256;          // channels handled at once
    const int PL = 256 / cpt;                      // pixel lanes
    const int c = threadIdx.x % cpt, pl = threadIdx.x / cpt;
    for (int c0 = 0; c0 < Cd; c0 += cpt) {
        float acc = 0.f;
        if (pl < PL && c0 + c < Cd)
            for (int m = m0 + pl; m < m1; m += PL) {
                int x = m % LW;
                int q = m / LW;
                int y = q % LH;
                int n = q / LH;
                acc += to_f32(((const T*)lo.ptr)[lo.off(n, y, x) + c0 + c]);
            }
        __syncthreads();
        red[threadIdx.x] = (pl < PL) ? acc : 0.f;
        __syncthreads();
        if (pl == 0 && c0 + c < Cd) {
            float s2 = 0.f;
            for (int i = 0; i < PL; ++i) s2 += red[i * cpt + c];
            if (part) part[(long long)blockIdx.x * Cd + c0 + c] = s2;        // deterministic form: colsum_partials_kernel adds the workgroups in order
            else out[c0 + c] = s2;                                         // single-workgroup launch (direct cross-check path)
        }
    }
}

// Wide-channel form (Cd % 8 == 0, 16-byte aligned view): 16-byte loads, thread = (8-channel vector, pixel lane), a
// workgroup covers `chunk` pixels; partial sums are combined through LDS and leave as one partial row per workgroup.
template <typename T>
__global__ __launch_bounds__(256) void view_colsum_vec(int N, int LH, int LW, int Cd, TView lo, float* __restrict__ part, int chunk) {
    constexpr int VN = 16 / sizeof(T);
    typedef __attribute__((__vector_size__(16))) T vec_t;
    __shared__ float red[256 * 8];
    const int M = N * LH * LW;
    const int m0 = blockIdx.x * chunk, m1 = m0 + chunk < M ? m0 + chunk : M;
    const int nvec = Cd / VN;                        // vectors per pixel (<= 256)
    const int PL = 256 / nvec;                       // pixel lanes
    const int cv = threadIdx.x % nvec, pl = threadIdx.x / nvec;
    float acc[VN];
#pragma unroll
    for (int k = 0; k < VN; ++k) acc[k] = 0.f;
    if (pl < PL)
        for (int m = m0 + pl; m < m1; m += PL) {
            const int x = m % LW, q = m / LW, y = q % LH, n = q / LH;
            const vec_t r = *(const vec_t*)((const T*)lo.ptr + lo.off(n, y, x) + cv * VN);
#pragma unroll
            for (int k = 0; k < VN; ++k) acc[k] += to_f32((T)r[k]);
        }
#pragma unroll
    for (int k = 0; k < VN; ++k) red[threadIdx.x * VN + k] = pl < PL ? acc[k] : 0.f;
    __syncthreads();
    for (int c = threadIdx.x; c < Cd; c += 256) {
        const int v = c / VN, k = c - v * VN;
        float s2 = 0.f;
        for (int i = 0; i < PL; ++i) s2 += red[(i * nvec + v) * VN + k];
        part[(long long)blockIdx.x * Cd + c] = s2;
    }
}

// Narrow form (8-channel pixels, Cd <= 8: the bias gradients of the two heads): one whole pixel per lane and step
// (16 / 32 bytes), 8 running sums per lane, one LDS tree and one partial row per workgroup.
template <typename T>
__global__ __launch_bounds__(256) void view_colsum_px8(int N, int LH, int LW, int Cd, TView lo, float* __restrict__ part, int chunk) {
    __shared__ float red[8][256];
    const int M = N * LH * LW;
    const int m0 = blockIdx.x * chunk, m1 = m0 + chunk < M ? m0 + chunk : M;
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    // four pixels per trip, their loads issued together (a 2048-pixel chunk is two trips: the launch is latency-bound)
    for (int mb = m0 + threadIdx.x; mb < m1; mb += 1024) {
        const T* p[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int m = mb + 256 * u < m1 ? mb + 256 * u : mb;
            const int x = m % LW, q = m / LW, y = q % LH, n = q / LH;
            p[u] = (const T*)lo.ptr + lo.off(n, y, x);
        }
        if (sizeof(T) == 2) {
            bf16x8 r[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) r[u] = *(const bf16x8*)p[u];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (mb + 256 * u < m1)
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc[k] += (float)r[u][k];
        } else {
            f32x4 r0[4], r1[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { r0[u] = *(const f32x4*)p[u]; r1[u] = *(const f32x4*)((const float*)p[u] + 4); }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (mb + 256 * u < m1)
#pragma unroll
                    for (int k = 0; k < 4; ++k) { acc[k] += r0[u][k]; acc[4 + k] += r1[u][k]; }
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) red[k][threadIdx.x] = acc[k];
    __syncthreads();
    for (int s2 = 128; s2 > 0; s2 >>= 1) {
        if (threadIdx.x < s2)
#pragma unroll
            for (int k = 0; k < 8; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + s2];
        __syncthreads();
    }
    if (threadIdx.x < Cd) part[(long long)blockIdx.x * Cd + threadIdx.x] = red[threadIdx.x][0];
}

// out[c] = sum_b part[b][c], b in workgroup order (one workgroup per channel): bit-reproducible, no float atomics
__global__ __launch_bounds__(256) void colsum_partials_kernel(const float* __restrict__ part, int nb, int C, float* __restrict__ out) {
    __shared__ float red[16];
    const int c = blockIdx.x;
    float s = 0.f;
    for (int b = threadIdx.x; b < nb; b += 256) s += part[(long long)b * C + c];
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[c] = s;
}

// out[d] = sum over all pixels of v[m][d]  (bias gradients of the two stride-1 heads, networks.py:47-48,75-78)
// pixels per workgroup of the form p2p_view_colsum picks (shared with p2p_view_colsum_workspace_bytes)
static bool view_colsum_is_px8(int C, const p2p_tensor* v) { return C <= 8 && v->ld == 8 && ((uintptr_t)v->ptr % 16) == 0; }
static int view_colsum_chunk(int dtype, int C, const p2p_tensor* v) {
    if (view_colsum_is_px8(C, v)) return 2048;     // 512 workgroups at 256 x 64 x 64
    const int vn = dtype == P2P_BF16 ? 8 : 4;
    if (C >= 64 && C % vn == 0 && C / vn <= 256 && 256 % (C / vn) == 0 && v->ld % vn == 0 && ((uintptr_t)v->ptr % 16) == 0) return 1024;
    return C >= 64 ? 256 : 2048;                  // wide pixels: few pixel lanes per workgroup -> more workgroups
}

extern "C" long long p2p_view_colsum_workspace_bytes(int dtype, int N, int H, int W, int C, const p2p_tensor* v) {
    if (!v || N <= 0 || H <= 0 || W <= 0 || C <= 0) return 0;
    const long long M = (long long)N * H * W;
    const int chunk = view_colsum_chunk(dtype, C, v);
    return ((M + chunk - 1) / chunk) * C * (long long)sizeof(float);
}

extern "C" int p2p_view_colsum(int dtype, int N, int H, int W, int C, const p2p_tensor* v, float* out, float* workspace,
                               void* stream) {
    P2P_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && v && v->ptr && out && workspace, "p2p_view_colsum: bad args");
    hipStream_t st = (hipStream_t)stream;
    long long M = (long long)N * H * W;
    P2P_REQUIRE(M < (1LL << 31), "p2p_view_colsum: too many pixels");
    const int chunk = view_colsum_chunk(dtype, C, v);
    const unsigned nb = (unsigned)((M + chunk - 1) / chunk);
    if (view_colsum_is_px8(C, v)) {
        P2P_DISPATCH_DTYPE(dtype, (view_colsum_px8<T><<<dim3(nb), 256, 0, st>>>(N, H, W, C, make_view(v), workspace, chunk)));
    } else if (chunk == 1024) {
        P2P_DISPATCH_DTYPE(dtype, (view_colsum_vec<T><<<dim3(nb), 256, 0, st>>>(N, H, W, C, make_view(v), workspace, chunk)));
    } else {
        P2P_DISPATCH_DTYPE(dtype, (view_colsum<T><<<dim3(nb), 256, 0, st>>>(N, H, W, C, make_view(v), out, chunk, workspace)));
    }
    colsum_partials_kernel<<<dim3((unsigned)C), 256, 0, st>>>(workspace, (int)nb, C, out);
    return p2p_check_launch("p2p_view_colsum");
}

template <typename T>
static int conv_direct_impl(int op, int stride, int N, int LH, int LW, int Cg, int Cd, TView hi, TView lo,
                            const void* w, const float* bias, float* dw, float* dbias, hipStream_t st) {
    const int TB = 256;
    if (op == P2P_OP_G) {
        long long total = (long long)N * LH * LW * Cd;
        conv_direct_G<T><<<dim3((unsigned)((total + TB - 1) / TB)), TB, 0, st>>>(stride, N, LH, LW, Cg, Cd, hi, lo,
                                                                               (const T*)w, bias);
    } else if (op == P2P_OP_P) {
        long long total = (long long)N * stride * LH * stride * LW * Cg;
        conv_direct_P<T><<<dim3((unsigned)((total + TB - 1) / TB)), TB, 0, st>>>(stride, N, LH, LW, Cg, Cd, hi, lo,
                                                                               (const T*)w);
    } else {
        long long M = (long long)N * LH * LW;
        int total = 16 * Cg * Cd;
        P2P_REQUIRE(M < (1LL << 31), "p2p_conv_direct: too many pixels");
        dim3 grid((total + TB - 1) / TB, 1);
        conv_direct_W<T><<<grid, TB, 0, st>>>(stride, N, LH, LW, Cg, Cd, hi, lo, dw, (int)M);
        if (dbias) view_colsum<T><<<dim3(1), TB, 0, st>>>(N, LH, LW, Cd, lo, dbias, (int)M);
    }
    return p2p_check_launch("p2p_conv_direct");
}

extern "C" int p2p_conv_direct(int op, int stride, int dtype, int N, int LH, int LW, int Cg, int Cd,
                               const p2p_tensor* hi, const p2p_tensor* lo, const void* w, const float* bias,
                               float* dw, float* dbias, void* stream) {
    P2P_REQUIRE(op >= 0 && op <= 2, "p2p_conv_direct: bad op %d", op);
    P2P_REQUIRE(stride == 1 || stride == 2, "p2p_conv_direct: stride must be 1 or 2, got %d", stride);
    P2P_REQUIRE(N > 0 && LH > 0 && LW > 0 && Cg > 0 && Cd > 0, "p2p_conv_direct: bad shape");
    P2P_REQUIRE(hi && lo && hi->ptr && lo->ptr, "p2p_conv_direct: null tensor");
    P2P_REQUIRE(op == P2P_OP_W ? dw != nullptr : w != nullptr, "p2p_conv_direct: null weight pointer");
    P2P_DISPATCH_DTYPE(dtype, return conv_direct_impl<T>(op, stride, N, LH, LW, Cg, Cd, make_view(hi), make_view(lo), w,
                                                         bias, dw, dbias, (hipStream_t)stream));
}
