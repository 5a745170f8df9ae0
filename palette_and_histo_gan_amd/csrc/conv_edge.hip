// LDS-resident convolution for the edge layers (networks.py:46-48,57,75-78: 4/8 -> 64, 36/33(+pad) -> 4, 64 -> 1
// channels and their data gradients).  These layers are HBM-bound (16-85 FLOP/B); the general implicit GEMM (igemm.hip)
// re-stages every input pixel once per tap and pads 1..4 output channels to a 128-row tile.  Here one workgroup
//   * keeps ALL weights of the layer in LDS for its whole life ([tap][out channel][in channel], <= 64 KB),
//   * walks over strips of output rows: brings the input strip (with its halo) into LDS once, then contracts all
//     taps out of LDS with MFMA (weights = A operand so a lane ends with 4 consecutive output channels of a pixel,
//     pixels = B operand read with per-lane addresses, so stride-2 gathers and sub-pixel phases are free),
//   * fuses bias + LeakyReLU and writes the output view directly.
// HBM traffic = input once + output once.  Forms (include/p2pgan.h): op G stride 1/2, op P stride 2 (4 phases),
// op P stride 1.  bf16: v_mfma_f32_32x32x16_bf16; f32: v_mfma_f32_32x32x2_f32 (exact).
#include "p2p_common.hpp"

struct CeArgs {
    const char* in; long long in_img; int in_row; int in_ld;
    char* out; long long out_img; int out_row; int out_ld;
    const char* w; int w_rows;           // [16][w_rows][C] in T
    const float* bias; int act; float alpha;
    int N, LH, LW, lgLW;
    int C, Cc;                           // contraction channels per tap (padded) and its 16-byte chunks
    int ncols;                           // real output channels
    int mode, S;                         // 0 = G (stride S), 1 = P stride 2, 2 = P stride 1
    int TH, strips_per_img, nstrips;
    int RH, RW;                          // strip size in pixels
    int PBp, WBp;                        // padded LDS strides: bytes per strip pixel / per weight row
    int w_lds_bytes;
};

template <typename T, int NT>
__global__ __launch_bounds__(512) void conv_edge_kernel(CeArgs a) {
    constexpr int ESZ = sizeof(T);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, r = lane & 31;
    char* wL = smem;                         // weights [16][32*NT][WBp]
    char* sL = smem + a.w_lds_bytes;         // strip   [RH*RW][PBp]
    const int NR = 32 * NT;
    const int Cc = a.Cc;
    const int n0 = blockIdx.y * NR;

    // ---- weights: once per workgroup, register-staged into padded rows (conflict-free ds_read_b128) ----------------
    {
        const int chunks = 16 * NR * Cc;
        for (int ci = tid; ci < chunks; ci += 512) {
            int cc = ci % Cc, row = (ci / Cc) % NR, tap = ci / (Cc * NR);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (n0 + row < a.w_rows)
                v = *(const f32x4*)(a.w + (((long long)tap * a.w_rows + n0 + row) * a.C) * ESZ + cc * 16);
            *(f32x4*)(wL + (tap * NR + row) * a.WBp + cc * 16) = v;
        }
    }
    const int gpB = a.in_ld * ESZ;           // HBM bytes per input pixel
    const int ntaps = a.mode == 1 ? 4 : 16;
    const int nphase = a.mode == 1 ? 4 : 1;
    const int tiles = (a.TH * a.LW) >> 5;    // 32-pixel MFMA column tiles per strip

    for (int strip = blockIdx.x; strip < a.nstrips; strip += gridDim.x) {
        const int n = strip / a.strips_per_img, y0 = (strip % a.strips_per_img) * a.TH;
        __syncthreads();                     // previous strip fully consumed (and weights written, first time)
        // ---- stage the input strip -----------------------------------------------------------------------------------
        {
            const int oy = a.mode == 0 ? a.S * y0 - 1 : (a.mode == 1 ? y0 - 1 : y0 - 2);
            const int ox = a.mode == 0 ? -1 : (a.mode == 1 ? -1 : -2);
            const int chunks = a.RH * a.RW * Cc;
            for (int ci = tid; ci < chunks; ci += 512) {
                int cc = ci % Cc, px = ci / Cc;
                int ry = px / a.RW, rx = px - ry * a.RW;
                f32x4 v = *(const f32x4*)(a.in + ((long long)n * a.in_img + (long long)(oy + ry) * a.in_row + (ox + rx)) * gpB + cc * 16);
                *(f32x4*)(sL + px * a.PBp + cc * 16) = v;
            }
        }
        __syncthreads();
        // ---- contract ------------------------------------------------------------------------------------------------
        for (int tile = wave; tile < tiles; tile += 8) {
            const int p = tile * 32 + r;                     // this lane's pixel (MFMA column) inside the strip
            const int yy = p >> a.lgLW, x = p & (a.LW - 1);
            for (int phs = 0; phs < nphase; ++phs) {
                const int ph = phs >> 1, pw = phs & 1;
                f32x16 acc[NT];
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
                const int kcs = ntaps * Cc;                  // 16-byte chunks of the whole contraction (even)
                if constexpr (ESZ == 2) {
                    for (int jm = 0; jm < kcs / 2; ++jm) {
                        const int kc = 2 * jm + h;
                        const int tq = kc / Cc, cc = kc - tq * Cc;
                        int ry, rx, widx;
                        if (a.mode == 0) { ry = a.S * yy + (tq >> 2); rx = a.S * x + (tq & 3); widx = tq; }
                        else if (a.mode == 1) {
                            int kh = (1 - ph) + 2 * (tq >> 1), kw = (1 - pw) + 2 * (tq & 1);
                            ry = yy + 1 + ((ph + 1 - kh) >> 1); rx = x + 1 + ((pw + 1 - kw) >> 1); widx = kh * 4 + kw;
                        } else { ry = yy + 3 - (tq >> 2); rx = x + 3 - (tq & 3); widx = tq; }
                        bf16x8 b = *(const bf16x8*)(sL + (ry * a.RW + rx) * a.PBp + cc * 16);
#pragma unroll
                        for (int j = 0; j < NT; ++j) {
                            bf16x8 w8 = *(const bf16x8*)(wL + (widx * NR + 32 * j + r) * a.WBp + cc * 16);
                            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w8, b, acc[j], 0, 0, 0);
                        }
                    }
                } else {
                    // f32: one channel per lane and MFMA (k = lane>>5): chunk kc holds 4 channels
                    for (int kc = 0; kc < kcs; ++kc) {
                        const int tq = kc / Cc, cc = kc - tq * Cc;
                        int ry, rx, widx;
                        if (a.mode == 0) { ry = a.S * yy + (tq >> 2); rx = a.S * x + (tq & 3); widx = tq; }
                        else if (a.mode == 1) {
                            int kh = (1 - ph) + 2 * (tq >> 1), kw = (1 - pw) + 2 * (tq & 1);
                            ry = yy + 1 + ((ph + 1 - kh) >> 1); rx = x + 1 + ((pw + 1 - kw) >> 1); widx = kh * 4 + kw;
                        } else { ry = yy + 3 - (tq >> 2); rx = x + 3 - (tq & 3); widx = tq; }
                        const float* bp = (const float*)(sL + (ry * a.RW + rx) * a.PBp + cc * 16);
#pragma unroll
                        for (int e2 = 0; e2 < 2; ++e2) {
                            float b = bp[2 * e2 + h];
#pragma unroll
                            for (int j = 0; j < NT; ++j) {
                                float wv = ((const float*)(wL + (widx * NR + 32 * j + r) * a.WBp + cc * 16))[2 * e2 + h];
                                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv, b, acc[j], 0, 0, 0);
                            }
                        }
                    }
                }
                // ---- epilogue: D[row = channel][col = pixel]; 4 consecutive channels per register group ------------------
                const int Y = y0 + yy;
                long long opix;
                if (a.mode == 1) opix = (long long)n * a.out_img + (long long)(2 * Y + ph) * a.out_row + (2 * x + pw);
                else opix = (long long)n * a.out_img + (long long)Y * a.out_row + x;
                T* op = (T*)a.out + opix * a.out_ld;
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int col = n0 + 32 * j + 8 * g + 4 * h;
                        if (col >= a.ncols) continue;
                        float v[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            v[k] = acc[j][4 * g + k];
                            if (a.bias && col + k < a.ncols) v[k] += a.bias[col + k];
                            if (a.act == P2P_ACT_LEAKY) v[k] = v[k] > 0.f ? v[k] : a.alpha * v[k];
                        }
                        if (col + 3 < a.ncols && (a.out_ld & 3) == 0) {      // base alignment checked on the host
                            typedef __attribute__((__vector_size__(4 * sizeof(T)))) T vec4_t;
                            vec4_t q;
#pragma unroll
                            for (int k = 0; k < 4; ++k) q[k] = from_f32<T>(v[k]);
                            *(vec4_t*)(op + col) = q;
                        } else {
#pragma unroll
                            for (int k = 0; k < 4; ++k)
                                if (col + k < a.ncols) op[col + k] = from_f32<T>(v[k]);
                        }
                    }
            }
        }
    }
}

struct CePlan { int ok, NT, nwin, TH, RH, RW, PBp, WBp, blocks, w_lds; size_t shm; };

static CePlan ce_plan(int op, int stride, int dtype, int N, int LH, int LW, int cin_pad, int ncols) {
    CePlan p = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int esz = dtype == P2P_BF16 ? 2 : 4;
    if ((LW & (LW - 1)) || LW < 16 || LW > 64) return p;
    if ((cin_pad * esz) % 16) return p;
    const int mode = op == P2P_OP_G ? 0 : (stride == 2 ? 1 : 2);
    const int nt_all = (ncols + 31) / 32;
    p.NT = nt_all >= 2 ? 2 : 1;
    p.nwin = (nt_all + p.NT - 1) / p.NT;
    if (p.nwin > 2) return p;
    const int pb = cin_pad * esz;
    p.PBp = (pb % 64 == 0) ? pb + 16 : pb;            // strides that are multiples of 64 B would alias LDS banks
    p.WBp = p.PBp;
    p.w_lds = 16 * 32 * p.NT * p.WBp;
    if (p.w_lds > 96 * 1024) return p;
    int TH = 512 / LW;
    if (TH > 8) TH = 8;
    if (TH > LH) TH = LH;
    for (;; TH >>= 1) {
        if (TH < 1 || LH % TH || (TH * LW) % 32) return p;
        p.RH = mode == 0 ? stride * TH + 3 : (mode == 1 ? TH + 2 : TH + 3);
        p.RW = mode == 0 ? stride * LW + 3 : (mode == 1 ? LW + 2 : LW + 3);
        p.shm = (size_t)p.w_lds + (size_t)p.RH * p.RW * p.PBp + 256;
        if (p.shm <= 150 * 1024) break;
        if (TH == 1) return p;
    }
    p.TH = TH;
    long long strips = (long long)N * (LH / TH);
    p.blocks = (int)(strips < 256 ? strips : 256);
    p.ok = 1;
    return p;
}

extern "C" int p2p_conv_edge_ok(int op, int stride, int dtype, int N, int LH, int LW, int cin_pad, int ncols) {
    return ce_plan(op, stride, dtype, N, LH, LW, cin_pad, ncols).ok;
}

extern "C" int p2p_conv_edge(int op, int stride, int dtype, int N, int LH, int LW, int cin_pad, int ncols, int w_rows,
                             const p2p_tensor* in, const p2p_tensor* out, const void* w, const float* bias, int act,
                             float alpha, void* stream) {
    P2P_REQUIRE(op == P2P_OP_G || op == P2P_OP_P, "p2p_conv_edge: op must be G or P");
    P2P_REQUIRE(stride == 1 || stride == 2, "p2p_conv_edge: stride must be 1 or 2");
    P2P_REQUIRE(in && out && in->ptr && out->ptr && w, "p2p_conv_edge: null pointer");
    const CePlan p = ce_plan(op, stride, dtype, N, LH, LW, cin_pad, ncols);
    P2P_REQUIRE(p.ok, "p2p_conv_edge: shape not supported (query p2p_conv_edge_ok)");
    const int esz = dtype == P2P_BF16 ? 2 : 4;
    P2P_REQUIRE((in->ld * esz) % 16 == 0 && ((uintptr_t)in->ptr % 16) == 0 && ((uintptr_t)w % 16) == 0 && in->ld >= cin_pad,
                "p2p_conv_edge: input pixels and weights must be 16-byte aligned");
    P2P_REQUIRE((out->ld & 3) != 0 || ((uintptr_t)out->ptr % (4 * esz)) == 0, "p2p_conv_edge: output view must be aligned to 4 channels");
    CeArgs a;
    a.in = (const char*)in->ptr; a.in_img = in->img_stride; a.in_row = in->row_stride; a.in_ld = in->ld;
    a.out = (char*)out->ptr; a.out_img = out->img_stride; a.out_row = out->row_stride; a.out_ld = out->ld;
    a.w = (const char*)w; a.w_rows = w_rows;
    a.bias = bias; a.act = act; a.alpha = alpha;
    a.N = N; a.LH = LH; a.LW = LW;
    a.lgLW = 0;
    while ((1 << a.lgLW) < LW) ++a.lgLW;
    a.C = cin_pad; a.Cc = cin_pad * esz / 16;
    a.ncols = ncols;
    a.mode = op == P2P_OP_G ? 0 : (stride == 2 ? 1 : 2);
    a.S = stride;
    a.TH = p.TH; a.strips_per_img = LH / p.TH; a.nstrips = N * a.strips_per_img;
    a.RH = p.RH; a.RW = p.RW; a.PBp = p.PBp; a.WBp = p.WBp; a.w_lds_bytes = p.w_lds;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(p.blocks, p.nwin);
#define CE_GO(NT_)                                                                                                     \
    do {                                                                                                               \
        static bool done = false;                                                                                      \
        if (!done) {                                                                                                   \
            (void)hipFuncSetAttribute((const void*)conv_edge_kernel<T, NT_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            done = true;                                                                                               \
        }                                                                                                              \
        conv_edge_kernel<T, NT_><<<grid, dim3(512), p.shm, st>>>(a);                                                   \
    } while (0)
    if (p.NT == 1) { P2P_DISPATCH_DTYPE(dtype, CE_GO(1)); }
    else { P2P_DISPATCH_DTYPE(dtype, CE_GO(2)); }
#undef CE_GO
    return p2p_check_launch("p2p_conv_edge");
}
