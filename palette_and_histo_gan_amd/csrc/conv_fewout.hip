// Convolutions with very few output channels (1..4) and a fat input: the generator's last stride-1 convolution
// 36(+4 pad) -> 4 (networks.py:75-78), the discriminator's last convolution 64 -> 1 (networks.py:57) and the data gradient
// of the discriminator's first convolution towards the fake image, 64 -> 4 (networks.py:46, transposed form).
//
// The implicit GEMM (igemm.hip) pads those 1..4 output channels to a 32-row MFMA tile and gathers every input pixel once
// per tap (16x).  Here the contraction is turned round: with Z[q][tap][o] = sum_c in[q][c] * W[tap][o][c] -- a dense GEMM
// whose M side is (16 taps x o) = 16 or 64 rows, so nothing is padded away and every input pixel is read ONCE straight
// from HBM into the MFMA B operand -- the output is the shifted sum out[p][o] = sum_tap Z[p + tap][tap][o].  One workgroup
// owns a strip of TH output rows: it computes Z for the input rows the strip touches, parks only the (input row, kh)
// pairs the strip needs in LDS ([TH*4 kh-rows][RW pixels][4 kw][OS outputs] f32, <= 72 KB so two workgroups share a CU
// and overlap each other's load latency), then gathers the 16 taps per output pixel in a fixed order (deterministic),
// adds the bias, applies LeakyReLU and writes the output view.  The weights (<= 64 rows x 64 channels) live in registers.
//
// Forms (same argument meaning as p2p_igemm_edge, include/p2pgan.h): op G stride 1 and op P stride 2, ncols <= 4,
// 32 < cin_pad <= 64.  bf16: v_mfma_f32_32x32x16_bf16, f32: v_mfma_f32_32x32x2_f32 (exact products).
#include "p2p_common.hpp"

struct FoArgs {
    const char* in; long long in_img; int in_row; int in_ld;
    char* out; long long out_img; int out_row; int out_ld;
    const char* w; int w_rows;           // [16][w_rows][C] in T
    const float* bias; int act; float alpha;
    int LH, LW, lgLW;
    int C;                               // contraction channels (multiple of 8, <= 16 * KS)
    int ncols;
    int mode;                            // 0 = G stride 1, 1 = P stride 2
    int TH, strips_per_img, nstrips;
    int RH, RW;                          // input pixels the strip touches
};

// one lane's share of a 16-channel K step of an MFMA operand
template <typename T> struct Frag;
template <> struct Frag<bf16_t> {
    bf16x8 v;                            // channels 16 ks + 8 h + [0, 8)
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (bf16_t)0.f;
    }
    __device__ __forceinline__ void load(const char* base, int ks, int h, int C) {
        if (16 * ks + 8 * h < C) v = *(const bf16x8*)(base + (16 * ks + 8 * h) * 2);
        else zero();
    }
};
template <> struct Frag<float> {
    float v[8];                          // channels 16 ks + 2 m + h, m = 0..7
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = 0.f;
    }
    __device__ __forceinline__ void load(const char* base, int ks, int h, int C) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int c = 16 * ks + 2 * m + h;
            v[m] = c < C ? ((const float*)base)[c] : 0.f;
        }
    }
};

__device__ __forceinline__ void fo_mma(f32x16& acc, const Frag<bf16_t>& a, const Frag<bf16_t>& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc, 0, 0, 0);
}
__device__ __forceinline__ void fo_mma(f32x16& acc, const Frag<float>& a, const Frag<float>& b) {
#pragma unroll
    for (int m = 0; m < 8; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[m], b.v[m], acc, 0, 0, 0);
}

// LDS slot of Z[(kh-row idx), pixel rx][kw][0..OS): float index.  OS = 4: 16 floats per pixel, the kw slot is XOR-swizzled
// with bits 2..3 of the pixel so 16 consecutive pixels cover all 64 banks with b128 accesses; OS = 1: 4 floats per pixel.
template <int OS>
__device__ __forceinline__ int fo_zoff(int idx, int RW, int rx, int kw) {
    if (OS == 4) return ((idx * RW + rx) << 4) + ((kw ^ ((rx >> 2) & 3)) << 2);
    return ((idx * RW + rx) << 2) + kw;
}

template <typename T, int KS, int OS>
__global__ __launch_bounds__(512) void conv_fewout_kernel(FoArgs a) {
    constexpr int ESZ = sizeof(T);
    constexpr int NF = OS == 4 ? 2 : 1;          // 32-row MFMA fragments covering the 16 * OS rows (tap, o)
    extern __shared__ __attribute__((aligned(16))) float Z[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, r = lane & 31;
    const int C = a.C, RW = a.RW, TH = a.TH;

    // ---- weights: row (tap, o) of fragment f is lane r's A row ---------------------------------------------------------
    Frag<T> wf[NF][KS];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int rr = 32 * f + r;
        const int tap = OS == 4 ? rr >> 2 : rr, o = OS == 4 ? rr & 3 : 0;
        const bool live = tap < 16 && o < a.ncols && o < a.w_rows;
        const char* wrow = a.w + ((long long)(live ? tap : 0) * a.w_rows + (live ? o : 0)) * C * ESZ;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (live) wf[f][ks].load(wrow, ks, h, C);
            else wf[f][ks].zero();
        }
    }

    // persistent workgroups (two per CU): the weights stay in registers, the workgroup walks over its strips
    for (int strip = blockIdx.x; strip < a.nstrips; strip += gridDim.x) {
    const int n = strip / a.strips_per_img, y0 = (strip - n * a.strips_per_img) * a.TH;
    __syncthreads();                 // the previous strip's taps have been gathered: Z may be overwritten
    // ---- Z = W^T x for every input pixel of the strip; keep the (input row, kh) pairs some output row of the strip uses ---
    const int gpB = a.in_ld * ESZ;
    const char* in0 = a.in + ((long long)n * a.in_img + (long long)(y0 - 1) * a.in_row - 1) * gpB;
    const int PX = a.RH * RW, tiles = (PX + 31) >> 5;
    for (int tile = wave; tile < tiles; tile += 8) {
        const int p = tile * 32 + r;
        const bool valid = p < PX;
        const int ry = p / RW, rx = p - ry * RW;
        const char* pb = in0 + ((long long)ry * a.in_row + rx) * gpB;
        Frag<T> bf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (valid) bf[ks].load(pb, ks, h, C);
            else bf[ks].zero();
        }
        f32x16 acc[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) {
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[f][e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) fo_mma(acc[f], wf[f][ks], bf[ks]);
        }
        if (!valid) continue;
        // D layout: column = pixel (lane & 31), row = 32 f + 8 g + 4 h + k for register e = 4 g + k
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int g = 0; g < (OS == 4 ? 4 : 2); ++g) {
                const int kh = OS == 4 ? 2 * f + (g >> 1) : 2 * g + h;
                const int kw = OS == 4 ? 2 * (g & 1) + h : 0;
                int yy;
                if (a.mode == 0) yy = ry - kh;
                else yy = ry - 1 - ((((1 - (kh & 1)) + 1 - kh)) >> 1);
                if (yy < 0 || yy >= TH) continue;
                const f32x4 q = {acc[f][4 * g], acc[f][4 * g + 1], acc[f][4 * g + 2], acc[f][4 * g + 3]};
                *(f32x4*)(Z + fo_zoff<OS>(yy * 4 + kh, RW, rx, kw)) = q;
            }
    }
    __syncthreads();

    // ---- gather the taps --------------------------------------------------------------------------------------------------
    const int LW = a.LW, lg = a.lgLW;
    const int items = a.mode == 0 ? TH * LW : 4 * TH * LW;
    float bv[OS];
#pragma unroll
    for (int o = 0; o < OS; ++o) bv[o] = (a.bias && o < a.ncols) ? a.bias[o] : 0.f;
    for (int it = tid; it < items; it += 512) {
        float v[OS];
#pragma unroll
        for (int o = 0; o < OS; ++o) v[o] = 0.f;
        int Y, X;                                    // output pixel
        if (a.mode == 0) {
            const int yy = it >> lg, x = it & (LW - 1);
#pragma unroll
            for (int kh = 0; kh < 4; ++kh)
#pragma unroll
                for (int kw = 0; kw < 4; ++kw) {
                    const float* z = Z + fo_zoff<OS>(yy * 4 + kh, RW, x + kw, OS == 4 ? kw : 0);
                    if (OS == 4) { const f32x4 q = *(const f32x4*)z; v[0] += q[0]; v[1 % OS] += q[1]; v[2 % OS] += q[2]; v[3 % OS] += q[3]; }
                    else v[0] += z[kw];
                }
            Y = y0 + yy; X = x;
        } else {
            const int pw = it & 1, x = (it >> 1) & (LW - 1), ph = (it >> (1 + lg)) & 1, yy = it >> (2 + lg);
#pragma unroll
            for (int ia = 0; ia < 2; ++ia)
#pragma unroll
                for (int ib = 0; ib < 2; ++ib) {
                    const int kh = (1 - ph) + 2 * ia, kw = (1 - pw) + 2 * ib;
                    const int rx = x + 1 + ((pw + 1 - kw) >> 1);
                    const float* z = Z + fo_zoff<OS>(yy * 4 + kh, RW, rx, OS == 4 ? kw : 0);
                    if (OS == 4) { const f32x4 q = *(const f32x4*)z; v[0] += q[0]; v[1 % OS] += q[1]; v[2 % OS] += q[2]; v[3 % OS] += q[3]; }
                    else v[0] += z[kw];
                }
            Y = 2 * (y0 + yy) + ph; X = 2 * x + pw;
        }
#pragma unroll
        for (int o = 0; o < OS; ++o) {
            v[o] += bv[o];
            if (a.act == P2P_ACT_LEAKY) v[o] = v[o] > 0.f ? v[o] : a.alpha * v[o];
        }
        T* op = (T*)a.out + ((long long)n * a.out_img + (long long)Y * a.out_row + X) * a.out_ld;
        if (OS == 4 && a.ncols == 4 && (a.out_ld & 3) == 0) {       // base alignment checked on the host
            typedef __attribute__((__vector_size__(4 * sizeof(T)))) T vec4_t;
            vec4_t q;
#pragma unroll
            for (int o = 0; o < 4; ++o) q[o] = from_f32<T>(v[o % OS]);
            *(vec4_t*)op = q;
        } else {
#pragma unroll
            for (int o = 0; o < OS; ++o)
                if (o < a.ncols) op[o] = from_f32<T>(v[o]);
        }
    }
    }
}

struct FoPlan { int ok, KS, OS, TH, RH, RW; size_t shm; };

static FoPlan fo_plan(int op, int stride, int dtype, int N, int LH, int LW, int cin_pad, int ncols) {
    FoPlan p = {0, 0, 0, 0, 0, 0, 0};
    if (!((op == P2P_OP_G && stride == 1) || (op == P2P_OP_P && stride == 2))) return p;
    if (dtype != P2P_BF16 && dtype != P2P_F32) return p;
    if (ncols < 1 || ncols > 4 || N < 1) return p;
    if (cin_pad % 8 || cin_pad <= 32 || cin_pad > 64) return p;
    if ((LW & (LW - 1)) || LW < 8 || LW > 128 || LH < 1) return p;
    p.KS = cin_pad > 48 ? 4 : 3;
    p.OS = ncols == 1 ? 1 : 4;
    const int mode = op == P2P_OP_G ? 0 : 1;
    p.RW = mode == 0 ? LW + 3 : LW + 2;
    int TH = 8;
    while (TH > 1 && (TH > LH || LH % TH || (size_t)TH * 4 * p.RW * 4 * p.OS * 4 > 72 * 1024)) TH >>= 1;
    if (LH % TH || (size_t)TH * 4 * p.RW * 4 * p.OS * 4 > 150 * 1024) return p;
    p.TH = TH;
    p.RH = mode == 0 ? TH + 3 : TH + 2;
    p.shm = (size_t)TH * 4 * p.RW * 4 * p.OS * 4;
    if ((long long)N * (LH / TH) > 0x7fffffffLL) return p;
    p.ok = 1;
    return p;
}

extern "C" int p2p_conv_fewout_ok(int op, int stride, int dtype, int N, int LH, int LW, int cin_pad, int ncols) {
    return fo_plan(op, stride, dtype, N, LH, LW, cin_pad, ncols).ok;
}

extern "C" int p2p_conv_fewout(int op, int stride, int dtype, int N, int LH, int LW, int cin_pad, int ncols, int w_rows,
                               const p2p_tensor* in, const p2p_tensor* out, const void* w, const float* bias, int act,
                               float alpha, void* stream) {
    P2P_REQUIRE(in && out && in->ptr && out->ptr && w, "p2p_conv_fewout: null pointer");
    P2P_REQUIRE(w_rows >= 1, "p2p_conv_fewout: w_rows must be positive");
    const FoPlan p = fo_plan(op, stride, dtype, N, LH, LW, cin_pad, ncols);
    P2P_REQUIRE(p.ok, "p2p_conv_fewout: shape not supported (query p2p_conv_fewout_ok)");
    const int esz = dtype == P2P_BF16 ? 2 : 4;
    P2P_REQUIRE((in->ld * esz) % 16 == 0 && ((uintptr_t)in->ptr % 16) == 0 && ((uintptr_t)w % 16) == 0 && in->ld >= cin_pad,
                "p2p_conv_fewout: input pixels and weights must be 16-byte aligned");
    P2P_REQUIRE((out->ld & 3) != 0 || ((uintptr_t)out->ptr % (4 * esz)) == 0, "p2p_conv_fewout: output view must be aligned to 4 channels");
    FoArgs a;
    a.in = (const char*)in->ptr; a.in_img = in->img_stride; a.in_row = in->row_stride; a.in_ld = in->ld;
    a.out = (char*)out->ptr; a.out_img = out->img_stride; a.out_row = out->row_stride; a.out_ld = out->ld;
    a.w = (const char*)w; a.w_rows = w_rows;
    a.bias = bias; a.act = act; a.alpha = alpha;
    a.LH = LH; a.LW = LW;
    a.lgLW = 0;
    while ((1 << a.lgLW) < LW) ++a.lgLW;
    a.C = cin_pad; a.ncols = ncols;
    a.mode = op == P2P_OP_G ? 0 : 1;
    a.TH = p.TH; a.strips_per_img = LH / p.TH; a.nstrips = N * a.strips_per_img;
    a.RH = p.RH; a.RW = p.RW;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)(a.nstrips < 512 ? a.nstrips : 512));
#define FO_GO(KS_, OS_)                                                                                                \
    do {                                                                                                               \
        static bool done = false;                                                                                      \
        if (!done) done = p2p_allow_lds((const void*)conv_fewout_kernel<T, KS_, OS_>, 160 * 1024, "conv_fewout_kernel");  \
        conv_fewout_kernel<T, KS_, OS_><<<grid, dim3(512), p.shm, st>>>(a);                                            \
    } while (0)
    if (p.KS == 3 && p.OS == 4) { P2P_DISPATCH_DTYPE(dtype, FO_GO(3, 4)); }
    else if (p.KS == 4 && p.OS == 4) { P2P_DISPATCH_DTYPE(dtype, FO_GO(4, 4)); }
    else if (p.KS == 3 && p.OS == 1) { P2P_DISPATCH_DTYPE(dtype, FO_GO(3, 1)); }
    else { P2P_DISPATCH_DTYPE(dtype, FO_GO(4, 1)); }
#undef FO_GO
    return p2p_check_launch("p2p_conv_fewout");
}
