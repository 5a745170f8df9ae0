// Convolutions with very few INPUT channels (1..8, stored as one 16-byte pixel of 8 bf16): the first convolution of the
// generator and of the discriminator, 4/8 -> 64 stride 2 (networks.py:10-16 down1, :45-46), and the data gradients of the
// two stride-1 heads, 4(+4) -> 32 and 1(+7) -> 64 (networks.py:57,75-78 transposed).  They write 8..64x more bytes than
// they read and the whole contraction is K = 16 taps x 8 channels = 128: the general implicit GEMM (igemm.hip) spends its
// time in tile prologues and epilogues (two K-blocks per tile).  Here
//   * the weights ([16 taps][<= 64 outputs][8]) live in registers as the MFMA A operand for the life of the workgroup,
//   * a strip of input rows (16 bytes per pixel) is staged in LDS once; one K step of v_mfma_f32_32x32x16_bf16 is exactly
//     two taps (lane half h takes tap 2 ks + h), so a B fragment is ONE ds_read_b128 of the pixel the tap points at --
//     the stride-2 gather and the flipped taps of the transposed form are per-lane addresses,
//   * the 32-pixel x ncols result tile is transposed through a per-wave LDS patch and leaves as whole 16-byte chunks of
//     each pixel's channel run (bias + LeakyReLU fused).
// Forms (same arguments as p2p_igemm_edge, include/p2pgan.h): op G stride 1/2 and op P stride 1, bf16, cin_pad == 8,
// ncols <= 64.  Everything else stays on p2p_igemm_edge.
#include "p2p_common.hpp"

struct FiArgs {
    const char* in; long long in_img; int in_row;          // 16-byte pixels (in_ld == 8)
    char* out; long long out_img; int out_row; int out_ld;
    const char* w; int w_rows;                             // [16][w_rows][8] bf16
    const float* bias; int act; float alpha;
    const char* gate; long long gate_img; int gate_row; int gate_ld;     // optional: out *= LeakyReLU'(gate) (gate > 0 ? 1 : alpha)
    int LH, LW, lgLW;
    int ncols;
    int mode, S;                                           // 0 = G (stride S), 2 = P stride 1
    int TH, strips_per_img, nstrips;
    int RH, RW;                                            // input pixels the strip touches
    int vec;                                               // output view allows 16-byte stores
};

template <int NT>
__global__ __launch_bounds__(256 * NT) void conv_fewin_kernel(FiArgs a) {
    // 4 * NT waves: wave = 4 f + wv owns output channels [32 f, 32 f + 32) of the 32-pixel tiles wv, wv + 4, ...
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int f = wave >> 2, wv = wave & 3;
    const int h = lane >> 5, r = lane & 31;
    const int RW = a.RW, S = a.S;
    constexpr int PROW = 64 + 16;                           // patch row: 32 bf16 channels + 16 B pad
    constexpr int PATCH = 32 * PROW;                        // per-wave transpose patch
    char* sL = smem;                                        // strip [RH*RW] x 16 B
    float* bL = (float*)(smem + ((a.RH * RW * 16 + 15) & ~15));      // bias [NT*32]
    char* pL = (char*)(bL + NT * 32) + wave * PATCH;

    // ---- weights -> registers: K step ks: row o = 32 f + r, taps 2 ks + h ---------------------------------------------
    bf16x8 wf[8];
    {
        const int o = 32 * f + r;
        const bool live = o < a.ncols && o < a.w_rows;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int tap = 2 * ks + h;
            bf16x8 v;
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = (bf16_t)0.f;
            if (live) v = *(const bf16x8*)(a.w + ((long long)(tap * a.w_rows + o) * 8) * 2);
            wf[ks] = v;
        }
    }
    if (tid < NT * 32) bL[tid] = (a.bias && tid < a.ncols) ? a.bias[tid] : 0.f;

    // persistent workgroups: the weights are fetched once, then the workgroup walks over its strips
    for (int strip = blockIdx.x; strip < a.nstrips; strip += gridDim.x) {
    const int n = strip / a.strips_per_img, y0 = (strip - n * a.strips_per_img) * a.TH;
    __syncthreads();                     // every wave is done with the previous strip (and the bias is in LDS)
    // ---- stage the strip ----------------------------------------------------------------------------------------------------
    {
        const int oy = a.mode == 0 ? S * y0 - 1 : y0 - 2;
        const int ox = a.mode == 0 ? -1 : -2;
        const int npx = a.RH * RW;
        const char* base = a.in + ((long long)n * a.in_img + (long long)oy * a.in_row + ox) * 16;
        for (int p = tid; p < npx; p += 256 * NT) {
            const int ry = p / RW, rx = p - ry * RW;
            *(f32x4*)(sL + p * 16) = *(const f32x4*)(base + ((long long)ry * a.in_row + rx) * 16);
        }
    }
    __syncthreads();

    // ---- contract and store -------------------------------------------------------------------------------------------------
    const int tiles = (a.TH * a.LW) >> 5;
    for (int tile = wv; tile < tiles; tile += 4) {
        const int p = tile * 32 + r;
        const int yy = p >> a.lgLW, x = p & (a.LW - 1);
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int tap = 2 * ks + h, kh = tap >> 2, kw = tap & 3;
            int ry, rx;
            if (a.mode == 0) { ry = S * yy + kh; rx = S * x + kw; }
            else { ry = yy + 3 - kh; rx = x + 3 - kw; }
            const bf16x8 b = *(const bf16x8*)(sL + (ry * RW + rx) * 16);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], b, acc, 0, 0, 0);
        }
        // D: column = pixel r, rows = channels 32 f + 8 g + 4 h + k -> patch[pixel][channel - 32 f] (bf16)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            typedef __attribute__((__vector_size__(4 * sizeof(bf16_t)))) bf16_t bf16x4;
            const f32x4 b4 = *(const f32x4*)(bL + 32 * f + 8 * g + 4 * h);
            bf16x4 q;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float v = acc[4 * g + k] + b4[k];
                if (a.act == P2P_ACT_LEAKY) v = v > 0.f ? v : a.alpha * v;
                q[k] = (bf16_t)v;
            }
            *(bf16x4*)(pL + r * PROW + (8 * g + 4 * h) * 2) = q;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        // 16-byte chunks: 4 per pixel, 16 pixels per pass
        const int ch = lane & 3, pq = lane >> 2;
        const int c0 = 32 * f + ch * 8;
#pragma unroll
        for (int ps = 0; ps < 32; ps += 16) {
            const int pix = ps + pq;
            const int pp = tile * 32 + pix;
            const int py = pp >> a.lgLW, px = pp & (a.LW - 1);
            if (c0 < a.ncols) {
                f32x4 v = *(const f32x4*)(pL + pix * PROW + ch * 16);
                if (a.gate) {
                    // backward of the LeakyReLU that produced `gate` (the layer's activation output), applied to the ROUNDED
                    // gradient exactly as p2p_act_bwd would apply it to the stored tensor: the results are bit-identical to
                    // convolution + p2p_act_bwd, without writing and re-reading the gradient (host checked ncols % 8 == 0)
                    const bf16x8 gt = *(const bf16x8*)((const bf16_t*)a.gate + ((long long)n * a.gate_img + (long long)(y0 + py) * a.gate_row + px) * a.gate_ld + c0);
                    bf16x8 e = *(const bf16x8*)&v;
#pragma unroll
                    for (int k = 0; k < 8; ++k) e[k] = (bf16_t)((float)e[k] * ((float)gt[k] > 0.f ? 1.f : a.alpha));
                    v = *(const f32x4*)&e;
                }
                bf16_t* op = (bf16_t*)a.out + ((long long)n * a.out_img + (long long)(y0 + py) * a.out_row + px) * a.out_ld + c0;
                if (a.vec && c0 + 8 <= a.ncols) *(f32x4*)op = v;
                else {
                    const bf16_t* e = (const bf16_t*)&v;
                    for (int k = 0; k < 8; ++k)
                        if (c0 + k < a.ncols) op[k] = e[k];
                }
            }
        }
        __builtin_amdgcn_wave_barrier();       // the patch is rewritten by the next tile
    }
    }
}

struct FiPlan { int ok, NT, TH, RH, RW; size_t shm; };

static FiPlan fi_plan(int op, int stride, int dtype, int N, int LH, int LW, int cin_pad, int ncols) {
    FiPlan p = {0, 0, 0, 0, 0, 0};
    if (dtype != P2P_BF16 || cin_pad != 8) return p;
    if (!((op == P2P_OP_G && (stride == 1 || stride == 2)) || (op == P2P_OP_P && stride == 1))) return p;
    if (ncols < 1 || ncols > 64 || N < 1) return p;
    if ((LW & (LW - 1)) || LW < 8 || LW > 128 || LH < 1) return p;
    p.NT = ncols > 32 ? 2 : 1;
    const int s = op == P2P_OP_G ? stride : 1;
    int TH = 8;
    while (TH > 1 && (TH > LH || LH % TH)) TH >>= 1;
    if (LH % TH || (TH * LW) % 32) return p;
    p.TH = TH;
    p.RH = s * TH + 3;
    p.RW = s * LW + 3;
    p.shm = (((size_t)p.RH * p.RW * 16 + 15) & ~(size_t)15) + (size_t)p.NT * 32 * 4 + (size_t)4 * p.NT * 32 * (64 + 16);
    if (p.shm > 64 * 1024) return p;
    if ((long long)N * (LH / TH) > 0x7fffffffLL) return p;
    p.ok = 1;
    return p;
}

extern "C" int p2p_conv_fewin_ok(int op, int stride, int dtype, int N, int LH, int LW, int cin_pad, int ncols) {
    return fi_plan(op, stride, dtype, N, LH, LW, cin_pad, ncols).ok;
}

static int conv_fewin_common(int op, int stride, int dtype, int N, int LH, int LW, int cin_pad, int ncols, int w_rows,
                             const p2p_tensor* in, const p2p_tensor* out, const void* w, const float* bias, int act,
                             float alpha, const p2p_tensor* gate, void* stream) {
    P2P_REQUIRE(in && out && in->ptr && out->ptr && w, "p2p_conv_fewin: null pointer");
    P2P_REQUIRE(w_rows >= 1, "p2p_conv_fewin: w_rows must be positive");
    const FiPlan p = fi_plan(op, stride, dtype, N, LH, LW, cin_pad, ncols);
    P2P_REQUIRE(p.ok, "p2p_conv_fewin: shape not supported (query p2p_conv_fewin_ok)");
    P2P_REQUIRE(in->ld == 8 && ((uintptr_t)in->ptr % 16) == 0 && ((uintptr_t)w % 16) == 0,
                "p2p_conv_fewin: the input view must hold 8-channel (16-byte) pixels, 16-byte aligned");
    FiArgs a;
    a.in = (const char*)in->ptr; a.in_img = in->img_stride; a.in_row = in->row_stride;
    a.out = (char*)out->ptr; a.out_img = out->img_stride; a.out_row = out->row_stride; a.out_ld = out->ld;
    a.w = (const char*)w; a.w_rows = w_rows;
    a.bias = bias; a.act = act; a.alpha = alpha;
    a.gate = nullptr; a.gate_img = 0; a.gate_row = a.gate_ld = 0;
    if (gate) {
        P2P_REQUIRE(gate->ptr && ncols % 8 == 0 && gate->ld % 8 == 0 && ((uintptr_t)gate->ptr % 16) == 0 && out->ld % 8 == 0 &&
                    ((uintptr_t)out->ptr % 16) == 0, "p2p_conv_fewin_actbwd: gate and output must hold whole 16-byte channel runs");
        a.gate = (const char*)gate->ptr; a.gate_img = gate->img_stride; a.gate_row = gate->row_stride; a.gate_ld = gate->ld;
    }
    a.LH = LH; a.LW = LW;
    a.lgLW = 0;
    while ((1 << a.lgLW) < LW) ++a.lgLW;
    a.ncols = ncols;
    a.mode = op == P2P_OP_G ? 0 : 2;
    a.S = op == P2P_OP_G ? stride : 1;
    a.TH = p.TH; a.strips_per_img = LH / p.TH; a.nstrips = N * a.strips_per_img;
    a.RH = p.RH; a.RW = p.RW;
    a.vec = out->ld % 8 == 0 && ((uintptr_t)out->ptr % 16) == 0;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)(a.nstrips < 1024 ? a.nstrips : 1024));      // <= 4 persistent workgroups per CU
    if (p.NT == 1) conv_fewin_kernel<1><<<grid, dim3(256), p.shm, st>>>(a);
    else conv_fewin_kernel<2><<<grid, dim3(512), p.shm, st>>>(a);
    return p2p_check_launch("p2p_conv_fewin");
}

extern "C" int p2p_conv_fewin(int op, int stride, int dtype, int N, int LH, int LW, int cin_pad, int ncols, int w_rows,
                              const p2p_tensor* in, const p2p_tensor* out, const void* w, const float* bias, int act,
                              float alpha, void* stream) {
    return conv_fewin_common(op, stride, dtype, N, LH, LW, cin_pad, ncols, w_rows, in, out, w, bias, act, alpha, nullptr, stream);
}

// The data gradient of a few-channel head followed by the backward of the LeakyReLU in front of it (networks.py:45-48: the
// discriminator's first block; tape gradient pix2pix_model.py:78-79): out = conv(in) * (gate > 0 ? 1 : alpha), gate = that
// block's activation output.  Same result, bit for bit, as p2p_conv_fewin followed by p2p_act_bwd.
extern "C" int p2p_conv_fewin_actbwd(int op, int stride, int dtype, int N, int LH, int LW, int cin_pad, int ncols, int w_rows,
                                     const p2p_tensor* in, const p2p_tensor* out, const void* w, const p2p_tensor* gate,
                                     float alpha, void* stream) {
    P2P_REQUIRE(gate, "p2p_conv_fewin_actbwd: null gate");
    return conv_fewin_common(op, stride, dtype, N, LH, LW, cin_pad, ncols, w_rows, in, out, w, nullptr, P2P_ACT_NONE, alpha, gate, stream);
}
