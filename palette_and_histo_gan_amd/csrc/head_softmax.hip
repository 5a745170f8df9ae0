// Palette-index head of Pix2PixIndexedModel in ONE kernel (pix2pix_model.py:261-325 over networks.py:75-78):
//     z[p][c]   = bias[c] + sum_{kh,kw,g} c6[p + (kh-1, kw-1)][g] * W[kh][kw][g][c]        Conv2D(256, 4, stride 1, SAME, bias)
//     probs     = softmax(z)            fake index = argmax(probs), ties -> lowest index     (:268, :286, :292)
//     seg       = mean_p -log probs[target]                                                  (:265, :274; log-sum-exp form)
//     dz[p][c]  = lambda_seg / #pixels * (probs - onehot(target))                            (the only gradient G receives, :263)
//     dbias     = per-workgroup column sums of dz (summed in fixed order by head_bias_sum_kernel)
// The 256-channel logits (268 MB at B = 128 in bf16) are never written: the generic path wrote them, read them back in
// p2p_softmax_cce_argmax and read dz a third time for the bias gradient (SURVEY.md 2.3 K13/K14 "fuse, never materialise").
//
// bf16, IMG_SIZE 64.  One workgroup (8 waves) owns 4 output rows x 64 pixels x all 256 channels:
//   * MFMA rows = channels, columns = pixels, so a lane holds 4 x 16 channels of ONE pixel per column tile and the softmax
//     reductions over channels are in-lane loops + one exchange with lane^32 + one with the partner wave (other channel half);
//   * the 7 x 67 pixel input strip (80-byte pixels: [up6 32 | source | zero pad]) is staged in LDS once; the four taps of a
//     kernel row are CONTIGUOUS there (pixel x-1+kw, channel g  <->  byte x*80 + 2*(40 kw + g)), so K runs over 160 values
//     per kernel row in 10 steps of 16 with no tap bookkeeping and no padding beyond the 40-channel pixels;
//   * the weights (327 KB) stream through a ring of 40 KB LDS stages (half a kernel row each) in MFMA-fragment order
//     (LDS-DMA with per-lane source addresses: lane (i, h) of block (step, channel tile) fetches its own 16 bytes), counted
//     s_waitcnt vmcnt + one raw s_barrier per stage;
//   * epilogue: softmax / CCE / argmax / gradient in registers, the bf16 gradient tile is transposed through LDS (padded
//     528-byte pixels) and leaves as whole 512-byte pixels.
#include "p2p_common.hpp"

#define HS_W 64
#define HS_CIN 40
#define HS_NCLS 256
#define HS_PIXB (HS_CIN * 2)                         // 80 bytes per input pixel
#define HS_STRIP_COLS (HS_W + 3)
#define HS_STRIP_ROWB (HS_STRIP_COLS * HS_PIXB)      // 5360
#define HS_KSTEPS_ROW 10                             // 4 taps x 40 channels = 160 values of K per kernel row = 10 steps of 16
#define HS_PATCH_PIXB 528                            // 512-byte gradient pixel + 16 bytes: conflict-free ds_write_b64 / ds_read_b128

// Two shapes of the same kernel (ROWS output rows per workgroup = ROWS x 2 waves; SS K steps per weight stage; RING stages):
//   <4, 5, 3>  8 waves, 160 KB of LDS: one workgroup per CU, the weights stream once per 4 rows;
//   <2, 2, 3>  4 waves, 76 KB: TWO workgroups per CU -- the long softmax epilogue (exponentials, argmax, gradient, transposition)
//              of one overlaps the MFMA loop of the other; the weights stream once per 2 rows (from L2).
template <int ROWS, int SS, int RING> struct HsCfg {
    static constexpr int NW = 2 * ROWS, NTHR = NW * 64, NPIX = ROWS * HS_W;
    static constexpr int STRIP_ROWS = ROWS + 3, STRIP_BYTES = STRIP_ROWS * HS_STRIP_ROWB;
    static constexpr int STAGE_BYTES = SS * 8 * 1024, NSTAGE = 4 * HS_KSTEPS_ROW / SS, PER_WAVE = SS * 8 / NW;
    static constexpr int SHM_MAIN = ((STRIP_BYTES + 255) & ~255) + RING * STAGE_BYTES;
    static constexpr int XCH_FLOATS = 6 * 2 * NPIX + 256 + 2 * NPIX + 4 * 256;        // exchange arrays, bias, per-pixel loss terms, bias-gradient partials
    static constexpr int SHM_EPI = NPIX * HS_PATCH_PIXB + XCH_FLOATS * 4;
    static constexpr int SHM = SHM_EPI > SHM_MAIN ? SHM_EPI : SHM_MAIN;
    static_assert(HS_KSTEPS_ROW % SS == 0 && (SS * 8) % NW == 0, "stage shape");
};

struct HsArgs {
    const char* in; long long in_img; int in_row;     // c6 view (element strides, ld = HS_CIN)
    const char* wt;                                   // [16][256][40] bf16
    const float* bias;
    const bf16_t* target; long long tg_img; int tg_row; int tg_ld;
    bf16_t* fake; long long fk_img; int fk_row; int fk_ld;
    char* dz; long long dz_img; int dz_row;           // ld = 256
    float* dbias_part;                                // [nwg][256] or null
    float* loss_part;                                 // [2][nwg]
    float grad_scale, inv_count;
    int H;
};

__device__ __forceinline__ void hs_glds16(const char* g, char* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <int ROWS, int SS, int RING>
__global__ __launch_bounds__(ROWS * 128) void head_softmax_kernel(HsArgs a) {
    typedef HsCfg<ROWS, SS, RING> K;
    constexpr int NW = K::NW, NTHR = K::NTHR, NPIX = K::NPIX, HS_STRIP_ROWS = K::STRIP_ROWS, HS_STRIP_BYTES = K::STRIP_BYTES;
    constexpr int HS_STAGE_BYTES = K::STAGE_BYTES, HS_NSTAGE = K::NSTAGE, HS_RING = RING, PER_WAVE = K::PER_WAVE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* strip = smem;
    char* ring = smem + ((HS_STRIP_BYTES + 255) & ~255);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cw = wave & 1, pr = wave >> 1;            // channel half, output row of the tile
    const int h = lane >> 5, li = lane & 31;
    const int tiles_per_img = a.H / ROWS;
    const int n = blockIdx.x / tiles_per_img, y0 = (blockIdx.x % tiles_per_img) * ROWS;

    // ---- staging ---------------------------------------------------------------------------------------------------
    {   // input strip: rows y0-1 .. y0+5, columns -1 .. 65, each row 5360 contiguous bytes in HBM
        constexpr int CPR = HS_STRIP_ROWB / 16, TOTAL = HS_STRIP_ROWS * CPR;      // 335 chunks per row
        const char* base = a.in + ((long long)n * a.in_img + (long long)(y0 - 1) * a.in_row - 1) * HS_PIXB;
        for (int c0 = wave * 64; c0 < TOTAL; c0 += NTHR) {
            const int ci = c0 + lane;
            if (ci < TOTAL) {
                const int r = ci / CPR, cc = ci - r * CPR;
                hs_glds16(base + (long long)r * a.in_row * HS_PIXB + cc * 16, strip + c0 * 16);
            }
        }
    }
    // weight stage st: SS consecutive K steps of one kernel row; block (s, ct) = 1 KB in fragment order, dealt round-robin to the waves
    constexpr int SPR = HS_KSTEPS_ROW / SS;            // stages per kernel row
    auto stage_w = [&](int st, char* buf) {
        const int kh = st / SPR, s0 = (st % SPR) * SS;
#pragma unroll
        for (int q = 0; q < PER_WAVE; ++q) {
            const int blk = q * NW + wave, s = blk >> 3, ct = blk & 7;
            const int kp = 16 * (s0 + s) + 8 * h;              // k' of this lane's 8 values inside the kernel row (0..159)
            const int kw = kp / HS_CIN, c = kp - kw * HS_CIN;
            const char* src = a.wt + ((((long long)(kh * 4 + kw)) * HS_NCLS + ct * 32 + li) * HS_CIN + c) * 2;
            hs_glds16(src, buf + blk * 1024);
        }
    };
    stage_w(0, ring);
    stage_w(1, ring + HS_STAGE_BYTES);

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // ---- main loop: 8 stages x 5 K steps x (4 channel tiles x 2 pixel tiles) ------------------------------------------
#pragma unroll 1
    for (int st = 0; st < HS_NSTAGE; ++st) {
        if (st + 1 < HS_NSTAGE) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_WAVE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (st + 2 < HS_NSTAGE) stage_w(st + 2, ring + ((st + 2) % HS_RING) * HS_STAGE_BYTES);
        const char* wbuf = ring + (st % HS_RING) * HS_STAGE_BYTES;
        const int kh = st / SPR, s0 = (st % SPR) * SS;
        const char* brow = strip + (pr + kh) * HS_STRIP_ROWB + li * HS_PIXB + 16 * h;
#pragma unroll
        for (int s = 0; s < SS; ++s) {
            bf16x8 bfr[2], afr[4];
#pragma unroll
            for (int j = 0; j < 2; ++j) bfr[j] = *(const bf16x8*)(brow + j * 32 * HS_PIXB + 32 * (s0 + s));
#pragma unroll
            for (int i = 0; i < 4; ++i) afr[i] = *(const bf16x8*)(wbuf + (s * 8 + cw * 4 + i) * 1024 + lane * 16);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();          // every wave is done with the strip and the ring: both are re-used below

    // ---- epilogue ------------------------------------------------------------------------------------------------------
    // LDS: gradient patch [NPIX pixels][528 B] from offset 0; exchange arrays behind it
    char* patch = smem;
    float* xch = (float*)(smem + NPIX * HS_PATCH_PIXB);
    constexpr int XS = 2 * NPIX;                              // [2 channel halves][NPIX]
    float* xmax = xch, *xsum = xch + XS, *xzt = xch + 2 * XS, *xbest = xch + 3 * XS, *xpt = xch + 4 * XS;
    int* xbidx = (int*)(xch + 5 * XS);
    float* sbias = xch + 6 * XS;                               // [256]
    float* spix = sbias + 256;                                 // [2][NPIX]: per-pixel loss terms
    float* sdb = spix + 2 * NPIX;                              // [4][256]: bias-gradient partials of the pixel groups
    for (int c = tid; c < HS_NCLS; c += NTHR) sbias[c] = a.bias ? a.bias[c] : 0.f;
    __syncthreads();
    const int chbase = cw * 128 + 4 * h;       // channel of (i, e): chbase + 32 i + (e & 3) + 8 (e >> 2)
    int tix[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int x = j * 32 + li, y = y0 + pr;
        tix[j] = (int)to_f32(a.target[((long long)n * a.tg_img + (long long)y * a.tg_row + x) * a.tg_ld]);
        float m = -INFINITY;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                acc[i][j][e] += sbias[chbase + 32 * i + (e & 3) + 8 * (e >> 2)];
                m = fmaxf(m, acc[i][j][e]);
            }
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        if (h == 0) xmax[cw * NPIX + pr * 64 + x] = m;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int px = pr * 64 + j * 32 + li;
        const float M = fmaxf(xmax[px], xmax[NPIX + px]);
        const int trel = tix[j] - chbase;              // the target's channel relative to this lane's first one
        float s = 0.f, zt = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float d = acc[i][j][e] - M;
                if (32 * i + (e & 3) + 8 * (e >> 2) == trel) zt = d;      // z_t - max, before the exponential
                const float ex = __expf(d);            // v_exp_f32: the bf16 mode's exponentials (f32 mode keeps the generic path)
                acc[i][j][e] = ex;
                s += ex;
            }
        s += __shfl_xor(s, 32, 64);
        zt += __shfl_xor(zt, 32, 64);
        if (h == 0) { xsum[cw * NPIX + px] = s; xzt[cw * NPIX + px] = zt; }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int px = pr * 64 + j * 32 + li;
        const float rS = 1.0f / (xsum[px] + xsum[NPIX + px]);
        const int trel = tix[j] - chbase;
        float best = -1.f, pt = 0.f;
        int bidx = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int cr = 32 * i + (e & 3) + 8 * (e >> 2);
                const float p = acc[i][j][e] * rS;             // exp(z - max) / sum
                if (p > best) { best = p; bidx = cr; }         // channels ascend with (i, e): strict '>' keeps the lowest index
                const bool hit = cr == trel;
                if (hit) pt = p;
                acc[i][j][e] = (p - (hit ? 1.f : 0.f)) * a.grad_scale;
            }
        bidx += chbase;
        {
            const float ob = __shfl_xor(best, 32, 64);
            const int oi = __shfl_xor(bidx, 32, 64);
            if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
            pt += __shfl_xor(pt, 32, 64);
        }
        if (h == 0) { xbest[cw * NPIX + px] = best; xbidx[cw * NPIX + px] = bidx; xpt[cw * NPIX + px] = pt; }
        // gradient tile -> LDS patch [pixel][channel] in bf16, 4 consecutive channels per store
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                typedef __attribute__((__vector_size__(4 * sizeof(bf16_t)))) bf16_t bf16x4;
                bf16x4 v4;
#pragma unroll
                for (int k = 0; k < 4; ++k) v4[k] = (bf16_t)acc[i][j][4 * g + k];
                *(bf16x4*)(patch + px * HS_PATCH_PIXB + (chbase + 32 * i + 8 * g) * 2) = v4;
            }
    }
    __syncthreads();
    // argmax across the channel halves, loss terms, fake index: one thread per pixel
    for (int px = tid; px < NPIX; px += NTHR) {
        const int r = px >> 6, x = px & 63, y = y0 + r;
        float b0 = xbest[px], b1 = xbest[NPIX + px];
        int i0 = xbidx[px], i1 = xbidx[NPIX + px];
        if (b1 > b0 || (b1 == b0 && i1 < i0)) { b0 = b1; i0 = i1; }
        a.fake[((long long)n * a.fk_img + (long long)y * a.fk_row + x) * a.fk_ld] = (bf16_t)(float)i0;
        const float pt = xpt[px] + xpt[NPIX + px];
        spix[px] = logf(xsum[px] + xsum[NPIX + px]) - (xzt[px] + xzt[NPIX + px]);      // -log softmax(z)[target]
        spix[NPIX + px] = 2.f * (1.f - pt);           // sum_c |onehot_c - p_c| = (1 - p_t) + sum_{c != t} p_c
    }
    // whole 512-byte gradient pixels: 32 lanes x 16 B per pixel
    {
        const int sub = tid & 31;
        for (int px = tid >> 5; px < NPIX; px += NTHR / 32) {
            const int r = px >> 6, x = px & 63, y = y0 + r;
            const f32x4 v = *(const f32x4*)(patch + px * HS_PATCH_PIXB + sub * 16);
            *(f32x4*)(a.dz + ((long long)n * a.dz_img + (long long)y * a.dz_row + x) * (HS_NCLS * 2) + sub * 16) = v;
        }
    }
    // bias gradient partial: column sums of the ROUNDED gradient tile; thread = (channel pair, pixel group), pixels in order
    if (a.dbias_part) {
        constexpr int NG = NTHR / 128;
        const int cp = tid & 127, pg = tid >> 7;
        float s0 = 0.f, s1 = 0.f;
#pragma unroll 8
        for (int px = pg * (NPIX / NG); px < (pg + 1) * (NPIX / NG); ++px) {
            const unsigned u = *(const unsigned*)(patch + px * HS_PATCH_PIXB + cp * 4);
            s0 += __uint_as_float(u << 16);
            s1 += __uint_as_float(u & 0xffff0000u);
        }
        sdb[pg * 256 + 2 * cp] = s0;
        sdb[pg * 256 + 2 * cp + 1] = s1;
    }
    __syncthreads();
    if (a.dbias_part)
        for (int c = tid; c < HS_NCLS; c += NTHR) {
            float s = 0.f;
#pragma unroll
            for (int g = 0; g < NTHR / 128; ++g) s += sdb[g * 256 + c];
            a.dbias_part[(long long)blockIdx.x * HS_NCLS + c] = s;
        }
    if (tid < 64) {           // fixed-order sums of the per-pixel loss terms
        float s0 = 0.f, s1 = 0.f;
        for (int k = 0; k < NPIX / 64; ++k) { s0 += spix[tid + 64 * k]; s1 += spix[NPIX + tid + 64 * k]; }
        s0 = wave_sum(s0);
        s1 = wave_sum(s1);
        if (tid == 0) {
            a.loss_part[blockIdx.x] = s0 * a.inv_count;
            a.loss_part[gridDim.x + blockIdx.x] = s1 * a.inv_count / (float)HS_NCLS;
        }
    }
}

// loss_out[k] = sum_b part[k][b]; dbias[c] = sum_b dbias_part[b][c] -- workgroup order, bit-reproducible
__global__ __launch_bounds__(256) void head_loss_sum_kernel(const float* __restrict__ part, int nb, float* __restrict__ loss_out) {
    __shared__ float red[16];
    for (int k = 0; k < 2; ++k) {
        float s = 0.f;
        for (int b = threadIdx.x; b < nb; b += 256) s += part[k * nb + b];
        s = block_sum(s, red);
        if (threadIdx.x == 0) loss_out[k] = s;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void head_bias_sum_kernel(const float* __restrict__ part, int nb, float* __restrict__ dbias) {
    __shared__ float red[16];
    const int c = blockIdx.x;
    float s = 0.f;
    for (int b = threadIdx.x; b < nb; b += 256) s += part[(long long)b * HS_NCLS + c];
    s = block_sum(s, red);
    if (threadIdx.x == 0) dbias[c] = s;
}

#include <stdlib.h>
// (one form: 4 output rows per workgroup.  The 2-row form -- two workgroups per CU -- streamed the weights twice as often and
// measured 0.428 against 0.289 ms at B = 128, r03; it and its switch are gone.)

// Both kernels of this file need the whole 160 KB of LDS of a gfx950 CU: the predicates answer 0 on a device that cannot give it,
// so the engine falls back to p2p_igemm_edge + p2p_softmax_cce_argmax instead of failing at the launch.
static bool hs_lds_ok(size_t need) {
    static int max_lds = -1;
    if (max_lds < 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) v = 0;
        max_lds = v;
    }
    return (size_t)max_lds >= need;
}

extern "C" int p2p_head_softmax_ok(int dtype, int N, int H, int W, int cin_pad, int ncls) {
    return dtype == P2P_BF16 && N > 0 && W == HS_W && H % 4 == 0 && cin_pad == HS_CIN && ncls == HS_NCLS && hs_lds_ok(HsCfg<4, 5, 3>::SHM);
}

extern "C" long long p2p_head_softmax_workspace_bytes(int N, int H) {
    const long long nwg = (long long)N * (H / 2);        // (sized for 2-row workgroups: callers' buffers keep their size)
    return nwg * (HS_NCLS + 2) * (long long)sizeof(float);
}

template <int ROWS, int SS, int RING>
static int hs_launch(const HsArgs& a, int nwg, hipStream_t st) {
    typedef HsCfg<ROWS, SS, RING> K;
    static_assert(K::SHM <= 160 * 1024, "LDS budget");
    static int attr_dev = -1;      // the attribute is per device: set (and checked) again when the current device changes
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (attr_dev != dev) {
        const hipError_t e = hipFuncSetAttribute((const void*)head_softmax_kernel<ROWS, SS, RING>, hipFuncAttributeMaxDynamicSharedMemorySize, K::SHM);
        P2P_REQUIRE(e == hipSuccess, "p2p_head_softmax_cce: %d bytes of dynamic LDS refused (%s)", (int)K::SHM, hipGetErrorString(e));
        attr_dev = dev;
    }
    head_softmax_kernel<ROWS, SS, RING><<<dim3(nwg), dim3(K::NTHR), K::SHM, st>>>(a);
    return 0;
}

extern "C" int p2p_head_softmax_cce(int dtype, int N, int H, int W, int cin_pad, int ncls, const p2p_tensor* in, const void* wt,
                                    const float* bias, const p2p_tensor* target, const p2p_tensor* fake_idx, float grad_scale,
                                    float inv_count, const p2p_tensor* dz, float* dbias, float* workspace, float* loss_out,
                                    void* stream) {
    P2P_REQUIRE(p2p_head_softmax_ok(dtype, N, H, W, cin_pad, ncls), "p2p_head_softmax_cce: shape not supported (query p2p_head_softmax_ok)");
    P2P_REQUIRE(in && in->ptr && wt && target && target->ptr && fake_idx && fake_idx->ptr && dz && dz->ptr && workspace && loss_out,
                "p2p_head_softmax_cce: null pointer");
    P2P_REQUIRE(in->ld == HS_CIN && dz->ld == HS_NCLS, "p2p_head_softmax_cce: input pixels hold %d channels, gradient pixels %d", HS_CIN, HS_NCLS);
    P2P_REQUIRE(((uintptr_t)in->ptr % 16) == 0 && ((uintptr_t)wt % 16) == 0 && ((uintptr_t)dz->ptr % 16) == 0, "p2p_head_softmax_cce: alignment");
    HsArgs a;
    a.in = (const char*)in->ptr; a.in_img = in->img_stride; a.in_row = in->row_stride;
    a.wt = (const char*)wt; a.bias = bias;
    a.target = (const bf16_t*)target->ptr; a.tg_img = target->img_stride; a.tg_row = target->row_stride; a.tg_ld = target->ld;
    a.fake = (bf16_t*)fake_idx->ptr; a.fk_img = fake_idx->img_stride; a.fk_row = fake_idx->row_stride; a.fk_ld = fake_idx->ld;
    a.dz = (char*)dz->ptr; a.dz_img = dz->img_stride; a.dz_row = dz->row_stride;
    const int nwg = N * (H / 4);
    a.loss_part = workspace;
    a.dbias_part = dbias ? workspace + 2 * (long long)nwg : nullptr;
    a.grad_scale = grad_scale; a.inv_count = inv_count; a.H = H;
    hipStream_t st = (hipStream_t)stream;
    int rc = hs_launch<4, 5, 3>(a, nwg, st);
    if (rc) return rc;
    rc = p2p_check_launch("p2p_head_softmax_cce");
    if (rc) return rc;
    head_loss_sum_kernel<<<1, 256, 0, st>>>(a.loss_part, nwg, loss_out);
    if (dbias) head_bias_sum_kernel<<<dim3(HS_NCLS), 256, 0, st>>>(a.dbias_part, nwg, dbias);
    return p2p_check_launch("p2p_head_softmax_cce sums");
}

// =====================================================================================================================
// Data gradient of the same head (pix2pix_model.py:78 through networks.py:75-78): op P, stride 1, 256 -> 32 channels
//     g6[n,y,x,g] = sum_{kh,kw,d} dz[n, y+1-kh, x+1-kw, d] * W[kh][kw][g][d]          g < 32 (the source image has no gradient)
// 137 GFLOP at B = 128 with K = 16 taps x 256 channels and only 32 output channels: every pixel fragment feeds ONE MFMA, so
// the kernel lives on LDS bandwidth.  One workgroup (8 waves) owns 8 output rows x 64 pixels; wave w owns row w.  K is cut
// into 8 chunks of 32 dz channels; per chunk the 11 x 67 pixel strip of 64-byte pixel slices (XOR-swizzled 16-byte slots:
// conflict-free ds_read_b128 at 64-byte pixel pitch) and the chunk's weights of all 16 taps (32 KB, MFMA-fragment order)
// are double-buffered in LDS; every wave issues the same number of LDS-DMA pieces per stage (counted s_waitcnt vmcnt).
#define HD_ROWS 8
#define HD_STRIP_ROWS (HD_ROWS + 3)
#define HD_COLS (HS_W + 3)
#define HD_STRIP_CHUNKS (HD_STRIP_ROWS * HD_COLS * 4)                 // 2948 16-byte slots
#define HD_STRIP_ROUNDS ((HD_STRIP_CHUNKS + 511) / 512)               // 6 LDS-DMA rounds of 512 lanes
#define HD_STRIP_BYTES (HD_STRIP_ROUNDS * 512 * 16)                   // 49152 (the tail is padding)
#define HD_W_BYTES (32 * 1024)                                        // 16 taps x 2 K steps x 1 KB
#define HD_STAGE (HD_STRIP_BYTES + HD_W_BYTES)
#define HD_NCHUNK (HS_NCLS / 32)

struct HdArgs {
    const char* dz; long long dz_img; int dz_row;      // haloed view, 256-channel pixels
    const char* wn;                                     // [16][w_rows][256] bf16
    int w_rows;
    char* out; long long out_img; int out_row; int out_ld;
    int H;
};

__global__ __launch_bounds__(512) void head_dgrad_kernel(HdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, li = lane & 31;
    const int tiles_per_img = a.H / HD_ROWS;
    const int n = blockIdx.x / tiles_per_img, y0 = (blockIdx.x % tiles_per_img) * HD_ROWS;
    const char* dzbase = a.dz + ((long long)n * a.dz_img + (long long)(y0 - 2) * a.dz_row - 2) * (HS_NCLS * 2);

    // per-lane source offsets of the strip slots this lane fills (the same for every channel chunk)
    long long soff[HD_STRIP_ROUNDS];
#pragma unroll
    for (int r = 0; r < HD_STRIP_ROUNDS; ++r) {
        int sl = r * 512 + wave * 64 + lane;
        if (sl >= HD_STRIP_CHUNKS) sl = HD_STRIP_CHUNKS - 1;            // padding slots re-fetch the last real one
        const int pidx = sl >> 2, qs = sl & 3;
        const int rr = pidx / HD_COLS, cidx = pidx - rr * HD_COLS;
        const int q = qs ^ ((cidx >> 2) & 3);                            // swizzle: slot qs of a pixel holds source chunk q
        soff[r] = ((long long)rr * a.dz_row + cidx) * (HS_NCLS * 2) + q * 16;
    }
    auto stage = [&](int dc, char* buf) {
#pragma unroll
        for (int r = 0; r < HD_STRIP_ROUNDS; ++r) hs_glds16(dzbase + soff[r] + dc * 64, buf + (r * 512 + wave * 64) * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int blk = q * 8 + wave, tap = blk >> 1, s = blk & 1;
            const char* src = a.wn + (((long long)tap * a.w_rows + li) * HS_NCLS + dc * 32 + s * 16 + 8 * h) * 2;
            hs_glds16(src, buf + HD_STRIP_BYTES + blk * 1024);
        }
    };
    constexpr int PER_WAVE = HD_STRIP_ROUNDS + 4;
    stage(0, smem);

    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

#pragma unroll 1
    for (int dc = 0; dc < HD_NCHUNK; ++dc) {
        char* cur = smem + (dc & 1) * HD_STAGE;
        // the other buffer was read during chunk dc-1: every wave is past that before anyone refills it
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (dc + 1 < HD_NCHUNK) {
            stage(dc + 1, smem + ((dc + 1) & 1) * HD_STAGE);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_WAVE) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();             // chunk dc has landed for every wave
        const char* wbuf = cur + HD_STRIP_BYTES;
#pragma unroll
        for (int tap = 0; tap < 16; ++tap) {
            const int kh = tap >> 2, kw = tap & 3;
            const int rr = wave + 3 - kh;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 afr = *(const bf16x8*)(wbuf + (tap * 2 + s) * 1024 + lane * 16);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int cidx = j * 32 + li + 3 - kw;
                    const int slot = ((rr * HD_COLS + cidx) << 2) | ((2 * s + h) ^ ((cidx >> 2) & 3));
                    const bf16x8 bfr = *(const bf16x8*)(cur + slot * 16);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr, bfr, acc[j], 0, 0, 0);
                }
            }
        }
    }
    // D[row = channel g][col = pixel]: g = (e & 3) + 8 (e >> 2) + 4 h; 4 consecutive channels per 8-byte store
    const int y = y0 + wave;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int x = j * 32 + li;
        char* op = a.out + ((long long)n * a.out_img + (long long)y * a.out_row + x) * a.out_ld * 2;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            typedef __attribute__((__vector_size__(4 * sizeof(bf16_t)))) bf16_t bf16x4;
            bf16x4 v4;
#pragma unroll
            for (int k = 0; k < 4; ++k) v4[k] = (bf16_t)acc[j][4 * g + k];
            *(bf16x4*)(op + (8 * g + 4 * h) * 2) = v4;
        }
    }
}

extern "C" int p2p_head_dgrad_ok(int dtype, int N, int H, int W, int ncls, int cout, int w_rows, int dz_ld, int out_ld) {
    return dtype == P2P_BF16 && N > 0 && W == HS_W && H % HD_ROWS == 0 && ncls == HS_NCLS && cout == 32 && w_rows >= 32 &&
           dz_ld == HS_NCLS && out_ld >= 32 && out_ld % 4 == 0 && hs_lds_ok(2 * HD_STAGE);
}

extern "C" int p2p_head_dgrad(int dtype, int N, int H, int W, int ncls, int cout, const p2p_tensor* dz, const void* wn, int w_rows,
                              const p2p_tensor* out, void* stream) {
    P2P_REQUIRE(dz && dz->ptr && wn && out && out->ptr, "p2p_head_dgrad: null pointer");
    P2P_REQUIRE(p2p_head_dgrad_ok(dtype, N, H, W, ncls, cout, w_rows, dz->ld, out->ld), "p2p_head_dgrad: shape not supported (query p2p_head_dgrad_ok)");
    P2P_REQUIRE(((uintptr_t)dz->ptr % 16) == 0 && ((uintptr_t)wn % 16) == 0 && ((uintptr_t)out->ptr % 8) == 0, "p2p_head_dgrad: alignment");
    HdArgs a;
    a.dz = (const char*)dz->ptr; a.dz_img = dz->img_stride; a.dz_row = dz->row_stride;
    a.wn = (const char*)wn; a.w_rows = w_rows;
    a.out = (char*)out->ptr; a.out_img = out->img_stride; a.out_row = out->row_stride; a.out_ld = out->ld;
    a.H = H;
    constexpr int SHM = 2 * HD_STAGE;
    static_assert(SHM <= 160 * 1024, "LDS budget");
    static int attr_dev = -1;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (attr_dev != dev) {
        const hipError_t e = hipFuncSetAttribute((const void*)head_dgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SHM);
        P2P_REQUIRE(e == hipSuccess, "p2p_head_dgrad: %d bytes of dynamic LDS refused (%s)", (int)SHM, hipGetErrorString(e));
        attr_dev = dev;
    }
    head_dgrad_kernel<<<dim3(N * (H / HD_ROWS)), dim3(512), SHM, (hipStream_t)stream>>>(a);
    return p2p_check_launch("p2p_head_dgrad");
}
