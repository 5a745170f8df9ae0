// The widest-map encoder/decoder block, up6 = Conv2DTranspose 128 -> 32 onto the full-resolution map (networks.py:26-27,
// 66-73), forward (op P) and data gradient (op G).  M = N*32*32 lo pixels is huge, K per output is only 512 and one side
// has 32 channels: in the implicit GEMM (igemm.hip) a 256x32 tile stages 9 KB of operands per 8 MFMAs and every lo pixel
// is fetched 16 times (4 phases x 4 taps), so the launch is bound by LDS-DMA issue (235 TFLOP/s).  Here, per workgroup:
//   * a strip of input rows (with halo) is brought into LDS ONCE (padded pixel stride -> conflict-free ds_read_b128),
//   * the weights live in registers as the MFMA A operand: 32 fragments = 128 VGPRs per wave; eight waves = four OUTPUT
//     slices x two tile parities: op P -> slice = sub-pixel phase (4 taps x 128 channels each), op G -> slice = 32 of the
//     128 output channels (16 taps x 32 channels each); workgroups are persistent (one per CU) and prefetch the next
//     strip into registers while the current one is multiplied,
//   * B fragments are per-lane ds_read_b128 of the pixel a tap points at (stride-2 gathers are free), one read per MFMA,
//   * the 32-pixel x 32-channel result goes through a per-wave LDS patch and leaves as 16-byte chunks.
// bf16 only (v_mfma_f32_32x32x16_bf16, f32 accumulate); other shapes and f32 stay on p2p_igemm.
#include "p2p_common.hpp"

struct CsArgs {
    const char* in; long long in_img; int in_row; int in_ld;
    char* out; long long out_img; int out_row; int out_ld;
    const char* w;                       // op P: wn [16][32][128]; op G: wt [16][128][32] (bf16)
    float* stat_part;                    // op P, optional: [N][slots][32][2] (mean, centred sum of squares), slots = strips_per_img
    int LH, LW, lgLW;                    // lo grid
    int TH, strips_per_img, nstrips;
    int RH, RW, PB;                      // strip pixels, padded LDS bytes per strip pixel
    float inv_RW;
};

#define CS_MAXC 7                        // 16-byte chunks of a strip staged per thread (register prefetch of the next strip)
#define CS_THREADS 512                   // 8 waves: wave & 3 = output slice (phase / channel block), wave >> 2 = tile parity

// MODE 1: op P stride 2 (lo 128 ch -> hi 32 ch, 4 phases);  MODE 0: op G stride 2 (hi 32 ch -> lo 128 ch)
template <int MODE>
__global__ __launch_bounds__(CS_THREADS) void conv_strip_kernel(CsArgs a) {
    constexpr int NTAP = MODE == 1 ? 4 : 16;         // taps contracted per output
    constexpr int KS = MODE == 1 ? 8 : 2;            // 16-channel K steps per tap (128 / 32 input channels)
    constexpr int CIN = KS * 16;
    constexpr int LGCPP = MODE == 1 ? 4 : 2;         // log2(16-byte chunks per input pixel)
    constexpr int PROW = 64 + 16;                    // patch row: 32 bf16 + pad
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, r = lane & 31;
    const int RW = a.RW, PB = a.PB;
    char* sL = smem;
    char* pL = smem + ((a.RH * RW * PB + 15) & ~15) + wave * (32 * PROW);
    const int slice = wave & 3, tpar = wave >> 2;    // output slice of this wave; it takes the tiles tpar, tpar + 2, ...
    const int ph = slice >> 1, pw = slice & 1;       // MODE 1: the slice is a sub-pixel phase

    // ---- weights -> registers, once per (persistent) workgroup --------------------------------------------------------------
    bf16x8 wf[NTAP][KS];
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
        int widx, row;
        if (MODE == 1) { widx = ((1 - ph) + 2 * (t >> 1)) * 4 + (1 - pw) + 2 * (t & 1); row = r; }       // wn[widx][g = r][d]
        else { widx = t; row = 32 * slice + r; }                                                          // wt[t][d = 32 slice + r][g]
        const int rows = MODE == 1 ? 32 : 128;
        const char* wp = a.w + ((long long)(widx * rows + row) * CIN) * 2;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) wf[t][ks] = *(const bf16x8*)(wp + (16 * ks + 8 * h) * 2);
    }

    // ---- strip staging through registers: the loads of strip s+1 are in flight while strip s is multiplied ---------------------
    const int nch = (a.RH * RW) << LGCPP;
    const int gpB = a.in_ld * 2;
    f32x4 stg[CS_MAXC];
    auto g_load = [&](int strip) {
        const int n = strip / a.strips_per_img, y0 = (strip - n * a.strips_per_img) * a.TH;
        const int oy = MODE == 1 ? y0 - 1 : 2 * y0 - 1;
        const char* base = a.in + ((long long)n * a.in_img + (long long)oy * a.in_row - 1) * gpB;
#pragma unroll
        for (int i = 0; i < CS_MAXC; ++i) {
            const int ci = tid + i * CS_THREADS;
            if (ci < nch) {
                const int px = ci >> LGCPP, cc = ci & ((1 << LGCPP) - 1);
                const int ry = (int)(((float)px + 0.5f) * a.inv_RW), rx = px - ry * RW;
                stg[i] = *(const f32x4*)(base + ((long long)ry * a.in_row + rx) * gpB + cc * 16);
            }
        }
    };
    auto l_store = [&]() {
#pragma unroll
        for (int i = 0; i < CS_MAXC; ++i) {
            const int ci = tid + i * CS_THREADS;
            if (ci < nch) *(f32x4*)(sL + (ci >> LGCPP) * PB + (ci & ((1 << LGCPP) - 1)) * 16) = stg[i];
        }
    };

    const int tiles = (a.TH * a.LW) >> 5;
    float* stL = (float*)(smem + ((a.RH * RW * PB + 15) & ~15) + 8 * (32 * PROW));      // [8 waves][32 ch][2] strip statistics
    int pend_n = -1, pend_s = 0;         // strip whose wave totals wait in stL
    auto flush_stats = [&]() {
        if (tid < 32) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int wv = 0; wv < 8; ++wv) { s1 += stL[(wv * 32 + tid) * 2]; s2 += stL[(wv * 32 + tid) * 2 + 1]; }
            const float cnt = (float)(4 * a.TH * a.LW);
            const float mean = s1 / cnt;
            float* dst = a.stat_part + (((long long)pend_n * a.strips_per_img + pend_s) * 32 + tid) * 2;
            dst[0] = mean;
            dst[1] = fmaxf(s2 - s1 * mean, 0.f);
        }
    };
    int strip = blockIdx.x;
    if (strip < a.nstrips) g_load(strip);
    for (; strip < a.nstrips; strip += gridDim.x) {
        __syncthreads();                 // every wave is done reading the previous strip (and has parked its statistics)
        if (MODE == 1 && a.stat_part && pend_n >= 0) flush_stats();
        l_store();
        __syncthreads();
        if (strip + (int)gridDim.x < a.nstrips) g_load(strip + gridDim.x);
        const int n = strip / a.strips_per_img, sidx = strip - n * a.strips_per_img, y0 = sidx * a.TH;
        float ssum = 0.f, ssq = 0.f;     // MODE 1 statistics: lane = (channel lane & 31, pixel half lane >> 5) of this wave's tiles
        for (int tile = tpar; tile < tiles; tile += 2) {
            const int p = tile * 32 + r;
            const int yy = p >> a.lgLW, x = p & (a.LW - 1);
            f32x16 acc;
            const float zero = p2p_valu_zero();      // not the MFMA's "C = 0": the previous tile's stores may still be reading these registers
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = zero;
#pragma unroll
            for (int t = 0; t < NTAP; ++t) {
                int ry, rx;
                if (MODE == 1) {
                    const int kh = (1 - ph) + 2 * (t >> 1), kw = (1 - pw) + 2 * (t & 1);
                    ry = yy + 1 + ((ph + 1 - kh) >> 1);
                    rx = x + 1 + ((pw + 1 - kw) >> 1);
                } else { ry = 2 * yy + (t >> 2); rx = 2 * x + (t & 3); }
                const char* bp = sL + (ry * RW + rx) * PB + 16 * h;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bf16x8 b = *(const bf16x8*)(bp + 32 * ks);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[t][ks], b, acc, 0, 0, 0);
                }
            }
            // D: column = pixel r, rows = this wave's 32 output channels 8 g + 4 h + k -> patch[pixel][channel]
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                typedef __attribute__((__vector_size__(4 * sizeof(bf16_t)))) bf16_t bf16x4;
                bf16x4 q;
#pragma unroll
                for (int k = 0; k < 4; ++k) q[k] = (bf16_t)acc[4 * g + k];
                *(bf16x4*)(pL + r * PROW + (8 * g + 4 * h) * 2) = q;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if (MODE == 1 && a.stat_part) {
                // statistics of the ROUNDED values the normalisation will read back, taken from the patch: lane sums
                // channel (lane & 31) over the 16 pixels of its half
                const bf16_t* col = (const bf16_t*)(pL + (16 * h) * PROW) + r;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float v = (float)col[i * (PROW / 2)];
                    ssum += v;
                    ssq += v * v;
                }
            }
            const int ch = lane & 3, pq = lane >> 2;     // 4 chunks of 16 B per pixel, 16 pixels per pass
#pragma unroll
            for (int ps = 0; ps < 32; ps += 16) {
                const int pix = ps + pq;
                const int pp = tile * 32 + pix;
                const int py = pp >> a.lgLW, px = pp & (a.LW - 1);
                const f32x4 v = *(const f32x4*)(pL + pix * PROW + ch * 16);
                long long opix;
                int c0;
                if (MODE == 1) { opix = (long long)n * a.out_img + (long long)(2 * (y0 + py) + ph) * a.out_row + (2 * px + pw); c0 = ch * 8; }
                else { opix = (long long)n * a.out_img + (long long)(y0 + py) * a.out_row + px; c0 = 32 * slice + ch * 8; }
                *(f32x4*)((bf16_t*)a.out + opix * a.out_ld + c0) = v;
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (MODE == 1 && a.stat_part) {
            // wave totals -> LDS; the 8 waves are added and written out at the top of the next iteration (after its first
            // barrier), so the strip loop keeps its two barriers
            ssum += __shfl_xor(ssum, 32, 64);
            ssq += __shfl_xor(ssq, 32, 64);
            if (h == 0) { stL[(wave * 32 + r) * 2] = ssum; stL[(wave * 32 + r) * 2 + 1] = ssq; }
            pend_n = n; pend_s = sidx;
        }
    }
    if (MODE == 1 && a.stat_part && pend_n >= 0) {
        __syncthreads();
        flush_stats();
    }
}

struct CsPlan { int ok, TH, RH, RW, PB, blocks; size_t shm; };

static CsPlan cs_plan(int op, int dtype, int N, int LH, int LW, int Cg, int Cd) {
    CsPlan p = {0, 0, 0, 0, 0, 0, 0};
    if (dtype != P2P_BF16 || Cg != 32 || Cd != 128 || N < 1) return p;
    if (op != P2P_OP_G && op != P2P_OP_P) return p;
    if ((LW & (LW - 1)) || LW < 32 || LW > 128 || LH < 1) return p;
    const int cin = op == P2P_OP_P ? 128 : 32;
    p.PB = cin * 2 + 16;
    // strip height: 4 or 2 rows within 72 KB (two workgroups per CU) if possible, else the tallest strip within 150 KB
    int TH = 0;
    for (int pass = 0; pass < 2 && !TH; ++pass)
        for (int t = 4; t >= (pass ? 1 : 2); t >>= 1) {
            if (t > LH || LH % t) continue;
            p.RH = op == P2P_OP_P ? t + 2 : 2 * t + 3;
            p.RW = op == P2P_OP_P ? LW + 2 : 2 * LW + 3;
            p.shm = (((size_t)p.RH * p.RW * p.PB + 15) & ~(size_t)15) + 8 * 32 * (size_t)(64 + 16) + 8 * 32 * 2 * sizeof(float);
            const long long nch = (long long)p.RH * p.RW * (cin / 8);
            if (nch > CS_MAXC * CS_THREADS) continue;            // staged through CS_MAXC registers per thread
            if (((t * LW) >> 5) & 1) continue;                   // the two waves of a slice split the 32-pixel tiles evenly
            if (p.shm <= (size_t)(pass ? 150 : 80) * 1024) { TH = t; break; }
        }
    if (!TH) return p;
    p.TH = TH;
    const long long strips = (long long)N * (LH / TH);
    if (strips > 0x7fffffffLL) return p;
    p.blocks = (int)(strips < 256 ? strips : 256);              // persistent workgroups, one per CU: weights are loaded once each
    p.ok = 1;
    return p;
}

extern "C" int p2p_conv_strip_ok(int op, int dtype, int N, int LH, int LW, int Cg, int Cd) {
    return cs_plan(op, dtype, N, LH, LW, Cg, Cd).ok;
}

// statistics slots per image written by op P (0 for op G / unsupported shapes): one per strip, each over the strip's
// 4 * TH * LW output pixels (same layout as p2p_igemm stat_part, consumed by p2p_norm_act_fwd with nsplit = -slots)
extern "C" int p2p_conv_strip_stat_slots(int op, int dtype, int N, int LH, int LW, int Cg, int Cd) {
    const CsPlan p = cs_plan(op, dtype, N, LH, LW, Cg, Cd);
    return (p.ok && op == P2P_OP_P) ? LH / p.TH : 0;
}

extern "C" int p2p_conv_strip(int op, int dtype, int N, int LH, int LW, int Cg, int Cd, const p2p_tensor* hi,
                              const p2p_tensor* lo, const void* w, float* stat_part, void* stream) {
    P2P_REQUIRE(hi && lo && hi->ptr && lo->ptr && w, "p2p_conv_strip: null pointer");
    const CsPlan p = cs_plan(op, dtype, N, LH, LW, Cg, Cd);
    P2P_REQUIRE(p.ok, "p2p_conv_strip: shape not supported (query p2p_conv_strip_ok)");
    const p2p_tensor* in = op == P2P_OP_G ? hi : lo;
    const p2p_tensor* out = op == P2P_OP_G ? lo : hi;
    const int cin = op == P2P_OP_P ? 128 : 32, cout = op == P2P_OP_P ? 32 : 128;
    P2P_REQUIRE(in->ld >= cin && in->ld % 8 == 0 && ((uintptr_t)in->ptr % 16) == 0 && ((uintptr_t)w % 16) == 0,
                "p2p_conv_strip: input view / weights must be 16-byte aligned");
    P2P_REQUIRE(out->ld >= cout && out->ld % 8 == 0 && ((uintptr_t)out->ptr % 16) == 0, "p2p_conv_strip: output view must be 16-byte aligned");
    CsArgs a;
    a.in = (const char*)in->ptr; a.in_img = in->img_stride; a.in_row = in->row_stride; a.in_ld = in->ld;
    a.out = (char*)out->ptr; a.out_img = out->img_stride; a.out_row = out->row_stride; a.out_ld = out->ld;
    a.w = (const char*)w;
    a.LH = LH; a.LW = LW;
    a.lgLW = 0;
    while ((1 << a.lgLW) < LW) ++a.lgLW;
    a.TH = p.TH; a.strips_per_img = LH / p.TH; a.nstrips = N * a.strips_per_img;
    a.RH = p.RH; a.RW = p.RW; a.PB = p.PB;
    a.inv_RW = 1.0f / (float)p.RW;
    a.stat_part = op == P2P_OP_P ? stat_part : nullptr;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)p.blocks);
    static bool attr_done = false;
    if (!attr_done)
        attr_done = (int)p2p_allow_lds((const void*)conv_strip_kernel<0>, 160 * 1024, "conv_strip_kernel<0>") &
                    (int)p2p_allow_lds((const void*)conv_strip_kernel<1>, 160 * 1024, "conv_strip_kernel<1>");
    if (op == P2P_OP_P) conv_strip_kernel<1><<<grid, dim3(CS_THREADS), p.shm, st>>>(a);
    else conv_strip_kernel<0><<<grid, dim3(CS_THREADS), p.shm, st>>>(a);
    return p2p_check_launch("p2p_conv_strip");
}
