// MFMA weight gradient of every 4x4 convolution (tape.gradient wrt the Conv2D / Conv2DTranspose kernels,
// pix2pix_model.py:78-79 over networks.py:10-16,26-27,46-48,75-78), s = stride:
//     dW[t][g][d] = sum_{m=(n,y,x)} hi[n, s*y+kh-1, s*x+kw-1, g] * lo[n, y, x, d]          t = (kh,kw)
// A "TN" GEMM: both operands have the reduction index (the pixel) as their slow dimension.  Tiles are
// staged [pixel][channel] exactly as they sit in HBM (global_load_lds_dwordx4, 16 B per lane); for bf16
// the MFMA operand (8 consecutive pixels of one channel per lane) is produced by the CDNA4 transposing LDS
// read ds_read_b64_tr_b16, for f32 (v_mfma_f32_32x32x2_f32, one pixel per lane) by a plain ds_read_b32.
// One workgroup = one tap x BG x BD output tile x one chunk of the pixel range; partial tiles go to f32
// slabs [msplit][16][Cg][Cd] and are summed in a fixed order (deterministic, no float atomics).
// Channel counts that are not tile multiples (edge layers: 4/8/36/33 channels, 1..4 output channels): the
// tile-row slots beyond the pixel's bytes re-read the pixel's first 16-byte chunk (never another pixel, never
// past the view) and the store is masked to g < Cg, d < Cd.
#include "p2p_common.hpp"
#include <stdlib.h>

struct WgemmArgs {
    const char* hi; long long hi_img; int hi_row; int hi_ld;
    const char* lo; long long lo_img; int lo_row; int lo_ld;
    float* part;            // [msplit][16][Cg][Cd]
    int M, LW, LH, lgLW, lgLH;
    int Cg, Cd;             // real channel counts (store mask, output strides)
    int stride;
    int chunk;              // pixels per workgroup (multiple of BK)
    int live_taps;          // 1: 1x1 lo map (2x2 hi map): only the 4 centre taps meet real pixels, the others write exact zeros
};

__device__ __forceinline__ void glds16w(const char* g, char* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// bank swizzle of a [pixel][RB bytes] tile for the transposing read: 64-byte column groups are XORed with
// a row-dependent value so that the 4 rows of one ds_read_b64_tr_b16 block fall on different banks.
template <int RB>
__device__ __forceinline__ int swz_group(int row) {
    constexpr int G = RB / 64;                 // 64-byte groups per row
    constexpr int R = RB >= 256 ? 1 : 256 / RB;  // rows per 256-byte bank line
    if (G == 1) return 0;
    return (row / R) & (G - 1) & 3;
}

template <typename T, int BG, int BD, int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(WM * WN * 64) void wgemm_kernel(WgemmArgs a) {
    constexpr int NW = WM * WN;
    constexpr int ESZ = sizeof(T);
    constexpr int BK = ESZ == 2 ? 64 : 32;          // pixels per stage
    constexpr int RBA = BG * ESZ, RBB = BD * ESZ;   // row bytes
    constexpr int A_BYTES = BK * RBA, B_BYTES = BK * RBB, STAGE = A_BYTES + B_BYTES;
    constexpr int NIA = A_BYTES / 1024 / NW, NIB = B_BYTES / 1024 / NW;   // glds instructions per wave
    constexpr int LPA = RBA / 16, LPB = RBB / 16;   // lanes (16-byte chunks) per row
    static_assert(NIA >= 1 && NIB >= 1, "tile too small");
    static_assert(WM * TM * 32 == BG && WN * TN * 32 == BD, "wave tiling must cover the block tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    // logical block order: (tap, g tile) fastest, then d tile, then pixel chunk -- all tiles of one pixel chunk read the
    // same pixels, so they are kept on one XCD (measured before the remap: 7x the algorithmic HBM bytes, r01 PMC run)
    const unsigned nblk = gridDim.x * gridDim.y * gridDim.z;
    const unsigned lin = xcd_remap(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), nblk);
    const int bx = lin % gridDim.x, by = (lin / gridDim.x) % gridDim.y, bz = lin / (gridDim.x * gridDim.y);
    const int gtiles = (a.Cg + BG - 1) / BG;
    const int t = bx / gtiles, g0 = (bx % gtiles) * BG;
    const int d0 = by * BD;
    const int kh = t >> 2, kw = t & 3;
    const int mbeg = bz * a.chunk;
    const int mend = min(mbeg + a.chunk, a.M);
    const bool dead_tap = a.live_taps && !(kh >= 1 && kh <= 2 && kw >= 1 && kw <= 2);     // sums over the zero halo only
    const int nst = (mend > mbeg && !dead_tap) ? (mend - mbeg + BK - 1) / BK : 0;
    const int s = a.stride;

    const int hi_img32 = (int)a.hi_img, lo_img32 = (int)a.lo_img;
    const int hi_pixB = a.hi_ld * ESZ, lo_pixB = a.lo_ld * ESZ;
    auto stage = [&](int st, char* buf) {
        const int mb = mbeg + st * BK;
#pragma unroll
        for (int i = 0; i < NIA; ++i) {
            int inst = i * NW + wave;
            int row = inst * (1024 / RBA) + lane / LPA;       // pixel within the stage
            int sl = lane % LPA;                               // physical 16-byte slot
            int lg = ((sl >> 2) ^ (ESZ == 2 ? swz_group<RBA>(row) : 0));
            int chunk = (lg << 2) | (sl & 3);
            int m = min(mb + row, a.M - 1);                    // clamped: the lo row of m >= M is a zero halo row
            int x = m & (a.LW - 1), y = (m >> a.lgLW) & (a.LH - 1), n = m >> (a.lgLW + a.lgLH);
            // 32-bit pixel index (views of < 2^31 pixels, checked on the host), then ONE 64-bit multiply-add: the 64-bit
            // products of the first version made the staging VALU-bound (25 % of the wave cycles, r01 SQ counters)
            const int pix = n * hi_img32 + (s * y + kh - 1) * a.hi_row + (s * x + kw - 1);
            // a tile row wider than the pixel (edge layers) must not leave the pixel: those slots only feed masked outputs,
            // so they re-read the pixel's first chunk (in bounds whatever lies behind the view; r01 read on into the
            // following pixels and, for the last pixel of a small map, past the allocation)
            int coff = g0 * ESZ + chunk * 16;
            if (coff >= hi_pixB) coff = 0;
            const char* src = a.hi + (long long)pix * hi_pixB + coff;
            glds16w(src, buf + inst * 1024);
        }
#pragma unroll
        for (int i = 0; i < NIB; ++i) {
            int inst = i * NW + wave;
            int row = inst * (1024 / RBB) + lane / LPB;
            int sl = lane % LPB;
            int lg = ((sl >> 2) ^ (ESZ == 2 ? swz_group<RBB>(row) : 0));
            int chunk = (lg << 2) | (sl & 3);
            int m = mb + row;
            const char* src;
            int coff = d0 * ESZ + chunk * 16;
            if (coff >= lo_pixB) coff = 0;
            if (m < mend) {
                int x = m & (a.LW - 1), y = (m >> a.lgLW) & (a.LH - 1), n = m >> (a.lgLW + a.lgLH);
                const int pix = n * lo_img32 + y * a.lo_row + x;
                src = a.lo + (long long)pix * lo_pixB + coff;
            } else {
                src = a.lo + ((long long)(-1) * a.lo_row - 1) * lo_pixB + coff;   // halo row -1: zeros
            }
            glds16w(src, buf + A_BYTES + inst * 1024);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    if (nst > 0) stage(0, smem);
    for (int st = 0; st < nst; ++st) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        char* cur = smem + (st & 1) * STAGE;
        if (st + 1 < nst) stage(st + 1, smem + ((st + 1) & 1) * STAGE);
        if constexpr (ESZ == 2) {
            // lane l: 16-lane group grp, li = l&15, q = li>>2, p = li&3.  Read rd of k-step kk covers rows
            // kk*16 + 8*(grp>>1) + 4*rd + q, columns c0 + 16*(grp&1) + 4p .. +3 and returns to lane li the
            // column c0 + 16*(grp&1) + li at those 4 rows  ==  MFMA operand element j = 4*rd + e, k = 8*(l>>5) + j.
            const int grp = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
#pragma unroll
            for (int kk = 0; kk < BK / 16; ++kk) {
                bf16x8 af[TM], bf[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    int cb = ((wm * TM + i) * 32 + 16 * (grp & 1) + 4 * p) * 2;     // byte column
                    s16x4 r[2];
#pragma unroll
                    for (int rd = 0; rd < 2; ++rd) {
                        int row = kk * 16 + 8 * (grp >> 1) + 4 * rd + q;
                        int off = row * RBA + ((((cb >> 6) ^ swz_group<RBA>(row)) << 6) | (cb & 63));
                        r[rd] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(cur + off));
                    }
                    union { s16x4 h[2]; bf16x8 v; } u;
                    u.h[0] = r[0]; u.h[1] = r[1];
                    af[i] = u.v;
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    int cb = ((wn * TN + j) * 32 + 16 * (grp & 1) + 4 * p) * 2;
                    s16x4 r[2];
#pragma unroll
                    for (int rd = 0; rd < 2; ++rd) {
                        int row = kk * 16 + 8 * (grp >> 1) + 4 * rd + q;
                        int off = row * RBB + ((((cb >> 6) ^ swz_group<RBB>(row)) << 6) | (cb & 63));
                        r[rd] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(cur + A_BYTES + off));
                    }
                    union { s16x4 h[2]; bf16x8 v; } u;
                    u.h[0] = r[0]; u.h[1] = r[1];
                    bf[j] = u.v;
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
        } else {
            const int kl = lane >> 5, cl = lane & 31;
#pragma unroll 4
            for (int kk = 0; kk < BK / 2; ++kk) {
                float af[TM], bf[TN];
                int row = kk * 2 + kl;
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = *(const float*)(cur + row * RBA + ((wm * TM + i) * 32 + cl) * 4);
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[j] = *(const float*)(cur + A_BYTES + row * RBB + ((wn * TN + j) * 32 + cl) * 4);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
        }
    }

    // D[row = g][col = d]
    float* outp = a.part + ((long long)bz * 16 + t) * a.Cg * a.Cd;
    const int h = lane >> 5;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            int g = g0 + (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (g >= a.Cg) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                int d = d0 + (wn * TN + j) * 32 + (lane & 31);
                if (d < a.Cd) outp[(long long)g * a.Cd + d] = acc[i][j][e];
            }
        }
}

// out = sum of the msplit partial slabs, fixed order; four outputs per lane (n = 16 * Cg * Cd is a multiple of 16)
__global__ void slab_sum_kernel(const float* __restrict__ part, int nslabs, long long n, float* __restrict__ out) {
    long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const long long stride = (long long)gridDim.x * blockDim.x * 4;
    for (; i < n; i += stride) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < nslabs; ++k) s += *(const f32x4*)(part + (long long)k * n + i);
        *(f32x4*)(out + i) = s;
    }
}

static int ilog2_exact_w(long long v) {
    int l = 0;
    while ((1LL << l) < v) ++l;
    return (1LL << l) == v ? l : -1;
}

extern "C" long long p2p_wgemm_workspace_bytes(int N, int LH, int LW, int Cg, int Cd, int msplit) {
    (void)N; (void)LH; (void)LW;
    if (msplit <= 1) return 0;
    return (long long)msplit * 16 * Cg * Cd * (long long)sizeof(float);
}

template <typename T, int BG, int BD, int WM, int WN, int TM, int TN>
static void wgemm_go(WgemmArgs& a, int msplit, hipStream_t st) {
    constexpr int ESZ = sizeof(T);
    constexpr int BK = ESZ == 2 ? 64 : 32;
    dim3 grid(16 * ((a.Cg + BG - 1) / BG), (a.Cd + BD - 1) / BD, msplit);
    wgemm_kernel<T, BG, BD, WM, WN, TM, TN><<<grid, dim3(WM * WN * 64), 2 * BK * (BG + BD) * ESZ, st>>>(a);
}

template <typename T>
static int wgemm_launch(WgemmArgs& a, int msplit, hipStream_t st) {
    constexpr int ESZ = sizeof(T);
    constexpr int BK = ESZ == 2 ? 64 : 32;
    a.chunk = ((a.M + msplit - 1) / msplit + BK - 1) / BK * BK;
    const int cg = a.Cg, cd = a.Cd;
    if (cd > 64) {
        static int w8 = -1;
        if (w8 < 0) { const char* e = getenv("P2P_WGEMM_W8"); w8 = e ? atoi(e) : 1; }     // eight waves (32x64 each) per 128x128 tile: -6.5 % on the kernel (r02 A/B), as in p2p_igemm
        if (cg > 64 && w8) wgemm_go<T, 128, 128, 4, 2, 1, 2>(a, msplit, st);
        else if (cg > 64) wgemm_go<T, 128, 128, 2, 2, 2, 2>(a, msplit, st);
        else if (cg > 32) wgemm_go<T, 64, 128, 2, 2, 1, 2>(a, msplit, st);
        else wgemm_go<T, 32, 128, 1, 4, 1, 1>(a, msplit, st);
    } else if (cd > 32) {
        if (cg > 32) wgemm_go<T, 64, 64, 2, 2, 1, 1>(a, msplit, st);
        else wgemm_go<T, 32, 64, 1, 2, 1, 1>(a, msplit, st);
    } else {
        if (cg > 32) wgemm_go<T, 64, 32, 2, 1, 1, 1>(a, msplit, st);
        else wgemm_go<T, 32, 32, 1, 1, 1, 1>(a, msplit, st);
    }
    return p2p_check_launch("p2p_wgemm");
}

static int wgemm_common(int dtype, int stride, int N, int LH, int LW, int Cg, int Cd, const p2p_tensor* hi,
                        const p2p_tensor* lo, float* dw, int msplit, void* workspace, void* stream) {
    P2P_REQUIRE(N > 0 && LH > 0 && LW > 0 && Cg > 0 && Cd > 0, "p2p_wgemm: bad shape");
    P2P_REQUIRE(hi && lo && hi->ptr && lo->ptr && dw, "p2p_wgemm: null pointer");
    P2P_REQUIRE(msplit >= 1 && (msplit == 1 || workspace), "p2p_wgemm: msplit > 1 needs a workspace");
    const int esz = dtype == P2P_BF16 ? 2 : 4;
    P2P_REQUIRE((hi->ld * esz) % 16 == 0 && (lo->ld * esz) % 16 == 0 && ((uintptr_t)hi->ptr % 16) == 0 && ((uintptr_t)lo->ptr % 16) == 0,
                "p2p_wgemm: pixels must be 16-byte aligned (pad the channel count of the view)");
    P2P_REQUIRE((long long)N * hi->img_stride < (1LL << 31) && (long long)N * lo->img_stride < (1LL << 31),
                "p2p_wgemm: views of 2^31 pixels or more are not supported");
    WgemmArgs a;
    a.hi = (const char*)hi->ptr; a.hi_img = hi->img_stride; a.hi_row = hi->row_stride; a.hi_ld = hi->ld;
    a.lo = (const char*)lo->ptr; a.lo_img = lo->img_stride; a.lo_row = lo->row_stride; a.lo_ld = lo->ld;
    a.M = N * LH * LW; a.LW = LW; a.LH = LH;
    a.lgLW = ilog2_exact_w(LW); a.lgLH = ilog2_exact_w(LH);
    P2P_REQUIRE(a.lgLW >= 0 && a.lgLH >= 0, "p2p_wgemm: LH=%d, LW=%d must be powers of two", LH, LW);
    a.Cg = Cg; a.Cd = Cd; a.stride = stride;
    a.live_taps = (stride == 2 && LH == 1 && LW == 1) ? 1 : 0;
    a.part = msplit == 1 ? dw : (float*)workspace;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    P2P_DISPATCH_DTYPE(dtype, rc = wgemm_launch<T>(a, msplit, st));
    if (rc) return rc;
    if (msplit > 1) {
        long long n = 16LL * Cg * Cd;
        long long blocks = (n / 4 + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        slab_sum_kernel<<<dim3((unsigned)blocks), 256, 0, st>>>((const float*)workspace, msplit, n, dw);
        return p2p_check_launch("p2p_wgemm reduce");
    }
    return 0;
}

extern "C" int p2p_wgemm(int dtype, int N, int LH, int LW, int Cg, int Cd, const p2p_tensor* hi, const p2p_tensor* lo,
                         float* dw, int msplit, void* workspace, void* stream) {
    P2P_REQUIRE(Cg % 32 == 0 && Cd % 32 == 0, "p2p_wgemm: need Cg %% 32 == 0 and Cd %% 32 == 0 (Cg=%d, Cd=%d); use p2p_wgemm_edge", Cg, Cd);
    return wgemm_common(dtype, 2, N, LH, LW, Cg, Cd, hi, lo, dw, msplit, workspace, stream);
}

extern "C" int p2p_wgemm_edge(int dtype, int stride, int N, int LH, int LW, int Cg, int Cd, const p2p_tensor* hi,
                              const p2p_tensor* lo, float* dw, int msplit, void* workspace, void* stream) {
    P2P_REQUIRE(stride == 1 || stride == 2, "p2p_wgemm_edge: stride must be 1 or 2");
    return wgemm_common(dtype, stride, N, LH, LW, Cg, Cd, hi, lo, dw, msplit, workspace, stream);
}
