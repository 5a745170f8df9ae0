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
#include <utility>

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

// ---------------------------------------------------------------------------------------------------------------------------------
// Software-pipelined form (round 4) for the deep bf16 layers (Cg, Cd multiples of 128; whole 64-pixel K-blocks).  The kernel
// above re-derives every pixel's address per 16-byte slot per stage (~25 vector instructions per LDS-DMA piece: as much vector
// issue time as the stage's MFMAs take), reads its fragments and waits for them in front of every pair of MFMAs, and gives a
// wave a 32x64 tile (1.5 KB of LDS reads per MFMA).  Here:
//   * a stage is 64 consecutive pixels that share ONE wave-uniform base: 64 divides the map or the map divides 64, so
//     pixel = (image n0 + nl, row y0 + dy, column x0 + x) with (nl, dy, x) fixed per lane -- an LDS-DMA piece costs scalar
//     arithmetic only (p2p_glds16_sv: base in scalar registers, 32-bit lane offset);
//   * 128x128 tile of one tap, eight waves = two K groups x (2 x 2) waves of 64x64 (1 KB of LDS reads per MFMA); group kg takes
//     the 16-pixel k-steps 2 kg, 2 kg + 1 of every stage, the groups exchange half of their accumulators through LDS at the end
//     (fixed order: deterministic); the second wave of every SIMD comes from inside the workgroup, so a launch needs half the
//     pixel split (msplit) and half the f32 partial slabs to fill the chip;
//   * ring of four 32 KB stages, three in flight behind counted s_waitcnt vmcnt; per stage and wave: MFMAs of k-step 0 | barrier |
//     transposing reads of the next stage's two fragment sets | MFMAs of k-step 1 with the LDS-DMA pieces between them.
template <int... I, typename F>
__device__ __forceinline__ void wgemm_static_for(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}

// one MFMA operand (8 consecutive pixels of a channel per lane) = two transposing reads 4 pixel rows (4 * 256 bytes) apart
template <int OFF>
__device__ __forceinline__ bf16x8 wgemm_tr8(unsigned addr) {
    union { s16x4 h[2]; bf16x8 v; } u;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(u.h[0]) : "v"(addr), "n"(OFF));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(u.h[1]) : "v"(addr), "n"(OFF + 1024));
    return u.v;
}

template <int NST>
__global__ __launch_bounds__(512) void wgemm_pipe_kernel(WgemmArgs a) {
    constexpr int BG = 128, BD = 128, BK = 64, NW = 8, NWG = 4, WN = 2, TM = 2, TN = 2, NMF = TM * TN;
    constexpr int RB = 256;                                       // bytes per pixel row of a tile (128 bf16 channels)
    constexpr int A_BYTES = BK * RB, STAGE = 2 * A_BYTES;         // 16 KB + 16 KB
    constexpr int NL = 4;                                         // LDS-DMA pieces per wave per stage: 2 of hi, 2 of lo
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = wave / NWG, wv = wave % NWG, wm = wv / WN, wn = wv % WN;
    const unsigned nblk = gridDim.x * gridDim.y * gridDim.z;
    const unsigned lin = xcd_remap(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), nblk);
    const int bx = lin % gridDim.x, by = (lin / gridDim.x) % gridDim.y, bz = lin / (gridDim.x * gridDim.y);
    const int gtiles = a.Cg / BG;
    const int t = bx / gtiles, g0 = (bx % gtiles) * BG, d0 = by * BD;
    const int kh = t >> 2, kw = t & 3;
    const int mbeg = bz * a.chunk, mend = min(mbeg + a.chunk, a.M);
    const bool dead_tap = a.live_taps && !(kh >= 1 && kh <= 2 && kw >= 1 && kw <= 2);     // sums over the zero halo only
    const int nit = (mend > mbeg && !dead_tap) ? (mend - mbeg) / BK : 0;                 // launcher: chunk and M are multiples of 64
    const int s = a.stride;
    const int lgHW = a.lgLW + a.lgLH, HW = 1 << lgHW;
    const long long hi_pixB = a.hi_ld * 2, lo_pixB = a.lo_ld * 2;

    // ---- staging: lane-constant 32-bit offsets; piece `inst` = 4 pixels x 256 bytes, lane = (pixel in piece, 16-byte slot) -------------
    unsigned hoff[2], loff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int inst = i * NW + wave;
        const int q = inst * 4 + (lane >> 4), sl = lane & 15;
        const int chunk = (((sl >> 2) ^ (q & 3)) << 2) | (sl & 3);            // swz_group<256>(row) = row & 3
        const int nl = q >> lgHW, pq = q & (HW - 1), dy = pq >> a.lgLW, x = pq & (a.LW - 1);
        hoff[i] = (unsigned)(((long long)nl * a.hi_img + (long long)s * dy * a.hi_row + s * x) * hi_pixB + chunk * 16);
        loff[i] = (unsigned)(((long long)nl * a.lo_img + (long long)dy * a.lo_row + x) * lo_pixB + chunk * 16);
    }
    const unsigned smem32 = p2p_lds32(smem);
    const char* s_hi;                   // wave-uniform bases of the stage being requested
    const char* s_lo;
    auto stage_prepare = [&](int it) {
        const int mb = mbeg + it * BK;
        const int n0 = mb >> lgHW, pb = mb & (HW - 1), y0 = pb >> a.lgLW, x0 = pb & (a.LW - 1);
        s_hi = a.hi + ((long long)n0 * a.hi_img + (long long)(s * y0 + kh - 1) * a.hi_row + (s * x0 + kw - 1)) * hi_pixB + g0 * 2;
        s_lo = a.lo + ((long long)n0 * a.lo_img + (long long)y0 * a.lo_row + x0) * lo_pixB + d0 * 2;
    };
    auto stage_piece = [&](unsigned dst, auto pc) {
        constexpr int p = decltype(pc)::value;
        if constexpr (p < 2) p2p_glds16_sv(s_hi, hoff[p], dst + (p * NW + wave) * 1024);
        else p2p_glds16_sv(s_lo, loff[p - 2], dst + A_BYTES + ((p - 2) * NW + wave) * 1024);
    };
    auto stage_all = [&](int it, unsigned dst) {
        stage_prepare(it);
        wgemm_static_for(std::make_integer_sequence<int, NL>{}, [&](auto pc) { stage_piece(dst, pc); });
    };
    auto vm_wait_stages = [&](int n) {      // all but the n youngest stages have landed (n wave-uniform)
        if (NST >= 4 && n >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NL) : "memory");
        else if (NST >= 3 && n == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // ---- fragment reads (ds_read_b64_tr_b16): lane l, 16-lane group grp, li = l & 15, q4 = li >> 2, p = li & 3.  Read rd of k-step kk
    // covers rows kk*16 + 8*(grp>>1) + 4*rd + q4, byte columns cb .. cb+7 with cb = (tile*32 + 16*(grp&1) + 4p)*2; the 64-byte column
    // group is XORed with row & 3 = q4 (the same for every kk and rd), so kk and rd are plain byte offsets.
    const int grp = lane >> 4, li = lane & 15, q4 = li >> 2, p4 = li & 3;
    const int rbase = (8 * (grp >> 1) + q4) * RB;
    unsigned abase[TM];
    unsigned bbase[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int cb = ((wm * TM + i) * 32 + 16 * (grp & 1) + 4 * p4) * 2;
        abase[i] = smem32 + rbase + ((((cb >> 6) ^ q4) << 6) | (cb & 63)) + kg * (2 * 16 * RB);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int cb = ((wn * TN + j) * 32 + 16 * (grp & 1) + 4 * p4) * 2;
        bbase[j] = smem32 + A_BYTES + rbase + ((((cb >> 6) ^ q4) << 6) | (cb & 63)) + kg * (2 * 16 * RB);
    }
    bf16x8 fa[4][TM], fb[4][TN];       // sets {0, 1} and {2, 3} alternate between stages; set = one 16-pixel k-step
    // (inline asm + counted waits: with the builtin hipcc put s_waitcnt lgkmcnt(0) between the first and the second deferred MFMA of
    // every other stage -- right behind the sixteen reads it had just issued for the NEXT stage)
    auto load_set = [&](auto sc, auto kc, unsigned bufoff) {       // set sc <- k-step kc (0 / 1 of this group) of the stage at bufoff
        constexpr int s4 = decltype(sc)::value, k2 = decltype(kc)::value;
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[s4][i] = wgemm_tr8<k2 * 16 * RB>(abase[i] + bufoff);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[s4][j] = wgemm_tr8<k2 * 16 * RB>(bbase[j] + bufoff);
    };
    auto mfma_set = [&](auto sc) {
        constexpr int s4 = decltype(sc)::value;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s4][i], fb[s4][j], acc[i][j], 0, 0, 0);
    };
    auto mfma_dma = [&](auto sc, const bool more, unsigned dst) {   // MFMAs of one set, an LDS-DMA piece behind each (unconditional MFMAs)
        constexpr int s4 = decltype(sc)::value;
        wgemm_static_for(std::make_integer_sequence<int, NMF>{}, [&](auto mc) {
            constexpr int m = decltype(mc)::value, i = m / TN, j = m % TN;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s4][i], fb[s4][j], acc[i][j], 0, 0, 0);
            if constexpr (m < NL) { if (more) stage_piece(dst, mc); }
        });
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;

#pragma unroll
    for (int st = 0; st < NST - 1; ++st)
        if (st < nit) stage_all(st, smem32 + st * STAGE);
    vm_wait_stages(min(NST - 2, nit - 1));
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (NST - 1 < nit) stage_all(NST - 1, smem32 + (NST - 1) * STAGE);
    auto body = [&](auto pc, int it) {
        constexpr int P = decltype(pc)::value;
        using X0 = std::integral_constant<int, 2 * P>; using X1 = std::integral_constant<int, 2 * P + 1>;
        using Y0 = std::integral_constant<int, 2 - 2 * P>; using Y1 = std::integral_constant<int, 3 - 2 * P>;
        const unsigned nbuf = smem32 + (it % NST) * STAGE;
        const bool more = it + NST < nit;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * (TM + TN)) : "memory");      // set X0 has landed (X1's eight reads behind it)
        __builtin_amdgcn_sched_barrier(0);
        mfma_set(X0{});
        __builtin_amdgcn_sched_barrier(0);
        if (more) stage_prepare(it + NST);
        vm_wait_stages(min(NST - 2, nit - 2 - it));                 // stage it + 1 has landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // both sets of this stage are in registers: nobody reads it any more
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (it + 1 < nit) {
            const unsigned nxt = ((it + 1) % NST) * STAGE;
            load_set(Y0{}, I0{}, nxt);
            load_set(Y1{}, I1{}, nxt);
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma_dma(X1{}, more, nbuf);
    };
    if (nit > 0) {
        load_set(I0{}, I0{}, 0);
        load_set(I1{}, I1{}, 0);
    }
    for (int it = 0; it < nit; it += 2) {
        body(I0{}, it);
        if (it + 1 < nit) body(I1{}, it + 1);
    }
    __builtin_amdgcn_sched_barrier(0);
    // (no fragment read or LDS-DMA piece is outstanding after the last stage; said explicitly because the swap below re-uses the
    // fragment registers and nothing interlocks a vector write against a pending LDS return: tools/exp/asm_checks.py)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);

    // ---- K groups: every wave sends its odd fragment row and keeps the even one (group 1 swaps first: static register indices) -------
    if (kg) {
#pragma unroll
        for (int j = 0; j < TN; ++j) { const f32x16 tt = acc[0][j]; acc[0][j] = acc[1][j]; acc[1][j] = tt; }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    {
        const int slot = ((1 - kg) * NWG + wv) * TN;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v = {acc[1][j][4 * q], acc[1][j][4 * q + 1], acc[1][j][4 * q + 2], acc[1][j][4 * q + 3]};
                *(f32x4*)(smem + (slot + j) * 4096 + q * 1024 + lane * 16) = v;
            }
    }
    __syncthreads();
    {
        const int slot = (kg * NWG + wv) * TN;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = *(const f32x4*)(smem + (slot + j) * 4096 + q * 1024 + lane * 16);
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[0][j][4 * q + k] += v[k];
            }
    }
    // D[row = g][col = d]: this wave owns logical fragment row kg of its 64-row block
    float* outp = a.part + ((long long)bz * 16 + t) * a.Cg * a.Cd;
    const int h = lane >> 5;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int g = g0 + (wm * TM + kg) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int d = d0 + (wn * TN + j) * 32 + (lane & 31);
            outp[(long long)g * a.Cd + d] = acc[0][j][e];
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

static int wgemm_pipe_on() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("P2P_WGEMM_PIPE"); v = e ? atoi(e) : 1; }
    return v;
}

// The pipelined kernel takes bf16 launches with 128-multiple channel counts, whole 64-pixel stages per workgroup and stages that
// share one base (64 divides the map's pixels with rows no wider than 64, or the map divides 64); views below 4 GB.
static bool wgemm_pipe_ok(const WgemmArgs& a, int esz, int msplit) {
    if (!wgemm_pipe_on() || esz != 2 || a.Cg % 128 || a.Cd % 128 || a.M % 64 || a.chunk % 64) return false;
    // One workgroup per CU (128 KB of LDS, 138 VGPRs: a two-stage form for two per CU spilled inside the loop).  Launches with two
    // workgroups per CU worth of tiles stay with the kernel above, whose 64 KB workgroups share a CU (up2: 33.5 vs 39.0 us, r04)
    static int cap = -1;
    if (cap < 0) { const char* e = getenv("P2P_WGEMM_PIPE_MAXWG"); cap = e ? atoi(e) : 448; }
    if (16LL * (a.Cg / 128) * (a.Cd / 128) * msplit > cap) return false;
    const int hw = a.LH * a.LW;
    if (!((hw % 64 == 0 && a.LW <= 64) || 64 % hw == 0)) return false;
    const long long hspan = ((long long)(a.M / hw + 64) * a.hi_img) * a.hi_ld * 2, lspan = ((long long)(a.M / hw + 64) * a.lo_img) * a.lo_ld * 2;
    return hspan < 0xffffffffLL && lspan < 0xffffffffLL;
}

template <typename T>
static int wgemm_launch(WgemmArgs& a, int msplit, hipStream_t st) {
    constexpr int ESZ = sizeof(T);
    constexpr int BK = ESZ == 2 ? 64 : 32;
    a.chunk = ((a.M + msplit - 1) / msplit + BK - 1) / BK * BK;
    const int cg = a.Cg, cd = a.Cd;
    if (wgemm_pipe_ok(a, ESZ, msplit)) {
        static bool attr = false;
        if (!attr) attr = p2p_allow_lds((const void*)wgemm_pipe_kernel<4>, 160 * 1024, "wgemm_pipe_kernel");
        dim3 grid(16 * (cg / 128), cd / 128, msplit);
        wgemm_pipe_kernel<4><<<grid, dim3(512), 4 * 32768, st>>>(a);
        return p2p_check_launch("p2p_wgemm");
    }
    if (cd > 64) {
        // eight waves (32x64 each) per 128x128 tile: -6.5 % against four 64x64 waves in this (unpipelined) kernel, r02
        if (cg > 64) wgemm_go<T, 128, 128, 4, 2, 1, 2>(a, msplit, st);
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
