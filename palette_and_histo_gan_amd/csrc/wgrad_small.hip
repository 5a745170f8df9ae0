// Weight gradient of the edge layers (networks.py:46-48,57,75-78: 4/8 -> 64, 36/33 -> 4, 64 -> 1 channels):
//     dW[t][g][d] = sum_{n,y,x} hi[n, s*y+kh-1, s*x+kw-1, g] * lo[n, y, x, d]
// The output is tiny (1 K - 8 K values) and the reduction runs over up to 2^20 pixels, so this is a streaming
// reduction, HBM-bound in principle.  The general kernel (wgemm.hip) re-stages the pixels once per tap; here one
// workgroup owns a strip of output rows of one image, brings the hi strip (with its halo) and the lo strip into LDS
// ONCE and contracts all 16 taps out of LDS: bytes from HBM = the algorithmic bytes.
//   * 16 waves, wave w owns tap w (or 8 waves with taps 2w, 2w+1) and all GT x DT 32x32 MFMA tiles of it;
//   * bf16: MFMA operands come from ds_read_b64_tr_b16 with PER-LANE row addresses (the 4 "rows" of a transposed
//     read are the 4 pixels s*x+kw-1 apart in the strip, so stride-2 gathers cost nothing);
//     f32: ds_read_b32 (one pixel per lane, v_mfma_f32_32x32x2_f32, exact);
//   * channel counts below 32 read past the pixel into the next pixel's bytes (finite), masked at the store;
//   * each workgroup accumulates over several strips and writes one f32 partial; a fixed-order reduction follows
//     (deterministic, no float atomics).
#include "p2p_common.hpp"

struct WsArgs {
    const char* hi; long long hi_img; int hi_row; int hi_ld;
    const char* lo; long long lo_img; int lo_row; int lo_ld;
    float* part;                   // [nblocks][16][Cg][Cd]
    int N, LH, LW, Cg, Cd;
    int TH;                        // lo rows per strip
    int strips_per_img, nstrips;   // LH / TH, N * strips_per_img
    int nbuf, buf_bytes;           // 1 or 2 strip buffers of buf_bytes each (hi strip | lo strip)
    int hrowB, lo_off;             // LDS bytes per hi strip row (a multiple of 256 when swizzled), offset of the lo strip in a buffer
    int abl;                       // diagnostics (P2P_WS_ABL): 1 = staging only, 2 = contraction only (results wrong)
    int mh, ml;                    // bank swizzle masks of the hi / lo strip (0 = natural order), see ws_swz
    int pack;                      // 0: one tap per MFMA tile row block.  Few-channel sides (8-channel = 16-byte pixels, bf16):
                                   // 1: a hi tile holds 4 taps x 8 hi channels, 2: a lo tile holds 4 taps x 8 lo channels
};

__device__ __forceinline__ void glds16s(const char* g, char* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// Bank swizzle of a strip.  One ds_read_b64_tr_b16 lane group (32 lanes) reads 4 pixels x 64 bytes; the pixels are
// u = (pixel bytes x pixel step) / 64 LDS "quarters" (64 B = 16 banks) apart, so for u = 2 / 4 they share 2 / 1 of the 4
// quarters of the 256-byte bank line: 2- / 4-way conflicts (r02 PMC: 60 % of this kernel's LDS cycles were conflict cycles).
// The quarter index (byte bits 6-7) is therefore XORed with the next bits (8-9) masked by m = u - 1 (checked exhaustively
// for every start pixel: conflict-free for (64 B, step 2) m=1, (128, 2) m=3, (128, 1) m=1, (256, 1) m=3).  The LDS-DMA
// writes lane-linear, so the stager applies the same involution to the SOURCE chunk index.
__device__ __forceinline__ int ws_swz(int b, int m) { return b ^ (((b >> 8) & m) << 6); }

template <typename T, int S, int GT, int DT, int TPW>
__global__ __launch_bounds__(1024 / TPW) __attribute__((amdgpu_waves_per_eu(TPW == 2 && GT * DT <= 2 ? 4 : 2)))
void wgrad_small_kernel(WsArgs a) {   // 8-wave variants with <= 2 tiles per tap: two workgroups per CU (<= 128 VGPRs)
    constexpr int NTHR = 1024 / TPW;                 // TPW taps per wave: 8 waves (TPW = 2) or 16 waves (TPW = 1)
    constexpr int ESZ = sizeof(T);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int TW = a.LW, TH = a.TH;
    const int RH = S * TH + 3, RW = S * TW + 3;                 // hi strip incl. halo, pixels
    const int hgB = a.hi_ld * ESZ, lgB = a.lo_ld * ESZ;         // pixel bytes in HBM (multiples of 16)
    // channel windows of this workgroup (blockIdx.y: d window, blockIdx.z: g window).  A pixel wider than the window is
    // staged as its window only (LDS pixel stride = window bytes); a narrower pixel is staged whole and over-read.
    const int g0 = blockIdx.z * 32 * GT, d0 = blockIdx.y * 32 * DT;
    const int hpB = hgB > 64 * GT * (ESZ / 2) ? 64 * GT * (ESZ / 2) : hgB;     // LDS bytes per hi pixel
    const int lpB = lgB > 64 * DT * (ESZ / 2) ? 64 * DT * (ESZ / 2) : lgB;     // LDS bytes per lo pixel
    const int hrowB = a.hrowB;                                   // bytes per strip row in LDS
    const int mh = a.mh, ml = a.ml;
    char* hiL = smem;
    char* loL = smem + a.lo_off;                                 // behind the hi strip + slack: short pixels are over-read by up to 64 B

    f32x16 acc[TPW][GT][DT];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int i = 0; i < GT; ++i)
#pragma unroll
            for (int j = 0; j < DT; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[t][i][j][e] = 0.f;

    // Strips are double-buffered when the plan left room (a.nbuf == 2): the LDS-DMA of strip s+1 is issued before strip s is
    // contracted and waited for after it, so staging (HBM latency + issue) hides behind the MFMAs instead of alternating with them.
    const int buf_bytes = a.buf_bytes;
    auto stage_strip = [&](int strip, char* hiB, char* loB) {
        if (a.abl == 2) return;
        const int n = strip / a.strips_per_img, y0 = (strip % a.strips_per_img) * TH;
        // ---- the hi strip (rows s*y0-1 .. , columns -1 ..) and the lo strip, 16 bytes per lane -----------------
        const int hchunks_row = hrowB >> 4, hchunks = RH * hchunks_row, hcpp = hpB >> 4;
        if (a.pack) {
            // Packed (few-channel) layers stage whole pixels without a swizzle: a strip row in LDS is a verbatim copy of a
            // contiguous run of HBM bytes.  Rows are dealt to the waves; the row base is wave-uniform (scalar registers) and a
            // lane only adds its 16-byte slot -- no per-slot integer divisions (they bounded these layers' staging, r03).
            constexpr int NWV = NTHR / 64;
            const int LWq = a.pack == 2 ? TW + 3 : TW, LHq = a.pack == 2 ? TH + 3 : TH, org = a.pack == 2 ? -2 : 0;
            const int lcr = (LWq * lpB) >> 4;
            const char* hrow0 = a.hi + ((long long)n * a.hi_img + (long long)(S * y0 - 1) * a.hi_row - 1) * hgB;
            const char* lrow0 = a.lo + ((long long)n * a.lo_img + (long long)(y0 + org) * a.lo_row + org) * lgB;
            for (int r = wave; r < RH + LHq; r += NWV) {
                const bool isHi = r < RH;
                const int rr = isHi ? r : r - RH;
                const int cpr = isHi ? hchunks_row : lcr;
                const char* src = isHi ? hrow0 + (long long)rr * a.hi_row * hgB : lrow0 + (long long)rr * a.lo_row * lgB;
                char* dst = (isHi ? hiB : loB) + rr * cpr * 16;
                for (int c0 = 0; c0 < cpr; c0 += 64)
                    if (c0 + lane < cpr) glds16s(src + (c0 + lane) * 16, dst + c0 * 16);
            }
            return;
        }
        for (int cI = wave * 64; cI < hchunks; cI += NTHR) {
            int ci = cI + lane;
            if (ci < hchunks) {
                int rr = ci / hchunks_row, cc = ci - rr * hchunks_row;
                cc ^= ((cc >> 4) & mh) << 2;                        // swizzled strips: this slot holds the chunk ws_swz maps here
                int px = cc / hcpp, ch = cc - px * hcpp;            // pixel of the strip row, 16-byte chunk inside its window
                if (px >= RW) px = RW - 1;                          // row padding of a swizzled strip (never read)
                const char* src = a.hi + ((long long)n * a.hi_img + (long long)(S * y0 - 1 + rr) * a.hi_row - 1 + px) * hgB +
                                  (hpB == hgB ? 0 : g0 * ESZ) + ch * 16;
                glds16s(src, hiB + cI * 16);       // wave-uniform base + lane*16
            }
        }
        const int LWp = a.pack == 2 ? TW + 3 : TW, LHp = a.pack == 2 ? TH + 3 : TH, lo_org = a.pack == 2 ? -2 : 0;
        const int lchunks_row = (LWp * lpB) >> 4, lchunks = LHp * lchunks_row, lcpp = lpB >> 4;
        for (int cI = wave * 64; cI < lchunks; cI += NTHR) {
            int ci = cI + lane;
            if (ci < lchunks) {
                const int cs = ci ^ (((ci >> 4) & ml) << 2);
                int rr = cs / lchunks_row, cc = cs - rr * lchunks_row;
                int px = cc / lcpp, ch = cc - px * lcpp;
                const char* src = a.lo + ((long long)n * a.lo_img + (long long)(y0 + lo_org + rr) * a.lo_row + lo_org + px) * lgB +
                                  (lpB == lgB ? 0 : d0 * ESZ) + ch * 16;
                glds16s(src, loB + cI * 16);
            }
        }
    };
    int cur = 0;
    if ((int)blockIdx.x < a.nstrips) stage_strip(blockIdx.x, hiL, loL);
    for (int strip = blockIdx.x; strip < a.nstrips; strip += gridDim.x) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();      // this strip has landed; the other buffer (previous strip) is fully consumed
        if (a.nbuf == 2) {
            const int nxt = strip + gridDim.x;
            if (nxt < a.nstrips) stage_strip(nxt, smem + (cur ^ 1) * buf_bytes, smem + (cur ^ 1) * buf_bytes + (loL - hiL));
        }
        char* const hiC = smem + cur * buf_bytes;
        char* const loC = hiC + (loL - hiL);
        // ---- contract: k-steps of 16 (bf16) / 2 (f32) consecutive lo pixels of one row -------------------------------
        if (a.abl == 1) {
        } else if constexpr (ESZ == 2) {
            const int grp = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
            const int chB = (16 * (grp & 1) + 4 * p) * 2;            // byte offset of this lane's 4 channels inside a 32-channel tile
            if (a.pack) {
                // Packed taps (8 waves): the 32 channel slots of the few-channel operand's tile are 4 taps (kw = 0..3 of
                // kernel row kh = wave & 3) x 8 channels -- a lane's 4-channel chunk belongs to tap kw = 2 (grp & 1) + (p >> 1),
                // channels 4 (p & 1) .., and supplies that tap's pixel address to the transposing read.  The other operand
                // is read once per 16-pixel K step for all four taps; wave >> 2 picks its 32-channel tile.
                const int kh = wave & 3, tsel = wave >> 2;
                const int kw = 2 * (grp & 1) + (p >> 1);
                const int chP = 4 * (p & 1) * 2;
                for (int yy = 0; yy < TH; ++yy) {
                    for (int x0 = 0; x0 < TW; x0 += 16) {
                        s16x4 rh[2], rl[2];
#pragma unroll
                        for (int rd = 0; rd < 2; ++rd) {
                            const int x = x0 + 8 * (grp >> 1) + 4 * rd + q;
                            int offh, offl;
                            if (a.pack == 1) {       // K runs over lo pixels (yy, x); hi carries the taps
                                offh = ((S * yy + kh) * RW + S * x + kw) * hpB + chP;
                                offl = (yy * TW + x) * lpB + tsel * 64 + chB;
                            } else {                 // stride 1, K runs over hi INTERIOR pixels (yy, x); lo (haloed strip) carries the taps
                                offh = ((yy + 1) * RW + x + 1) * hpB + tsel * 64 + chB;
                                offl = ((yy - kh + 3) * (TW + 3) + (x - kw + 3)) * lpB + chP;
                            }
                            rh[rd] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(hiC + offh));
                            rl[rd] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(loC + offl));
                        }
                        union { s16x4 h[2]; bf16x8 v; } uh, ul;
                        uh.h[0] = rh[0]; uh.h[1] = rh[1];
                        ul.h[0] = rl[0]; ul.h[1] = rl[1];
                        acc[0][0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(uh.v, ul.v, acc[0][0][0], 0, 0, 0);
                    }
                }
            } else {
            // per-lane offsets of the reads of one 16-pixel K step at x0 = 0 of a row, swizzled once: a step of 16 pixels and a
            // row are multiples of 1024 bytes whenever a mask is set, so they do not change the swizzle bits
            int hb[TPW][GT][2], lb[DT][2];
#pragma unroll
            for (int rd = 0; rd < 2; ++rd) {
                const int xl = 8 * (grp >> 1) + 4 * rd + q;
#pragma unroll
                for (int t = 0; t < TPW; ++t)
#pragma unroll
                    for (int i = 0; i < GT; ++i) hb[t][i][rd] = ws_swz((S * xl + ((wave * TPW + t) & 3)) * hpB + i * 64 + chB, mh);
#pragma unroll
                for (int j = 0; j < DT; ++j) lb[j][rd] = ws_swz(xl * lpB + j * 64 + chB, ml);
            }
            for (int yy = 0; yy < TH; ++yy) {
                for (int x0 = 0; x0 < TW; x0 += 16) {
                    bf16x8 bfr[DT];
                    const int lbase = (yy * TW + x0) * lpB;
#pragma unroll
                    for (int j = 0; j < DT; ++j) {
                        s16x4 r[2];
#pragma unroll
                        for (int rd = 0; rd < 2; ++rd)
                            r[rd] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(loC + lbase + lb[j][rd]));
                        union { s16x4 h[2]; bf16x8 v; } u;
                        u.h[0] = r[0]; u.h[1] = r[1];
                        bfr[j] = u.v;
                    }
#pragma unroll
                    for (int t = 0; t < TPW; ++t) {
                        const int tap = wave * TPW + t, kh = tap >> 2;
                        const int hbase = (S * yy + kh) * hrowB + x0 * S * hpB;
#pragma unroll
                        for (int i = 0; i < GT; ++i) {
                            s16x4 r[2];
#pragma unroll
                            for (int rd = 0; rd < 2; ++rd)
                                r[rd] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(hiC + hbase + hb[t][i][rd]));
                            union { s16x4 h[2]; bf16x8 v; } u;
                            u.h[0] = r[0]; u.h[1] = r[1];
#pragma unroll
                            for (int j = 0; j < DT; ++j)
                                acc[t][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u.v, bfr[j], acc[t][i][j], 0, 0, 0);
                        }
                    }
                }
            }
            }
        } else {
            const int kl = lane >> 5, cl = lane & 31;
            for (int yy = 0; yy < TH; ++yy) {
                for (int x0 = 0; x0 < TW; x0 += 2) {
                    const int x = x0 + kl;
                    float bfr[DT];
#pragma unroll
                    for (int j = 0; j < DT; ++j) bfr[j] = *(const float*)(loC + (yy * TW + x) * lpB + (j * 32 + cl) * 4);
#pragma unroll
                    for (int t = 0; t < TPW; ++t) {
                        const int tap = wave * TPW + t, kh = tap >> 2, kw = tap & 3;
#pragma unroll
                        for (int i = 0; i < GT; ++i) {
                            float af = *(const float*)(hiC + ((S * yy + kh) * RW + S * x + kw) * hpB + (i * 32 + cl) * 4);
#pragma unroll
                            for (int j = 0; j < DT; ++j)
                                acc[t][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bfr[j], acc[t][i][j], 0, 0, 0);
                        }
                    }
                }
            }
        }
            if (a.nbuf == 2) cur ^= 1;
        else {
            const int nxt = strip + gridDim.x;
            if (nxt < a.nstrips) { __syncthreads(); stage_strip(nxt, hiL, loL); }     // single buffer: every wave is done with the strip
        }
    }
    // ---- partial store: D[row = g][col = d] ------------------------------------------------------------------------
    float* outp = a.part + (long long)blockIdx.x * 16 * a.Cg * a.Cd;
    const int gwin = (hpB == hgB) ? 0 : g0, dwin = (lpB == lgB) ? 0 : d0;      // whole-pixel staging covers window 0 only
    const int h = lane >> 5;
    if (a.pack) {
        // D[row][col]: the packed side's index is (kw = idx >> 3, channel = idx & 7) of kernel row kh = wave & 3
        if (wave < 8) {
            const int kh = wave & 3, tsel = wave >> 2;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * h, col = lane & 31;
                int tap, g, d;
                if (a.pack == 1) { tap = kh * 4 + (row >> 3); g = row & 7; d = dwin + tsel * 32 + col; }
                else { tap = kh * 4 + (col >> 3); g = gwin + tsel * 32 + row; d = col & 7; }
                if (g < a.Cg && d < a.Cd) outp[((long long)tap * a.Cg + g) * a.Cd + d] = acc[0][0][0][e];
            }
        }
        return;
    }
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int tap = wave * TPW + t;
#pragma unroll
        for (int i = 0; i < GT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                int g = gwin + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (g >= a.Cg) continue;
#pragma unroll
                for (int j = 0; j < DT; ++j) {
                    int d = dwin + j * 32 + (lane & 31);
                    if (d < a.Cd) outp[((long long)tap * a.Cg + g) * a.Cd + d] = acc[t][i][j][e];
                }
            }
    }
}

// Fixed-order sum of partial slabs, four outputs per lane: block (x, y) adds the slabs [y*chunk, (y+1)*chunk) of the 256
// outputs 256 x .. 256 x + 255 (4 slab lanes per output quad, combined through LDS) into out[y][.].  A second launch
// with gridDim.y = 1 folds the gridDim.y intermediate slabs, so small outputs (1 K - 8 K values x 512 slabs) are summed
// by hundreds of workgroups instead of a handful.
__global__ __launch_bounds__(256) void ws_slab_sum_kernel(const float* __restrict__ part, int nslabs, int n, int chunk,
                                                          float* __restrict__ out) {
    __shared__ f32x4 red[4][64];
    const int c4 = blockIdx.x * 64 + (threadIdx.x & 63), sg = threadIdx.x >> 6;
    const int s0 = blockIdx.y * chunk, s1 = min(nslabs, s0 + chunk);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (c4 * 4 < n)
        for (int k = s0 + sg; k < s1; k += 4) s += *(const f32x4*)(part + (long long)k * n + c4 * 4);
    red[sg][threadIdx.x & 63] = s;
    __syncthreads();
    if (sg == 0 && c4 * 4 < n) {
        const int l = threadIdx.x;
        *(f32x4*)(out + (long long)blockIdx.y * n + c4 * 4) = ((red[0][l] + red[1][l]) + red[2][l]) + red[3][l];
    }
}

// second-level split of the slab sum: enough workgroups to cover the chip, at least 4 slabs per workgroup
static int ws_sum_split(int nslabs, int n) {
    const int colblocks = (n / 4 + 63) / 64;
    int split = 1;
    while (colblocks * split < 512 && nslabs / (split * 2) >= 4 && split < 64) split *= 2;
    return split;
}

#include <stdlib.h>
// 16 waves (one tap each) instead of 8 (two taps each): 8 % slower alone, but +1.8 % on the whole step (three A/B
// repeats on one device, r01) -- this kernel runs on the side stream next to the data-gradient chain, where the extra
// waves keep it issuing while the other stream's workgroups hold most of the CU
static int ws_pack_on() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("P2P_WS_PACK"); v = e ? atoi(e) : 1; }
    return v;
}
static int ws_waves16() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("P2P_WS_W16"); v = e ? atoi(e) : 1; }
    return v;
}

// Tiling of the LDS-resident form: 32x32 MFMA tiles per workgroup (GT x DT <= 4), channel windows, strip height.
struct WsPlan { int ok, GT, DT, gwins, dwins, TH, blocks, pack, nbuf, buf_bytes, hrowB, lo_off, mh, ml; size_t shm; };

static WsPlan ws_plan(int dtype, int stride, int N, int LH, int LW, int Cg, int Cd, int hi_ld, int lo_ld) {
    WsPlan p = {};
    const int esz = dtype == P2P_BF16 ? 2 : 4;
    if ((LW & (LW - 1)) || LW < 16 || LW > 128) return p;       // 128: the stride-1 heads of 128x128 sprites (c5), strips of 4 rows
    if ((hi_ld * esz) % 16 || (lo_ld * esz) % 16) return p;
    const int gt_all = (Cg + 31) / 32, dt_all = (Cd + 31) / 32;
    // windows: at most 2 tiles along g, then as many along d as keep GT*DT <= 4
    p.GT = gt_all >= 2 ? 2 : 1;
    p.DT = dt_all >= 4 ? (p.GT == 1 ? 4 : 2) : (dt_all >= 2 ? 2 : 1);
    if (p.GT * p.DT > 4) p.DT = 4 / p.GT;
    p.gwins = (gt_all + p.GT - 1) / p.GT;
    p.dwins = (dt_all + p.DT - 1) / p.DT;
    if (p.gwins * p.dwins > 8) return p;                 // beyond that the general kernel (wgemm.hip) re-reads less
    // few-channel sides stored as one 16-byte pixel (bf16): pack 4 taps x 8 channels into one 32-slot tile
    if (ws_pack_on() && dtype == P2P_BF16 && p.gwins == 1 && p.dwins == 1) {
        if (hi_ld == 8 && Cg <= 8 && p.GT == 1 && p.DT == 2) p.pack = 1;                       // 4/8 -> 64 (down1, D.down)
        else if (stride == 1 && lo_ld == 8 && Cd <= 8 && p.GT == 2 && p.DT == 1) p.pack = 2;   // 36 -> 4, 64 -> 1 (the heads)
    }
    // strip height: the tallest of 8, 4, 2, 1 rows (<= 512 pixels) whose TWO buffers fit 150 KB (staging of the next strip under
    // the MFMAs of this one), unless that would leave strips of a single row where a single buffer allows >= 4 rows
    constexpr int dbuf = 0;      // double-buffered strips measured slower on c2 (r02: shorter strips, more halo); the kernel keeps the code path
    const size_t hpB = (size_t)hi_ld * esz > (size_t)64 * p.GT * (esz / 2) ? (size_t)64 * p.GT * (esz / 2) : (size_t)hi_ld * esz;
    const size_t lpB = (size_t)lo_ld * esz > (size_t)64 * p.DT * (esz / 2) ? (size_t)64 * p.DT * (esz / 2) : (size_t)lo_ld * esz;
    // bank swizzle (ws_swz) of the transposing reads: bf16, unpacked tiles, whole 64-byte quarters per pixel
    static int swz_on = -1;
    if (swz_on < 0) { const char* e = getenv("P2P_WS_SWIZZLE"); swz_on = e ? atoi(e) : 1; }
    auto mask_for = [&](size_t pixB, int step) {
        if (!swz_on || esz != 2 || p.pack || (pixB != 64 && pixB != 128 && pixB != 256)) return 0;
        const size_t u = pixB * step / 64;
        return u == 2 ? 1 : (u == 4 ? 3 : 0);
    };
    p.mh = mask_for(hpB, stride);
    p.ml = mask_for(lpB, 1);
    p.hrowB = (int)((size_t)(stride * LW + 3) * hpB);
    if (p.mh) p.hrowB = (p.hrowB + 255) & ~255;
    auto hi_bytes_for = [&](int th) { return (((size_t)(stride * th + 3) * p.hrowB + 255) & ~(size_t)255) + 256; };
    auto bytes_for = [&](int th) {
        const size_t lo_bytes = p.pack == 2 ? (size_t)(th + 3) * (LW + 3) * lpB : (size_t)th * LW * lpB;
        return hi_bytes_for(th) + ((lo_bytes + 255) & ~(size_t)255) + 512;
    };
    int TH = 512 / LW;
    if (TH > 8) TH = 8;
    if (TH > LH) TH = LH;
    // few-channel (packed) layers: the contraction is a handful of MFMAs per strip, the kernel is a stream of strips.  One
    // workgroup per CU alternates "stage, wait, contract" and leaves HBM idle most of the time; several co-resident workgroups
    // (their strips are 53-71 KB, their partial slabs 4-32 KB) keep loads in flight while one of them contracts.
    static int th_pack = -1, want_pack = -1;
    constexpr int dbuf_pack = 0;
    if (th_pack < 0) { const char* e = getenv("P2P_WS_TH_PACK"); th_pack = e ? atoi(e) : 8; }
    if (want_pack < 0) { const char* e = getenv("P2P_WS_WANT_PACK"); want_pack = e ? atoi(e) : 512; }
    if (p.pack && TH > th_pack) TH = th_pack;
    int th1 = 0, th2 = 0;       // tallest strip with one / two buffers
    for (int t = TH; t >= 1; t >>= 1) {
        if (LH % t) continue;
        if (!th1 && bytes_for(t) <= 150 * 1024) th1 = t;
        if (!th2 && 2 * bytes_for(t) <= 150 * 1024) th2 = t;
    }
    if (!th1) return p;
    p.nbuf = (dbuf && th2 && (th2 >= 2 || th1 < 4)) ? 2 : 1;
    if (p.pack && dbuf_pack && th2 == th1) p.nbuf = 2;
    TH = p.nbuf == 2 ? th2 : th1;
    p.buf_bytes = (int)bytes_for(TH);
    p.lo_off = (int)hi_bytes_for(TH);
    p.shm = (size_t)p.nbuf * p.buf_bytes;
    p.TH = TH;
    long long strips = (long long)N * (LH / TH);
    // workgroups wanted per launch: one per CU -- half the partial slabs of 512, c2 step 1 % faster (r02)
    static int want_env = -1;
    if (want_env < 0) { const char* e = getenv("P2P_WS_WANT"); want_env = e ? atoi(e) : 256; }
    long long want = (p.pack ? want_pack : want_env) / (p.gwins * p.dwins);
    const long long slab_bytes = 16LL * Cg * Cd * 4;
    const long long cap = (64LL << 20) / slab_bytes;         // keep the partial slabs within 64 MB
    if (want > cap) want = cap;
    if (want < 32) want = 32;
    p.blocks = (int)(strips < want ? strips : want);
    p.ok = 1;
    return p;
}

// Applicability / workspace of the LDS-resident form.  Returns the number of partial slabs (workgroups along x) or 0.
extern "C" int p2p_wgrad_small_blocks(int dtype, int stride, int N, int LH, int LW, int Cg, int Cd, int hi_ld, int lo_ld) {
    WsPlan p = ws_plan(dtype, stride, N, LH, LW, Cg, Cd, hi_ld, lo_ld);
    if (!p.ok) return 0;
    const int split = ws_sum_split(p.blocks, 16 * Cg * Cd);
    return p.blocks + (split > 1 ? split : 0);
}

template <typename T>
static int ws_launch(WsArgs& a, int stride, const WsPlan& p, hipStream_t st) {
    dim3 grid(p.blocks, p.dwins, p.gwins);
#define WS_GO(S_, G_, D_)                                                                                              \
    do {                                                                                                               \
        static bool done = false;                                                                                      \
        if (!done)                                                                                                     \
            done = (int)p2p_allow_lds((const void*)wgrad_small_kernel<T, S_, G_, D_, 2>, 160 * 1024, "wgrad_small_kernel") &   \
                   (int)p2p_allow_lds((const void*)wgrad_small_kernel<T, S_, G_, D_, 1>, 160 * 1024, "wgrad_small_kernel");    \
        if (ws_waves16() && !p.pack) wgrad_small_kernel<T, S_, G_, D_, 1><<<grid, dim3(1024), p.shm, st>>>(a);         \
        else wgrad_small_kernel<T, S_, G_, D_, 2><<<grid, dim3(512), p.shm, st>>>(a);                                  \
    } while (0)
#define WS_SEL(S_)                                                                                                     \
    do {                                                                                                               \
        if (p.GT == 1 && p.DT == 1) WS_GO(S_, 1, 1);                                                                   \
        else if (p.GT == 1 && p.DT == 2) WS_GO(S_, 1, 2);                                                              \
        else if (p.GT == 1 && p.DT == 4) WS_GO(S_, 1, 4);                                                              \
        else if (p.GT == 2 && p.DT == 1) WS_GO(S_, 2, 1);                                                              \
        else WS_GO(S_, 2, 2);                                                                                          \
    } while (0)
    if (stride == 1) WS_SEL(1); else WS_SEL(2);
#undef WS_SEL
#undef WS_GO
    return p2p_check_launch("p2p_wgrad_small");
}

extern "C" int p2p_wgrad_small(int dtype, int stride, int N, int LH, int LW, int Cg, int Cd, const p2p_tensor* hi,
                               const p2p_tensor* lo, float* dw, void* workspace, void* stream) {
    P2P_REQUIRE(hi && lo && hi->ptr && lo->ptr && dw && workspace, "p2p_wgrad_small: null pointer");
    P2P_REQUIRE(stride == 1 || stride == 2, "p2p_wgrad_small: stride must be 1 or 2");
    const WsPlan plan = ws_plan(dtype, stride, N, LH, LW, Cg, Cd, hi->ld, lo->ld);
    P2P_REQUIRE(plan.ok, "p2p_wgrad_small: shape not supported (query p2p_wgrad_small_blocks)");
    const int nblocks = plan.blocks;
    P2P_REQUIRE(((uintptr_t)hi->ptr % 16) == 0 && ((uintptr_t)lo->ptr % 16) == 0, "p2p_wgrad_small: views must be 16-byte aligned");
    const int esz = dtype == P2P_BF16 ? 2 : 4;
    WsArgs a;
    a.hi = (const char*)hi->ptr; a.hi_img = hi->img_stride; a.hi_row = hi->row_stride; a.hi_ld = hi->ld;
    a.lo = (const char*)lo->ptr; a.lo_img = lo->img_stride; a.lo_row = lo->row_stride; a.lo_ld = lo->ld;
    a.part = (float*)workspace;
    a.N = N; a.LH = LH; a.LW = LW; a.Cg = Cg; a.Cd = Cd;
    (void)esz;
    a.TH = plan.TH;
    a.pack = plan.pack;
    a.nbuf = plan.nbuf; a.buf_bytes = plan.buf_bytes;
    static int abl = -1;
    if (abl < 0) { const char* e = getenv("P2P_WS_ABL"); abl = e ? atoi(e) : 0; }
    a.abl = abl;
    a.hrowB = plan.hrowB; a.lo_off = plan.lo_off; a.mh = plan.mh; a.ml = plan.ml;
    a.strips_per_img = LH / plan.TH;
    a.nstrips = N * a.strips_per_img;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    P2P_DISPATCH_DTYPE(dtype, rc = ws_launch<T>(a, stride, plan, st));
    if (rc) return rc;
    const int n = 16 * Cg * Cd;
    const int colblocks = (n / 4 + 63) / 64, split = ws_sum_split(nblocks, n);
    const float* part = (const float*)workspace;
    if (split == 1) {
        ws_slab_sum_kernel<<<dim3(colblocks, 1), 256, 0, st>>>(part, nblocks, n, nblocks, dw);
    } else {
        float* part2 = (float*)workspace + (long long)nblocks * n;
        const int chunk = (nblocks + split - 1) / split;
        ws_slab_sum_kernel<<<dim3(colblocks, split), 256, 0, st>>>(part, nblocks, n, chunk, part2);
        ws_slab_sum_kernel<<<dim3(colblocks, 1), 256, 0, st>>>(part2, split, n, split, dw);
    }
    return p2p_check_launch("p2p_wgrad_small reduce");
}
