// MFMA implicit GEMM for every 4x4 convolution of the step that maps pixels -> pixels: Conv2D /
// Conv2DTranspose forward and their data gradients (reference call sites networks.py:10-16,26-27,46-48,75-78
// and the tape gradients taken at pix2pix_model.py:78-79).  Forms (include/p2pgan.h), s = stride:
//   op G: lo[m][d] = sum_{t=(kh,kw)} sum_g hi[n,s*y+kh-1,s*x+kw-1,g] * Wt[t][d][g]                (16 taps)
//   op P, s=2: hi[n,2y+ph,2x+pw][g] = sum_{2x2 taps of phase (ph,pw)} sum_d lo[n,y+dy,x+dx,d] * Wn[t][g][d]
//   op P, s=1: hi[n,y,x][g]         = sum_{t} sum_d lo[n,y+1-kh,x+1-kw,d] * Wn[t][g][d]           (16 taps)
// All are "NT" GEMMs  C[m][n] = sum_k A[m][k] B[n][k]:  A rows are gathered 16-byte chunks of one input
// pixel's channels (the zero halo around every image supplies the SAME padding, so no bounds checks),
// B rows are one output channel's weights.
//
// Tiling (CDNA4, wave64): 4 or 8 waves; block tile BM x BN, wave tile (TM x TN) 32x32 MFMA tiles,
// K-block = 128 bytes per row (64 bf16 / 32 f32).  A and B tiles are staged global->LDS with
// global_load_lds_dwordx4 (no VGPR round trip); the LDS image is lane-linear, so the bank swizzle
// (16-byte slot ^= (row>>1)&7, conflict-free for ds_read_b128 MFMA operand reads) is applied to the
// per-lane SOURCE address and again on the read.  Double-buffered: one barrier per K-block, the loads of
// block k+1 are in flight while block k is multiplied.
//   bf16: v_mfma_f32_32x32x16_bf16, f32 accumulate.   f32 (parity mode): v_mfma_f32_32x32x2_f32 (exact f32).
// GEN = false: channels-per-tap bytes are a power of two (all encoder/decoder blocks) -> shifts only.
// GEN = true : any multiple of 16 bytes per tap (edge layers: 8-channel images, the 36(+4)-channel concat, the
//              33(+7)-channel indexed concat), plus bias / LeakyReLU / column mask in the epilogue.
#include "p2p_common.hpp"
#include <stdlib.h>
#include <utility>

// Diagnostic builds only (tools/ubench/igemm_abl.py): 1 = staging without the LDS reads / MFMAs, 2 = LDS reads + MFMAs
// without the staging.  The product library is always built with 0.
#ifndef P2P_ABL
#define P2P_ABL 0
#endif

struct IgemmArgs {
    const char* in; long long in_img; int in_row; int in_ld;      // gathered input view (element strides)
    char* out; long long out_img; int out_row; int out_ld;        // output view, splitk == 1
    float* slabs; long long slab_stride;                           // splitk > 1: f32 [ks][pixels][ncols]
    const char* w;                                                  // [taps][ncols_pad][C] in T
    const float* bias;                                              // GEN epilogue, may be null
    int M, lgLW, lgLH, LW, LH;
    int C, lgCB, Cc;     // contraction channels per tap, log2(C*sizeof(T)) (pow2 path), 16-byte chunks per tap
    unsigned cc_inv;     // ceil(2^24 / Cc): chunk index -> tap by one multiply (exact for every chunk index of a launch while Cc <= 1024; GEN path)
    int w_rows;          // rows of a weight tap slab (>= the launched column tiles)
    int ncols;           // real output channels (stores are masked to col < ncols)
    int mode;            // 0 = G, 1 = P stride 2 (4 phases), 2 = P stride 1
    int si;              // input scale of op G (stride)
    int splitk, taps_per;
    int live_taps;       // 1: a 1x1 low-resolution map -- only the taps that touch real pixels are contracted (op G: the 4
                         // centre taps of the 2x2 input, op P: the one tap per sub-pixel phase); the others only ever
                         // multiply the zero halo
    int act; float alpha;
    int w_major;         // logical block order: 0 = column tile fastest (an XCD owns a range of pixel tiles and reads ALL weights),
                         // 1 = pixel tile fastest (an XCD owns a range of (column tile, phase, K split) and reads all pixels but
                         // only its share of the weights): taken when the weights are the larger operand (deep layers)
    // fused InstanceNorm statistics (VEPI epilogue): per (image, slot, channel) the mean and the centred sum of squares
    // of `stat_rows` consecutive output pixels of one image; slot = phase * tiles_per_image + tile_in_image
    float* stat_part; int stat_rows; int stat_slots; int lgHW;
};

template <typename T> struct Frag;
template <> struct Frag<bf16_t> { typedef bf16x8 type; };
template <> struct Frag<float> { typedef f32x4 type; };

__device__ __forceinline__ void mfma_step(f32x16& acc, const bf16x8& a, const bf16x8& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ void mfma_step(f32x16& acc, const f32x4& a, const f32x4& b) {
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc, 0, 0, 0);
}

__device__ __forceinline__ void glds16(const char* g, char* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// Epilogue shared by both kernel forms: D[row = n][col = m]: col = lane&31 (pixel), row = (e&3) + 8*(e>>2) + 4*(lane>>5) (channel).
// KG = 2 (igemm_pipe_kernel): after the exchange the EVEN physical fragment rows of K group `kg` hold the logical rows i + kg; a wave
// stores those only.
template <typename T, bool GEN, bool VEPI, int BM, int BN, int NTHR, int TM, int TN, int KG>
__device__ __forceinline__ void igemm_epilogue(const IgemmArgs& a, f32x16 (&acc)[TM][TN], char* smem, const int tid, const int lane,
                                               const int h, const int wm, const int wn, const int kg, const int m0, const int n0,
                                               const int ks, const int phase) {
    const int ph = phase >> 1, pw = phase & 1;
    auto out_pixel = [&](int m) -> long long {
        int x = m & (a.LW - 1);
        int y = (m >> a.lgLW) & (a.LH - 1);
        int n = m >> (a.lgLW + a.lgLH);
        if (a.mode == 1) return (long long)n * a.out_img + (long long)(2 * y + ph) * a.out_row + (2 * x + pw);
        return (long long)n * a.out_img + (long long)y * a.out_row + x;
    };
    const bool to_slabs = a.splitk > 1;
    if constexpr (VEPI) {
        // stage the tile through LDS as [pixel][channel] in the OUTPUT type, then store whole 16-byte chunks of
        // each pixel's channel run (coalesced rows instead of 2-byte scatter).
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();      // every wave is done reading the operand tiles
        auto run = [&](auto tag) {
            typedef decltype(tag) TO;
            constexpr int osz = sizeof(TO);
            constexpr int RS = BN * osz + 16;     // padded row stride (16-byte aligned rows; the 32 pixel lanes' 4-channel writes hit distinct banks)
            typedef __attribute__((__vector_size__(4 * sizeof(TO)))) TO vec4_t;
            constexpr int CE = 16 / osz, CPR = BN / CE, RPP = NTHR / CPR;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if (KG > 1 && (i % KG) != 0) continue;          // K groups: even physical rows only, logical row = i + kg
                int ml = (wm * TM + i + (KG > 1 ? kg : 0)) * 32 + (lane & 31);
#pragma unroll
                for (int j = 0; j < TN; ++j) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        int nl = (wn * TN + j) * 32 + 8 * g + 4 * h;
                        vec4_t v4;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            float v = acc[i][j][4 * g + k];
                            if (GEN && !to_slabs) {
                                int col = n0 + nl + k;
                                if (a.bias && col < a.ncols) v += a.bias[col];
                                if (a.act == P2P_ACT_LEAKY) v = v > 0.f ? v : a.alpha * v;
                            }
                            v4[k] = from_f32<TO>(v);
                        }
                        *(vec4_t*)(smem + ml * RS + nl * osz) = v4;
                    }
                }
            }
            __syncthreads();
            if (a.stat_part && !to_slabs) {
                // InstanceNorm statistics of this tile (networks.py:18,29), taken from the ROUNDED values the consumer
                // will read back: two passes over the LDS tile (mean, then centred squares), per group of stat_rows rows.
                constexpr int NP = NTHR / BN;                      // row parts per column
                float* scr = (float*)(smem + BM * RS);             // [NP][BN] scratch behind the tile
                const int colL = tid % BN, part = tid / BN;
                const int rows = a.stat_rows, ngrp = BM / rows;
                for (int gI = 0; gI < ngrp; ++gI) {
                    const int rbeg = gI * rows;
                    float sacc = 0.f;
                    for (int r = rbeg + part; r < rbeg + rows; r += NP) sacc += to_f32(*(const TO*)(smem + r * RS + colL * osz));
                    scr[part * BN + colL] = sacc;
                    __syncthreads();
                    float mean = 0.f;
#pragma unroll
                    for (int q2 = 0; q2 < NP; ++q2) mean += scr[q2 * BN + colL];
                    mean /= (float)rows;
                    __syncthreads();
                    float qacc = 0.f;
                    for (int r = rbeg + part; r < rbeg + rows; r += NP) {
                        float d = to_f32(*(const TO*)(smem + r * RS + colL * osz)) - mean;
                        qacc += d * d;
                    }
                    scr[part * BN + colL] = qacc;
                    __syncthreads();
                    if (part == 0 && n0 + colL < a.ncols) {
                        float m2 = 0.f;
#pragma unroll
                        for (int q2 = 0; q2 < NP; ++q2) m2 += scr[q2 * BN + colL];
                        const int mrow = m0 + rbeg;
                        const int img = mrow >> a.lgHW;
                        const int tile_in_img = (mrow & ((1 << a.lgHW) - 1)) / rows;
                        const int slot = phase * (a.stat_slots / (a.mode == 1 ? 4 : 1)) + tile_in_img;
                        float* dst = a.stat_part + (((long long)img * a.stat_slots + slot) * a.ncols + n0 + colL) * 2;
                        dst[0] = mean;
                        dst[1] = m2;
                    }
                    __syncthreads();
                }
            }
            TO* obase = to_slabs ? (TO*)(a.slabs + (long long)ks * a.slab_stride) : (TO*)a.out;
            const int chunk = tid % CPR, r0 = tid / CPR;
            const int col = n0 + chunk * CE;
#pragma unroll
            for (int ps = 0; ps < BM / RPP; ++ps) {
                int ml = ps * RPP + r0;
                int m = m0 + ml;
                if (m < a.M && col < a.ncols) {
                    f32x4 v = *(const f32x4*)(smem + ml * RS + chunk * 16);
                    *(f32x4*)(obase + out_pixel(m) * a.out_ld + col) = v;
                }
            }
        };
        if (to_slabs) run(float()); else run(T());
    } else {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if (KG > 1 && (i % KG) != 0) continue;
            int m = m0 + (wm * TM + i + (KG > 1 ? kg : 0)) * 32 + (lane & 31);
            if (m >= a.M) continue;
            long long pix = out_pixel(m);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    int col = n0 + (wn * TN + j) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (col >= a.ncols) continue;
                    float v = acc[i][j][e];
                    if (!to_slabs) {
                        if (GEN) {
                            if (a.bias) v += a.bias[col];
                            if (a.act == P2P_ACT_LEAKY) v = v > 0.f ? v : a.alpha * v;
                        }
                        ((T*)a.out)[pix * a.out_ld + col] = from_f32<T>(v);
                    } else {
                        a.slabs[(long long)ks * a.slab_stride + pix * a.out_ld + col] = v;
                    }
                }
        }
    }
}

template <typename T, int WM, int WN, int TM, int TN, bool GEN, bool VEPI, int NST>
__global__ __launch_bounds__(WM * WN * 64) void igemm_kernel(IgemmArgs a) {
    constexpr int NW = WM * WN, NTHR = NW * 64;     // waves / threads per workgroup (4 or 8 waves)
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    constexpr int NA = BM / (8 * NW), NB = BN / (8 * NW);       // 8-row pieces (1 KB per wave-instruction) staged per thread
    static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "every wave stages the same number of pieces (counted vmcnt)");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    // logical block order: N tile fastest, then phase / K split, then M tile: the blocks that gather the same input
    // pixels (all output-channel tiles, all 4 sub-pixel phases) sit next to each other and share an XCD's L2
    const unsigned nblk = gridDim.x * gridDim.y * gridDim.z;
    const unsigned lin = xcd_remap(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), nblk);
    int bx, by, bz;
    if (a.w_major) { bx = lin % gridDim.x; by = (lin / gridDim.x) % gridDim.y; bz = lin / (gridDim.x * gridDim.y); }
    else { by = lin % gridDim.y; bz = (lin / gridDim.y) % gridDim.z; bx = lin / (gridDim.y * gridDim.z); }
    const int m0 = bx * BM, n0 = by * BN;
    const int ks = bz % a.splitk, phase = bz / a.splitk;
    const int ph = phase >> 1, pw = phase & 1;
    const int tap_begin = ks * a.taps_per;
    const int CBbytes = GEN ? a.Cc * 16 : (1 << a.lgCB);          // bytes per tap row
    const int nkb = (a.taps_per * CBbytes) >> 7;                  // K-blocks of 128 bytes
    constexpr int esz = sizeof(T);
    const int pixB = a.in_ld * esz, rowB = a.in_row * pixB;       // byte strides of the gathered view

    // ---- per-thread staging rows -------------------------------------------------------------------
    const char* abase[NA];   // address of the row's base pixel in the input view
    int aq[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        int r = (i * NW + wave) * 8 + (lane >> 3);
        int m = min(m0 + r, a.M - 1);
        int x = m & (a.LW - 1);
        int y = (m >> a.lgLW) & (a.LH - 1);
        int n = m >> (a.lgLW + a.lgLH);
        int by = a.mode == 0 ? a.si * y - 1 : y;
        int bx = a.mode == 0 ? a.si * x - 1 : x;
        abase[i] = a.in + ((long long)n * a.in_img + (long long)by * a.in_row + bx) * pixB;
        aq[i] = (lane & 7) ^ ((r >> 1) & 7);
    }
    int bq[NB];
    int bbase[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        int r = (j * NW + wave) * 8 + (lane >> 3);
        bq[j] = (lane & 7) ^ ((r >> 1) & 7);
        bbase[j] = (n0 + r) * a.C * esz;
    }
    const int wtap = a.w_rows * a.C * esz;   // bytes per weight tap slab

    auto tap_geom = [&](int tl, int& dy, int& dx, int& widx) {
        if (a.live_taps) tl = a.mode == 0 ? ((1 + (tl >> 1)) << 2) + 1 + (tl & 1) : 2 * ph + pw;
        if (a.mode == 0) { dy = tl >> 2; dx = tl & 3; widx = tl; }
        else if (a.mode == 1) {
            int kh = (1 - ph) + 2 * (tl >> 1), kw = (1 - pw) + 2 * (tl & 1);
            dy = (ph + 1 - kh) >> 1; dx = (pw + 1 - kw) >> 1; widx = kh * 4 + kw;
        } else { dy = 1 - (tl >> 2); dx = 1 - (tl & 3); widx = tl; }
    };
    auto split_k = [&](int kb, int q, int& tl, int& cbyte) {
        if (GEN) {
            int kc = kb * 8 + q;
            int tp = (int)(((unsigned)kc * a.cc_inv) >> 24);      // kc / Cc without the ~20-instruction integer division per piece
            tl = tap_begin + tp;
            cbyte = (kc - tp * a.Cc) << 4;
        } else {
            int kbyte = (kb << 7) + (q << 4);
            tl = tap_begin + (kbyte >> a.lgCB);
            cbyte = kbyte & (CBbytes - 1);
        }
    };

    constexpr int NL = NA + NB;                   // LDS-DMA instructions ("pieces") per thread per stage
    auto stage_piece = [&](int kb, char* buf, int p) {
        int tl, cbyte, dy, dx, widx;
        if (p < NA) {
            split_k(kb, aq[p], tl, cbyte);
            tap_geom(tl, dy, dx, widx);
            glds16(abase[p] + (dy * rowB + dx * pixB + cbyte), buf + (p * NW + wave) * 1024);
        } else {
            const int j = p - NA;
            split_k(kb, bq[j], tl, cbyte);
            tap_geom(tl, dy, dx, widx);
            glds16(a.w + (widx * wtap + bbase[j] + cbyte), buf + A_BYTES + (j * NW + wave) * 1024);
        }
    };
    // When a K-block (128 bytes per row) lies inside one tap (channels-per-tap bytes >= 128, all blocks but the 32-channel
    // ones), the tap and its geometry are the same for every lane: they are derived once per K-block from wave-uniform
    // values (scalar registers) and each piece costs one 64-bit vector add on a per-lane address that already holds the
    // row base and the swizzled 16-byte slot.  (The per-lane form above spent ~25 vector instructions per piece.)
    const bool uni = !GEN && a.lgCB >= 7;
    const char* abq[NA];
    int bbq[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) abq[i] = abase[i] + (aq[i] << 4);
#pragma unroll
    for (int j = 0; j < NB; ++j) bbq[j] = bbase[j] + (bq[j] << 4);
    auto stage = [&](int kb, char* buf) {
        if (uni) {
            const int kbyte0 = kb << 7;
            const int tl = tap_begin + (kbyte0 >> a.lgCB), cb0 = kbyte0 & (CBbytes - 1);
            int dy, dx, widx;
            tap_geom(tl, dy, dx, widx);
            const int aoff = dy * rowB + dx * pixB + cb0;
            const char* wk = a.w + ((long long)widx * wtap + cb0);
#pragma unroll
            for (int i = 0; i < NA; ++i) glds16(abq[i] + aoff, buf + (i * NW + wave) * 1024);
#pragma unroll
            for (int j = 0; j < NB; ++j) glds16(wk + bbq[j], buf + A_BYTES + (j * NW + wave) * 1024);
            return;
        }
#pragma unroll
        for (int p = 0; p < NL; ++p) stage_piece(kb, buf, p);
    };

    // accumulators: weights are the MFMA "A" operand (rows = output channel), pixels the "B" operand
    // (columns), so a lane ends up with 4 CONSECUTIVE output channels of one pixel per register group.
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // fragment read offsets (row part); the k-step part is XORed in
    int arow[TM], brow[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) arow[i] = (wm * TM + i) * 32 + (lane & 31);
#pragma unroll
    for (int j = 0; j < TN; ++j) brow[j] = (wn * TN + j) * 32 + (lane & 31);
    const int h = lane >> 5;

    typedef typename Frag<T>::type frag_t;

    // NST = 2: one K-block of prefetch, vmcnt(0) + barrier per K-block (two workgroups per CU hide each other's waits).
    // NST = 3: ring of three LDS stages, two K-blocks of loads in flight across the barrier: a counted s_waitcnt leaves
    //          the younger stage's LDS-DMA outstanding (cdna_hip_programming.md "Pipelining across barriers") -- for
    //          launches that put a single workgroup on each CU.
    if (P2P_ABL != 2) stage(0, smem);
    if (P2P_ABL != 2 && NST == 3 && nkb > 1) stage(1, smem + STAGE);
    for (int kb = 0; kb < nkb; ++kb) {
        if (NST == 3) {
            if (kb + 1 < nkb) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        char* cur = smem + (NST == 3 ? (kb % 3) : (kb & 1)) * STAGE;
        const int kn = NST == 3 ? kb + 2 : kb + 1;                   // K-block to prefetch
        char* nxt = smem + (NST == 3 ? (kn % 3) : (kn & 1)) * STAGE;
        if (P2P_ABL != 2 && kn < nkb) stage(kn, nxt);
#pragma unroll
        for (int s = 0; P2P_ABL != 1 && s < 4; ++s) {
            frag_t af[TM], bf[TN];
            const int q = 2 * s + h;
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[i] = *(const frag_t*)(cur + arow[i] * 128 + ((q ^ ((arow[i] >> 1) & 7)) << 4));
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bf[j] = *(const frag_t*)(cur + A_BYTES + brow[j] * 128 + ((q ^ ((brow[j] >> 1) & 7)) << 4));
            __builtin_amdgcn_s_setprio(1);       // MFMA issue ahead of the partner wave's loads (+1 % measured, r01)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) mfma_step(acc[i][j], bf[j], af[i]);
            __builtin_amdgcn_s_setprio(0);
        }
    }

    igemm_epilogue<T, GEN, VEPI, BM, BN, NTHR, TM, TN, 1>(a, acc, smem, tid, lane, h, wm, wn, 0, m0, n0, ks, phase);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Software-pipelined form (round 4) for the encoder / decoder blocks whose K-blocks lie inside one tap (channels-per-tap bytes
// >= 128) and whose output channels fill 128-wide tiles: the deep layers (maps <= 4x4) in bf16, every main layer in f32.
// What the form above left on the table (ISA of r03: three ds_read_b128, s_waitcnt lgkmcnt(0), two MFMAs, four times per K-block,
// so every 64 matrix cycles paid one full LDS round trip, and a 32x64 wave tile reads 1.5 KB of LDS per MFMA -- with the LDS-DMA
// writes exactly the LDS array's capacity at the matrix peak):
//   * wave tile 64x64 (TM = TN = 2): 1 KB of LDS reads per MFMA;
//   * four fragment sets, one per 16-deep k-step of a K-block, loaded with ds_read_b128 in inline asm and retired by COUNTED
//     s_waitcnt lgkmcnt(TM + TN): the reads of step s+1 are in flight under the MFMAs of step s;
//   * the MFMAs of a K-block's last two k-steps are deferred across the block's single barrier, behind the first reads of the
//     next block: the matrix pipe has 2 x TM x TN instructions of work while the barrier, the LDS latency and the LDS-DMA issue
//     of the next stage (one piece between two MFMAs) pass;
//   * KG = 2: two K groups of four waves inside one workgroup (8 waves, 2 per SIMD) take alternate K-blocks of the SAME 128x128
//     output tile and exchange half of their accumulators through LDS at the end (fixed order: deterministic).  The deep layers
//     have exactly one 128x128 tile per CU at batch 256, so the second wave of every SIMD comes from splitting K inside the
//     workgroup instead of from f32 slabs in HBM.
template <int... I, typename F>
__device__ __forceinline__ void igemm_static_for(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
__device__ __forceinline__ unsigned lds_addr32(const void* p) {
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}
template <typename F> __device__ __forceinline__ void lds_read16(F& d, unsigned addr) {
    static_assert(sizeof(F) == 16, "one ds_read_b128");
    asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(addr));
}
template <int OFF, typename F> __device__ __forceinline__ void lds_read16_off(F& d, unsigned addr) {
    static_assert(sizeof(F) == 16 && OFF >= 0 && OFF < 65536, "one ds_read_b128 with an immediate offset");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
template <int N> __device__ __forceinline__ void lgkm_wait() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);      // register-only MFMAs must not be hoisted above the wait (cdna_hip_programming.md rule 18)
}

template <typename T, int MODE, int WM, int WN, int TM, int TN, int KG, int NST, bool VEPI>
__global__ __launch_bounds__(KG * WM * WN * 64) void igemm_pipe_kernel(IgemmArgs a) {
    constexpr int NWG = WM * WN, NW = KG * NWG, NTHR = NW * 64;
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;      // one K-block (128 bytes per row) per stage
    constexpr int NA = BM / (8 * NW), NB = BN / (8 * NW), NL = NA + NB;                 // LDS-DMA pieces per wave per stage
    constexpr int NF = TM + TN, NMF = TM * TN;
    constexpr int KS = 4 / KG;                                                          // 16-deep k-steps of a K-block per wave
    static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "every wave stages the same number of pieces (counted vmcnt)");
    static_assert(KG == 1 || (KG == 2 && TM % 2 == 0), "accumulator rows are split between two K groups");
    static_assert(NST >= 2 && NST <= 4, "ring of 2..4 stages");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = wave / NWG, wv = wave % NWG;
    const int wm = wv / WN, wn = wv % WN;
    const unsigned nblk = gridDim.x * gridDim.y * gridDim.z;
    const unsigned lin = xcd_remap(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), nblk);
    int bx, by, bz;
    if (a.w_major) { bx = lin % gridDim.x; by = (lin / gridDim.x) % gridDim.y; bz = lin / (gridDim.x * gridDim.y); }
    else { by = lin % gridDim.y; bz = (lin / gridDim.y) % gridDim.z; bx = lin / (gridDim.y * gridDim.z); }
    const int m0 = bx * BM, n0 = by * BN;
    const int ks = bz % a.splitk, phase = bz / a.splitk;
    const int ph = phase >> 1, pw = phase & 1;
    const int tap_begin = ks * a.taps_per;
    const int CBbytes = 1 << a.lgCB;                               // bytes per tap row (>= 128: a K-block lies inside one tap)
    const int nkb = (a.taps_per * CBbytes) >> 7;                  // K-blocks of 128 bytes = stages
    constexpr int esz = sizeof(T);
    const int pixB = a.in_ld * esz, rowB = a.in_row * pixB;

    // ---- staging: 32-bit per-lane byte offsets (row base + swizzled 16-byte slot) from the lowest address a tap can reach (one
    // row and one pixel in front of the view); a K-block adds a wave-uniform 64-bit base (p2p_glds16_sv: no vector address math) ----
    const char* const in_lo = a.in - (rowB + pixB);
    unsigned abq[NA];
    unsigned bbq[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int r = (i * NW + wave) * 8 + (lane >> 3);
        const int m = min(m0 + r, a.M - 1);
        const int x = m & (a.LW - 1), y = (m >> a.lgLW) & (a.LH - 1), n = m >> (a.lgLW + a.lgLH);
        const int byy = MODE == 0 ? a.si * y - 1 : y, bxx = MODE == 0 ? a.si * x - 1 : x;
        abq[i] = (unsigned)(((long long)n * a.in_img + (long long)(byy + 1) * a.in_row + bxx + 1) * pixB + (((lane & 7) ^ ((r >> 1) & 7)) << 4));
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int r = (j * NW + wave) * 8 + (lane >> 3);
        bbq[j] = (unsigned)((n0 + r) * a.C * esz + (((lane & 7) ^ ((r >> 1) & 7)) << 4));
    }
    const int wtap = a.w_rows * a.C * esz;
    const char* s_ak;                   // wave-uniform (scalar registers): bases of the K-block being requested
    const char* s_wk;
    auto stage_prepare = [&](int kb) {
        const int kbyte0 = kb << 7;
        int tl = tap_begin + (kbyte0 >> a.lgCB);
        const int cb0 = kbyte0 & (CBbytes - 1);
        int dy, dx, widx;
        if (a.live_taps) tl = MODE == 0 ? ((1 + (tl >> 1)) << 2) + 1 + (tl & 1) : 2 * ph + pw;
        if (MODE == 0) { dy = tl >> 2; dx = tl & 3; widx = tl; }
        else {
            const int kh = (1 - ph) + 2 * (tl >> 1), kw = (1 - pw) + 2 * (tl & 1);
            dy = (ph + 1 - kh) >> 1; dx = (pw + 1 - kw) >> 1; widx = kh * 4 + kw;
        }
        s_ak = in_lo + (dy * rowB + dx * pixB + cb0);
        s_wk = a.w + ((long long)widx * wtap + cb0);
    };
    // LDS image: the ring's stages are interleaved at the granularity of one LDS-DMA piece (8 rows x 128 B = 1 KiB): piece q of stage
    // st lies at (q * NST + st) * 1024, A pieces first.  A fragment row's address is then the same register for every stage and
    // the stage is an immediate offset of the ds_read (st * 1024 < 64 K): the K loop, unrolled over the ring, carries no vector
    // address arithmetic (r04: 8 v_add_u32 per 8 MFMAs before; vector and matrix instructions of a SIMD do not overlap --
    // tools/ubench/clock_probe.hip).  Bank behaviour is that of the flat image: address bits 0..9 are unchanged.
    constexpr int A_ALL = NST * A_BYTES;
    const unsigned smem32 = p2p_lds32(smem);
    auto stage_piece = [&](int st, auto pc) {             // piece p (0 .. NL-1) of this wave, into ring slot st
        constexpr int p = decltype(pc)::value;
        if (P2P_ABL == 2) return;
        if constexpr (p < NA) p2p_glds16_sv(s_ak, abq[p], smem32 + (((p * NW + wave) * NST + st) << 10));
        else p2p_glds16_sv(s_wk, bbq[p - NA], smem32 + A_ALL + ((((p - NA) * NW + wave) * NST + st) << 10));
    };
    auto stage_all = [&](int kb, int st) {
        stage_prepare(kb);
        igemm_static_for(std::make_integer_sequence<int, NL>{}, [&](auto pc) { stage_piece(st, pc); });
    };
    // all but the `n` youngest stages of LDS-DMA have landed (n wave-uniform, 0 .. NST - 2)
    auto vm_wait_stages = [&](int n) {
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

    // fragment read addresses inside ring slot 0: piece(row) * NST KiB + (row & 7) * 128 + ((2s + h) ^ swizzle(row)) * 16, the last two terms
    // = ((row & 7) * 128 + ((h ^ swizzle) << 4)) ^ (s << 5);  K group kg takes the k-steps s = KS * kg .. KS * kg + KS - 1 of every K-block
    typedef typename Frag<T>::type frag_t;
    const int h = lane >> 5;
    unsigned afo[KS][TM], bfo[KS][TN];
    const unsigned sm0 = lds_addr32(smem);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = (wm * TM + i) * 32 + (lane & 31);
        const unsigned b = sm0 + (row >> 3) * (NST * 1024) + (row & 7) * 128 + ((h ^ ((row >> 1) & 7)) << 4);
#pragma unroll
        for (int s4 = 0; s4 < KS; ++s4) afo[s4][i] = b ^ ((KS * kg + s4) << 5);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = (wn * TN + j) * 32 + (lane & 31);
        const unsigned b = sm0 + A_ALL + (row >> 3) * (NST * 1024) + (row & 7) * 128 + ((h ^ ((row >> 1) & 7)) << 4);
#pragma unroll
        for (int s4 = 0; s4 < KS; ++s4) bfo[s4][j] = b ^ ((KS * kg + s4) << 5);
    }
    frag_t fa[4][TM], fb[4][TN];          // KG = 1: one set per k-step; KG = 2: sets {0, 1} and {2, 3} alternate between K-blocks
    auto load_set = [&](auto sc, auto kc, auto stc) {              // set sc <- k-step kc (of this group's KS) of ring slot stc
        constexpr int s4 = decltype(sc)::value, k4 = decltype(kc)::value, OFF = decltype(stc)::value * 1024;
        if (P2P_ABL == 1) return;
#pragma unroll
        for (int i = 0; i < TM; ++i) lds_read16_off<OFF>(fa[s4][i], afo[k4][i]);
#pragma unroll
        for (int j = 0; j < TN; ++j) lds_read16_off<OFF>(fb[s4][j], bfo[k4][j]);
    };
    auto mfma_set = [&](auto sc) {
        constexpr int s4 = decltype(sc)::value;
        if (P2P_ABL == 1) return;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) mfma_step(acc[i][j], fb[s4][j], fa[s4][i]);
    };
    // MFMAs of the sets S0 .. S0 + NS - 1 with one LDS-DMA piece of the stage being requested behind each of the first NL of
    // them.  The MFMAs are unconditional (one code path: accumulators that live across a branch with MFMAs in both arms were
    // copied register by register at the join); only the tiny LDS-DMA issues sit under the wave-uniform `more`.
    auto mfma_dma = [&](auto s0c, auto nsc, const bool more, int buf) {
        constexpr int S0 = decltype(s0c)::value, NS = decltype(nsc)::value;
        igemm_static_for(std::make_integer_sequence<int, NS * NMF>{}, [&](auto mc) {
            constexpr int m = decltype(mc)::value;
            constexpr int s4 = S0 + m / NMF, i = (m % NMF) / TN, j = m % TN;
            if (P2P_ABL != 1) mfma_step(acc[i][j], fb[s4][j], fa[s4][i]);
            if constexpr (m < NL) { if (more) stage_piece(buf, mc); }
        });
        if constexpr (NL > NS * NMF) {
            if (more)
                igemm_static_for(std::make_integer_sequence<int, (NL > NS * NMF ? NL - NS * NMF : 0)>{},
                                 [&](auto mc) { stage_piece(buf, std::integral_constant<int, decltype(mc)::value + NS * NMF>{}); });
        }
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;

    // ---- prologue: stages 0 .. NST-2 requested, stage 0 landed (barrier), stage NST-1 requested, first fragment sets requested -------
#pragma unroll
    for (int st = 0; st < NST - 1; ++st)
        if (st < nkb) stage_all(st, st);
    vm_wait_stages(min(NST - 2, nkb - 1));
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (NST - 1 < nkb) stage_all(NST - 1, NST - 1);
    // The K loop is unrolled over the ring: K-block `it` lives in ring slot it % NST, a compile-time constant in every copy.
    if constexpr (KG == 1) {
        // four sets, one per k-step; the MFMAs of k-steps 2 and 3 run behind the barrier that ends the K-block
        auto kblock = [&](auto stc, int it) {
            constexpr int ST = decltype(stc)::value;
            using NX = std::integral_constant<int, (ST + 1) % NST>;
            __builtin_amdgcn_sched_barrier(0);
            load_set(I1{}, I1{}, stc); lgkm_wait<NF>(); mfma_set(I0{});
            __builtin_amdgcn_sched_barrier(0);
            load_set(I2{}, I2{}, stc); lgkm_wait<NF>(); mfma_set(I1{});
            __builtin_amdgcn_sched_barrier(0);
            load_set(I3{}, I3{}, stc);
            const bool more = it + NST < nkb;                           // wave-uniform
            if (more) stage_prepare(it + NST);                          // scalar work under the waits below
            // stage it + 1 must have landed before anybody reads it; the younger stages may stay in flight
            vm_wait_stages(min(NST - 2, nkb - 2 - it));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // sets 2 and 3 are in registers: this slot may be overwritten
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (it + 1 < nkb) load_set(I0{}, I0{}, NX{});
            __builtin_amdgcn_sched_barrier(0);
            mfma_dma(I2{}, I2{}, more, ST);                             // the slot this K-block releases takes K-block it + NST
        };
        load_set(I0{}, I0{}, I0{});
        for (int it0 = 0; it0 < nkb; it0 += NST)
            igemm_static_for(std::make_integer_sequence<int, NST>{}, [&](auto stc) {
                if (it0 + decltype(stc)::value < nkb) kblock(stc, it0 + decltype(stc)::value);
            });
    } else {
        // two K groups share every K-block (group kg: k-steps 2 kg, 2 kg + 1): stages stay 32 KB, so a ring of four leaves three
        // K-blocks (96 KB per CU) in flight -- the launch is fed from HBM / the Infinity Cache, whose latency one 64 KB stage in
        // flight did not cover (r04: 43 us with warm caches but 60 us inside the step, against 52 / 57 us for the r03 kernel).
        // Per K-block and wave: MFMAs of k-step 0 | barrier | reads of the next block's two sets | MFMAs of k-step 1 + LDS-DMA.
        static_assert(KG == 1 || NST % 2 == 0, "the fragment sets alternate with the K-block's parity");
        auto kblock = [&](auto stc, int it) {
            constexpr int ST = decltype(stc)::value, P = ST & 1;
            using NX = std::integral_constant<int, (ST + 1) % NST>;
            using X0 = std::integral_constant<int, 2 * P>; using X1 = std::integral_constant<int, 2 * P + 1>;
            using Y0 = std::integral_constant<int, 2 - 2 * P>; using Y1 = std::integral_constant<int, 3 - 2 * P>;
            const bool more = it + NST < nkb;
            __builtin_amdgcn_sched_barrier(0);
            lgkm_wait<NF>();                                            // X0 has landed (X1 behind it)
            mfma_set(X0{});
            __builtin_amdgcn_sched_barrier(0);
            if (more) stage_prepare(it + NST);
            vm_wait_stages(min(NST - 2, nkb - 2 - it));                 // stage it + 1 has landed
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // X1 too: nobody reads stage it any more
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (it + 1 < nkb) {
                load_set(Y0{}, I0{}, NX{});
                load_set(Y1{}, I1{}, NX{});
            }
            __builtin_amdgcn_sched_barrier(0);
            mfma_dma(X1{}, I1{}, more, ST);
        };
        load_set(I0{}, I0{}, I0{});
        load_set(I1{}, I1{}, I0{});
        for (int it0 = 0; it0 < nkb; it0 += NST)
            igemm_static_for(std::make_integer_sequence<int, NST>{}, [&](auto stc) {
                if (it0 + decltype(stc)::value < nkb) kblock(stc, it0 + decltype(stc)::value);
            });
    }
    __builtin_amdgcn_sched_barrier(0);
    // No fragment read is outstanding here (the last K-block issues none) -- but that rests on two correlated branches the register
    // allocator knows nothing about: it re-uses the fragment registers right below.  The explicit wait costs nothing and makes the
    // property visible to tools/exp/asm_checks.py (a vector write is not interlocked against a pending LDS return).
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);

    // ---- K groups: group g keeps fragment rows i with i % KG == g and receives the other group's partial sums for them --------------
    if constexpr (KG == 2) {
        // (behind the loop's last barrier nobody reads a stage buffer any more and no LDS-DMA is outstanding.)  Group 1 first
        // swaps its fragment rows pairwise, so that every wave SENDS the odd physical rows and KEEPS the even ones with static
        // register indices (a runtime row choice put the accumulators into scratch): physical row 2r of group g is logical row 2r + g.
        constexpr int TMO = TM / 2;
        if (kg) {
#pragma unroll
            for (int r = 0; r < TMO; ++r)
#pragma unroll
                for (int j = 0; j < TN; ++j) { const f32x16 t = acc[2 * r][j]; acc[2 * r][j] = acc[2 * r + 1][j]; acc[2 * r + 1][j] = t; }
        }
#pragma unroll
        for (int r = 0; r < TMO; ++r) {
            const int slot = (((1 - kg) * NWG + wv) * TMO + r) * TN;           // the other group's wave with the same (wm, wn)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    f32x4 v = {acc[2 * r + 1][j][4 * q4], acc[2 * r + 1][j][4 * q4 + 1], acc[2 * r + 1][j][4 * q4 + 2], acc[2 * r + 1][j][4 * q4 + 3]};
                    *(f32x4*)(smem + (slot + j) * 4096 + q4 * 1024 + lane * 16) = v;
                }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < TMO; ++r) {
            const int slot = ((kg * NWG + wv) * TMO + r) * TN;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const f32x4 v = *(const f32x4*)(smem + (slot + j) * 4096 + q4 * 1024 + lane * 16);
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[2 * r][j][4 * q4 + k] += v[k];
                }
        }
    }
    igemm_epilogue<T, false, VEPI, BM, BN, NTHR, TM, TN, KG>(a, acc, smem, tid, lane, h, wm, wn, kg, m0, n0, ks, phase);
}

static int ilog2_exact(long long v) {
    int l = 0;
    while ((1LL << l) < v) ++l;
    return (1LL << l) == v ? l : -1;
}

static int igemm_env(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
// minimum number of 256x128 workgroups for the 8-wave tile (0 = never): one per CU
static int igemm_big() { static int v = -1; if (v < 0) v = igemm_env("P2P_IGEMM_BIG", 256); return v; }

// Row-tile height of the launch: 128-column layers take the 8-wave 256x128 tile (three LDS stages, one workgroup per
// CU) once that still yields igemm_big() workgroups, else the 4-wave 128x128 tile; narrower layers take 256 rows when the
// pixel count fills the chip.  Shared by the launcher and p2p_igemm_stat_slots.
static int igemm_bm(long long M, int ctiles, unsigned gz) {
    if (ctiles % 128 == 0) {
        if (igemm_big() > 0 && ((M + 255) / 256) * (ctiles / 128) * (long long)gz >= igemm_big()) return 256;
        return 128;
    }
    return M >= 256 * 512 ? 256 : 128;
}

template <typename T, int WM, int WN, int TM, int TN, bool GEN>
static void igemm_go(IgemmArgs& a, unsigned gz, bool vepi, hipStream_t st) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NTHR = WM * WN * 64;
    const int ctiles = (a.ncols + 31) / 32 * 32;
    dim3 grid((a.M + BM - 1) / BM, (ctiles + BN - 1) / BN, gz);
    // three LDS stages when the launch cannot put two workgroups on every CU anyway (<= 320 workgroups) and the
    // K loop is long enough to fill the ring
    const long long nblk = (long long)grid.x * grid.y * grid.z;
    const int nkb_host = (a.taps_per * a.C * (int)sizeof(T)) >> 7;
    int nst = ((nblk <= 320 || NTHR == 512) && nkb_host >= 4) ? 3 : 2;   // 8-wave tiles: one workgroup per CU by design
    if (3 * (size_t)(BM + BN) * 128 > 150 * 1024) nst = 2;
    size_t stage = (size_t)nst * (BM + BN) * 128;
    size_t epi = (size_t)BM * (BN * 4 + 16) + 4096;    // f32 staging of the epilogue is the larger case (+ stats scratch)
    size_t shm = vepi ? (stage > epi ? stage : epi) : stage;
    static bool attr_done = false;      // per template instantiation: allow > 64 KB of dynamic LDS
    if (!attr_done)     // (a refusal is reported by the caller's p2p_check_launch)
        attr_done = (int)p2p_allow_lds((const void*)igemm_kernel<T, WM, WN, TM, TN, GEN, true, 2>, 160 * 1024, "igemm_kernel") &
                    (int)p2p_allow_lds((const void*)igemm_kernel<T, WM, WN, TM, TN, GEN, false, 2>, 160 * 1024, "igemm_kernel") &
                    (int)p2p_allow_lds((const void*)igemm_kernel<T, WM, WN, TM, TN, GEN, true, 3>, 160 * 1024, "igemm_kernel") &
                    (int)p2p_allow_lds((const void*)igemm_kernel<T, WM, WN, TM, TN, GEN, false, 3>, 160 * 1024, "igemm_kernel");
    if (nst == 3) {
        if (vepi) igemm_kernel<T, WM, WN, TM, TN, GEN, true, 3><<<grid, dim3(NTHR), shm, st>>>(a);
        else igemm_kernel<T, WM, WN, TM, TN, GEN, false, 3><<<grid, dim3(NTHR), shm, st>>>(a);
    } else {
        if (vepi) igemm_kernel<T, WM, WN, TM, TN, GEN, true, 2><<<grid, dim3(NTHR), shm, st>>>(a);
        else igemm_kernel<T, WM, WN, TM, TN, GEN, false, 2><<<grid, dim3(NTHR), shm, st>>>(a);
    }
}

// Software-pipelined form: 0 = off (the r03 kernel for every shape), 1 = automatic choice, 2 = always two K groups per 128x128
// tile, 3 = always four-wave 128x128 workgroups (two per CU), 4 = the 256x128 tile wherever its workgroups fill the chip
static int igemm_pipe_mode() { static int v = -1; if (v < 0) v = igemm_env("P2P_IGEMM_PIPE", 1); return v; }

template <typename T, int MODE, int WM, int WN, int TM, int TN, int KG, int NST>
static void igemm_pipe_go(IgemmArgs& a, unsigned gz, bool vepi, hipStream_t st) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NTHR = KG * WM * WN * 64;
    const int ctiles = (a.ncols + 31) / 32 * 32;
    dim3 grid((a.M + BM - 1) / BM, (ctiles + BN - 1) / BN, gz);
    size_t shm = (size_t)NST * (BM + BN) * 128;
    const size_t epi = (size_t)BM * (BN * 4 + 16) + 4096;      // f32 staging of the epilogue is the larger case (+ stats scratch)
    if (vepi && epi > shm) shm = epi;
    static bool attr_done = false;
    if (!attr_done)
        attr_done = (int)p2p_allow_lds((const void*)igemm_pipe_kernel<T, MODE, WM, WN, TM, TN, KG, NST, true>, 160 * 1024, "igemm_pipe_kernel") &
                    (int)p2p_allow_lds((const void*)igemm_pipe_kernel<T, MODE, WM, WN, TM, TN, KG, NST, false>, 160 * 1024, "igemm_pipe_kernel");
    if (vepi) igemm_pipe_kernel<T, MODE, WM, WN, TM, TN, KG, NST, true><<<grid, dim3(NTHR), shm, st>>>(a);
    else igemm_pipe_kernel<T, MODE, WM, WN, TM, TN, KG, NST, false><<<grid, dim3(NTHR), shm, st>>>(a);
}

// The pipelined kernel takes the launch when K-blocks lie inside one tap, the columns fill 128-wide tiles and the op is a
// stride-2 G or P; returns false otherwise (the caller falls back to igemm_kernel).
template <typename T, int MODE>
static bool igemm_pipe_try(IgemmArgs& a, unsigned gz, bool vepi, hipStream_t st) {
    const int ctiles = (a.ncols + 31) / 32 * 32;
    const int nkb = (a.taps_per * a.C * (int)sizeof(T)) >> 7;
    const int bm = igemm_bm(a.M, ctiles, gz);
    const int pm = igemm_pipe_mode();
    // f32 (parity mode) without fused statistics: ONE variant whatever the pixel count, so that a pixel's sum does not depend on
    // the batch it sits in (engine.batch_invariant)
    const bool fixed = sizeof(T) == 4 && a.stat_part == nullptr && pm == 1;
    if (bm == 256 && !fixed) { igemm_pipe_go<T, MODE, 4, 2, 2, 2, 1, 3>(a, gz, vepi, st); return true; }
    const long long tiles = ((a.M + 127) / 128) * (long long)(ctiles / 128) * gz;
    (void)nkb;
    const bool kg2 = fixed || pm == 2 || (pm != 3 && tiles <= 384);      // one workgroup per CU anyway: the second wave of a SIMD splits K
    if (kg2) igemm_pipe_go<T, MODE, 2, 2, 2, 2, 2, 4>(a, gz, vepi, st);
    else igemm_pipe_go<T, MODE, 2, 2, 2, 2, 1, 2>(a, gz, vepi, st);
    return true;
}

template <typename T, bool GEN>
static int igemm_launch(IgemmArgs& a, int phases, bool vepi, hipStream_t st) {
    unsigned gz = (unsigned)(phases * a.splitk);
    const int ctiles = (a.ncols + 31) / 32 * 32;      // launched columns (<= w_rows)
    const bool bigM = a.M >= 256 * 512;                // enough rows to fill the chip with 256-row tiles
    if constexpr (!GEN) {
        // (32-bit per-lane gather offsets: the gathered view and the weights must each span less than 4 GB)
        const long long in_span = ((long long)((a.M >> (a.lgLW + a.lgLH)) + 1) * a.in_img + 4LL * a.in_row + 4) * a.in_ld * (long long)sizeof(T);
        const long long w_span = 16LL * a.w_rows * a.C * (long long)sizeof(T);
        if (igemm_pipe_mode() && a.lgCB >= 7 && ctiles % 128 == 0 && a.mode <= 1 && in_span < 0xffffffffLL && w_span < 0xffffffffLL) {
            const bool ok = a.mode == 0 ? igemm_pipe_try<T, 0>(a, gz, vepi, st) : igemm_pipe_try<T, 1>(a, gz, vepi, st);
            if (ok) return p2p_check_launch("p2p_igemm");
        }
    }
    if (igemm_bm(a.M, ctiles, gz) == 256 && ctiles % 128 == 0) igemm_go<T, 4, 2, 2, 2, GEN>(a, gz, vepi, st);
    // 128x128: eight waves (32x64 each), i.e. twice the waves per SIMD for the same LDS: +12 % over four 64x64 waves
    // (r01, A/B on one device); the 64- and 32-column tiles measured no better with eight waves and keep four
    else if (ctiles % 128 == 0) igemm_go<T, 4, 2, 1, 2, GEN>(a, gz, vepi, st);
    else if (ctiles % 64 == 0) {
        if (bigM) igemm_go<T, 4, 1, 2, 2, GEN>(a, gz, vepi, st);
        else igemm_go<T, 2, 2, 2, 1, GEN>(a, gz, vepi, st);
    } else {
        if (bigM) igemm_go<T, 4, 1, 2, 1, GEN>(a, gz, vepi, st);
        else igemm_go<T, 4, 1, 1, 1, GEN>(a, gz, vepi, st);
    }
    return p2p_check_launch("p2p_igemm");
}

extern "C" int p2p_igemm_stat_slots(int op, int N, int LH, int LW, int ncols);

// Shared implementation.  C = channels of the gathered operand as laid out in `w` and read from the input
// view (whole 16-byte chunks), w_rows = rows per weight tap slab (multiple of 32, >= the launched columns).
static int igemm_common(int op, int stride, int dtype, int N, int LH, int LW, int C, int ncols, int w_rows,
                        const p2p_tensor* in, const p2p_tensor* out, const void* w, const float* bias, int act,
                        float alpha, int splitk, float* slabs, float* stat_part, void* stream) {
    const int esz = dtype == P2P_BF16 ? 2 : 4;
    IgemmArgs a;
    a.C = C; a.ncols = ncols; a.w_rows = w_rows;
    a.bias = bias; a.act = act; a.alpha = alpha;
    a.lgLW = ilog2_exact(LW);
    a.lgLH = ilog2_exact(LH);
    P2P_REQUIRE(a.lgLW >= 0 && a.lgLH >= 0, "p2p_igemm: LH=%d, LW=%d must be powers of two", LH, LW);
    P2P_REQUIRE(((long long)C * esz) % 16 == 0, "p2p_igemm: contraction channels must fill whole 16-byte chunks (C=%d)", C);
    P2P_REQUIRE(w_rows % 32 == 0 && ncols >= 1 && (ncols + 31) / 32 * 32 <= w_rows, "p2p_igemm: bad column counts %d/%d", ncols, w_rows);
    P2P_REQUIRE((in->ld * esz) % 16 == 0 && ((uintptr_t)in->ptr % 16) == 0 && ((uintptr_t)w % 16) == 0,
                "p2p_igemm: input pixels and weights must be 16-byte aligned");
    a.Cc = C * esz / 16;
    a.cc_inv = (unsigned)(((1u << 24) + a.Cc - 1) / a.Cc);
    P2P_REQUIRE(a.Cc <= 1024, "p2p_igemm: more than 1024 16-byte chunks per tap (C=%d)", C);      // cc_inv is exact up to there (checked exhaustively)
    a.lgCB = ilog2_exact((long long)C * esz);
    const bool pow2 = a.lgCB >= 6 && ncols == w_rows && !bias && act == P2P_ACT_NONE;
    a.mode = op == P2P_OP_G ? 0 : (stride == 2 ? 1 : 2);
    a.si = stride;
    // a 1x1 lo map (2x2 hi map) only meets real pixels through 4 of the 16 taps (op G) / 1 of the 4 taps of a phase (op P)
    a.live_taps = (stride == 2 && LH == 1 && LW == 1) ? 1 : 0;
    const int ntaps = a.live_taps ? (a.mode == 1 ? 1 : 4) : (a.mode == 1 ? 4 : 16);
    P2P_REQUIRE(splitk >= 1 && ntaps % splitk == 0, "p2p_igemm: splitk=%d must divide %d", splitk, ntaps);
    a.taps_per = ntaps / splitk;
    P2P_REQUIRE(((long long)a.taps_per * C * esz) % 128 == 0, "p2p_igemm: K per split must be a multiple of 128 bytes");
    P2P_REQUIRE(splitk == 1 || slabs, "p2p_igemm: splitk > 1 needs a slab workspace");
    P2P_REQUIRE(splitk == 1 || (!bias && act == P2P_ACT_NONE), "p2p_igemm: bias/activation need splitk == 1");
    a.in = (const char*)in->ptr; a.in_img = in->img_stride; a.in_row = in->row_stride; a.in_ld = in->ld;
    a.out = (char*)out->ptr; a.out_img = out->img_stride; a.out_row = out->row_stride; a.out_ld = out->ld;
    a.slabs = slabs;
    const int os = a.mode == 1 ? 2 : 1;
    const int OH = os * LH, OW = os * LW;
    a.slab_stride = 0;
    if (splitk > 1) {   // slabs are dense [pixels][ncols]
        a.out_img = (long long)OH * OW; a.out_row = OW; a.out_ld = ncols;
        a.slab_stride = (long long)N * OH * OW * ncols;
    }
    a.w = (const char*)w;
    a.M = N * LH * LW; a.LW = LW; a.LH = LH;
    a.splitk = splitk;
    {   // every XCD has its own L2: whichever operand an XCD's blocks share is fetched once per XCD, the other 8 times
        // (r02 PMC: the deep layers fetched 89 MB per launch for 8-17 MB of weights).  Give the XCDs shares of the LARGER one.
        static int wm = -1;
        if (wm < 0) { const char* e = getenv("P2P_IGEMM_WMAJOR"); wm = e ? atoi(e) : 1; }
        const int in_pix_per_lo = a.mode == 0 ? stride * stride : 1;
        const long long a_bytes = (long long)N * LH * LW * in_pix_per_lo * C * esz;
        const long long w_bytes = (long long)(a.live_taps ? 4 : 16) * ncols * C * esz;
        a.w_major = (wm && w_bytes > a_bytes) ? 1 : 0;
    }
    const int phases = a.mode == 1 ? 4 : 1;
    hipStream_t st = (hipStream_t)stream;
    // vector epilogue: whole 16-byte chunks of each pixel's channel run must be addressable
    const int osz = splitk > 1 ? 4 : esz, ce = 16 / osz;
    const bool vepi = ncols % ce == 0 && (a.out_ld * osz) % 16 == 0 &&
                      (splitk > 1 ? ((uintptr_t)slabs % 16) == 0 : ((uintptr_t)out->ptr % 16) == 0);
    // fused statistics: whole tiles, each tile row group inside one image (see p2p_igemm_stat_slots)
    a.stat_part = nullptr; a.stat_rows = 0; a.stat_slots = 0; a.lgHW = a.lgLW + a.lgLH;
    if (stat_part) {
        const int slots = p2p_igemm_stat_slots(op, N, LH, LW, ncols);
        P2P_REQUIRE(slots > 0 && vepi && splitk == 1, "p2p_igemm: fused statistics not available for this shape (query p2p_igemm_stat_slots)");
        const int ctl = (ncols + 31) / 32 * 32;
        const int bm = igemm_bm((long long)N * LH * LW, ctl, (unsigned)phases);
        const int hw = LH * LW;
        a.stat_part = stat_part;
        a.stat_rows = hw < bm ? hw : bm;
        a.stat_slots = slots;
    }
    if (pow2) { P2P_DISPATCH_DTYPE(dtype, return (igemm_launch<T, false>(a, phases, vepi, st))); }
    else { P2P_DISPATCH_DTYPE(dtype, return (igemm_launch<T, true>(a, phases, vepi, st))); }
}

// Number of statistics slots per image that the fused epilogue writes for this launch shape, or 0 if the statistics
// cannot be fused (ragged tiles): slot partials are (mean, centred sum of squares) over LH*LW*phases/slots pixels each.
extern "C" int p2p_igemm_stat_slots(int op, int N, int LH, int LW, int ncols) {
    const long long M = (long long)N * LH * LW;
    const int ctl = (ncols + 31) / 32 * 32;
    const int phases_q = op == P2P_OP_P ? 4 : 1;
    const int bm = igemm_bm(M, ctl, (unsigned)phases_q);
    const int hw = LH * LW;
    if (hw * phases_q <= 16) return 0;      // output maps of <= 16 pixels: p2p_norm_act_fwd takes its own statistics there
    if (M % bm != 0) return 0;
    if (!(hw % bm == 0 || bm % hw == 0)) return 0;
    const int phases = op == P2P_OP_P ? 4 : 1;
    return phases * (hw >= bm ? hw / bm : 1);
}

// block-resident form for the wide maps (brig.hip)
extern "C" int p2p_brig_ok(int op, int dtype, int N, int LH, int LW, int Cg, int Cd);
extern "C" int p2p_brig_stat_slots(int op, int dtype, int N, int LH, int LW, int Cg, int Cd);
struct BrigNorm;
int brig_launch(int op, int dtype, int N, int LH, int LW, int Cg, int Cd, const p2p_tensor* hi, const p2p_tensor* lo,
                const void* w, float* stat_part, void* stream, const BrigNorm* norm = nullptr);

// Statistics slots of p2p_igemm for a layer, whichever of its two kernels takes the shape (block-resident form on the
// wide bf16 maps, im2col form otherwise).
extern "C" int p2p_igemm_layer_stat_slots(int op, int dtype, int N, int LH, int LW, int Cg, int Cd) {
    if (p2p_brig_ok(op, dtype, N, LH, LW, Cg, Cd)) return p2p_brig_stat_slots(op, dtype, N, LH, LW, Cg, Cd);
    return p2p_igemm_stat_slots(op, N, LH, LW, op == P2P_OP_G ? Cd : Cg);
}

extern "C" int p2p_igemm(int op, int dtype, int N, int LH, int LW, int Cg, int Cd, const p2p_tensor* hi,
                         const p2p_tensor* lo, const void* w, int splitk, float* slabs, float* stat_part, void* stream) {
    P2P_REQUIRE(op == P2P_OP_G || op == P2P_OP_P, "p2p_igemm: op must be G or P");
    P2P_REQUIRE(N > 0 && LH > 0 && LW > 0, "p2p_igemm: bad shape");
    P2P_REQUIRE(Cg % 32 == 0 && Cd % 32 == 0 && Cg > 0 && Cd > 0, "p2p_igemm: Cg=%d, Cd=%d must be multiples of 32", Cg, Cd);
    P2P_REQUIRE(hi && lo && hi->ptr && lo->ptr && w, "p2p_igemm: null pointer");
    if (splitk == 1 && p2p_brig_ok(op, dtype, N, LH, LW, Cg, Cd))
        return brig_launch(op, dtype, N, LH, LW, Cg, Cd, hi, lo, w, stat_part, stream);
    const p2p_tensor* in = op == P2P_OP_G ? hi : lo;
    const p2p_tensor* out = op == P2P_OP_G ? lo : hi;
    const int C = op == P2P_OP_G ? Cg : Cd, ncols = op == P2P_OP_G ? Cd : Cg;
    return igemm_common(op, 2, dtype, N, LH, LW, C, ncols, ncols, in, out, w, nullptr, P2P_ACT_NONE, 0.f, splitk, slabs, stat_part, stream);
}

extern "C" int p2p_igemm_edge(int op, int stride, int dtype, int N, int LH, int LW, int cin_pad, int ncols,
                              int w_rows, const p2p_tensor* in, const p2p_tensor* out, const void* w,
                              const float* bias, int act, float alpha, void* stream) {
    P2P_REQUIRE(op == P2P_OP_G || op == P2P_OP_P, "p2p_igemm_edge: op must be G or P");
    P2P_REQUIRE(stride == 1 || stride == 2, "p2p_igemm_edge: stride must be 1 or 2");
    P2P_REQUIRE(N > 0 && LH > 0 && LW > 0 && cin_pad > 0, "p2p_igemm_edge: bad shape");
    P2P_REQUIRE(in && out && in->ptr && out->ptr && w, "p2p_igemm_edge: null pointer");
    return igemm_common(op, stride, dtype, N, LH, LW, cin_pad, ncols, w_rows, in, out, w, bias, act, alpha, 1, nullptr, nullptr, stream);
}
