// Block-resident implicit GEMM for the stride-2 4x4 blocks on WIDE maps (lo grid 8x8 .. 64x64; bf16): Conv2D /
// Conv2DTranspose forward and their data gradients (reference call sites networks.py:10-16,26-27 and the tape
// gradients taken at pix2pix_model.py:78-79), same math as igemm.hip:
//   op P: hi[n,2y+ph,2x+pw][g] = sum_{2x2 taps of phase (ph,pw)} sum_d lo[n,y+dy,x+dx,d] * Wn[t][g][d]
//   op G: lo[n,y,x][d]         = sum_{16 taps} sum_g hi[n,2y+kh-1,2x+kw-1,g] * Wt[t][d][g]
// Why a second kernel: in the im2col form (igemm.hip) every input pixel travels L2 -> LDS once per tap that touches it
// (16 (phase, tap) pairs for op P, 4 for op G) and once more per output-channel tile; measured (tools/ubench, r02) that
// kernel's LDS-DMA side alone runs at 20-26 B/clk/CU, most of it served from beyond L2, and bounds the launch.  Here
// a workgroup owns 256 lo pixels (whole images on maps <= 16x16, a strip of rows otherwise):
//   * the input block those pixels need (with its halo) is brought into LDS ONCE per K chunk (CK input channels) and
//     every tap / sub-pixel phase reads it at a per-lane offset: op P merges the four phases, so the 16 (phase, tap)
//     products share one copy of the lo block (9 distinct offsets); op G reads a column-parity de-interleaved image
//     so that its stride-2 gathers are unit-stride in LDS;
//   * only the weights stream: a ring of four 16 KB stages ([4 phases][64 channels][CK] for op P, [2 taps][256
//     channels][CK] for op G), LDS-DMA (global_load_lds_dwordx4) with a counted s_waitcnt so that two stages and the
//     next input block stay in flight across the single barrier of a step (16 MFMAs per wave);
//   * eight waves = 4 (phase | 64-channel quarter) x 2 (pixel halves), wave tile 128 pixels x 64 channels
//     (4 x 2 MFMA tiles, 128 accumulator registers), weights are the MFMA A operand so a lane ends with 4 consecutive
//     channels of one pixel;
//   * LDS images are lane-linear DMA targets, so the bank swizzle (16-byte slot ^= row / rows-per-256-bytes) goes on
//     the per-lane SOURCE chunk and on the ds_read_b128 address; the lane <-> pixel assignment of a 32-pixel MFMA
//     column block is permuted (rotation on 16-wide maps, group split on 8-wide maps) so that the 16 lanes of every
//     ds_read_b128 lane group hit 16 distinct 16-byte slots for every tap offset;
//   * epilogue: per-wave LDS patch -> whole 128-byte lines per pixel; InstanceNorm statistics (networks.py:18,29) of the
//     rounded values are taken from the patch -- with whole images in a workgroup they are complete per (image, channel).
#include "p2p_common.hpp"
#include <stdlib.h>
#include <utility>

// Diagnostic builds only (tools/ubench): 1 = DMA and barriers without LDS reads / MFMAs, 2 = LDS reads + MFMAs without DMA
#ifndef P2P_ABL
#define P2P_ABL 0
#endif

struct BrigArgs {
    const char* in; long long in_img; int in_row; int in_ld;      // gathered view (lo for op P, hi for op G), element strides
    char* out; long long out_img; int out_row; int out_ld;        // output view (hi for op P, lo for op G)
    const char* w;                                                  // [16][ncols][C] bf16 (wn for op P, wt for op G)
    float* stat_part; int stat_slots;                               // optional [N][slots][ncols][2]
    int C, ncols;            // contraction channels per tap; output channels (= rows of a weight tap slab)
    int N, LH, LW, lgLW;
    int ipt, rpt, tiles_per_img, ntiles, nnt;   // images / lo rows per tile, tiles per image, M tiles, N tiles
    int BR, PITCH, HALF, BP, npix, npieces;    // input block per image: rows, row pitch (LDS pixels), parity-plane offset (op G), pixels; per tile: pixels, DMA pieces
    int abytes;              // bytes of one input block buffer (npieces KB)
    int nkc;                 // K chunks
    int rot;                 // lane rotation per block row on 16-wide maps (row pitch mod 16)
    long long in_lo;         // most negative byte offset from `in` that the block gathers (halo of the first image): per-lane offsets are unsigned from there
};

__device__ __forceinline__ void brig_glds16(const char* g, char* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <int N> __device__ __forceinline__ void brig_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int... I, typename F>
__device__ __forceinline__ void brig_static_for(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}

// pixel `i` (0..31) of MFMA column block `pb` (0..7) of a tile -> (image in tile, lo row in tile, lo column).  The
// assignment makes the LDS pixel indices of each ds_read_b128 lane group {0-3,12-15,20-27} / {4-11,16-19,28-31}
// distinct modulo 16 (MI355X_MICROARCH.md, LDS) whatever the tap offset.
__device__ __forceinline__ void brig_lane_pixel(const BrigArgs& a, int pb, int i, int& img, int& ly, int& lx) {
    if (a.lgLW >= 5) {                      // one row segment of 32 consecutive columns
        const int lgb = a.lgLW - 5;
        ly = pb >> lgb; lx = ((pb & ((1 << lgb) - 1)) << 5) + i; img = 0;
    } else if (a.lgLW == 4) {               // two rows of 16: the second row is rotated by the row pitch excess
        const int k = i >> 4;
        ly = pb * 2 + k; lx = ((i & 15) - k * a.rot) & 15; img = 0;
    } else {                                // four rows of 8: lane group 0 takes rows 0 and 2, lane group 1 rows 1 and 3
        const int g = (0x96 >> (i >> 2)) & 1;      // lanes 0-3,12-15,20-27 -> group 0; 4-11,16-19,28-31 -> group 1
        int o;
        if (g == 0) o = i < 4 ? i : (i < 16 ? i - 8 : i - 12);
        else o = i < 12 ? i - 4 : (i < 20 ? i - 8 : i - 16);
        const int k = 2 * (o >> 3) + g;
        const int row = pb * 4 + k;         // 0..31 over the 4 images of the tile
        img = row >> 3; ly = row & 7; lx = o & 7;
    }
}

// MODE 1: op P (lo -> hi, four phases merged), CK = 32.   MODE 0: op G (hi -> lo), CK = 16.
template <int MODE>
__global__ __launch_bounds__(512) void brig_kernel(BrigArgs a) {
    constexpr int CK = MODE == 1 ? 32 : 16;
    constexpr int RB = CK * 2;              // bytes per LDS row (one pixel / one weight row of the chunk)
    constexpr int NQ = RB / 16;             // 16-byte chunks per row
    constexpr int LGRPB = MODE == 1 ? 2 : 3;   // log2(rows per 256 bytes)
    constexpr int LGPPP = MODE == 1 ? 4 : 5;
    constexpr int LGNQ = MODE == 1 ? 2 : 1;
    constexpr int NT = MODE == 1 ? 4 : 8;   // steps per K chunk (op P: one tap of each phase; op G: two taps)
    constexpr int NWST = 4, WST = 16384;    // weight ring
    constexpr int PA = MODE == 1 ? 4 : 6;   // input-block pieces a wave issues per K chunk (duplicates pad the count)
    constexpr int PATCH = 32 * 144;         // epilogue patch per wave: 32 pixels x (64 channels bf16 + 16 B pad)
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int quarter = wave & 3, half = wave >> 2;
    const int r = lane & 31, h = lane >> 5;
    const int ph = quarter >> 1, pw = quarter & 1;       // op P: the quarter is a sub-pixel phase
    char* const Abuf = smem;
    char* const Wring = smem + 2 * a.abytes;

    const unsigned nblk = gridDim.x;
    const unsigned lin = xcd_remap(blockIdx.x, nblk);
    const int nt_i = lin % a.nnt, tile = lin / a.nnt;
    const int n0c = nt_i * (MODE == 1 ? 64 : 256);       // first output channel of this workgroup
    int img0, y0;
    if (a.tiles_per_img > 1) { img0 = tile / a.tiles_per_img; y0 = (tile - img0 * a.tiles_per_img) * a.rpt; }
    else { img0 = tile * a.ipt; y0 = 0; }

    const int esz = 2;
    const long long pixB = (long long)a.in_ld * esz;

    // ---- input block DMA: per-lane source bases (channel offset of the K chunk is added per issue) --------------------------
    unsigned abase[PA];     // byte offsets from a.in (the launcher checks that the view spans < 4 GB)
    int adst[PA];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        int p = wave + 8 * i;
        while (p >= a.npieces) p -= 8;
        int idx = (p << LGPPP) + (lane >> LGNQ);
        const int s = lane & (NQ - 1);
        const int q = s ^ ((idx >> LGRPB) & (NQ - 1));
        if (idx >= a.npix) idx = 0;
        const int img = idx / a.BP, rem = idx - img * a.BP;
        const int by = rem / a.PITCH, bi = rem - by * a.PITCH;
        int n = img0 + img;
        if (n >= a.N) n = a.N - 1;
        int y, x;
        if (MODE == 1) { const int bx = bi < a.LW + 2 ? bi : 0; y = y0 + by - 1; x = bx - 1; }
        else { const int par = bi >= a.HALF ? 1 : 0; const int bx = 2 * (bi - par * a.HALF) + par; y = 2 * y0 + by - 1; x = bx - 1; }
        abase[i] = (unsigned)(((long long)n * a.in_img + (long long)y * a.in_row + x) * pixB + q * 16 - a.in_lo);
        adst[i] = p * 1024;
    }
    // ---- weight stage DMA: two pieces per wave ---------------------------------------------------------------------------------
    unsigned wbase[2];      // byte offsets from a.w
    const long long tapB = (long long)a.ncols * a.C * esz;       // bytes per weight tap slab
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int p = wave + 8 * j;
        const int row = (p << LGPPP) + (lane >> LGNQ);
        const int s = lane & (NQ - 1);
        const int q = s ^ ((row >> LGRPB) & (NQ - 1));
        long long wrow;
        if (MODE == 1) {
            const int phs = row >> 6, n = row & 63;
            const int widx0 = (1 - (phs >> 1)) * 4 + (1 - (phs & 1));       // tap t = 0 of that phase
            wrow = (long long)widx0 * a.ncols + n0c + n;
        } else {
            const int tsel = row >> 8, n = row & 255;
            wrow = (long long)tsel * a.ncols + n0c + n;
        }
        wbase[j] = (unsigned)(wrow * a.C * esz + q * 16);
    }
    auto issue_w = [&](int step) {          // stage of global step `step` (clamped past the end: lands in a free slot, never read)
        const int total = a.nkc * NT;
        if (step >= total) step = total - 1;
        const int kc = step / NT, t = step - kc * NT;
        const long long off = (MODE == 1 ? (long long)((t >> 1) * 8 + (t & 1) * 2) : (long long)(2 * t)) * tapB + (long long)kc * CK * esz;
        char* dst = Wring + (step & (NWST - 1)) * WST;
        const char* src = a.w + off;
        brig_glds16(src + wbase[0], dst + wave * 1024);
        brig_glds16(src + wbase[1], dst + (wave + 8) * 1024);
    };
    auto issue_a = [&](int kc, int i) {     // piece i of this wave for K chunk kc
        if (kc >= a.nkc) kc = a.nkc - 1;
        brig_glds16(a.in + a.in_lo + (long long)kc * CK * esz + abase[i], Abuf + (kc & 1) * a.abytes + adst[i]);
    };

    // ---- fragment addresses ----------------------------------------------------------------------------------------------------
    int aidx[4];            // LDS pixel index of this lane's pixel in each of the wave's 4 column blocks (tap offset 0)
#pragma unroll
    for (int pbi = 0; pbi < 4; ++pbi) {
        int img, ly, lx;
        brig_lane_pixel(a, half * 4 + pbi, r, img, ly, lx);
        aidx[pbi] = MODE == 1 ? img * a.BP + ly * a.PITCH + lx : img * a.BP + 2 * ly * a.PITCH + lx;
    }
    // weight fragment offsets inside a stage: row = [u * 256 (op G)] + quarter * 64 + cb * 32 + r, chunk q swizzled by the row
    int woff[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const int row = (MODE == 1 ? 0 : u * 256) + quarter * 64 + cb * 32 + r;
            const int q = MODE == 1 ? 2 * u + h : h;
            woff[u][cb] = row * RB + ((q ^ ((row >> LGRPB) & (NQ - 1))) << 4);
        }

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // ---- prologue: first input block, first NWST-1 weight stages ----------------------------------------------------------------
#pragma unroll
    for (int i = 0; P2P_ABL != 2 && i < PA; ++i) issue_a(0, i);
#pragma unroll
    for (int s = 0; P2P_ABL != 2 && s < NWST - 1; ++s) issue_w(s);

    for (int kc = 0; kc < a.nkc; ++kc) {
        const char* Acur = Abuf + (kc & 1) * a.abytes;
        brig_static_for(std::make_integer_sequence<int, NT>{}, [&](auto tt) {
            constexpr int t = decltype(tt)::value;
            // pieces issued after this step's weight stage: the two following steps' stages + their input-block pieces
            constexpr int NA1 = MODE == 1 ? (((t + NT - 1) % NT) < 2 ? 2 : 0) : (((t + NT - 1) % NT) < PA ? 1 : 0);
            constexpr int NA2 = MODE == 1 ? (((t + NT - 2) % NT) < 2 ? 2 : 0) : (((t + NT - 2) % NT) < PA ? 1 : 0);
            brig_wait_vm<4 + NA1 + NA2>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            // every wave has finished step s-1: its weight slot and (at t = 0) the other input buffer are free
            if (P2P_ABL != 2) {
                if (MODE == 1) { if (t < 2) { issue_a(kc + 1, 2 * t); issue_a(kc + 1, 2 * t + 1); } }
                else { if (t < PA) issue_a(kc + 1, t); }
                issue_w(kc * NT + t + NWST - 1);
            }
            const char* Wcur = Wring + (t & (NWST - 1)) * WST;
#pragma unroll
            for (int u = 0; P2P_ABL != 1 && u < 2; ++u) {
                int toff, q;
                if (MODE == 1) {
                    const int kh = (1 - ph) + 2 * (t >> 1), kw = (1 - pw) + 2 * (t & 1);
                    const int dy = (ph + 1 - kh) >> 1, dx = (pw + 1 - kw) >> 1;
                    toff = (1 + dy) * a.PITCH + 1 + dx;
                    q = 2 * u + h;
                } else {
                    const int tap = 2 * t + u, kh = tap >> 2, kw = tap & 3;
                    toff = kh * a.PITCH + (kw & 1) * a.HALF + (kw >> 1);
                    q = h;
                }
                bf16x8 wf[2], af[4];
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) wf[cb] = *(const bf16x8*)(Wcur + woff[u][cb]);
#pragma unroll
                for (int pbi = 0; pbi < 4; ++pbi) {
                    int base = aidx[pbi];
                    // op G has 16 tap offsets x 4 column blocks: hoisted out of the K loop they would cost 64 registers (spills)
                    if (MODE == 0) asm volatile("" : "+v"(base));
                    const int idx = base + toff;
                    af[pbi] = *(const bf16x8*)(Acur + idx * RB + ((q ^ ((idx >> LGRPB) & (NQ - 1))) << 4));
                }
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int pbi = 0; pbi < 4; ++pbi)
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
                        acc[pbi][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[cb], af[pbi], acc[pbi][cb], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
            }
        });
    }

    // ---- epilogue ----------------------------------------------------------------------------------------------------------------
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();                        // every DMA (the clamped tail ones too) has landed, every wave is done with the operands
    char* const pL = smem + wave * PATCH;
    float* const stL = (float*)(smem + 8 * PATCH);           // [8 waves][2 image selectors][64 channels][2]
    const bool want_stats = a.stat_part != nullptr;
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};            // lane = channel (0..63): sums over the wave's pixels, per image selector
    typedef __attribute__((__vector_size__(4 * sizeof(bf16_t)))) bf16_t bf16x4;
#pragma unroll
    for (int pbi = 0; pbi < 4; ++pbi) {
        const int pb = half * 4 + pbi;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 qv;
#pragma unroll
                for (int k = 0; k < 4; ++k) qv[k] = (bf16_t)acc[pbi][cb][4 * g + k];
                *(bf16x4*)(pL + r * 144 + (cb * 32 + 8 * g + 4 * h) * 2) = qv;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (want_stats) {
            const int sel = a.lgLW == 3 ? (pbi >> 1) : 0;
            const bf16_t* col = (const bf16_t*)pL + lane;
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const float v = (float)col[i * 72];
                t1 += v;
                t2 += v * v;
            }
            if (sel == 0) { s1[0] += t1; s2[0] += t2; } else { s1[1] += t1; s2[1] += t2; }
        }
        const int ch8 = lane & 7;
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int pix = ps * 8 + (lane >> 3);
            int img, ly, lx;
            brig_lane_pixel(a, pb, pix, img, ly, lx);
            const int n = img0 + img;
            if (n < a.N) {
                const f32x4 v = *(const f32x4*)(pL + pix * 144 + ch8 * 16);
                long long opix;
                if (MODE == 1) opix = (long long)n * a.out_img + (long long)(2 * (y0 + ly) + ph) * a.out_row + (2 * lx + pw);
                else opix = (long long)n * a.out_img + (long long)(y0 + ly) * a.out_row + lx;
                const int c0 = n0c + (MODE == 1 ? 0 : quarter * 64) + ch8 * 8;
                *(f32x4*)((bf16_t*)a.out + opix * a.out_ld + c0) = v;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
    if (want_stats) {
        stL[((wave * 2 + 0) * 64 + lane) * 2 + 0] = s1[0];
        stL[((wave * 2 + 0) * 64 + lane) * 2 + 1] = s2[0];
        stL[((wave * 2 + 1) * 64 + lane) * 2 + 0] = s1[1];
        stL[((wave * 2 + 1) * 64 + lane) * 2 + 1] = s2[1];
        __syncthreads();
        // groups of (image, channel): 8-wide maps hold 4 images per tile (image = half * 2 + selector), wider maps one image /
        // strip per tile.  op P: the four phases (quarters) of a half cover the same channels; op G: each quarter has its own 64.
        const int nimg = a.lgLW == 3 ? 4 : 1;
        const int nch = MODE == 1 ? 64 : 256;
        for (int e = tid; e < nimg * nch; e += 512) {
            const int img = e / nch, ch = e - img * nch;
            float t1 = 0.f, t2 = 0.f;
            const int c = ch & 63;
            if (a.lgLW == 3) {
                const int hf = img >> 1, sel = img & 1;
                if (MODE == 1) {
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) { const int wv = hf * 4 + qd; t1 += stL[((wv * 2 + sel) * 64 + c) * 2]; t2 += stL[((wv * 2 + sel) * 64 + c) * 2 + 1]; }
                } else {
                    const int wv = hf * 4 + (ch >> 6);
                    t1 = stL[((wv * 2 + sel) * 64 + c) * 2]; t2 = stL[((wv * 2 + sel) * 64 + c) * 2 + 1];
                }
            } else {
                if (MODE == 1) {
#pragma unroll
                    for (int wv = 0; wv < 8; ++wv) { t1 += stL[((wv * 2) * 64 + c) * 2]; t2 += stL[((wv * 2) * 64 + c) * 2 + 1]; }
                } else {
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) { const int wv = hf * 4 + (ch >> 6); t1 += stL[((wv * 2) * 64 + c) * 2]; t2 += stL[((wv * 2) * 64 + c) * 2 + 1]; }
                }
            }
            const int n = img0 + img;
            if (n < a.N) {
                const float cnt = (float)((MODE == 1 ? 4 : 1) * a.rpt * a.LW);
                const float mean = t1 / cnt;
                const int slot = a.tiles_per_img > 1 ? tile - img0 * a.tiles_per_img : 0;
                float* dst = a.stat_part + (((long long)n * a.stat_slots + slot) * a.ncols + n0c + ch) * 2;
                dst[0] = mean;
                dst[1] = fmaxf(t2 - t1 * mean, 0.f);
            }
        }
    }
}

struct BrigPlan { int ok, ipt, rpt, tiles_per_img, ntiles, nnt, BR, PITCH, HALF, BP, npix, npieces, abytes, nkc, rot, slots; size_t shm; };

static int brig_enabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("P2P_BRIG"); v = e ? atoi(e) : 3; }      // bit 0: op G, bit 1: op P
    return v;
}

static BrigPlan brig_plan(int op, int dtype, int N, int LH, int LW, int Cg, int Cd) {
    BrigPlan p = {};
    if (dtype != P2P_BF16 || N < 1 || LH != LW) return p;
    if (LW != 8 && LW != 16 && LW != 32 && LW != 64) return p;
    const int mode = op == P2P_OP_P ? 1 : (op == P2P_OP_G ? 0 : -1);
    if (mode < 0) return p;
    if (!((brig_enabled() >> mode) & 1)) return p;
    const int C = mode == 1 ? Cd : Cg, ncols = mode == 1 ? Cg : Cd;
    const int CK = mode == 1 ? 32 : 16, BN = mode == 1 ? 64 : 256;
    if (C % 32 != 0 || ncols % BN != 0) return p;      // p2p_igemm's own contract: channel counts are multiples of 32
    int lg = 0;
    while ((1 << lg) < LW) ++lg;
    const int hw = LH * LW;
    p.ipt = hw >= 256 ? 1 : 256 / hw;
    p.rpt = hw >= 256 ? 256 / LW : LH;
    p.tiles_per_img = LH / p.rpt;
    p.ntiles = p.tiles_per_img > 1 ? N * p.tiles_per_img : (N + p.ipt - 1) / p.ipt;
    p.nnt = ncols / BN;
    if (mode == 1) {
        p.BR = p.rpt + 2;
        p.PITCH = LW == 8 ? 12 : LW + 2;      // 8-wide maps: pitch 12 balances the pixel indices modulo 16 over the lane groups
        p.HALF = 0;
        p.rot = p.PITCH & 15;
    } else {
        p.BR = 2 * p.rpt + 2;
        p.HALF = LW + 1;
        p.PITCH = 2 * p.HALF;
        p.rot = (2 * p.PITCH) & 15;
    }
    p.BP = p.BR * p.PITCH;
    p.npix = p.ipt * p.BP;
    const int ppp = 1024 / (CK * 2);
    p.npieces = (p.npix + ppp - 1) / ppp;
    // a lane may address up to one tap offset past its pixel: pad the buffer so that those reads stay inside it
    p.abytes = (p.npieces + 1) * 1024;
    if (p.npieces > 8 * (mode == 1 ? 4 : 6)) return p;
    p.nkc = C / CK;
    p.slots = p.tiles_per_img;
    p.shm = 2 * (size_t)p.abytes + 4 * 16384;
    const size_t epi = 8 * 32 * 144 + 8 * 2 * 64 * 2 * sizeof(float);
    if (p.shm < epi) p.shm = epi;
    if (p.shm > 160 * 1024) return p;
    p.ok = 1;
    return p;
}

extern "C" int p2p_brig_ok(int op, int dtype, int N, int LH, int LW, int Cg, int Cd) {
    return brig_plan(op, dtype, N, LH, LW, Cg, Cd).ok;
}

extern "C" int p2p_brig_stat_slots(int op, int dtype, int N, int LH, int LW, int Cg, int Cd) {
    const BrigPlan p = brig_plan(op, dtype, N, LH, LW, Cg, Cd);
    return p.ok ? p.slots : 0;
}

int brig_launch(int op, int dtype, int N, int LH, int LW, int Cg, int Cd, const p2p_tensor* hi, const p2p_tensor* lo,
                const void* w, float* stat_part, void* stream) {
    const BrigPlan p = brig_plan(op, dtype, N, LH, LW, Cg, Cd);
    P2P_REQUIRE(p.ok, "p2p_brig: shape not supported (query p2p_brig_ok)");
    const p2p_tensor* in = op == P2P_OP_G ? hi : lo;
    const p2p_tensor* out = op == P2P_OP_G ? lo : hi;
    P2P_REQUIRE((in->ld * 2) % 16 == 0 && ((uintptr_t)in->ptr % 16) == 0 && ((uintptr_t)w % 16) == 0,
                "p2p_brig: input pixels and weights must be 16-byte aligned");
    P2P_REQUIRE((out->ld * 2) % 16 == 0 && ((uintptr_t)out->ptr % 16) == 0, "p2p_brig: output pixels must be 16-byte aligned");
    BrigArgs a;
    a.in = (const char*)in->ptr; a.in_img = in->img_stride; a.in_row = in->row_stride; a.in_ld = in->ld;
    a.out = (char*)out->ptr; a.out_img = out->img_stride; a.out_row = out->row_stride; a.out_ld = out->ld;
    a.w = (const char*)w;
    a.stat_part = stat_part; a.stat_slots = p.slots;
    a.C = op == P2P_OP_P ? Cd : Cg; a.ncols = op == P2P_OP_P ? Cg : Cd;
    a.N = N; a.LH = LH; a.LW = LW;
    a.lgLW = 0;
    while ((1 << a.lgLW) < LW) ++a.lgLW;
    a.ipt = p.ipt; a.rpt = p.rpt; a.tiles_per_img = p.tiles_per_img; a.ntiles = p.ntiles; a.nnt = p.nnt;
    a.BR = p.BR; a.PITCH = p.PITCH; a.HALF = p.HALF; a.BP = p.BP; a.npix = p.npix; a.npieces = p.npieces;
    a.abytes = p.abytes; a.nkc = p.nkc; a.rot = p.rot;
    // per-lane gather offsets are 32-bit, counted from the lowest address a block touches (row -1, column -1 of image 0)
    a.in_lo = -((long long)in->row_stride + 1) * in->ld * 2;
    const long long span = ((long long)(N - 1) * in->img_stride + (long long)(op == P2P_OP_P ? LH + 1 : 2 * LH + 1) * in->row_stride +
                            (op == P2P_OP_P ? LW + 1 : 2 * LW + 1)) * in->ld * 2 - a.in_lo;
    P2P_REQUIRE(span < 0xffffffffLL && (long long)16 * a.ncols * a.C * 2 < 0xffffffffLL, "p2p_brig: view larger than 4 GB");
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)brig_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)brig_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    const dim3 grid((unsigned)(p.ntiles * p.nnt));
    hipStream_t st = (hipStream_t)stream;
    if (op == P2P_OP_P) brig_kernel<1><<<grid, dim3(512), p.shm, st>>>(a);
    else brig_kernel<0><<<grid, dim3(512), p.shm, st>>>(a);
    return p2p_check_launch("p2p_igemm(block-resident)");
}
