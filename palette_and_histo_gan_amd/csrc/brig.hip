// Block-resident implicit GEMM for the stride-2 4x4 blocks on WIDE maps (lo grid 8x8 .. 64x64; bf16): Conv2D /
// Conv2DTranspose forward and their data gradients (reference call sites networks.py:10-16,26-27 and the tape
// gradients taken at pix2pix_model.py:78-79), same math as igemm.hip:
//   op P: hi[n,2y+ph,2x+pw][g] = sum_{2x2 taps of phase (ph,pw)} sum_d lo[n,y+dy,x+dx,d] * Wn[t][g][d]
//   op G: lo[n,y,x][d]         = sum_{16 taps} sum_g hi[n,2y+kh-1,2x+kw-1,g] * Wt[t][d][g]
// Why a second kernel: in the im2col form (igemm.hip) every input pixel travels L2 -> LDS once per tap that touches it
// (16 (phase, tap) pairs for op P, 4 for op G) and once more per output-channel tile; measured (tools/ubench, r02) that
// kernel's LDS-DMA side alone runs at 20-26 B/clk/CU, most of it served from beyond L2, and bounds the launch.  Here
// a workgroup owns 256 lo pixels (whole images on maps <= 16x16, a strip of rows otherwise):
//   * op P: the lo block those pixels need (with its one-pixel halo) is brought into LDS ONCE per K chunk (32 channels)
//     and the 16 (phase, tap) products read it at 9 distinct per-lane offsets: the four sub-pixel phases are merged
//     (wave & 3 = phase);
//   * op G is the adjoint: hi is split into its four (row, column)-parity planes; a K chunk = one plane x 32 channels,
//     its block is (rows + 1) x (LW + 1) pixels, and the four taps that fall on that plane read it at unit stride
//     (kh = 1 - p + 2a, kw = 1 - q + 2b reads plane pixel (y + a, x + b): the same index algebra as a phase of op P);
//   * only the weights stream: a ring of four 16 KB stages (one tap x [4 phases | 4 channel quarters] x 64 output
//     channels x 32 input channels, or two taps x 32 output channels: CBW below), LDS-DMA (global_load_lds_dwordx4)
//     with a counted s_waitcnt so that two stages and the next input block stay in flight across the single barrier
//     of a step (16 MFMAs per wave);
//   * eight waves = 4 (phase | channel quarter) x 2 (pixel halves), wave tile 128 pixels x 64 or 32 channels; weights
//     are the MFMA A operand so a lane ends with 4 consecutive channels of one pixel.  CBW = 1 (32 channels per wave)
//     is chosen when the 64-channel form would launch fewer than ~3/4 of the CUs (8x8 maps at batch 256);
//   * LDS images are lane-linear DMA targets, so the bank swizzle (16-byte slot ^= (row >> 2) & 3 on 64-byte rows)
//     goes on the per-lane SOURCE chunk and on the ds_read_b128 address; the lane <-> pixel assignment of a 32-pixel
//     MFMA column block is permuted (rotation on 16-wide maps, group split on 8-wide maps) so that the 16 lanes of
//     every ds_read_b128 lane group hit 16 distinct 16-byte slots for every tap offset;
//   * epilogue: per-wave LDS patch -> whole 64/128-byte runs per pixel; InstanceNorm statistics (networks.py:18,29) of
//     the rounded values are taken from the patch -- with whole images in a workgroup they are complete per (image, channel).
#include "p2p_common.hpp"
#include <stdlib.h>
#include <utility>

// Diagnostic builds only (tools/ubench/build_abl.sh): 1 = DMA and barriers without LDS reads / MFMAs, 2 = LDS reads + MFMAs
// without DMA, 3 = everything but never wait for the DMA (stale operands), 4 / 5 = as 1 with the input blocks / the weights only,
// 6 / 7 = everything (MFMAs included) but without the weight / input-block DMA
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
    int BR, PITCH, BP, npix, npieces;          // input block per image: rows, row pitch (LDS pixels), pixels; per tile: pixels, DMA pieces
    int abytes;              // bytes of one input block buffer
    int nkc;                 // K chunks (op P: C / 32; op G: 4 planes x C / 32)
    int rot;                 // lane rotation per block row on 16-wide maps (row pitch mod 16)
    int stagger;             // waves 4-7 issue their DMA after their first MFMA group (P2P_BRIG_STAGGER, default on)
    // fused InstanceNorm + activation (whole images per tile): y = act((x - mean) * rstd * gamma + beta) written to act_out,
    // (mean, rstd) to norm_stats[N][ncols][2]; the rounded conv output still goes to `out` (the backward pass reads it)
    const float* gamma; const float* beta; float eps; int act; float alpha;
    char* act_out; long long act_img; int act_row; int act_ld;
    float* norm_stats;
    long long in_lo;         // most negative byte offset from `in` that a block gathers: per-lane offsets are unsigned from there
};

__device__ __forceinline__ void brig_glds16(const char* g, char* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <int N> __device__ __forceinline__ void brig_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// Fragment reads in inline asm, retired by COUNTED waits (round 4).  With plain loads hipcc waited lgkmcnt(0) in front of every
// second MFMA group -- i.e. for the reads it had just issued for the NEXT sub-step (the late-DMA branch between the two groups
// splits the basic block and its wait-count bookkeeping gives up at the join): a full LDS round trip exposed per 16 MFMAs.
__device__ __forceinline__ unsigned brig_lds32(const void* p) { return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)p; }
__device__ __forceinline__ void brig_read16(bf16x8& d, unsigned addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(addr)); }
template <int OFF> __device__ __forceinline__ void brig_read16_off(bf16x8& d, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
template <int N> __device__ __forceinline__ void brig_wait_lgkm() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);      // the register-only MFMAs stay behind the wait
}

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
    } else {                                // four rows of 8 (pitch 12): lane group 0 takes rows 0 and 2, lane group 1 rows 1 and 3
        const int g = (0x96 >> (i >> 2)) & 1;      // lanes 0-3,12-15,20-27 -> group 0; 4-11,16-19,28-31 -> group 1
        int o;
        if (g == 0) o = i < 4 ? i : (i < 16 ? i - 8 : i - 12);
        else o = i < 12 ? i - 4 : (i < 20 ? i - 8 : i - 16);
        const int k = 2 * (o >> 3) + g;
        const int row = pb * 4 + k;         // 0..31 over the 4 images of the tile
        img = row >> 3; ly = row & 7; lx = o & 7;
    }
}

// pieces of the input block issued in step t of a K chunk (ahead of the step's weight stage)
template <int NT> __host__ __device__ constexpr int brig_a_at(int t) { return NT == 4 ? (t < 2 ? 2 : 0) : (t < 1 ? 4 : 0); }
// s_waitcnt count of step t: everything issued after this step's weight stage (NWST - 1 steps ahead) may stay in
// flight; at t = 0 the chunk's input block (issued in the previous chunk) must have landed too
template <int NT, int NWST> __host__ __device__ constexpr int brig_vm(int t) {
    int nw = 0;
    for (int j = 1; j <= NWST - 2; ++j) nw += brig_a_at<NT>((t + 8 * NT - j) % NT) + 2;
    if (t != 0) return nw;
    int na = 2;                                      // the stage issued behind the block's last pieces ...
    const int last = NT == 4 ? 1 : 0;                // ... in step `last` of the previous chunk
    for (int s = last + 1; s < NT; ++s) na += brig_a_at<NT>(s) + 2;
    return nw < na ? nw : na;
}

// MODE 1: op P (lo -> hi, four phases merged).   MODE 0: op G (hi -> lo, K chunks walk the four parity planes).
// CBW: 32-channel blocks per wave (2: one tap per step; 1: two taps per step).
template <int MODE, int CBW, int NWST>
__global__ __launch_bounds__(512) void brig_kernel(BrigArgs a) {
    constexpr int CK = 32, RB = 64;         // channels per K chunk, bytes per LDS row (one pixel / one weight row of the chunk)
    constexpr int TPS = CBW == 2 ? 1 : 2;   // taps per step
    constexpr int NT = 4 / TPS;             // steps per K chunk
    constexpr int WST = 16384;              // weight ring: NWST stages of 16 KB
    constexpr int PA = 4;                   // input-block pieces a wave issues per K chunk (duplicates pad the count)
    constexpr int CW = 32 * CBW;            // output channels per wave
    constexpr int PROW = CW * 2 + 16;       // epilogue patch row: CW channels bf16 + pad
    constexpr int PATCH = 32 * PROW;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int quarter = wave & 3, half = wave >> 2;
    const int r = lane & 31, h = lane >> 5;
    char* const Abuf = smem;
    char* const Wring = smem + 2 * a.abytes;

    const unsigned lin = xcd_remap(blockIdx.x, gridDim.x);
    const int nt_i = lin % a.nnt, tile = lin / a.nnt;
    const int n0c = nt_i * (MODE == 1 ? CW : 4 * CW);     // first output channel of this workgroup
    int img0, y0;
    if (a.tiles_per_img > 1) { img0 = tile / a.tiles_per_img; y0 = (tile - img0 * a.tiles_per_img) * a.rpt; }
    else { img0 = tile * a.ipt; y0 = 0; }

    constexpr int esz = 2;
    const long long pixB = (long long)a.in_ld * esz;
    const long long rowB = (long long)a.in_row * pixB;

    // ---- input block DMA: per-lane source offsets (the chunk's channel / plane offset is added per issue) ---------------------
    unsigned abase[PA];
    int adst[PA];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        int p = wave + 8 * i;
        while (p >= a.npieces) p -= 8;
        int idx = (p << 4) + (lane >> 2);
        const int s = lane & 3;
        const int q = s ^ ((idx >> 2) & 3);
        if (idx >= a.npix) idx = 0;
        const int img = idx / a.BP, rem = idx - img * a.BP;
        const int by = rem / a.PITCH;
        int bi = rem - by * a.PITCH;
        if (bi >= a.LW + (MODE == 1 ? 2 : 1)) bi = 0;     // pad columns of the LDS pitch
        int n = img0 + img;
        if (n >= a.N) n = a.N - 1;
        long long y, x;
        if (MODE == 1) { y = y0 + by - 1; x = bi - 1; }
        else { y = 2 * (y0 + by); x = 2 * bi; }           // plane (0,0); plane (p,q) lies p rows / q columns before
        abase[i] = (unsigned)(((long long)n * a.in_img + y * a.in_row + x) * pixB + q * 16 - a.in_lo);
        adst[i] = p * 1024;
    }
    // ---- weight stage DMA: two pieces per wave; stage rows = [tap in step][quarter][32 * CBW channels] -------------------------
    unsigned wbase[2];
    int wq[2];              // op P: the phase of the row decides which tap slab it comes from
    const long long tapB = (long long)a.ncols * a.C * esz;       // bytes per weight tap slab
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int p = wave + 8 * j;
        const int row = (p << 4) + (lane >> 2);
        const int s = lane & 3;
        const int q = s ^ ((row >> 2) & 3);
        const int tsel = row / (4 * CW), qd = (row / CW) & 3, n = row & (CW - 1);
        const long long wrow = MODE == 1 ? (long long)n0c + n : (long long)n0c + qd * CW + n;
        wbase[j] = (unsigned)(wrow * a.C * esz + q * 16 + (long long)tsel * 2 * tapB);     // second tap of a step: kw + 2
        wq[j] = ((p << 4) / CW) & 3;       // = qd for every row of the piece (16 rows inside one CW block): wave-uniform
    }
    // tap slab of (phase | plane) pq for tap index tt = 2a + b: kh = 1 - p + 2a, kw = 1 - q + 2b
    auto widx = [](int pq, int tt) { return ((1 - (pq >> 1)) + 2 * (tt >> 1)) * 4 + (1 - (pq & 1)) + 2 * (tt & 1); };
    auto issue_w = [&](int step) {          // stage of global step `step` (clamped past the end: lands in a free slot, never read)
        if (P2P_ABL == 4 || P2P_ABL == 6) return;
        const int total = a.nkc * NT;
        if (step >= total) step = total - 1;
        const int kc = step / NT, t = step - kc * NT;
        const int tt0 = t * TPS;            // first tap of the step (TPS = 2: taps (a, 0) and (a, 1))
        char* dst = Wring + (step % NWST) * WST;
        if (MODE == 1) {
            const long long coff = (long long)kc * CK * esz;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const long long off = (long long)widx(wq[j], tt0) * tapB + coff;      // the piece's phase picks the slab (wave-uniform)
                p2p_glds16_sv(a.w + off, wbase[j], p2p_lds32(dst) + (wave + 8 * j) * 1024);
            }
        } else {
            const int plane = kc & 3, cc = kc >> 2;
            const char* src = a.w + (long long)widx(plane, tt0) * tapB + (long long)cc * CK * esz;      // wave-uniform
            p2p_glds16_sv(src, wbase[0], p2p_lds32(dst) + wave * 1024);
            p2p_glds16_sv(src, wbase[1], p2p_lds32(dst) + (wave + 8) * 1024);
        }
    };
    auto issue_a = [&](int kc, int i) {     // piece i of this wave for K chunk kc
        if (P2P_ABL == 5 || P2P_ABL == 7) return;
        if (kc >= a.nkc) kc = a.nkc - 1;
        long long off;
        if (MODE == 1) off = (long long)kc * CK * esz;
        else { const int plane = kc & 3, cc = kc >> 2; off = (long long)cc * CK * esz - (plane >> 1) * rowB - (plane & 1) * pixB; }
        p2p_glds16_sv(a.in + a.in_lo + off, abase[i], p2p_lds32(Abuf) + (kc & 1) * a.abytes + adst[i]);
    };

    // ---- fragment addresses ----------------------------------------------------------------------------------------------------
    int aidx[4];            // LDS pixel index of this lane's pixel in each of the wave's 4 column blocks (tap offset 0)
#pragma unroll
    for (int pbi = 0; pbi < 4; ++pbi) {
        int img, ly, lx;
        brig_lane_pixel(a, half * 4 + pbi, r, img, ly, lx);
        aidx[pbi] = img * a.BP + ly * a.PITCH + lx;
    }
    // weight fragment offsets inside a stage: row = (tap in step * 4 + quarter) * CW + cb * 32 + r, chunk q swizzled by the row
    int woff[TPS][2];      // channel block cb of the wave lies 32 rows = 2048 bytes further on (same swizzle): an immediate offset
#pragma unroll
    for (int tp = 0; tp < TPS; ++tp)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int row = (tp * 4 + quarter) * CW + r;
            const int q = 2 * ks + h;
            woff[tp][ks] = row * RB + ((q ^ ((row >> 2) & 3)) << 4);
        }

    f32x16 acc[4][CBW];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < CBW; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // ---- prologue: first input block, first NWST-1 weight stages ----------------------------------------------------------------
#pragma unroll
    for (int i = 0; P2P_ABL != 2 && i < PA; ++i) issue_a(0, i);
#pragma unroll
    for (int s = 0; P2P_ABL != 2 && s < NWST - 1; ++s) issue_w(s);

    // Software pipeline: a step has NS = 2 * TPS sub-steps (one 16-channel MFMA K step of one tap each).  Two fragment sets
    // ping-pong, the loads of sub-step i+1 are issued before the MFMAs of sub-step i, and the MFMAs of a step's LAST sub-step
    // are deferred to the start of the next step, behind its barrier and its first loads: the matrix pipe then has work while
    // both waves of a SIMD wait for the barrier, the DMA issue and the first LDS reads (MI355X_MICROARCH.md, two waves per SIMD, item 9).
    constexpr int NS = 2 * TPS;
    const bool stagger = a.stagger != 0;
    bf16x8 wfA[CBW], afA[4], wfB[CBW], afB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) afB[i][e] = (bf16_t)0.f;
#pragma unroll
    for (int i = 0; i < CBW; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) wfB[i][e] = (bf16_t)0.f;
    auto mfma_set = [&](const bf16x8 (&wf)[CBW], const bf16x8 (&af)[4]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int pbi = 0; pbi < 4; ++pbi)
#pragma unroll
            for (int cb = 0; cb < CBW; ++cb)
                acc[pbi][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[cb], af[pbi], acc[pbi][cb], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    for (int kc = 0; kc < a.nkc; ++kc) {
        const unsigned Acur32 = brig_lds32(Abuf) + (kc & 1) * a.abytes;
        // op P: the phase of this wave; op G: the parity plane of this chunk
        const int pq = MODE == 1 ? quarter : (kc & 3);
        brig_static_for(std::make_integer_sequence<int, NT>{}, [&](auto tt) {
            constexpr int t = decltype(tt)::value;
            if (P2P_ABL != 3 && (P2P_ABL < 4 || P2P_ABL >= 6)) brig_wait_vm<brig_vm<NT, NWST>(t)>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            // every wave has finished reading step s-1: its weight slot and (at t = 0) the other input buffer are free
            auto issue_dma = [&]() {
                if (P2P_ABL != 2) {
                    constexpr int na = brig_a_at<NT>(t);
                    constexpr int a0 = NT == 4 ? 2 * t : 0;
#pragma unroll
                    for (int i = 0; i < na; ++i) issue_a(kc + 1, a0 + i);
                    issue_w(kc * NT + t + NWST - 1);
                }
            };
            const unsigned Wcur32 = brig_lds32(Wring) + ((kc * NT + t) % NWST) * WST;
            auto load_set = [&](int sub, bf16x8 (&wf)[CBW], bf16x8 (&af)[4]) {
                const int tp = sub >> 1, ks = sub & 1;
                // tap (a, b) of the phase / plane reads block pixel (ly + da, lx + db): op P da = 1 + dy with dy = 0 / -1 (ph = 0),
                // +1 / 0 (ph = 1); op G da = a
                const int ta = (t * TPS + tp) >> 1, tb = (t * TPS + tp) & 1;
                int da, db;
                if (MODE == 1) { da = 1 + (((pq >> 1) + 1 - ((1 - (pq >> 1)) + 2 * ta)) >> 1); db = 1 + (((pq & 1) + 1 - ((1 - (pq & 1)) + 2 * tb)) >> 1); }
                else { da = ta; db = tb; }
                const int toff = da * a.PITCH + db;
                const int q = 2 * ks + h;
                __builtin_amdgcn_sched_barrier(0);
                {
                    const unsigned wa = Wcur32 + woff[tp][ks];
                    brig_read16(wf[0], wa);
                    if constexpr (CBW == 2) brig_read16_off<32 * RB>(wf[CBW - 1], wa);
                }
#pragma unroll
                for (int pbi = 0; pbi < 4; ++pbi) {
                    const int idx = aidx[pbi] + toff;
                    brig_read16(af[pbi], Acur32 + idx * RB + ((q ^ ((idx >> 2) & 3)) << 4));
                }
            };
            constexpr int NR = CBW + 4;      // reads per fragment set
            // The two waves of a SIMD (w and w + 4) would otherwise run the same program in lockstep behind the barrier: both
            // issue their LDS-DMA (tens of cycles of issue time per instruction) while the matrix pipe idles.  The second half
            // issues its DMA after its first MFMA group instead, so one partner multiplies while the other stages.
            const bool late = stagger && half != 0;       // wave-uniform
            if (!late || P2P_ABL == 1 || P2P_ABL == 4 || P2P_ABL == 5) issue_dma();
            if (P2P_ABL != 1 && P2P_ABL != 4 && P2P_ABL != 5) {
                load_set(0, wfA, afA);
                __builtin_amdgcn_sched_barrier(0);
                mfma_set(wfB, afB);              // last sub-step of the previous step (zeros before the first one): landed before the barrier
                __builtin_amdgcn_sched_barrier(0);
                if (late) issue_dma();
                load_set(1, wfB, afB);
                brig_wait_lgkm<NR>();            // set A has landed, set B stays in flight under its MFMAs
                mfma_set(wfA, afA);
                if (NS == 4) {
                    load_set(2, wfA, afA);
                    brig_wait_lgkm<NR>();
                    mfma_set(wfB, afB);
                    load_set(3, wfB, afB);
                    brig_wait_lgkm<NR>();
                    mfma_set(wfA, afA);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        });
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (P2P_ABL != 1 && P2P_ABL != 4 && P2P_ABL != 5) mfma_set(wfB, afB);

    // ---- epilogue ----------------------------------------------------------------------------------------------------------------
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();                        // every DMA (the clamped tail ones too) has landed, every wave is done with the operands
    char* const pL = smem + wave * PATCH;
    float* const stL = (float*)(smem + 8 * PATCH);           // [8 waves][2 image selectors][CW channels][2]
    const bool want_stats = a.stat_part != nullptr;
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};            // lane = channel (< CW): sums over the wave's pixels, per image selector
    typedef __attribute__((__vector_size__(4 * sizeof(bf16_t)))) bf16_t bf16x4;
    const int ph = quarter >> 1, pw = quarter & 1;           // op P: the quarter is a sub-pixel phase
#pragma unroll
    for (int pbi = 0; pbi < 4; ++pbi) {
        const int pb = half * 4 + pbi;
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 qv;
#pragma unroll
                for (int k = 0; k < 4; ++k) qv[k] = (bf16_t)acc[pbi][cb][4 * g + k];
                *(bf16x4*)(pL + r * PROW + (cb * 32 + 8 * g + 4 * h) * 2) = qv;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if ((want_stats || a.act_out != nullptr) && lane < CW) {
            const int sel = a.lgLW == 3 ? (pbi >> 1) : 0;
            const bf16_t* col = (const bf16_t*)pL + lane;
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const float v = (float)col[i * (PROW / 2)];
                t1 += v;
                t2 += v * v;
            }
            if (sel == 0) { s1[0] += t1; s2[0] += t2; } else { s1[1] += t1; s2[1] += t2; }
        }
        constexpr int LPP = CW / 8, PPS = 64 / LPP;          // lanes per pixel (16 bytes each), pixels per pass
        const int ch8 = lane & (LPP - 1);
#pragma unroll
        for (int ps = 0; ps < 32 / PPS; ++ps) {
            const int pix = ps * PPS + lane / LPP;
            int img, ly, lx;
            brig_lane_pixel(a, pb, pix, img, ly, lx);
            const int n = img0 + img;
            if (n < a.N) {
                const f32x4 v = *(const f32x4*)(pL + pix * PROW + ch8 * 16);
                long long opix;
                if (MODE == 1) opix = (long long)n * a.out_img + (long long)(2 * (y0 + ly) + ph) * a.out_row + (2 * lx + pw);
                else opix = (long long)n * a.out_img + (long long)(y0 + ly) * a.out_row + lx;
                const int c0 = n0c + (MODE == 1 ? 0 : quarter * CW) + ch8 * 8;
                *(f32x4*)((bf16_t*)a.out + opix * a.out_ld + c0) = v;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
    const bool fuse = a.act_out != nullptr;
    if (want_stats || fuse) {
        if (lane < CW) {
            stL[((wave * 2 + 0) * CW + lane) * 2 + 0] = s1[0];
            stL[((wave * 2 + 0) * CW + lane) * 2 + 1] = s2[0];
            stL[((wave * 2 + 1) * CW + lane) * 2 + 0] = s1[1];
            stL[((wave * 2 + 1) * CW + lane) * 2 + 1] = s2[1];
        }
        __syncthreads();
        // groups of (image, channel): 8-wide maps hold 4 images per tile (image = half * 2 + selector), wider maps one image /
        // strip per tile.  op P: the four phases (quarters) of a half cover the same channels; op G: each quarter has its own.
        const int nimg = a.lgLW == 3 ? 4 : 1;
        constexpr int NCH = MODE == 1 ? CW : 4 * CW;
        float* const scL = stL + 8 * 2 * CW * 2;             // [nimg][NCH][2]: scale, shift of the fused normalisation
        for (int e = tid; e < nimg * NCH; e += 512) {
            const int img = e / NCH, ch = e - img * NCH;
            const int c = ch & (CW - 1), qd = ch / CW;       // op G: the quarter that owns the channel
            float t1 = 0.f, t2 = 0.f;
            const int hf0 = a.lgLW == 3 ? (img >> 1) : 0, hf1 = a.lgLW == 3 ? (img >> 1) : 1, sel = a.lgLW == 3 ? (img & 1) : 0;
            for (int hf = hf0; hf <= hf1; ++hf) {
                if (MODE == 1) {
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) { const int wv = hf * 4 + q4; t1 += stL[((wv * 2 + sel) * CW + c) * 2]; t2 += stL[((wv * 2 + sel) * CW + c) * 2 + 1]; }
                } else {
                    const int wv = hf * 4 + qd;
                    t1 += stL[((wv * 2 + sel) * CW + c) * 2]; t2 += stL[((wv * 2 + sel) * CW + c) * 2 + 1];
                }
            }
            const int n = img0 + img;
            const float cnt = (float)((MODE == 1 ? 4 : 1) * a.rpt * a.LW);
            const float mean = t1 / cnt;
            const float m2 = fmaxf(t2 - t1 * mean, 0.f);
            if (n < a.N && want_stats) {
                const int slot = a.tiles_per_img > 1 ? tile - img0 * a.tiles_per_img : 0;
                float* dst = a.stat_part + (((long long)n * a.stat_slots + slot) * a.ncols + n0c + ch) * 2;
                dst[0] = mean;
                dst[1] = m2;
            }
            if (fuse) {
                const float rstd = rsqrtf(m2 / cnt + a.eps);
                const float ga = a.gamma[n0c + ch] * rstd;
                scL[(img * NCH + ch) * 2] = ga;
                scL[(img * NCH + ch) * 2 + 1] = a.beta[n0c + ch] - mean * ga;
                if (n < a.N) {
                    a.norm_stats[((long long)n * a.ncols + n0c + ch) * 2] = mean;
                    a.norm_stats[((long long)n * a.ncols + n0c + ch) * 2 + 1] = rstd;
                }
            }
        }
        if (fuse) {
            // second pass over the accumulators: the same rounded values through the patch, normalised and activated on the
            // way out, into the (haloed, channel-sliced) activation view -- the block is complete without a separate kernel
            __syncthreads();
            constexpr int LPP = CW / 8, PPS = 64 / LPP;
            const int ch8 = lane & (LPP - 1);
            const int cw0 = (MODE == 1 ? 0 : quarter * CW) + ch8 * 8;     // first of this lane's 8 channels inside the workgroup's range
#pragma unroll
            for (int pbi = 0; pbi < 4; ++pbi) {
                const int pb = half * 4 + pbi;
#pragma unroll
                for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        bf16x4 qv;
#pragma unroll
                        for (int k = 0; k < 4; ++k) qv[k] = (bf16_t)acc[pbi][cb][4 * g + k];
                        *(bf16x4*)(pL + r * PROW + (cb * 32 + 8 * g + 4 * h) * 2) = qv;
                    }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int ps = 0; ps < 32 / PPS; ++ps) {
                    const int pix = ps * PPS + lane / LPP;
                    int img, ly, lx;
                    brig_lane_pixel(a, pb, pix, img, ly, lx);
                    const int n = img0 + img;
                    if (n < a.N) {
                        const bf16x8 xv = *(const bf16x8*)(pL + pix * PROW + ch8 * 16);
                        const float* sc = scL + (img * NCH + cw0) * 2;
                        bf16x8 yv;
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            float y = (float)xv[k] * sc[2 * k] + sc[2 * k + 1];
                            if (a.act == P2P_ACT_LEAKY) y = y > 0.f ? y : a.alpha * y;
                            else if (a.act == P2P_ACT_RELU) y = y > 0.f ? y : 0.f;
                            yv[k] = (bf16_t)y;
                        }
                        long long opix;
                        if (MODE == 1) opix = (long long)n * a.act_img + (long long)(2 * (y0 + ly) + ph) * a.act_row + (2 * lx + pw);
                        else opix = (long long)n * a.act_img + (long long)(y0 + ly) * a.act_row + lx;
                        *(bf16x8*)((bf16_t*)a.act_out + opix * a.act_ld + n0c + cw0) = yv;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
}

struct BrigPlan { int ok, ring, cbw, ipt, rpt, tiles_per_img, ntiles, nnt, BR, PITCH, BP, npix, npieces, abytes, nkc, rot, slots; size_t shm; };

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
    if (C % 32 != 0 || ncols % 32 != 0) return p;      // p2p_igemm's own contract: channel counts are multiples of 32
    const int hw = LH * LW;
    p.ipt = hw >= 256 ? 1 : 256 / hw;
    p.rpt = hw >= 256 ? 256 / LW : LH;
    p.tiles_per_img = LH / p.rpt;
    p.ntiles = p.tiles_per_img > 1 ? N * p.tiles_per_img : (N + p.ipt - 1) / p.ipt;
    // output channels per workgroup: 64 per wave (op P: 64, op G: 256) unless that leaves a quarter of the CUs idle and the
    // 32-per-wave form does better.  P2P_BRIG_CBW = 1 / 2 forces a form (tests).
    const int bn2 = mode == 1 ? 64 : 256, bn1 = bn2 / 2;
    const char* force = getenv("P2P_BRIG_CBW");
    const int fc = force ? atoi(force) : 0;
    p.cbw = 0;
    if (ncols % bn2 == 0 && fc != 1) p.cbw = 2;
    if (ncols % bn1 == 0 && (p.cbw == 0 || (fc != 2 && (long long)p.ntiles * (ncols / bn2) < 192))) p.cbw = 1;
    if (!p.cbw) return p;
    p.nnt = ncols / (p.cbw == 2 ? bn2 : bn1);
    // A workgroup walks ALL of K alone (no split): with few pixel tiles the launch is a handful of long-running workgroups on an
    // idle chip (r04: 2 workgroups and 40 us for the 8x8 layers at batch 4, 64 workgroups and 46 us at batch 128, where the
    // pipelined im2col kernel with its K split takes 8-17 us).  Below P2P_BRIG_MIN_WG workgroups (default 160) the shape goes to
    // p2p_igemm's im2col path (and the fused conv + InstanceNorm form is not offered).  The tests set 1 to reach this kernel with
    // small batches.
    {
        const char* e = getenv("P2P_BRIG_MIN_WG");
        const long long min_wg = e ? atoi(e) : 160;
        // (r04 kept the fused block below batch 32: the step was issued from Python then and the extra normalisation launch cost
        // host time, c1 0.935 vs 1.026 ms.  With the step replayed by the library and the one-pass normalisation kernels of
        // norm_act.hip the im2col route wins at every small batch: c1 0.848 -> 0.775 ms, profiles/r05_exp_small_batch.txt)
        if ((long long)p.ntiles * p.nnt < min_wg) return p;
    }
    if (mode == 1) {
        p.BR = p.rpt + 2;
        p.PITCH = LW == 8 ? 12 : LW + 2;      // 8-wide maps: pitch 12 balances the pixel indices modulo 16 over the lane groups
    } else {
        p.BR = p.rpt + 1;
        p.PITCH = LW == 8 ? 12 : LW + 1;
    }
    p.rot = p.PITCH & 15;
    p.BP = p.BR * p.PITCH;
    p.npix = p.ipt * p.BP;
    p.npieces = (p.npix + 15) / 16;
    if (p.npieces > 32) return p;
    p.abytes = (p.npieces + 1) * 1024;         // one spare piece: tap offsets of clamped lanes stay inside the buffer
    p.nkc = (mode == 1 ? 1 : 4) * (C / 32);
    p.slots = p.tiles_per_img;
    p.ring = 4;       // weight ring of four 16 KB stages (six measured no better, r02: that form and its switch are gone)
    p.shm = 2 * (size_t)p.abytes + (size_t)p.ring * 16384;
    const size_t epi = 8 * 32 * 144 + 8 * 2 * 64 * 2 * sizeof(float) + 4 * 256 * 2 * sizeof(float);
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

struct BrigNorm { const float* gamma; const float* beta; float eps; int act; float alpha; const p2p_tensor* act_out; float* stats; };

int brig_launch(int op, int dtype, int N, int LH, int LW, int Cg, int Cd, const p2p_tensor* hi, const p2p_tensor* lo,
                const void* w, float* stat_part, void* stream, const BrigNorm* norm = nullptr) {
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
    a.BR = p.BR; a.PITCH = p.PITCH; a.BP = p.BP; a.npix = p.npix; a.npieces = p.npieces;
    a.abytes = p.abytes; a.nkc = p.nkc; a.rot = p.rot;
    a.gamma = a.beta = nullptr; a.eps = 0.f; a.act = 0; a.alpha = 0.f; a.act_out = nullptr; a.act_img = 0; a.act_row = a.act_ld = 0; a.norm_stats = nullptr;
    if (norm) {
        P2P_REQUIRE(p.tiles_per_img == 1, "p2p_igemm_norm_act: the fused block needs whole images per workgroup (query p2p_igemm_norm_act_ok)");
        P2P_REQUIRE(norm->gamma && norm->beta && norm->act_out && norm->act_out->ptr && norm->stats, "p2p_igemm_norm_act: null pointer");
        P2P_REQUIRE((norm->act_out->ld * 2) % 16 == 0 && ((uintptr_t)norm->act_out->ptr % 16) == 0, "p2p_igemm_norm_act: activation view must be 16-byte aligned");
        a.gamma = norm->gamma; a.beta = norm->beta; a.eps = norm->eps; a.act = norm->act; a.alpha = norm->alpha;
        a.act_out = (char*)norm->act_out->ptr; a.act_img = norm->act_out->img_stride; a.act_row = norm->act_out->row_stride; a.act_ld = norm->act_out->ld;
        a.norm_stats = norm->stats;
    }
    { static int sg = -1; if (sg < 0) { const char* e = getenv("P2P_BRIG_STAGGER"); sg = e ? atoi(e) : 1; } a.stagger = sg; }
    // per-lane gather offsets are 32-bit, counted from the lowest address a block touches (row -1, column -1 of image 0)
    a.in_lo = -((long long)in->row_stride + 1) * in->ld * 2;
    const long long span = ((long long)(N - 1) * in->img_stride + (long long)(op == P2P_OP_P ? LH + 1 : 2 * LH + 1) * in->row_stride +
                            (op == P2P_OP_P ? LW + 1 : 2 * LW + 1)) * in->ld * 2 - a.in_lo;
    P2P_REQUIRE(span < 0xffffffffLL && (long long)16 * a.ncols * a.C * 2 < 0xffffffffLL, "p2p_brig: view larger than 4 GB");
    const dim3 grid((unsigned)(p.ntiles * p.nnt));
    hipStream_t st = (hipStream_t)stream;
    const int key = (op == P2P_OP_P ? 2 : 0) + (p.cbw == 2 ? 1 : 0);
#define BRIG_GO(M, CB, R)                                                                                                          \
    do {                                                                                                                           \
        static bool attr = false;                                                                                                  \
        if (!attr) attr = p2p_allow_lds((const void*)brig_kernel<M, CB, R>, 160 * 1024, "brig_kernel");                          \
        brig_kernel<M, CB, R><<<grid, dim3(512), p.shm, st>>>(a);                                                                  \
    } while (0)
    switch (key) {
        case 0: BRIG_GO(0, 1, 4); break;
        case 1: BRIG_GO(0, 2, 4); break;
        case 2: BRIG_GO(1, 1, 4); break;
        default: BRIG_GO(1, 2, 4); break;
    }
#undef BRIG_GO
    return p2p_check_launch("p2p_igemm(block-resident)");
}

// Fused block (networks.py:7-21,24-36 without dropout): convolution + InstanceNorm + LeakyReLU / ReLU in one launch, on the
// shapes whose workgroups hold whole images.  1 if p2p_igemm_norm_act takes the shape.
extern "C" int p2p_igemm_norm_act_ok(int op, int dtype, int N, int LH, int LW, int Cg, int Cd) {
    const BrigPlan p = brig_plan(op, dtype, N, LH, LW, Cg, Cd);
    static int en = -1;
    if (en < 0) { const char* e = getenv("P2P_BRIG_FUSE_NORM"); en = e ? atoi(e) : 1; }
    return p.ok && p.tiles_per_img == 1 && en;
}

extern "C" int p2p_igemm_norm_act(int op, int dtype, int N, int LH, int LW, int Cg, int Cd, const p2p_tensor* hi,
                                  const p2p_tensor* lo, const void* w, const float* gamma, const float* beta, float eps,
                                  int act, float alpha, const p2p_tensor* act_out, float* stats, void* stream) {
    P2P_REQUIRE(op == P2P_OP_G || op == P2P_OP_P, "p2p_igemm_norm_act: op must be G or P");
    P2P_REQUIRE(hi && lo && hi->ptr && lo->ptr && w, "p2p_igemm_norm_act: null pointer");
    BrigNorm nm = {gamma, beta, eps, act, alpha, act_out, stats};
    return brig_launch(op, dtype, N, LH, LW, Cg, Cd, hi, lo, w, nullptr, stream, &nm);
}
