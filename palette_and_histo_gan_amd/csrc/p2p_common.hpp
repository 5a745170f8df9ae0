// Shared device/host helpers for libp2pgan_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/p2pgan.h"

typedef __bf16 bf16_t;
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((__vector_size__(4 * sizeof(short)))) short s16x4;

#define P2P_WAVE 64

void p2p_set_error(const char* fmt, ...);
int p2p_check_launch(const char* what);
// Raises a kernel's dynamic-LDS limit above the default 64 KB (hipFuncAttributeMaxDynamicSharedMemorySize).  Returns true on
// success; a refusal is remembered and reported, with the kernel's name and the byte count, by the p2p_check_launch that follows
// the launch -- which would otherwise fail with an unexplained "invalid argument".  Callers keep their "done" flag false on
// failure, so the next call tries (and reports) again.
bool p2p_allow_lds(const void* kernel, int bytes, const char* name);

#define P2P_REQUIRE(cond, ...)                 \
    do {                                       \
        if (!(cond)) {                         \
            p2p_set_error(__VA_ARGS__);        \
            return -1;                         \
        }                                      \
    } while (0)

// Device-side copy of p2p_tensor (plain ints so it can be passed by value as a kernel argument).
struct TView {
    char* ptr;
    long long img;
    int row;
    int ld;
    __host__ __device__ long long off(int n, int y, int x) const {
        return ((long long)n * img + (long long)y * row + x) * ld;
    }
};

static inline TView make_view(const p2p_tensor* t) {
    TView v;
    v.ptr = (char*)t->ptr;
    v.img = t->img_stride;
    v.row = t->row_stride;
    v.ld = t->ld;
    return v;
}

struct GSrc {
    const char* ptr;
    int kind;     // 0 none, 1 activation dtype, 2 f32 (nslabs slabs)
    int nslabs;
    long long slab;
    int ld;
    int coff;
};

static inline GSrc make_gsrc(const p2p_gsrc* g) {
    GSrc s;
    if (!g) {
        s.ptr = nullptr; s.kind = 0; s.nslabs = 0; s.slab = 0; s.ld = 0; s.coff = 0;
        return s;
    }
    s.ptr = (const char*)g->ptr; s.kind = g->ptr ? g->kind : 0; s.nslabs = g->nslabs;
    s.slab = g->slab_stride; s.ld = g->ld; s.coff = g->coff;
    return s;
}

template <typename T> __device__ __forceinline__ float to_f32(T x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x) { return (T)x; }

// gradient source value at dense pixel index `pix`, channel c
template <typename T>
__device__ __forceinline__ float gsrc_load(const GSrc& g, long long pix, int c) {
    if (g.kind == 0) return 0.f;
    long long e = pix * g.ld + g.coff + c;
    if (g.kind == 1) return to_f32(((const T*)g.ptr)[e]);
    float s = 0.f;
    const float* p = (const float*)g.ptr + e;
    for (int k = 0; k < g.nslabs; ++k) s += p[(long long)k * g.slab];
    return s;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// block-wide sum for blockDim.x <= 1024, result valid in every thread; `red` is >= 16 floats of LDS
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float s = 0.f;
    for (int i = 0; i < nw; ++i) s += red[i];
    return s;
}

// Pixel index -> (n, y, x) with 32-bit arithmetic (callers guarantee N*H*W < 2^31): shifts for power-of-two maps
// (every map of the reference models), one 32-bit division pair otherwise.  64-bit '%' and '/' cost ~100 instructions each.
struct PixDec {
    int W, H, lgW, lgH;   // lg* = -1 when not a power of two
    __host__ __device__ static int lg2(int v) { int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1; }
    __host__ static PixDec make(int H_, int W_) { PixDec d; d.W = W_; d.H = H_; d.lgW = lg2(W_); d.lgH = lg2(H_); return d; }
    __device__ __forceinline__ void operator()(unsigned p, int& n, int& y, int& x) const {
        if (lgW >= 0 && lgH >= 0) { x = p & (W - 1); y = (p >> lgW) & (H - 1); n = p >> (lgW + lgH); }
        else { unsigned q = p / (unsigned)W; x = p - q * W; n = q / (unsigned)H; y = q - n * H; }
    }
};

// XCD-aware block index (cdna_hip_programming.md T1): workgroups are dealt round-robin over the 8 XCDs, each with its
// own L2, so the blocks b, b+8, b+16, ... share an L2.  This bijective remap hands every XCD one CONTIGUOUS range of the
// logical block order, so blocks that the caller orders next to each other (same input pixels) hit the same L2.
// Speed only: any placement gives the same results.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk) {
    const unsigned q = nblk >> 3, r = nblk & 7u, xcd = bid & 7u;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// LDS-DMA (global_load_lds_dwordx4) with a WAVE-UNIFORM 64-bit base in scalar registers and a 32-bit per-lane byte offset: the
// saddr form of the instruction.  The builtin only takes a flat per-lane pointer, which costs a 64-bit vector add per piece
// and keeps the per-lane offsets as 64-bit register pairs (r04: that pushed the block-resident kernel over 256 VGPRs).
// `lds_dst` = wave-uniform LDS byte address of the piece (lane l lands at lds_dst + 16 l).  M0 is written and restored inside
// the statement (cdna_hip_programming.md 5.7; the s_nop also covers a v_readfirstlane that wrote the base just before the
// statement); the load is invisible to hipcc's wait-count bookkeeping: callers count vmcnt.
__device__ __forceinline__ void p2p_glds16_sv(const char* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 2\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
// A zero that reaches an accumulator through VECTOR moves.  hipcc starts an accumulation with the first MFMA's "C = 0" form, whose
// write-back can land in registers that a global store issued just before still has to read as its data: the matrix pipe's
// write-back is not interlocked against that read (measured in round 4, tools/exp/hist_repro.py and tools/exp/store_mfma_hazard.py:
// one workgroup in a thousand stored a wrong dword), a vector instruction's write is.  Kernels that store a tile and start the next
// tile's MFMAs in the same wave zero their accumulators from this value.
__device__ __forceinline__ float p2p_valu_zero() {
    float z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    return z;
}

__device__ __forceinline__ unsigned p2p_lds32(const void* p) {
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}

// A fork of a side stream costs the forking stream a marker packet of its own (~4.8 us between two kernels in the step's trace, 17 per
// step).  An entry point whose LAST launch goes through P2P_LAUNCH_LAST hands an armed event (p2p_arm_stop_event, streams.hip) to that
// kernel's own dispatch packet as its completion signal (hipExtLaunchKernelGGL stop event): the side stream waits for the event, the
// forking stream carries no extra packet.  Not armed: a plain launch.
hipEvent_t p2p_take_stop_event();
#define P2P_LAUNCH_LAST(kernel, grid, block, shm, st, ...)                                                  \
    do {                                                                                                    \
        hipEvent_t stop_ev_ = p2p_take_stop_event();                                                        \
        if (stop_ev_) hipExtLaunchKernelGGL(kernel, grid, block, shm, st, nullptr, stop_ev_, 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(kernel, grid, block, shm, st, __VA_ARGS__);                                 \
    } while (0)

#define P2P_DISPATCH_DTYPE(dtype, CALL)                         \
    do {                                                        \
        if ((dtype) == P2P_F32) { typedef float T; CALL; }      \
        else if ((dtype) == P2P_BF16) { typedef bf16_t T; CALL; } \
        else { p2p_set_error("bad dtype %d", (int)(dtype)); return -1; } \
    } while (0)
