// Microbenchmarks that size the implicit-GEMM design: not part of the product library.
//   feed:  L2 -> LDS rate per CU of global_load_lds_dwordx4 by access shape (full 128-B lines ... 16-B pieces)
//   mfma:  MFMA rate with operand fragments read from LDS, by wave tile shape (LDS read bandwidth ceiling)
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench/feed.hip -o tools/ubench/feed ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

__device__ __forceinline__ void glds16(const char* g, char* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// ROWB = bytes per row a wave-instruction takes (128 = whole lines, 8 rows per instruction ... 16 = 64 rows);
// rows are `stride` bytes apart.  Every wave keeps DEPTH instructions in flight.
template <int ROWB, int DEPTH>
__global__ __launch_bounds__(512) void feed_kernel(const char* src, unsigned region_mask, int stride, int iters, int rot) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int LPR = ROWB / 16, RPI = 64 / LPR;       // lanes per row, rows per instruction
    const int row = lane / LPR, ch = lane % LPR;
    char* ring = smem + wave * (DEPTH * 1024);
    // workgroups of one XCD (blockIdx & 7 equal) walk the same region -> L2 hits after the first touch
    // rot: workgroups start at different 128-byte pieces of the stride-long row (K rotation of a GEMM's row reads)
    unsigned off = ((blockIdx.x >> 3) * 8 + wave) * RPI * stride + (rot ? ((blockIdx.x >> 3) * 128) % stride : 0);
    for (int it = 0; it < iters; ++it) {
        const unsigned o = (off + row * stride) & region_mask;
        glds16(src + o + ch * 16, ring + (it % DEPTH) * 1024);
        off += 64 * RPI * stride;     // 8 waves x 8... spread
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int ROWB, int DEPTH>
static void run_feed(const char* src, size_t region, int stride, int nblk, int wg_per_cu, int rot = 0) {
    const int iters = 4096;
    size_t shm = wg_per_cu == 1 ? 96 * 1024 : 8 * DEPTH * 1024;
    if (shm < 8 * DEPTH * 1024) shm = 8 * DEPTH * 1024;
    CK(hipFuncSetAttribute((const void*)feed_kernel<ROWB, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    feed_kernel<ROWB, DEPTH><<<nblk, 512, shm>>>(src, (unsigned)(region - 1), stride, 64, rot);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    feed_kernel<ROWB, DEPTH><<<nblk, 512, shm>>>(src, (unsigned)(region - 1), stride, iters, rot);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    double bytes = (double)nblk * 8 * iters * 1024;
    if (rot) printf("[rot] ");
    printf("feed rowB=%3d depth=%2d stride=%4d region=%5zuKB blocks=%4d (%d/CU): %8.1f GB/s total, %6.1f GB/s per CU, %5.1f B/clk/CU @2.4GHz\n",
           ROWB, DEPTH, stride, region >> 10, nblk, wg_per_cu, bytes / ms * 1e-6, bytes / ms * 1e-6 / 256.0, bytes / ms * 1e-6 / 256.0 / 2.4);
}

// MFMA with fragments from LDS: wave tile TM x TN (32x32 tiles), 8 waves; reads are conflict-free (XOR swizzle)
template <int TM, int TN>
__global__ __launch_bounds__(512) void mfma_kernel(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 16384; i += 512) ((float*)smem)[i] = (float)((i * 2654435761u) >> 20) * 1e-3f;     // 64 KB of operands
    __syncthreads();
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int r = lane & 31, h = lane >> 5;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bf16x8 af[TM], bf[TN];
            const int q = 2 * s + h;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = ((wave & 3) * TM + i) * 32 + r;
                af[i] = *(const bf16x8*)(smem + (row & 255) * 128 + ((q ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = ((wave >> 2) * TN + j) * 32 + r;
                bf[j] = *(const bf16x8*)(smem + 32768 + (row & 255) * 128 + ((q ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    out[blockIdx.x * 512 + tid] = s;
}

template <int TM, int TN>
static void run_mfma(float* out, int wg_per_cu) {
    const int iters = 2048, nblk = 256 * wg_per_cu;
    const size_t shm = wg_per_cu == 1 ? 96 * 1024 : 65536;
    CK(hipFuncSetAttribute((const void*)mfma_kernel<TM, TN>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    mfma_kernel<TM, TN><<<nblk, 512, shm>>>(out, 16);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    mfma_kernel<TM, TN><<<nblk, 512, shm>>>(out, iters);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    double fl = (double)nblk * 8 * iters * 4 * TM * TN * 32768.0;
    printf("mfma wave tile %dx%d (x32), %d WG/CU (%d waves/SIMD): %7.1f TFLOP/s, LDS reads %.2f per MFMA\n", TM, TN, wg_per_cu,
           2 * wg_per_cu, fl / ms * 1e-9, (double)(TM + TN) / (TM * TN));
}

int main(int argc, char** argv) {
    const size_t big = 256u << 20;
    char* src;
    CK(hipMalloc(&src, big));
    CK(hipMemset(src, 1, big));
    float* out;
    CK(hipMalloc(&out, 512 * 512 * sizeof(float)));
    if (argc > 1 && argv[1][0] == 's') {
        printf("== row-stride sweep (channel camping?): 128-byte rows, region 8 MB ==\n");
        run_feed<128, 8>(src, 8u << 20, 128, 256, 1);
        run_feed<128, 8>(src, 8u << 20, 256, 256, 1);
        run_feed<128, 8>(src, 8u << 20, 512, 256, 1);
        run_feed<128, 8>(src, 8u << 20, 1024, 256, 1);
        run_feed<128, 8>(src, 8u << 20, 2048, 256, 1);
        run_feed<128, 8>(src, 8u << 20, 4096, 256, 1);
        run_feed<128, 8>(src, 8u << 20, 8192, 256, 1);
        run_feed<128, 8>(src, 8u << 20, 1024, 256, 1, 1);
        run_feed<128, 8>(src, 8u << 20, 2048, 256, 1, 1);
        run_feed<128, 8>(src, 8u << 20, 4096, 256, 1, 1);
        run_feed<128, 8>(src, 8u << 20, 8192, 256, 1, 1);
        run_feed<128, 8>(src, 8u << 20, 2048 + 128, 256, 1);
        run_feed<128, 8>(src, 8u << 20, 2048 + 256, 256, 1);
        run_feed<128, 8>(src, 64u << 20, 2048, 256, 1);
        run_feed<128, 8>(src, 64u << 20, 2048, 256, 1, 1);
        run_feed<128, 8>(src, 64u << 20, 2048 + 128, 256, 1);
        return 0;
    }

    printf("== LDS-DMA feed, one workgroup (8 waves) per CU, region L2-resident (2 MB per XCD walk) ==\n");
    run_feed<128, 8>(src, 2u << 20, 128, 256, 1);
    run_feed<128, 16>(src, 2u << 20, 128, 256, 1);
    run_feed<128, 8>(src, 2u << 20, 512, 256, 1);
    run_feed<64, 8>(src, 2u << 20, 512, 256, 1);
    run_feed<32, 8>(src, 2u << 20, 512, 256, 1);
    run_feed<16, 8>(src, 2u << 20, 512, 256, 1);
    run_feed<16, 16>(src, 2u << 20, 512, 256, 1);
    printf("== same, two workgroups per CU ==\n");
    run_feed<128, 8>(src, 2u << 20, 128, 512, 2);
    run_feed<64, 8>(src, 2u << 20, 512, 512, 2);
    run_feed<16, 8>(src, 2u << 20, 512, 512, 2);
    printf("== region 32 MB (Infinity Cache) and 256 MB (HBM) ==\n");
    run_feed<128, 8>(src, 32u << 20, 128, 256, 1);
    run_feed<128, 16>(src, 32u << 20, 128, 256, 1);
    run_feed<128, 8>(src, big, 128, 256, 1);
    run_feed<128, 16>(src, big, 128, 256, 1);
    run_feed<16, 16>(src, 32u << 20, 512, 256, 1);
    printf("== MFMA fed from LDS ==\n");
    run_mfma<1, 2>(out, 1);
    run_mfma<1, 2>(out, 2);
    run_mfma<2, 2>(out, 1);
    run_mfma<2, 2>(out, 2);
    run_mfma<2, 4>(out, 1);
    run_mfma<4, 2>(out, 1);
    return 0;
}
