// Shader clock under load: s_memtime (shader clock) against the constant-rate wall clock, for a matrix-pipe loop, a vector loop and
// both mixed.  Also the issue overlap of MFMA and VALU from DIFFERENT waves of one SIMD (3 waves per SIMD as in the histogram kernels).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/clock_probe tools/ubench/clock_probe.hip && tools/ubench/clock_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

// mode bit 0: waves with (wave % 3 == 0) or all run MFMAs; bit 1: VALU
// kind 0: every wave MFMA only; 1: every wave VALU only; 2: every wave alternates 1 MFMA + 8 VALU; 3: wave slot 0 of each SIMD MFMA only,
// slots 1, 2 VALU only
// vector instruction flavours for the overlap question: 0 v_pk_fma_f32, 1 v_fma_f32, 2 v_add_u32 / v_xor_b32, 3 v_rcp_f32
template <int FL>
__device__ __forceinline__ void valu8(float& v0, float& v1, float& v2, float& v3, float& v4, float& v5, float& v6, float& v7) {
    if (FL == 1) {
        asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
                     "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"
                     : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(1.0001f), "v"(0.5f));
    } else if (FL == 2) {
        asm volatile("v_add_u32 %0, %0, %8\n\tv_xor_b32 %1, %1, %8\n\tv_add_u32 %2, %2, %8\n\tv_xor_b32 %3, %3, %8\n\t"
                     "v_add_u32 %4, %4, %8\n\tv_xor_b32 %5, %5, %8\n\tv_add_u32 %6, %6, %8\n\tv_xor_b32 %7, %7, %8"
                     : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(12345));
    } else {
        asm volatile("v_rcp_f32 %0, %0\n\tv_rcp_f32 %1, %1\n\tv_rcp_f32 %2, %2\n\tv_rcp_f32 %3, %3\n\t"
                     "v_rcp_f32 %4, %4\n\tv_rcp_f32 %5, %5\n\tv_rcp_f32 %6, %6\n\tv_rcp_f32 %7, %7"
                     : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
    }
}

// kind 1: every wave 64 vector instructions per iteration; kind 3: wave slot 0 of each SIMD 8 MFMAs, slots 1, 2 the vector instructions
template <int FL>
__global__ __launch_bounds__(768) void probe_fl(int kind, int iters, float* sink) {
    const int slot = threadIdx.x >> 8;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(threadIdx.x * 0.001f + e); b[e] = (__bf16)(1.0f); }
    f32x16 acc0 = {0}, acc1 = {0};
    float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f, v4 = 4.f, v5 = 5.f, v6 = 6.f, v7 = 7.f;
    const bool do_m = kind == 3 && slot == 0;
    for (int i = 0; i < iters; ++i) {
        if (do_m) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) valu8<FL>(v0, v1, v2, v3, v4, v5, v6, v7);
        }
    }
    float s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    for (int e = 0; e < 16; ++e) s += acc0[e] + acc1[e];
    if (s == 12345.678f) sink[0] = s;
}

template <int FL>
static void run_fl(const char* name, float* sink, float mfma_ms) {
    float t[2];
    for (int ki = 0; ki < 2; ++ki) {
        const int kind = ki == 0 ? 1 : 3;
        for (int rep = 0; rep < 2; ++rep) {
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0);
            probe_fl<FL><<<256, 768>>>(kind, 20000, sink);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&t[ki], e0, e1);
        }
    }
    printf("%-14s three waves per SIMD of it: %7.3f ms;  one MFMA wave + two waves of it: %7.3f ms   (no overlap: %.3f, full overlap: %.3f)\n", name, t[0], t[1],
           mfma_ms / 3 + t[0] * 2 / 3, (mfma_ms / 3 > t[0] * 2 / 3 ? mfma_ms / 3 : t[0] * 2 / 3));
}

__global__ __launch_bounds__(768) void probe(int kind, int iters, long long* out, float* sink) {
    const int wave = threadIdx.x >> 6, slot = wave >> 2;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(threadIdx.x * 0.001f + e); b[e] = (__bf16)(1.0f); }
    f32x16 acc0 = {0}, acc1 = {0};
    float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f, v4 = 4.f, v5 = 5.f, v6 = 6.f, v7 = 7.f;
    const bool do_m = kind == 0 || kind == 2 || (kind == 3 && slot == 0);
    const bool do_v = kind == 1 || kind == 2 || (kind == 3 && slot != 0);
    long long t0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        if (do_m && do_v) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
                v0 = fmaf(v0, 1.0001f, 0.5f); v1 = fmaf(v1, 1.0001f, 0.5f); v2 = fmaf(v2, 1.0001f, 0.5f); v3 = fmaf(v3, 1.0001f, 0.5f);
                v4 = fmaf(v4, 1.0001f, 0.5f); v5 = fmaf(v5, 1.0001f, 0.5f); v6 = fmaf(v6, 1.0001f, 0.5f); v7 = fmaf(v7, 1.0001f, 0.5f);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
                v0 = fmaf(v0, 1.0001f, 0.5f); v1 = fmaf(v1, 1.0001f, 0.5f); v2 = fmaf(v2, 1.0001f, 0.5f); v3 = fmaf(v3, 1.0001f, 0.5f);
                v4 = fmaf(v4, 1.0001f, 0.5f); v5 = fmaf(v5, 1.0001f, 0.5f); v6 = fmaf(v6, 1.0001f, 0.5f); v7 = fmaf(v7, 1.0001f, 0.5f);
            }
        } else if (do_m) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                v0 = fmaf(v0, 1.0001f, 0.5f); v1 = fmaf(v1, 1.0001f, 0.5f); v2 = fmaf(v2, 1.0001f, 0.5f); v3 = fmaf(v3, 1.0001f, 0.5f);
                v4 = fmaf(v4, 1.0001f, 0.5f); v5 = fmaf(v5, 1.0001f, 0.5f); v6 = fmaf(v6, 1.0001f, 0.5f); v7 = fmaf(v7, 1.0001f, 0.5f);
            }
        }
    }
    long long t1 = clock64(), w1 = wall_clock64();
    float s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    for (int e = 0; e < 16; ++e) s += acc0[e] + acc1[e];
    if (s == 12345.678f) sink[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; }
}

int main() {
    long long* out; float* sink;
    hipMalloc(&out, 16); hipMalloc(&sink, 4);
    int wall_khz = 0;
    hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    const char* names[4] = {"MFMA only (8 per iteration per wave)", "VALU only (64 per iteration per wave)", "1 MFMA + 8 VALU alternating, every wave (8 + 64 per iteration)",
                            "one wave per SIMD MFMA only, two waves VALU only"};
    for (int kind = 0; kind < 4; ++kind) {
        const int iters = 20000;
        for (int rep = 0; rep < 2; ++rep) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            probe<<<256, 768>>>(kind, iters, out, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long h[2]; hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
            if (rep == 1)
                printf("%-70s %8.3f ms   s_memtime ticks %lld, wall ticks %lld (%d kHz) -> s_memtime %.0f MHz;  cycles/iteration at 2.4 GHz %.0f\n", names[kind], ms,
                       h[0], h[1], wall_khz, (double)h[0] / ((double)h[1] / wall_khz) / 1000.0, ms * 1e-3 * 2.4e9 / iters);
        }
    }
    float mfma_ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        probe<<<256, 768>>>(0, 20000, out, sink);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&mfma_ms, e0, e1);
    }
    printf("\nMFMA in one wave of a SIMD against vector instructions in the other two (8 MFMAs | 64 vector instructions per iteration); MFMA alone, 3 waves: %.3f ms\n", mfma_ms);
    run_fl<1>("v_fma_f32", sink, mfma_ms);
    run_fl<2>("v_add/xor_b32", sink, mfma_ms);
    run_fl<3>("v_rcp_f32", sink, mfma_ms);
    return 0;
}
