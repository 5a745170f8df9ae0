// Input pipeline of the RGBA and indexed models (dataset_utils.py:11-20,39-49,66-120,209-246), device-resident.
// The whole sprite set (294 sprites x 4 directions x 64x64x4 bytes = 19 MB) is decoded ONCE on the host (PNG: zlib inflate in
// Python, scanline un-filtering below) and kept in HBM as uint8; a train batch is then one launch that gathers the shuffled
// sprite pair, blackens fully transparent pixels, applies the pair's augmentation (same hue rotation and the same nearest,
// zero-filled translation for source and target), normalises to [-1, 1] and writes the two f32 NHWC batches train_step takes.
// HBM-bound byte work: 2 x 4 B read and 2 x 16 B written per pixel.
#include "p2p_common.hpp"

// ---- host: PNG scanline un-filtering (PNG 1.2 section 6: None, Sub, Up, Average, Paeth), 8-bit samples --------------------
extern "C" int p2p_png_unfilter(const unsigned char* filtered, int height, int row_bytes, int bpp, unsigned char* out) {
    P2P_REQUIRE(filtered && out && height > 0 && row_bytes > 0 && bpp > 0 && bpp <= 8, "p2p_png_unfilter: bad args");
    for (int y = 0; y < height; ++y) {
        const unsigned char* src = filtered + (size_t)y * (row_bytes + 1);
        unsigned char* cur = out + (size_t)y * row_bytes;
        const unsigned char* up = y ? cur - row_bytes : nullptr;
        const int f = src[0];
        P2P_REQUIRE(f >= 0 && f <= 4, "p2p_png_unfilter: filter type %d in row %d", f, y);
        for (int i = 0; i < row_bytes; ++i) {
            const int a = i >= bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0;
            int pred = 0;
            if (f == 1) pred = a;
            else if (f == 2) pred = b;
            else if (f == 3) pred = (a + b) >> 1;
            else if (f == 4) {
                const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
                pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
            }
            cur[i] = (unsigned char)(src[1 + i] + pred);
        }
    }
    return 0;
}

// ---- device ------------------------------------------------------------------------------------------------------------------
// tf.image.adjust_hue on float RGB of any scale (dataset_utils.py:80-84 -> stateless_random_hue): rotate the hue angle by
// delta (in turns), keep value and chroma.  Piecewise-linear HSV form; a grey pixel (chroma 0) is unchanged.
__device__ __forceinline__ void hue_rotate(float& r, float& g, float& b, float delta) {
    const float vmax = fmaxf(r, fmaxf(g, b)), vmin = fminf(r, fminf(g, b)), c = vmax - vmin;
    if (c <= 0.f) return;
    float h6;
    if (vmax == r) h6 = (g - b) / c;
    else if (vmax == g) h6 = 2.f + (b - r) / c;
    else h6 = 4.f + (r - g) / c;
    h6 += 6.f * delta;
    h6 -= 6.f * floorf(h6 * (1.f / 6.f));
    if (h6 >= 6.f) h6 -= 6.f;
    if (h6 < 0.f) h6 = 0.f;
    const int sector = min(5, (int)h6);
    const float f = h6 - (float)(sector & ~1);          // position inside the pair of sectors, [0, 2)
    const float x = c * (1.f - fabsf(f - 1.f));
    float rr, gg, bb;
    switch (sector) {
        case 0: rr = c; gg = x; bb = 0.f; break;
        case 1: rr = x; gg = c; bb = 0.f; break;
        case 2: rr = 0.f; gg = c; bb = x; break;
        case 3: rr = 0.f; gg = x; bb = c; break;
        case 4: rr = x; gg = 0.f; bb = c; break;
        default: rr = c; gg = 0.f; bb = x; break;
    }
    r = rr + vmin; g = gg + vmin; b = bb + vmin;
}

// aug[b] = (apply, hue delta in turns, dy, dx in pixels); out pixel (y, x) samples the input at (round(y - dy), round(x - dx))
// (Keras RandomTranslation -> ImageProjectiveTransformV3, nearest, fill_mode constant 0: dataset_utils.py:87-92).
__global__ __launch_bounds__(256) void sprites_rgba_kernel(const uchar4* __restrict__ sprites, long long sprite_pixels,
                                                           const int* __restrict__ src_idx, const int* __restrict__ tgt_idx,
                                                           const float* __restrict__ aug, int B, int S, int lgS, int normalise,
                                                           f32x4* __restrict__ source, f32x4* __restrict__ target, int n_sprites) {
    const long long npix = (long long)B * S * S;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(p & (S - 1)), y = (int)((p >> lgS) & (S - 1)), b = (int)(p >> (2 * lgS));
        int sy = y, sx = x;
        bool on = false;
        float delta = 0.f;
        if (aug) {
            const f32x4 a = *(const f32x4*)(aug + 4 * b);
            on = a[0] != 0.f;
            if (on) {
                delta = a[1];
                sy = (int)roundf((float)y - a[2]);
                sx = (int)roundf((float)x - a[3]);
            }
        }
        const bool inside = sy >= 0 && sy < S && sx >= 0 && sx < S;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int sprite = (k ? tgt_idx : src_idx)[b];
            uchar4 u = make_uchar4(0, 0, 0, 0);
            if (inside && sprite >= 0 && sprite < n_sprites) u = sprites[(long long)sprite * sprite_pixels + sy * S + sx];
            float r = u.x, g = u.y, bl = u.z, al = u.w;
            if (u.w == 0) { r = 0.f; g = 0.f; bl = 0.f; }          // blacken_transparent_pixels (dataset_utils.py:11-20)
            if (on) hue_rotate(r, g, bl, delta);
            f32x4 o = {r, g, bl, al};
            if (normalise) { o[0] = o[0] / 127.5f - 1.f; o[1] = o[1] / 127.5f - 1.f; o[2] = o[2] / 127.5f - 1.f; o[3] = o[3] / 127.5f - 1.f; }
            (k ? target : source)[p] = o;
        }
    }
}

extern "C" int p2p_sprites_rgba_batch(const void* sprites, int n_sprites, int S, const int* src_idx, const int* tgt_idx,
                                      const float* aug, int B, int normalise, float* source, float* target, void* stream) {
    P2P_REQUIRE(sprites && src_idx && tgt_idx && source && target && n_sprites > 0 && B > 0, "p2p_sprites_rgba_batch: bad args");
    P2P_REQUIRE(S >= 4 && (S & (S - 1)) == 0, "p2p_sprites_rgba_batch: S must be a power of two");
    P2P_REQUIRE(((uintptr_t)source % 16) == 0 && ((uintptr_t)target % 16) == 0 && ((uintptr_t)sprites % 4) == 0 &&
                (!aug || ((uintptr_t)aug % 16) == 0), "p2p_sprites_rgba_batch: alignment");
    int lgS = 0;
    while ((1 << lgS) < S) ++lgS;
    long long blocks = ((long long)B * S * S + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    sprites_rgba_kernel<<<dim3((unsigned)blocks), 256, 0, (hipStream_t)stream>>>(
        (const uchar4*)sprites, (long long)S * S, src_idx, tgt_idx, aug, B, S, lgS, normalise, (f32x4*)source, (f32x4*)target,
        n_sprites);
    return p2p_check_launch("p2p_sprites_rgba_batch");
}

// Indexed batches (dataset_utils.py:123-164,232-246): the union palette of a sprite pair and its two index maps do not depend
// on the step (no augmentation on this path), so they are extracted once at load time and a batch is a gather of rows:
// table k has rows of row_ints[k] int32 values; out[k][b] = table[k][sel[b]].
__global__ __launch_bounds__(256) void gather_rows_kernel(const int* __restrict__ table, int row_ints, const int* __restrict__ sel,
                                                          int B, int* __restrict__ out, int n_rows) {
    const int b = blockIdx.y;
    const int row = min(max(sel[b], 0), n_rows - 1);           // a selector outside the table never leaves it (the host also checks)
    const int4* src = (const int4*)(table + (long long)row * row_ints);
    int4* dst = (int4*)(out + (long long)b * row_ints);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < row_ints / 4; i += gridDim.x * blockDim.x) dst[i] = src[i];
}

extern "C" int p2p_gather_rows_i32(const int* table, int n_rows, int row_ints, const int* sel, int B, int* out, void* stream) {
    P2P_REQUIRE(table && sel && out && n_rows > 0 && B > 0 && row_ints > 0 && row_ints % 4 == 0, "p2p_gather_rows_i32: bad args");
    P2P_REQUIRE(((uintptr_t)table % 16) == 0 && ((uintptr_t)out % 16) == 0, "p2p_gather_rows_i32: alignment");
    int bx = (row_ints / 4 + 255) / 256;
    if (bx > 16) bx = 16;
    gather_rows_kernel<<<dim3(bx, B), 256, 0, (hipStream_t)stream>>>(table, row_ints, sel, B, out, n_rows);
    return p2p_check_launch("p2p_gather_rows_i32");
}

// palette_ordering = "shuffled" (io_utils.py:53-55: tf.random.shuffle(colors) inside the dataset map, i.e. a NEW permutation
// every time a sample is loaded): the batch is gathered from the tables in first-appearance order and re-labelled here.
//   idx_out[b][i]    = inv[b][ idx_in[b][i] ]         (both index maps: the colour that sat at position c now sits at inv[c])
//   pal_out[b][j][:] = pal_in[b][ perm[b][j] ][:]     (position j holds the colour that sat at perm[j]);  inv[perm[j]] = j
// so indexed_to_rgba(idx_out, pal_out) == indexed_to_rgba(idx_in, pal_in) pixel for pixel.  Values outside [0, P) are clamped.
__global__ __launch_bounds__(256) void palette_relabel_kernel(const int* __restrict__ src_in, const int* __restrict__ tgt_in,
                                                              const int* __restrict__ pal_in, const int* __restrict__ perm,
                                                              const int* __restrict__ inv, int n, int P, int C,
                                                              int* __restrict__ src_out, int* __restrict__ tgt_out,
                                                              int* __restrict__ pal_out) {
    const int b = blockIdx.y;
    const int* iv = inv + (long long)b * P;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const long long o = (long long)b * n + i;
        src_out[o] = iv[min(max(src_in[o], 0), P - 1)];
        tgt_out[o] = iv[min(max(tgt_in[o], 0), P - 1)];
    }
    if (blockIdx.x == 0)
        for (int e = threadIdx.x; e < P * C; e += blockDim.x) {
            const int j = e / C, c = e - j * C;
            const int from = min(max(perm[(long long)b * P + j], 0), P - 1);
            pal_out[((long long)b * P + j) * C + c] = pal_in[((long long)b * P + from) * C + c];
        }
}

extern "C" int p2p_palette_relabel_batch(const int* src_idx, const int* tgt_idx, const int* palette, const int* perm, const int* inv,
                                         int B, int n, int P, int C, int* src_out, int* tgt_out, int* pal_out, void* stream) {
    P2P_REQUIRE(src_idx && tgt_idx && palette && perm && inv && src_out && tgt_out && pal_out, "p2p_palette_relabel_batch: null pointer");
    P2P_REQUIRE(B > 0 && n > 0 && P > 0 && C > 0, "p2p_palette_relabel_batch: bad shape");
    int bx = (n + 255) / 256;
    if (bx > 16) bx = 16;
    palette_relabel_kernel<<<dim3(bx, B), 256, 0, (hipStream_t)stream>>>(src_idx, tgt_idx, palette, perm, inv, n, P, C, src_out,
                                                                        tgt_out, pal_out);
    return p2p_check_launch("p2p_palette_relabel_batch");
}
