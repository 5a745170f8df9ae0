/*
 * p2pgan.h -- C ABI of libp2pgan_hip.so: the MI355X (gfx950) kernels behind the Pix2Pix side2side
 * training step of fegemo/palette-and-histo-gan.
 *
 * The reference has no FFI seam: its hot path bottoms out in TensorFlow/Keras ops called from Python
 * (SURVEY.md section 8b).  Each entry point below replaces the TF op(s) at the cited reference call
 * site; the Python host in palette_and_histo_gan_amd/ re-creates the reference class API
 * (Pix2PixModel.train_step etc.) on top of them through ctypes.
 *
 * Conventions (all entry points):
 *   - raw DEVICE pointers; the library never allocates or frees, workspaces are passed in;
 *   - tensors are NHWC "views" (p2p_tensor): a pointer to element (n=0,y=0,x=0,c=0) of the view plus
 *     pixel strides.  Activation buffers are allocated with a zero halo of 2 pixels around every image
 *     so kernels read the SAME-padding taps without bounds checks, and several views may alias one
 *     buffer with different channel offsets (concat-by-slice, networks.py:45,94);
 *   - dtype selects the storage/MFMA-input type of activations and weight copies: P2P_F32 (parity
 *     mode, exact-f32 MFMA) or P2P_BF16 (throughput mode, bf16 in / f32 accumulate).  Master weights,
 *     gradients, optimizer state, statistics and loss partials are always f32;
 *   - no entry point reads outside the pixels of the views it is given: where a tile row of an edge layer is wider than
 *     a pixel (4/8/36/33-channel tensors) the surplus slots re-read bytes of the same pixel.  What a view must provide is
 *     its zero halo: the stride-2 kernels gather rows/columns -1 .. 2*LH (hi) and -1 .. LH (lo) of every image, the
 *     stride-1 heads -1 .. H+1 (p2p_view_halo_pixels() = 2 pixels on every side covers all of them);
 *   - `stream` is a hipStream_t passed as void*; all work is stream-ordered, no hidden syncs;
 *   - return 0 on success, otherwise a hipError_t / negative argument-check code; the message is
 *     available from p2p_last_error() (thread-local).
 *
 * Stride-2 4x4 layers are described once by (hi, lo, W[16][Cg][Cd]):
 *   Conv2D          (networks.py:10-16): hi = input  (2LH x 2LW x Cg), lo = output, W = HWIO kernel
 *   Conv2DTranspose (networks.py:26-27): hi = output (2LH x 2LW x Cg), lo = input,  W = (kh,kw,Cout,Cin)
 * so the Keras layouts of both layer kinds are the same [tap][Cg][Cd] array and
 *   op G (hi -> lo): lo[n,y,x,d]  = sum_{kh,kw,g} hi[n,s*y+kh-1,s*x+kw-1,g] W[kh,kw,g,d]   conv fwd / convT dgrad
 *   op P (lo -> hi): hi[n,Y,X,g]  = sum_{Y=s*y+kh-1, X=s*x+kw-1, d} lo[n,y,x,d] W[kh,kw,g,d] convT fwd / conv dgrad
 *   op W           : dW[kh,kw,g,d] = sum_{n,y,x} hi[n,s*y+kh-1,s*x+kw-1,g] lo[n,y,x,d]     weight gradient
 * The stride-1 layers (networks.py:47-48,75-78; TF SAME pad 1 before / 2 after) use the same forms with s=1.
 */
#ifndef P2PGAN_H
#define P2PGAN_H

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { P2P_F32 = 0, P2P_BF16 = 1 } p2p_dtype;
typedef enum { P2P_OP_G = 0, P2P_OP_P = 1, P2P_OP_W = 2 } p2p_convop;
typedef enum { P2P_ACT_NONE = 0, P2P_ACT_LEAKY = 1, P2P_ACT_RELU = 2 } p2p_act;

typedef struct {
    void* ptr;            /* element (0,0,0,0) of the view */
    long long img_stride; /* pixels between images */
    int row_stride;       /* pixels between rows */
    int ld;               /* elements between pixels (channels of the underlying buffer) */
} p2p_tensor;

/* A gradient source read pointwise at (pixel, channel): kind 0 = absent, 1 = activation dtype,
 * 2 = f32, possibly `nslabs` split-K slabs `slab_stride` elements apart that are summed on load.
 * Dense [pixels][ld] layout, channel offset coff. */
typedef struct {
    const void* ptr;
    int kind;
    int nslabs;
    long long slab_stride;
    int ld;
    int coff;
} p2p_gsrc;

const char* p2p_last_error(void);
int p2p_version(void);
/* Zero halo (pixels on every side of every image) that the gathering kernels rely on; see Conventions. */
int p2p_view_halo_pixels(void);

/* ---- convolutions -------------------------------------------------------------------------- */

/* Direct (non-MFMA) 4x4 convolution, any channel counts, stride 1 or 2: the edge layers
 * (Cin 1..8, Cout 1..4; networks.py:46-48,57,75-78) and the on-device cross-check of the MFMA paths.
 * op G/P write the output view in `dtype`; op W writes f32 dW[16][Cg][Cd] (and dbias[Cd] if non-null,
 * = sum of lo).  `w` is the [16][Cg][Cd] weight copy in `dtype`; `bias` (f32[Cd], op G only) may be null. */
int p2p_conv_direct(int op, int stride, int dtype, int N, int LH, int LW, int Cg, int Cd,
                    const p2p_tensor* hi, const p2p_tensor* lo, const void* w, const float* bias,
                    float* dw, float* dbias, void* stream);

/* MFMA implicit-GEMM for the stride-2 blocks (networks.py:7-36), op G or P.
 * Requires Cg % 32 == 0 and Cd % 32 == 0.  `w` is Wt[16][Cd][Cg] for op G and Wn[16][Cg][Cd] for op P
 * (both produced by p2p_weight_prep).  The input view must carry the zero halo.  With splitk == 1 the
 * result is written to the output view in `dtype`; with splitk > 1 (a divisor of 16 for G, of 4 for P)
 * f32 partial slabs [splitk][pixels][Cout] are written to `slabs` and summed by the consumer
 * (p2p_norm_act_fwd / p2p_gsrc). */
int p2p_igemm(int op, int dtype, int N, int LH, int LW, int Cg, int Cd,
              const p2p_tensor* hi, const p2p_tensor* lo, const void* w,
              int splitk, float* slabs, float* stat_part, void* stream);
/* InstanceNorm statistics fused into the epilogue (splitk == 1): stat_part[N][slots][Cout][2] receives, per image and
 * slot, the mean and the centred sum of squares of an equal share of that image's output pixels;
 * slots = p2p_igemm_stat_slots(op, N, LH, LW, Cout) (0 = this shape cannot fuse them: pass stat_part = null and let
 * p2p_norm_act_fwd compute them).  p2p_norm_act_fwd consumes them with ws = stat_part, nsplit = -slots. */
int p2p_igemm_stat_slots(int op, int N, int LH, int LW, int ncols);

/* Block-resident form of p2p_igemm for the wide maps (bf16, LH == LW in {8,16,32,64}, splitk == 1; op P: Cd % 32 == 0 and
 * Cg % 64 == 0, op G: Cg % 16 == 0 and Cd % 256 == 0): a workgroup keeps the input block of 256 lo pixels (with halo) in
 * LDS, all taps / sub-pixel phases read it there and only the weights stream (csrc/brig.hip).  p2p_igemm takes this path
 * by itself whenever p2p_brig_ok says the shape qualifies (environment P2P_BRIG=0 keeps the im2col kernel);
 * p2p_igemm_layer_stat_slots is the statistics-slot query that matches the kernel p2p_igemm will use for the layer. */
int p2p_brig_ok(int op, int dtype, int N, int LH, int LW, int Cg, int Cd);
/* The whole block of networks.py:7-21 / 24-36 (without dropout) in ONE launch where a workgroup holds whole images (lo maps up
 * to 16x16): convolution, InstanceNorm statistics, normalisation and LeakyReLU / ReLU.  Writes the rounded convolution result
 * to the output view of the op (dense raw tensor: the backward pass reads it), (mean, rstd) to stats[N][Cout][2] and
 * act(gamma * (x - mean) * rstd + beta) into the (haloed, channel-sliced) view act_out.  p2p_igemm_norm_act_ok tells whether
 * the shape qualifies; otherwise p2p_igemm + p2p_norm_act_fwd do the same in two launches. */
int p2p_igemm_norm_act_ok(int op, int dtype, int N, int LH, int LW, int Cg, int Cd);
int p2p_igemm_norm_act(int op, int dtype, int N, int LH, int LW, int Cg, int Cd, const p2p_tensor* hi, const p2p_tensor* lo,
                       const void* w, const float* gamma, const float* beta, float eps, int act, float alpha,
                       const p2p_tensor* act_out, float* stats, void* stream);
int p2p_brig_stat_slots(int op, int dtype, int N, int LH, int LW, int Cg, int Cd);
int p2p_igemm_layer_stat_slots(int op, int dtype, int N, int LH, int LW, int Cg, int Cd);

/* Edge-layer form of the same kernel (Cin 1..8, the 36/33-channel concat, Cout 1..4; networks.py:46-48,57,75-78):
 * stride 1 or 2, any contraction width that fills whole 16-byte chunks (`cin_pad` = channels of the gathered
 * view as padded in HBM and in `w`), output columns masked to `ncols` (launched in 32-wide tiles; `w_rows`, a
 * multiple of 32 >= those tiles, = rows per tap slab of `w`), optional f32 bias[ncols] and LeakyReLU fused in
 * the epilogue.  `w` is [16][w_rows][cin_pad] in `dtype` (p2p_weight_prep_pad). */
int p2p_igemm_edge(int op, int stride, int dtype, int N, int LH, int LW, int cin_pad, int ncols, int w_rows,
                   const p2p_tensor* in, const p2p_tensor* out, const void* w, const float* bias,
                   int act, float alpha, void* stream);

/* LDS-strip form of p2p_igemm for the widest-map block (Cg == 32, Cd == 128, bf16: up6 forward = op P, its data gradient
 * = op G; networks.py:26-27,66-73): the input strip is staged once, the weights live in registers, the four waves split
 * the phases (op P) or the output channels (op G); persistent workgroups prefetch the next strip into registers.  Same
 * tensors and weight copies as p2p_igemm (wn for op P, wt for op G), no split-K.  op P optionally writes InstanceNorm
 * statistics like p2p_igemm: stat_part [N][slots][32][2] with slots = p2p_conv_strip_stat_slots (0 = not available).
 * p2p_conv_strip_ok tells whether the shape is supported. */
int p2p_conv_strip_ok(int op, int dtype, int N, int LH, int LW, int Cg, int Cd);
int p2p_conv_strip_stat_slots(int op, int dtype, int N, int LH, int LW, int Cg, int Cd);
int p2p_conv_strip(int op, int dtype, int N, int LH, int LW, int Cg, int Cd,
                   const p2p_tensor* hi, const p2p_tensor* lo, const void* w, float* stat_part, void* stream);

/* Few-output form (ncols <= 4, 32 < cin_pad <= 64; op G stride 1 or op P stride 2): the generator's 36 -> 4 head, the
 * discriminator's 64 -> 1 head and d(D first conv)/d(fake image) 64 -> 4 (networks.py:46,57,75-78).  Contracts the
 * channels first (rows = 16 taps x outputs, no padding of the 1..4 outputs to a 32-row tile, every input pixel read
 * once), then sums the 16 shifted taps out of LDS in a fixed order.  Same arguments and semantics as p2p_igemm_edge;
 * p2p_conv_fewout_ok tells whether the shape is supported. */
int p2p_conv_fewout_ok(int op, int stride, int dtype, int N, int LH, int LW, int cin_pad, int ncols);
int p2p_conv_fewout(int op, int stride, int dtype, int N, int LH, int LW, int cin_pad, int ncols, int w_rows,
                    const p2p_tensor* in, const p2p_tensor* out, const void* w, const float* bias,
                    int act, float alpha, void* stream);

/* Few-input form (cin_pad == 8: one 16-byte pixel, bf16 only; op G stride 1/2 or op P stride 1; ncols <= 64): the first
 * convolution of both networks and the data gradients of the two stride-1 heads (networks.py:10-16,45-46,57,75-78).
 * Weights live in registers, a strip of input rows in LDS, one MFMA K step = two taps, LDS-transposed 16-byte stores.
 * Same arguments and semantics as p2p_igemm_edge; p2p_conv_fewin_ok tells whether the shape is supported. */
int p2p_conv_fewin_ok(int op, int stride, int dtype, int N, int LH, int LW, int cin_pad, int ncols);
int p2p_conv_fewin(int op, int stride, int dtype, int N, int LH, int LW, int cin_pad, int ncols, int w_rows,
                   const p2p_tensor* in, const p2p_tensor* out, const void* w, const float* bias,
                   int act, float alpha, void* stream);

/* p2p_conv_fewin followed by the backward of the LeakyReLU in front of the layer, in one launch: out = conv(in) * (gate > 0 ?
 * 1 : alpha) with `gate` = the activation output of that block (same shape as out; whole 16-byte channel runs).  Bit-identical
 * to p2p_conv_fewin + p2p_act_bwd; the gradient tensor between them is never written (reference: the tape gradient through
 * the discriminator's first block, networks.py:45-48, pix2pix_model.py:78-79). */
int p2p_conv_fewin_actbwd(int op, int stride, int dtype, int N, int LH, int LW, int cin_pad, int ncols, int w_rows,
                          const p2p_tensor* in, const p2p_tensor* out, const void* w, const p2p_tensor* gate, float alpha,
                          void* stream);

/* MFMA weight gradient of a stride-2 block: dw[16][Cg][Cd] (f32) = sum over pixels.  The pixel sum is
 * split over `msplit` workgroups per tile; partial slabs go to `workspace`
 * (p2p_wgemm_workspace_bytes) and are reduced deterministically. */
long long p2p_wgemm_workspace_bytes(int N, int LH, int LW, int Cg, int Cd, int msplit);
int p2p_wgemm(int dtype, int N, int LH, int LW, int Cg, int Cd,
              const p2p_tensor* hi, const p2p_tensor* lo, float* dw,
              int msplit, void* workspace, void* stream);

/* Edge-layer form: stride 1 or 2, any Cg/Cd (store masked to the real counts); both views must have 16-byte
 * pixels (pad the channel count) and a zero halo. */
int p2p_wgemm_edge(int dtype, int stride, int N, int LH, int LW, int Cg, int Cd,
                   const p2p_tensor* hi, const p2p_tensor* lo, float* dw,
                   int msplit, void* workspace, void* stream);

/* LDS-resident form for the edge layers (Cg, Cd <= 64 with at most two 32x32 tiles, LW in {16,32,64}): every strip of
 * pixels is staged once and all 16 taps are contracted out of LDS, so HBM traffic = the algorithmic bytes.
 * p2p_wgrad_small_blocks returns the number of f32 slabs of 16*Cg*Cd floats the call needs in `workspace` (one partial
 * per workgroup plus the intermediate slabs of the two-level fixed-order sum), or 0 if the shape is not supported
 * (use p2p_wgemm_edge). */
int p2p_wgrad_small_blocks(int dtype, int stride, int N, int LH, int LW, int Cg, int Cd, int hi_ld, int lo_ld);
int p2p_wgrad_small(int dtype, int stride, int N, int LH, int LW, int Cg, int Cd,
                    const p2p_tensor* hi, const p2p_tensor* lo, float* dw, void* workspace, void* stream);

/* out[c] = sum over all pixels of v[n,y,x,c] (f32): bias gradients of the stride-1 heads.  One partial per workgroup and
 * channel goes to `workspace` (p2p_view_colsum_workspace_bytes), the partials are added in workgroup order: bit-reproducible. */
long long p2p_view_colsum_workspace_bytes(int dtype, int N, int H, int W, int C, const p2p_tensor* v);
int p2p_view_colsum(int dtype, int N, int H, int W, int C, const p2p_tensor* v, float* out, float* workspace, void* stream);

/* ---- InstanceNorm + activation + dropout (networks.py:18-19,29-34), fused ---------------------- */

/* raw: conv output, dense [N*H*W][C]; raw_kind 1 = `dtype`, 2 = f32 with nslabs split-K slabs.
 * gamma/beta null => no normalisation (down1, D.down: networks.py:46,58).
 * y = act(drop(gamma*(x-mu)*rsqrt(var+eps)+beta)), dropout keep-mask `mask` (u8 dense [N*H*W][C], may be
 * null) scales kept values by 2.  Writes y into the (haloed, possibly channel-sliced) view `out`,
 * mean/rstd into stats[N][C][2] and, if raw_out != null, the summed raw tensor in `dtype`.
 * nsplit > 1 splits every image's pixel range over nsplit workgroups (two launches: partial sums to the f32
 * workspace ws, >= N*nsplit*C*2*4 bytes, then apply); nsplit in {0, 1} or ws == null = one launch;
 * nsplit < 0: ws holds -nsplit statistics slots per image written by p2p_igemm's epilogue (apply only). */
int p2p_norm_act_fwd(int dtype, int N, int H, int W, int C,
                     const void* raw, int raw_kind, int nslabs, long long slab_stride,
                     const float* gamma, const float* beta, float eps, int act, float alpha,
                     const unsigned char* mask, const p2p_tensor* out, void* raw_out, float* stats,
                     float* ws, long long ws_bytes, int nsplit, void* stream);
/* As p2p_norm_act_fwd, and copies tail_ch channels per pixel from the view `tail` (same N, H, W) into the channels that
 * FOLLOW this layer's C channels in `out`: the last concat of the generator is [up6 | input image] (networks.py:92-94) and
 * written this way every pixel of the concat buffer leaves one wave complete (no partial 32-byte sectors: the separate
 * copy cost 41 of 71 us with cold caches).  Workgroup (vector) form only: H*W > 16, C % 8 == 0, tail_ch whole 16-byte vectors. */
int p2p_norm_act_fwd_tail(int dtype, int N, int H, int W, int C,
                          const void* raw, int raw_kind, int nslabs, long long slab_stride,
                          const float* gamma, const float* beta, float eps, int act, float alpha,
                          const unsigned char* mask, const p2p_tensor* out, void* raw_out, float* stats,
                          float* ws, long long ws_bytes, int nsplit, const p2p_tensor* tail, int tail_ch, void* stream);

/* Backward of the fused block: dact = g1 + g2 (pointwise gradient sources), through dropout/activation
 * (sign recomputed from raw+stats) and the InstanceNorm closed form.  Writes d(raw)
 * into the haloed view `draw` and per-image partials dgamma_part/dbeta_part [N][C] (f32).  ws/nsplit as above;
 * maps of up to 64x64 pixels whose (image, channel group) slice fits a workgroup's registers take one launch that
 * reads every operand once, whatever nsplit says (the same holds for p2p_norm_act_fwd without conv-epilogue statistics).
 * nsplit | 0x100 (nsplit > 0, forward and backward): the two-pass forms only, whose order of the per-channel sums does
 * not depend on N -- an image's result then does not depend on the batch it sits in (the engine's f32 parity mode);
 * nsplit | 0x200: the one-launch forms with the geometry they would pick for N = 256 (wide channel groups, several
 * pixels per thread), so that tests reach those variants with a small batch. */
int p2p_norm_act_bwd(int dtype, int N, int H, int W, int C,
                     const void* raw, const float* stats, const float* gamma, const float* beta,
                     int act, float alpha, const unsigned char* mask,
                     const p2p_gsrc* g1, const p2p_gsrc* g2,
                     const p2p_tensor* draw, float* dgamma_part, float* dbeta_part,
                     float* ws, long long ws_bytes, int nsplit, void* stream);

/* Backward of a LeakyReLU that was fused into a conv epilogue (only its OUTPUT is stored):
 * draw = (g1 + g2) * (act_out > 0 ? 1 : alpha). */
int p2p_act_bwd(int dtype, int N, int H, int W, int C, const p2p_tensor* act_out, const p2p_gsrc* g1,
                const p2p_gsrc* g2, float alpha, const p2p_tensor* draw, void* stream);

/* out[c] = scale * sum_r part[r][c] (f32): batch reduction of dgamma/dbeta partials, loss partials. */
int p2p_colsum(const float* part, int rows, int cols, float scale, float* out, void* stream);

/* Batched form: task t sums the dense [rows][cols] block at part + table[t][0] over rows into out + table[t][3];
 * table = device int32[ntasks][4] = {part_off, rows, cols, out_off}.  One launch reduces the dgamma/dbeta
 * partials of every InstanceNorm layer of a backward pass. */
int p2p_colsum_batched(const float* part, const int* table, int ntasks, int max_cols, float* out, void* stream);

/* ---- losses (pix2pix_model.py:44-56) ------------------------------------------------------------ */

/* Loss sums are deterministic: the loss kernels launch P2P_LOSS_BLOCKS workgroups and write one partial per workgroup
 * and loss term, partials[k * P2P_LOSS_BLOCKS + b]; p2p_loss_partials_sum adds K consecutive rows in workgroup order,
 * out[k] = sum_b partials[k * P2P_LOSS_BLOCKS + b] (one launch for all loss terms of a step). */
#define P2P_LOSS_BLOCKS 256
int p2p_loss_partials_sum(const float* partials, int K, float* out, void* stream);

/* logits: view [N2][H][W][1]; images [0,n_real) are D(real), the rest D(fake).
 * partials rows 0..2 = BCE(1,real), BCE(0,fake), BCE(1,fake) scaled by inv_count.
 * dlogits_d (all N2 images): d(real+fake loss)/dlogit; dlogits_g (fake images only): d(adv)/dlogit. */
int p2p_bce_logits(int dtype, int N2, int n_real, int H, int W, const p2p_tensor* logits,
                   float inv_count, const p2p_tensor* dlogits_d, const p2p_tensor* dlogits_g,
                   float* partials, void* stream);
/* As p2p_bce_logits for gradient views of 8-channel pixels [g | 7 padding channels] (the operand layout of the few-channel
 * convolution kernels): whole pixels are stored -- a 2-byte store into a 16-byte pixel is a partial sector write. */
int p2p_bce_logits_pad8(int dtype, int N2, int n_real, int H, int W, const p2p_tensor* logits,
                        float inv_count, const p2p_tensor* dlogits_d, const p2p_tensor* dlogits_g,
                        float* partials, void* stream);

/* fake = tanh(z) written to `fake` view; partials row 0 = inv_count * sum |real - fake| over all C channels.
 * fake_f32 (may be null): dense f32 [N*H*W][C] copy of tanh(z) before rounding to `dtype` -- the histogram loss reads its
 * images in f32 in every mode (log-chroma of dark colours does not survive 8 significant bits). */
int p2p_tanh_l1_fwd(int dtype, int N, int H, int W, int C, const p2p_tensor* z, const p2p_tensor* real,
                    const p2p_tensor* fake, float inv_count, float* partials, float* fake_f32, void* stream);
/* The train step's form for 4-channel images whose discriminator inputs are 8-channel pixels [image | source]
 * (networks.py:45): real_pair = [target | source] is read whole and the WHOLE fake pixel [tanh(z) | source] is written. */
int p2p_tanh_l1_fwd_pair(int dtype, int N, int H, int W, const p2p_tensor* z, const p2p_tensor* real_pair,
                         const p2p_tensor* fake_pair, float inv_count, float* partials, float* fake_f32, void* stream);

/* dz = (g_d + g_extra + lambda_l1*inv_count*sign(fake-real)) * (1 - fake^2) into the haloed view dz. */
int p2p_tanh_l1_bwd(int dtype, int N, int H, int W, int C, const p2p_tensor* fake, const p2p_tensor* real,
                    const p2p_gsrc* g_d, const p2p_gsrc* g_extra, float l1_scale,
                    const p2p_tensor* dz, void* stream);
/* As p2p_tanh_l1_bwd (C = 4) for a dz view of 8-channel pixels [dz | 4 padding channels]: whole-pixel stores. */
int p2p_tanh_l1_bwd_pad8(int dtype, int N, int H, int W, const p2p_tensor* fake, const p2p_tensor* real,
                         const p2p_gsrc* g_d, const p2p_gsrc* g_extra, float l1_scale, const p2p_tensor* dz,
                         void* stream);

/* ---- RGB-uv histogram + Hellinger loss (histogram.py:4-89, pix2pix_model.py:242-250) ------------------------- */

/* The histogram entry points take `dtype` = the element type of the IMAGE view they read; the engine always hands them f32
 * views (the f32 input batch for the real image, p2p_tanh_l1_fwd's fake_f32 for the generated one), also in bf16 mode.
 * Raw (unnormalised) histogram of the RGB channels of `img` ([-1,1] images): hist[N][3][64][64] f32 with
 * hist[n][c][i][j] = sum_p Iy[p] k(u_p - d_i) k(v_p - d_j) for component c (the reference's (B,64,64,3) tensor
 * transposed and before its division by the per-image total, histogram.py:75-79). */
int p2p_rgbuv_hist_fwd(int dtype, int N, int H, int W, const p2p_tensor* img, float* hist, void* stream);
/* The same with the reference function's other arguments (histogram.py:36): hist[N][3][size][size], size 2..128, bin centres
 * linspace(-3, 3, size), method 0 = "inverse-quadratic", 1 = "RBF", 2 = any other string (the reference then applies no kernel
 * function: histogram.py:20-27 has no third branch), sigma > 0.  General f32 kernel for evaluation code; no reference call site
 * leaves the defaults, which the specialised kernels serve. */
int p2p_rgbuv_hist_general(int dtype, int N, int H, int W, const p2p_tensor* img, int size, int method, float sigma, float* hist,
                           void* stream);

/* The same raw histograms [N][3][64][64], all three colour components of an image from THREE shared kernel rows per pixel
 * (the components' (u, v) are (a, b), (-a, c), (-b, -c) of three log-chroma differences and the bin grid is symmetric,
 * histogram.py:54-55,72-74), contracted over the image's pixels or -- where `points`/`npoints` (p2p_rgbuv_points) list them
 * -- over its DISTINCT colours weighted by their pixel counts (identical up to f32 summation order).  workspace:
 * p2p_rgbuv_hist_fwd3_workspace_bytes(N) bytes (partial histograms of the pixel ranges, summed in fixed order). */
long long p2p_rgbuv_hist_fwd3_workspace_bytes(int N);
int p2p_rgbuv_hist_fwd3(int dtype, int N, int H, int W, const p2p_tensor* img, const float* points, const int* npoints, int cap,
                        float* hist, float* workspace, void* stream);
/* Distinct colours of every image with their pixel counts: points[n][k] = (r, g, b, count) for k < npoints[n], in order of
 * first appearance within tiles of 1024 pixels (bitwise equality of the f32 / bf16 RGB values; colours of different tiles
 * are not merged).  npoints[n] = -1 where the list would exceed `cap` entries (the image is then contracted densely).
 * Deterministic (integer LDS atomics only).  points: f32 [N][cap][4], 16-byte aligned. */
int p2p_rgbuv_points(int dtype, int N, int H, int W, const p2p_tensor* img, int cap, float* points, int* npoints, void* stream);

/* out[N][64][64][3] = raw[N][3][64][64] transposed and divided by the per-image total (histogram.py:75-79). */
int p2p_hist_normalize(const float* raw, int N, float* out, void* stream);

/* Per-image totals of both raw histograms and the LOCAL Hellinger sum of squares
 * sq_sum[0] = sum_{n,c,i,j} (sqrt(pred/tot_pred) - sqrt(true/tot_true))^2 (histogram.py:88-89): one partial per image in
 * sq_part[N], added in image order (bit-reproducible, no float atomics).  Under data parallelism sq_sum is all-reduced
 * (SUM) before the two calls below. */
int p2p_hellinger_fwd(const float* hist_true, const float* hist_pred, int N, float* tot_true, float* tot_pred,
                      float* sq_part, float* sq_sum, void* stream);
/* loss_out[0] = sqrt(sq_sum) / (sqrt(2) * B_global). */
int p2p_hellinger_finish(const float* sq_sum, float inv_global_batch, float* loss_out, void* stream);
/* d(coef' * hellinger)/d(fake image) with coef = lambda_hist / (2*sqrt(2)*B_global): writes three f32 slabs
 * dimg[3][N*H*W][4] (one per colour component, alpha gradient 0) that the consumer sums; gh_ws is a
 * [N][3][64][64] f32 workspace. */
int p2p_rgbuv_hist_hellinger_bwd(int dtype, int N, int H, int W, const p2p_tensor* fake, const float* hist_true,
                                 const float* hist_pred, const float* tot_true, const float* tot_pred,
                                 const float* sq_sum, float coef, float* gh_ws, float* dimg, void* stream);
/* The same gradient, the three colour components evaluated together from three shared kernel rows per pixel and summed in
 * the kernel: ONE f32 slab [N*H*W][4] is written to `dimg` (consumer: p2p_tanh_l1_bwd with a one-slab gradient source). */
int p2p_rgbuv_hist_hellinger_bwd3(int dtype, int N, int H, int W, const p2p_tensor* fake, const float* hist_true,
                                  const float* hist_pred, const float* tot_true, const float* tot_pred, const float* sq_sum,
                                  float coef, float* gh_ws, float* dimg, void* stream);

/* ---- palette-index head (pix2pix_model.py:261-325) ------------------------------------------------------------ */

/* z: logits view [N][H][W][C]; target: view holding the real palette index of every pixel (as a value of `dtype`).
 * Writes argmax(softmax(z)) (ties -> lowest index) into fake_idx as a value of `dtype`, optionally
 * dz = grad_scale * (softmax(z) - onehot(target)) and the f32 probabilities; loss_out[0] = inv_count * sum CCE with
 * CCE = log(sum_c exp(z_c - max)) - (z_target - max) (the logits form Keras 2.9 uses for a softmax-activated output,
 * finite for every input), loss_out[1] = inv_count / C * sum |onehot - p|.  loss_part: workspace of
 * 2 * P2P_SOFTMAX_MAX_BLOCKS floats (one partial per workgroup, summed in workgroup order: bit-reproducible). */
#define P2P_SOFTMAX_MAX_BLOCKS 8192
int p2p_softmax_cce_argmax(int dtype, int N, int H, int W, int C, const p2p_tensor* z, const p2p_tensor* target,
                           const p2p_tensor* fake_idx, float grad_scale, float inv_count, const p2p_tensor* dz,
                           float* probs_out, float* loss_part, float* loss_out, void* stream);
/* tf.argmax(probs, axis=-1, output_type=int32) on dense f32 probabilities [M][C]; ties -> lowest index. */
int p2p_argmax_lastdim(const float* probs, long long M, int C, int* out, void* stream);

/* The palette-index head in one launch (bf16, IMG_SIZE 64: query p2p_head_softmax_ok): Conv2D(256, 4, stride 1, SAME, bias) of
 * the haloed concat view `in` (pixels of cin_pad = 40 channels: [up6 32 | source | zero pad], networks.py:75-78,92-94) with
 * the op-G weight copy wt[16][256][40], softmax over the 256 palette slots, argmax (ties -> lowest index) written as an
 * activation-dtype value into `fake_idx`, CategoricalCrossentropy against the index image `target` in the log-sum-exp form,
 * and its gradient grad_scale * (probs - onehot) into the haloed view `dz` (256-channel pixels) -- what p2p_igemm_edge +
 * p2p_softmax_cce_argmax + p2p_view_colsum compute, without ever writing the logits (pix2pix_model.py:268,273-293,300-301).
 * loss_out[0] = inv_count * sum of -log p_target, loss_out[1] = inv_count / 256 * sum |onehot - p|; dbias (may be NULL) =
 * column sums of dz.  workspace: p2p_head_softmax_workspace_bytes(N, H) bytes (per-workgroup partials, summed in order). */
int p2p_head_softmax_ok(int dtype, int N, int H, int W, int cin_pad, int ncls);
long long p2p_head_softmax_workspace_bytes(int N, int H);
int p2p_head_softmax_cce(int dtype, int N, int H, int W, int cin_pad, int ncls, const p2p_tensor* in, const void* wt,
                         const float* bias, const p2p_tensor* target, const p2p_tensor* fake_idx, float grad_scale,
                         float inv_count, const p2p_tensor* dz, float* dbias, float* workspace, float* loss_out, void* stream);
/* Data gradient of that head through its first 32 input channels (the up6 slice of the concat; the source image has no
 * gradient): op P, stride 1, out[n,y,x,g] = sum_{kh,kw,d} dz[n,y+1-kh,x+1-kw,d] * W[kh][kw][g][d], g < 32, with the op-P weight
 * copy wn[16][w_rows][256] (bf16, IMG_SIZE 64: query p2p_head_dgrad_ok).  dz: haloed view of 256-channel pixels (zero halo of
 * 2 pixels); out: view with >= 32 channels per pixel, channels [0, 32) are written. */
int p2p_head_dgrad_ok(int dtype, int N, int H, int W, int ncls, int cout, int w_rows, int dz_ld, int out_ld);
int p2p_head_dgrad(int dtype, int N, int H, int W, int ncls, int cout, const p2p_tensor* dz, const void* wn, int w_rows,
                   const p2p_tensor* out, void* stream);

/* ---- optimizer / parameter plumbing (pix2pix_model.py:28-29,81-83) ------------------------------- */

/* Keras Adam over a flat f32 buffer; t = iteration AFTER increment; grads are multiplied by gscale first. */
int p2p_adam_flat(float* p, const float* g, float* m, float* v, long long n, int t,
                  float lr, float beta1, float beta2, float eps, float gscale, void* stream);

/* Device-resident step state (lets a whole step be captured in a hipGraph and replayed): t_dev[0] += 1 and
 * lr_t_dev[0] = lr*sqrt(1-beta2^t)/(1-beta1^t); p2p_adam_flat_dev reads the step size from lr_t_dev. */
int p2p_adam_tick(int* t_dev, float* lr_t_dev, float lr, float beta1, float beta2, void* stream);
int p2p_adam_flat_dev(float* p, const float* g, float* m, float* v, long long n, const float* lr_t_dev,
                      float beta1, float beta2, float eps, float gscale, void* stream);
int p2p_counter_add(long long* counter_dev, long long inc, void* stream);

/* master f32 W[16][Cg][Cd] -> wn (dtype, same layout; may be null) and wt (dtype, [16][Cd][Cg]; may be null). */
int p2p_weight_prep(int dtype, const float* w, int Cg, int Cd, void* wn, void* wt, void* stream);

/* Padded variant for the edge layers: wn is [16][wn_rows][wn_cols], wt is [16][wt_rows][wt_cols]; entries
 * outside the real [Cg][Cd] block are zero. */
int p2p_weight_prep_pad(int dtype, const float* w, int Cg, int Cd, void* wn, int wn_rows, int wn_cols,
                        void* wt, int wt_rows, int wt_cols, void* stream);

/* Batched form: one launch for every layer.  `tasks_dev` is a DEVICE array of ntasks (<= 64) descriptors with the
 * arguments of p2p_weight_prep_pad (null wn / wt = not wanted); tiles_g, tiles_d and the task's workgroup count come from
 * p2p_weight_prep_task_blocks, first_block is the running sum of the preceding tasks' counts, total_blocks the sum. */
typedef struct p2p_prep_task {
    const float* w;
    void* wn;
    void* wt;
    int Cg, Cd, wn_rows, wn_cols, wt_rows, wt_cols, tiles_g, tiles_d;
    long long first_block;
} p2p_prep_task;
long long p2p_weight_prep_task_blocks(int Cg, int Cd, int wn_rows, int wn_cols, int wt_rows, int wt_cols,
                                      int have_wn, int have_wt, int* tiles_g, int* tiles_d);
int p2p_weight_prep_batched(int dtype, const p2p_prep_task* tasks_dev, int ntasks, long long total_blocks, void* stream);
/* Keras Adam step (as p2p_adam_flat_dev, same expressions) on the masters of the listed layers AND their operand copies in one
 * pass (tf.keras.optimizers.Adam.apply_gradients, pix2pix_model.py:81-83, followed by what p2p_weight_prep_batched derives):
 * every task's master `w` lies inside `params`; grads / m / v are the flat buffers parallel to it.  Each kernel tensor must be
 * listed ONCE; n_elems = sum of 16 * Cg * Cd over the tasks.  The small tensors (gamma, beta, bias) go through
 * p2p_adam_flat_dev. */
int p2p_adam_prep_batched(int dtype, long long n_elems, const p2p_prep_task* tasks_dev, int ntasks, long long total_blocks,
                          float* params, const float* grads, float* m, float* v, const float* lr_t_dev, float beta1, float beta2,
                          float eps, void* stream);


/* dense f32 (or i32 if src_is_int) [N][H][W][C] host-layout batch -> view in `dtype` (dataset_utils.py:39-48 contract). */
int p2p_pack_input(int dtype, int N, int H, int W, int C, const void* src, int src_is_int,
                   const p2p_tensor* dst, void* stream);
/* same batch into ndst (1..4) views with one read (dsts = array of p2p_tensor). */
int p2p_pack_input_multi(int dtype, int N, int H, int W, int C, const void* src, int src_is_int,
                         const p2p_tensor* dsts, int ndst, void* stream);
/* The RGBA (4-channel) batch of a train step in one launch: source and target read once, whole-pixel stores into
 * down1's input [source|0], channels 32..39 of the last concat buffer [source|0], the real half of the discriminator
 * input [target|source] and the source half of its fake half.  Every view starts on an 8-channel boundary.
 * v_c6 and v_dfake may be NULL: those two are PARTIAL-pixel stores (16 of 80 bytes, 8 of 16), which the train step leaves to
 * the kernels that write the rest of the pixel (p2p_norm_act_fwd_tail, p2p_tanh_l1_fwd_pair). */
int p2p_pack_pair(int dtype, int N, int H, int W, const float* source, const float* target,
                  const p2p_tensor* v_src, const p2p_tensor* v_c6, const p2p_tensor* v_dreal, const p2p_tensor* v_dfake,
                  void* stream);
/* The palette-index batch of a train step (one int32 index per pixel, dataset_utils.py:232-246; 8-channel pixels): whole-pixel
 * stores of [source 0..] into v_src (and v_c6 unless NULL), [target source 0..] into v_dreal, [0 source 0..] into v_dfake
 * (its channel 0 is written later by the head's argmax). */
int p2p_pack_pair_idx(int dtype, int N, int H, int W, const int* source, const int* target,
                      const p2p_tensor* v_src, const p2p_tensor* v_c6, const p2p_tensor* v_dreal, const p2p_tensor* v_dfake,
                      void* stream);
/* out[7] = [g_total, g_adv, g_l1, g_aux, d_total, d_real, d_fake] from the loss slots written by the loss kernels:
 * slots[0..2] = BCE(1,real), BCE(0,fake), BCE(1,fake); slots[l1_slot] = L1; slots[aux_slot] = histogram /
 * segmentation loss (aux_slot < 0: none); g_total = adv + lambda_l1*l1 + lambda_aux*aux. */
int p2p_finish_losses(const float* slots, int aux_slot, int l1_slot, float lambda_l1, float lambda_aux, float* out,
                      void* stream);
/* view in `dtype` -> dense f32 [N][H][W][C] (for generate()/tests). */
int p2p_unpack(int dtype, int N, int H, int W, int C, const p2p_tensor* src, float* dst, void* stream);

/* Bernoulli(0.5) keep mask of Dropout(0.5) (networks.py:31-32): counter-based RNG, one byte per element. */
int p2p_dropout_mask(unsigned char* mask, long long n, long long seed, long long counter, void* stream);
/* same with the call counter = counter_dev[0]*16 + salt read on the device; the random stream is indexed by
 * elem_offset + i (elem_offset = elements of this layer's mask that belong to the samples in front of the shard, a
 * multiple of 8), so the ranks of a data-parallel step draw the masks of the single-process global batch. */
int p2p_dropout_mask_dev(unsigned char* mask, long long n, long long seed, const long long* counter_dev,
                         long long salt, long long elem_offset, void* stream);

/* ---- collectives (RCCL over xGMI; build-added data parallelism) ----------------------------------- */

/* For hosts without PyTorch (the Python host of this repository reaches the same RCCL through torch.distributed).  Rank 0
 * obtains 128 opaque bytes with p2p_comm_unique_id and distributes them; every rank (one process per GPU, hipSetDevice done)
 * calls p2p_comm_init with them.  p2p_comm_allreduce_sum: in-place SUM of n f32 values, stream-ordered -- per step the flat
 * generator gradient buffer in buckets, the tail [small tensors | discriminator gradients | loss slots] and, for the
 * histogram model, the one Hellinger scalar between p2p_hellinger_fwd and p2p_hellinger_finish.  librccl.so is loaded on the
 * first call (no link-time dependency). */
int p2p_comm_unique_id(void* id_out_128_bytes);
int p2p_comm_init(const void* id_128_bytes, int rank, int world, void** comm_out);
int p2p_comm_allreduce_sum(void* comm, float* buf, long long n, void* stream);
int p2p_comm_destroy(void* comm);

/* ---- stream ordering (the drop-in's own schedule; the reference's train_step is one tf.function, pix2pix_model.py:63-87) ---
 * Events that order kernels of ONE device between the engine's HIP streams (weight gradients and histograms run beside the
 * data-gradient chain).  Created with hipEventDisableTiming | hipEventDisableSystemFence: no host-visible release, so a record
 * does not write back / invalidate L2.  Not for host synchronisation (use the stream's own synchronise for that). */
int p2p_event_create(void** ev_out);
int p2p_event_destroy(void* ev);
int p2p_event_record(void* ev, void* stream);            /* marks the work issued so far on `stream` */
int p2p_stream_wait_event(void* stream, void* ev);       /* later work on `stream` waits for the event's latest record */
/* A record costs the recording stream a packet of its own between two kernels.  p2p_arm_stop_event(ev) instead hands `ev` to the
 * LAST kernel that the next supporting entry point of this thread launches (p2p_norm_act_bwd, p2p_act_bwd: the calls the weight-
 * gradient forks follow), as that dispatch's completion signal: equivalent to p2p_event_record(ev, stream) right behind the call.
 * p2p_disarm_stop_event(&was_pending) afterwards tells whether the event is still unclaimed (record it the ordinary way then). */
int p2p_arm_stop_event(void* ev);
int p2p_disarm_stop_event(int* was_pending);

/* ---- step replay (the reference runs train_step as ONE traced tf.function, pix2pix_model.py:62: a step costs the host one
 * call, not one per op; side2side_model.py:73,114 is the loop that issues it) -------------------------------------------------
 * A train step is ~105 kernel entry points and ~40 stream operations whose arguments do not change from step to step (persistent
 * buffers, explicit stream handles).  The host records them once as an array of p2p_replay_call and re-issues the whole step
 * with ONE p2p_replay call: the same entry points, in the same order, on the same streams -- bit-identical results, no
 * interpreter between two launches.  `fn` = p2p_replay_fn_index(name of an entry point of this header that returns int and takes
 * only scalars and pointers); a[k] = argument k in an 8-byte slot (ints and floats in the low bytes, little endian).  Bit k of
 * `ind64` / `ind32` set: a[k] is the ADDRESS of a host variable of 8 / 4 bytes that is read when the call is replayed (the batch
 * pointers, the result pointer, the optimizer's hyper-parameters: what may change between two steps of one recording).
 * Structures passed by pointer (p2p_tensor, p2p_gsrc) are read at replay time like any other pointer argument: the host keeps
 * them alive.  Returns 0, or the first failing call's code with p2p_last_error() naming the call's index and entry point. */
#define P2P_REPLAY_MAX_ARGS 24
typedef struct {
    int fn;
    int nargs;
    unsigned ind64;
    unsigned ind32;
    unsigned long long a[P2P_REPLAY_MAX_ARGS];
} p2p_replay_call;
int p2p_replay_fn_index(const char* name);      /* -1: not a replayable entry point */
int p2p_replay_fn_nargs(int fn);
int p2p_replay(const p2p_replay_call* calls, int n);

/* ---- input pipeline (reference dataset_utils.py:11-20,39-49,66-120,209-246) ---------------------------- */

/* Host helper (no GPU): undo the PNG scanline filters of an inflated IDAT stream (height rows of 1 + row_bytes bytes, 8-bit
 * samples, bpp bytes per pixel) into out[height][row_bytes].  Replaces libpng behind tf.image.decode_png (dataset_utils.py:68). */
int p2p_png_unfilter(const unsigned char* filtered, int height, int row_bytes, int bpp, unsigned char* out);

/* One RGBA train/test batch from the HBM-resident sprite set `sprites` (uint8 [n_sprites][S][S][4], S a power of two).
 * src_idx/tgt_idx: device int32 [B] sprite numbers; aug: device f32 [B][4] = (apply != 0, hue delta in turns, dy, dx in
 * pixels) or NULL (test set / augment=False).  Per pixel: translate (nearest, fill 0) -> blacken alpha == 0
 * (dataset_utils.py:11-20) -> hue rotation of RGB (tf.image.adjust_hue semantics, :80-84) -> x / 127.5 - 1 if `normalise`
 * (:39-49) -> source/target f32 [B][S][S][4], the two tensors train_step takes (pix2pix_model.py:63-64). */
int p2p_sprites_rgba_batch(const void* sprites, int n_sprites, int S, const int* src_idx, const int* tgt_idx, const float* aug,
                           int B, int normalise, float* source, float* target, void* stream);

/* Row gather out[b] = table[sel[b]] (rows of row_ints int32, row_ints % 4 == 0): indexed batches are rows of the index maps
 * and palettes extracted once at load time (dataset_utils.py:123-164). */
int p2p_gather_rows_i32(const int* table, int n_rows, int row_ints, const int* sel, int B, int* out, void* stream);

/* palette_ordering = "shuffled" (io_utils.py:53-55 calls tf.random.shuffle inside the dataset map: a new permutation of the
 * palette's colours every time a sample is loaded).  Re-labels one gathered indexed batch: src_out/tgt_out[b][i] =
 * inv[b][idx[b][i]] over the n index values of each sample, pal_out[b][j] = palette[b][perm[b][j]] over the P palette rows of C
 * ints; perm/inv: device int32 [B][P], inverse of each other (the padding rows map to themselves).  The decoded image
 * indexed_to_rgba(idx, palette) (io_utils.py:96-103) is unchanged. */
int p2p_palette_relabel_batch(const int* src_idx, const int* tgt_idx, const int* palette, const int* perm, const int* inv,
                              int B, int n, int P, int C, int* src_out, int* tgt_out, int* pal_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
