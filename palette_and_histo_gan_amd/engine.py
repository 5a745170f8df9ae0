"""Device engine of the Pix2Pix side2side training step on MI355X.

Owns the HBM layout (haloed NHWC activation buffers, concat-by-slice, flat f32 parameter / gradient /
Adam buffers, per-layer weight copies for the MFMA kernels) and issues the HIP kernels of
libp2pgan_hip.so in the order of the reference's train_step (pix2pix_model.py:62-89, 295-325).  PyTorch
tensors are only device-memory holders here; every arithmetic op on the path is a call into the C ABI
(include/p2pgan.h).  There is no CPU fallback.
"""
from collections import OrderedDict
import ctypes as C
import math

import numpy as np
import torch

from . import _lib as L

HALO = 2
IN_EPS = 1e-3           # tfa InstanceNormalization default (networks.py:18,29)
LEAKY_ALPHA = 0.3       # keras LeakyReLU default (networks.py:19)
DOWN_FILTERS = (64, 128, 256, 512, 512, 512)      # networks.py:57-64
UP_FILTERS = (512, 512, 256, 128, 64, 32)         # networks.py:66-73
UP_DROPOUT = (True, True, True, False, False, False)
MAX_PALETTE_SIZE = 256


def _torch_dtype(dtype):
    return torch.float32 if dtype == L.F32 else torch.bfloat16


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class HaloBuf:
    """NHWC activation buffer with a zero halo of HALO pixels around every image."""

    def __init__(self, n, h, w, c, dtype, device):
        self.n, self.h, self.w, self.c, self.dtype = n, h, w, c, dtype
        self.hp, self.wp = h + 2 * HALO, w + 2 * HALO
        self.t = torch.zeros((n, self.hp, self.wp, c), dtype=_torch_dtype(dtype), device=device)
        self.esz = self.t.element_size()

    def view(self, coff=0, n0=0):
        off = ((n0 * self.hp + HALO) * self.wp + HALO) * self.c + coff
        return L.Tensor(self.t.data_ptr() + off * self.esz, self.hp * self.wp, self.wp, self.c)


class DenseBuf:
    """Dense [N*H*W][C] tensor (conv raw outputs, gradient sources)."""

    def __init__(self, n, h, w, c, torch_dtype, device):
        self.n, self.h, self.w, self.c = n, h, w, c
        self.t = torch.empty((n * h * w, c), dtype=torch_dtype, device=device)
        self.esz = self.t.element_size()

    def view(self, coff=0, n0=0):
        return L.Tensor(self.t.data_ptr() + (n0 * self.h * self.w * self.c + coff) * self.esz,
                        self.h * self.w, self.w, self.c)

    def ptr(self, n0=0):
        return C.c_void_p(self.t.data_ptr() + n0 * self.h * self.w * self.c * self.esz)

    def gsrc(self, coff=0, kind=1, nslabs=1, n0=0):
        return L.GSrc(self.t.data_ptr() + n0 * self.h * self.w * self.c * self.esz, kind, nslabs,
                      self.n * self.h * self.w * self.c, self.c, coff)


def _p(t, off_elems=0):
    return C.c_void_p(t.data_ptr() + off_elems * t.element_size())


NULL = C.c_void_p(0)


class ParamStore:
    """Flat f32 parameter / gradient / Adam-moment buffers with named views (Keras variable order)."""

    def __init__(self, shapes, device):
        self.shapes = OrderedDict(shapes)
        self.offsets = OrderedDict()
        off = 0
        for k, s in self.shapes.items():
            self.offsets[k] = off
            off += int(np.prod(s))
            off = (off + 3) // 4 * 4          # keep every tensor 16-byte aligned
        self.numel = off
        self.params = torch.zeros(off, dtype=torch.float32, device=device)
        self.grads = torch.zeros(off, dtype=torch.float32, device=device)
        self.m = torch.zeros(off, dtype=torch.float32, device=device)
        self.v = torch.zeros(off, dtype=torch.float32, device=device)
        self.t = 0                            # Adam iteration count

    def count(self):
        return int(sum(int(np.prod(s)) for s in self.shapes.values()))

    def view(self, buf, name):
        o = self.offsets[name]
        return buf[o:o + int(np.prod(self.shapes[name]))].view(self.shapes[name])

    def p(self, name):
        return _p(self.params, self.offsets[name])

    def g(self, name):
        return _p(self.grads, self.offsets[name])

    def load(self, values):
        for k in self.shapes:
            self.view(self.params, k).copy_(torch.as_tensor(np.asarray(values[k]), dtype=torch.float32))

    def export(self, buf=None):
        buf = self.params if buf is None else buf
        return OrderedDict((k, self.view(buf, k).detach().cpu().numpy().copy()) for k in self.shapes)


def generator_param_shapes(in_ch, out_ch):
    """Variable order / shapes of UnetGenerator (networks.py:53-98); conv kernels keep the Keras layouts
    HWIO (Conv2D) and (kh,kw,Cout,Cin) (Conv2DTranspose) == [tap][Cg][Cd] in both cases."""
    shapes = OrderedDict()
    c = in_ch
    for i, f in enumerate(DOWN_FILTERS, start=1):
        shapes[f"down{i}.kernel"] = (4, 4, c, f)
        if i > 1:
            shapes[f"down{i}.gamma"] = (f,)
            shapes[f"down{i}.beta"] = (f,)
        c = f
    skips = list(reversed(DOWN_FILTERS[:-1])) + [in_ch]
    for i, (f, s) in enumerate(zip(UP_FILTERS, skips), start=1):
        shapes[f"up{i}.kernel"] = (4, 4, f, c)
        shapes[f"up{i}.gamma"] = (f,)
        shapes[f"up{i}.beta"] = (f,)
        c = f + s
    shapes["last.kernel"] = (4, 4, c, out_ch)
    shapes["last.bias"] = (out_ch,)
    return shapes


def discriminator_param_shapes(in_ch):
    """PatchDiscriminator variables (networks.py:39-50)."""
    return OrderedDict([("down.kernel", (4, 4, 2 * in_ch, 64)), ("last.kernel", (4, 4, 64, 1)), ("last.bias", (1,))])


class Pix2PixEngine:
    """One generator + one discriminator + their optimizers on one GPU."""

    def __init__(self, in_ch=4, out_ch=4, head="tanh", img_size=64, dtype=L.BF16, device="cuda:0", seed=47,
                 use_mfma=True):
        assert img_size % 64 == 0 and (img_size & (img_size - 1)) == 0, "IMG_SIZE must be a power of two >= 64"
        L.lib()       # fail loudly now if the HIP library is missing
        self.in_ch, self.out_ch, self.head, self.S = in_ch, out_ch, head, img_size
        self.dtype, self.device, self.use_mfma = dtype, torch.device(device), use_mfma
        self.tdt = _torch_dtype(dtype)
        self.G = ParamStore(generator_param_shapes(in_ch, out_ch), self.device)
        self.D = ParamStore(discriminator_param_shapes(in_ch), self.device)
        self.rng = np.random.default_rng(seed)
        self.seed, self.mask_counter = int(seed), 0
        self._init_params()
        self.wcopies = {}           # (store id, name) -> dict(wn=tensor, wt=tensor)
        self._alloc_weight_copies()
        self.plans = {}
        self.lr, self.beta1, self.beta2, self.adam_eps = 2e-4, 0.5, 0.999, 1e-7   # pix2pix_model.py:28-29
        self.losses = torch.zeros(16, dtype=torch.float32, device=self.device)
        self.step_count = 0
        self.refresh_weight_copies()

    # ------------------------------------------------------------------ parameters
    def _init_params(self):
        """tf.random_normal_initializer(0., 0.02) kernels, zero biases, gamma=1, beta=0 (networks.py:7,24,40,54)."""
        for store in (self.G, self.D):
            vals = {}
            for k, s in store.shapes.items():
                if k.endswith(".kernel"):
                    vals[k] = self.rng.normal(0.0, 0.02, size=s).astype(np.float32)
                elif k.endswith(".gamma"):
                    vals[k] = np.ones(s, np.float32)
                else:
                    vals[k] = np.zeros(s, np.float32)
            store.load(vals)

    def _layers(self, store):
        return [k[:-7] for k in store.shapes if k.endswith(".kernel")]

    def _alloc_weight_copies(self):
        for sid, store in (("G", self.G), ("D", self.D)):
            for name in self._layers(store):
                kh, kw, cg, cd = store.shapes[name + ".kernel"]
                n = 16 * cg * cd
                ent = {"cg": cg, "cd": cd}
                if self.dtype == L.F32:
                    ent["wn"] = None           # the f32 master is its own native copy
                else:
                    ent["wn"] = torch.empty(n, dtype=self.tdt, device=self.device)
                ent["wt"] = torch.empty(n, dtype=self.tdt, device=self.device) if (cg % 32 == 0 and cd % 32 == 0) else None
                self.wcopies[(sid, name)] = ent

    def refresh_weight_copies(self):
        """Re-derives the per-layer weight copies ([16][Cg][Cd] native and [16][Cd][Cg] transposed, in the
        activation dtype) from the f32 masters; runs after every Adam step."""
        for (sid, name), ent in self.wcopies.items():
            store = self.G if sid == "G" else self.D
            wn = _p(ent["wn"]) if ent["wn"] is not None else NULL
            wt = _p(ent["wt"]) if ent["wt"] is not None else NULL
            if ent["wn"] is None and ent["wt"] is None:
                continue
            L.call("p2p_weight_prep", self.dtype, store.p(name + ".kernel"), ent["cg"], ent["cd"], wn, wt, _stream())

    def wn(self, sid, name):
        ent = self.wcopies[(sid, name)]
        if ent["wn"] is None:
            store = self.G if sid == "G" else self.D
            return store.p(name + ".kernel")
        return _p(ent["wn"])

    def wt(self, sid, name):
        return _p(self.wcopies[(sid, name)]["wt"])

    def set_params(self, g_values=None, d_values=None):
        if g_values is not None:
            self.G.load(g_values)
        if d_values is not None:
            self.D.load(d_values)
        self.refresh_weight_copies()

    # ------------------------------------------------------------------ buffers
    def plan(self, B):
        if B in self.plans:
            return self.plans[B]
        S, dt, dev, tdt = self.S, self.dtype, self.device, self.tdt
        P = {"B": B}
        # concat buffers c1..c6: [up_k output | skip]   (networks.py:92-94)
        skips = list(reversed(DOWN_FILTERS[:-1])) + [self.in_ch]
        P["c"] = [None]
        for k in range(1, 7):
            res = S // 64 * (2 ** k)
            P["c"].append(HaloBuf(B, res, res, UP_FILTERS[k - 1] + skips[k - 1], dt, dev))
        r6 = S // 64
        P["a6"] = HaloBuf(B, r6, r6, 512, dt, dev)
        # raw conv outputs, stats, dropout masks, d(raw)
        P["rd"], P["ru"], P["sd"], P["su"], P["dd"], P["du"], P["mask"] = {}, {}, {}, {}, {}, {}, {}
        for i, f in enumerate(DOWN_FILTERS, start=1):
            res = S // (2 ** i)
            P["rd"][i] = DenseBuf(B, res, res, f, tdt, dev)
            P["dd"][i] = HaloBuf(B, res, res, f, dt, dev)
            if i > 1:
                P["sd"][i] = torch.empty((B, f, 2), dtype=torch.float32, device=dev)
        for i, f in enumerate(UP_FILTERS, start=1):
            res = S // 64 * (2 ** i)
            P["ru"][i] = DenseBuf(B, res, res, f, tdt, dev)
            P["du"][i] = HaloBuf(B, res, res, f, dt, dev)
            P["su"][i] = torch.empty((B, f, 2), dtype=torch.float32, device=dev)
            if UP_DROPOUT[i - 1]:
                P["mask"][i] = torch.empty((B * res * res, f), dtype=torch.uint8, device=dev)
        # gradient sources: d(concat_k) for k=1..6, d(a_k) from the down path, d(a6)
        P["gc"] = [None] + [DenseBuf(B, P["c"][k].h, P["c"][k].w, P["c"][k].c, tdt, dev) for k in range(1, 7)]
        P["ga"] = {i: DenseBuf(B, S // 2 ** i, S // 2 ** i, DOWN_FILTERS[i - 1], tdt, dev) for i in range(1, 7)}
        P["part"] = torch.empty((2, B, 1024), dtype=torch.float32, device=dev)     # dgamma/dbeta partials
        # generator head
        P["z"] = DenseBuf(B, S, S, self.out_ch, tdt, dev)
        P["dz"] = HaloBuf(B, S, S, self.out_ch, dt, dev)
        # discriminator: images [0,B) = [real | source], [B,2B) = [fake | source]   (networks.py:45)
        ic = self.in_ch
        P["dcat"] = HaloBuf(2 * B, S, S, 2 * ic, dt, dev)
        P["d_raw"] = DenseBuf(2 * B, S // 2, S // 2, 64, tdt, dev)
        P["d_act"] = HaloBuf(2 * B, S // 2, S // 2, 64, dt, dev)
        P["logits"] = DenseBuf(2 * B, S // 2, S // 2, 1, tdt, dev)
        P["dld"] = HaloBuf(2 * B, S // 2, S // 2, 1, dt, dev)
        P["dlg"] = HaloBuf(B, S // 2, S // 2, 1, dt, dev)
        P["g_dact"] = DenseBuf(2 * B, S // 2, S // 2, 64, tdt, dev)
        P["d_draw"] = HaloBuf(2 * B, S // 2, S // 2, 64, dt, dev)
        P["g_dcat"] = DenseBuf(B, S, S, 2 * ic, tdt, dev)
        # split-K / wgrad workspaces
        P["slabs"] = torch.empty(self._max_slab_elems(B), dtype=torch.float32, device=dev)
        P["wws"] = torch.empty(self._max_wgrad_ws(B) // 4 + 4, dtype=torch.float32, device=dev)
        self.plans[B] = P
        return P

    # -- heuristics for the MFMA kernels --------------------------------------------------------------
    def _mfma_ok(self, cg, cd, lh):
        return self.use_mfma and cg % 32 == 0 and cd % 32 == 0 and (lh & (lh - 1)) == 0

    def _splitk(self, op, B, lh, cg, cd):
        ntaps = 16 if op == L.OP_G else 4
        ncols = cd if op == L.OP_G else cg
        cc = cg if op == L.OP_G else cd
        esz = 2 if self.dtype == L.BF16 else 4
        bn = 128 if ncols % 128 == 0 else (64 if ncols % 64 == 0 else 32)
        blocks = ((B * lh * lh + 127) // 128) * (ncols // bn) * (1 if op == L.OP_G else 4)
        sk = 1
        while blocks * sk < 256 and sk * 2 <= ntaps and ((ntaps // (sk * 2)) * cc * esz) % 128 == 0:
            sk *= 2
        return sk

    def _msplit(self, B, lh, cg, cd):
        bg = 128 if cg % 128 == 0 else (64 if cg % 64 == 0 else 32)
        tiles = 16 * (cg // bg) * (cd // 128)
        m = B * lh * lh
        ms = 1
        while tiles * ms < 512 and m // (ms * 2) >= 256:
            ms *= 2
        return ms

    def _max_slab_elems(self, B):
        S, best = self.S, 4
        for i in range(2, 7):
            lh = S // 2 ** i
            cg, cd = DOWN_FILTERS[i - 2], DOWN_FILTERS[i - 1]
            best = max(best, self._splitk(L.OP_G, B, lh, cg, cd) * B * lh * lh * cd)
            best = max(best, self._splitk(L.OP_P, B, lh, cg, cd) * B * 4 * lh * lh * cg)
        cin = 512
        skips = list(reversed(DOWN_FILTERS[:-1])) + [self.in_ch]
        for i in range(1, 7):
            lh = S // 64 * 2 ** (i - 1)
            cg, cd = UP_FILTERS[i - 1], cin
            best = max(best, self._splitk(L.OP_P, B, lh, cg, cd) * B * 4 * lh * lh * cg)
            best = max(best, self._splitk(L.OP_G, B, lh, cg, cd) * B * lh * lh * cd)
            cin = cg + skips[i - 1]
        return best

    def _max_wgrad_ws(self, B):
        S, best = self.S, 16
        for i in range(2, 7):
            lh = S // 2 ** i
            cg, cd = DOWN_FILTERS[i - 2], DOWN_FILTERS[i - 1]
            best = max(best, self._msplit(B, lh, cg, cd) * 16 * cg * cd * 4)
        cin = 512
        skips = list(reversed(DOWN_FILTERS[:-1])) + [self.in_ch]
        for i in range(1, 7):
            lh = S // 64 * 2 ** (i - 1)
            cg, cd = UP_FILTERS[i - 1], cin
            best = max(best, self._msplit(B, lh, cg, cd) * 16 * cg * cd * 4)
            cin = cg + skips[i - 1]
        return best

    # ------------------------------------------------------------------ kernel wrappers
    def _conv(self, P, op, sid, name, N, lh, cg, cd, hi, lo, out_dense, stride=2, bias=None):
        """op G or P.  Returns (raw_kind, nslabs) describing where the result went: the dense output buffer
        in the activation dtype (1, 1) or f32 split-K slabs in P['slabs'] (2, nslabs)."""
        if stride == 2 and bias is None and self._mfma_ok(cg, cd, lh):
            sk = self._splitk(op, N, lh, cg, cd)
            w = self.wt(sid, name) if op == L.OP_G else self.wn(sid, name)
            L.call("p2p_igemm", op, self.dtype, N, lh, lh, cg, cd, C.byref(hi), C.byref(lo), w, sk,
                   _p(P["slabs"]) if sk > 1 else NULL, _stream())
            return (1, 1) if sk == 1 else (2, sk)
        L.call("p2p_conv_direct", op, stride, self.dtype, N, lh, lh, cg, cd, C.byref(hi), C.byref(lo),
               self.wn(sid, name), bias if bias is not None else NULL, NULL, NULL, _stream())
        return (1, 1)

    def _wgrad(self, P, sid, name, N, lh, cg, cd, hi, lo, stride=2, dbias=None):
        store = self.G if sid == "G" else self.D
        dw = store.g(name + ".kernel")
        if stride == 2 and dbias is None and self._mfma_ok(cg, cd, lh) and cd % 128 == 0:
            ms = self._msplit(N, lh, cg, cd)
            L.call("p2p_wgemm", self.dtype, N, lh, lh, cg, cd, C.byref(hi), C.byref(lo), dw, ms,
                   _p(P["wws"]) if ms > 1 else NULL, _stream())
        else:
            L.call("p2p_conv_direct", L.OP_W, stride, self.dtype, N, lh, lh, cg, cd, C.byref(hi), C.byref(lo),
                   NULL, NULL, dw, dbias if dbias is not None else NULL, _stream())

    def _norm_fwd(self, P, N, res, c, raw_buf, rk, gamma, beta, act, mask, out_view, stats):
        raw_kind, nslabs = rk
        raw = raw_buf.ptr() if raw_kind == 1 else _p(P["slabs"])
        slab = N * res * res * c
        L.call("p2p_norm_act_fwd", self.dtype, N, res, res, c, raw, raw_kind, nslabs, slab,
               gamma if gamma is not None else NULL, beta if beta is not None else NULL, IN_EPS, act, LEAKY_ALPHA,
               _p(mask) if mask is not None else NULL, C.byref(out_view),
               raw_buf.ptr() if raw_kind == 2 else NULL, _p(stats) if stats is not None else NULL, _stream())

    def _gs(self, P, buf, rk, coff=0):
        """gradient source for a conv result that went to `buf` (kind 1) or to the split-K slabs (kind 2)."""
        if rk[0] == 1:
            return buf.gsrc(coff=coff, kind=1)
        return L.GSrc(P["slabs"].data_ptr(), 2, rk[1], buf.n * buf.h * buf.w * buf.c, buf.c, coff)

    def _norm_bwd(self, P, store_name, N, res, c, raw_buf, stats, act, mask, g1, g2, draw_view, norm=True):
        part = P["part"]
        gam = self.G.p(store_name + ".gamma") if norm else NULL
        bet = self.G.p(store_name + ".beta") if norm else NULL
        L.call("p2p_norm_act_bwd", self.dtype, N, res, res, c, raw_buf.ptr(), _p(stats) if norm else NULL, gam, bet,
               act, LEAKY_ALPHA, _p(mask) if mask is not None else NULL, C.byref(g1),
               C.byref(g2) if g2 is not None else None, C.byref(draw_view),
               _p(part[1]) if norm else NULL, _p(part[0]) if norm else NULL, _stream())
        if norm:
            # batch reduction of the per-image partials (dense [N][c] at the start of each scratch plane)
            # into the flat gradient buffer: dgamma/dbeta sum over batch AND space
            L.call("p2p_colsum", _p(part[1]), N, c, 1.0, self.G.g(store_name + ".gamma"), _stream())
            L.call("p2p_colsum", _p(part[0]), N, c, 1.0, self.G.g(store_name + ".beta"), _stream())

    # ------------------------------------------------------------------ forward
    def _to_device(self, arr, c, B, is_int=False):
        """batch element as the reference hands it over (dataset_utils.py:209-246): dense NHWC f32 or i32."""
        t = torch.as_tensor(arr)
        t = t.to(device=self.device, dtype=torch.int32 if is_int else torch.float32).contiguous()
        if tuple(t.shape) != (B, self.S, self.S, c):
            raise ValueError(f"expected batch of shape {(B, self.S, self.S, c)}, got {tuple(t.shape)}")
        return t

    def _pack(self, P, t, view, c):
        """dense f32 / i32 device batch -> activation-dtype (haloed, channel-sliced) view."""
        is_int = t.dtype == torch.int32
        L.call("p2p_pack_input", self.dtype, P["B"], self.S, self.S, c, _p(t), 1 if is_int else 0, C.byref(view), _stream())

    def generator_forward(self, P, masks=None):
        """UnetGenerator forward up to the pre-activation head output z (networks.py:80-98)."""
        B, S = P["B"], self.S
        c = P["c"]
        # down path
        src_view, cin = c[6].view(coff=UP_FILTERS[5]), self.in_ch
        P["rk_d"], P["rk_u"] = {}, {}
        for i, f in enumerate(DOWN_FILTERS, start=1):
            res = S // 2 ** i
            rk = self._conv(P, L.OP_G, "G", f"down{i}", B, res, cin, f, src_view, P["rd"][i].view(), P["rd"][i])
            P["rk_d"][i] = rk
            out_view = P["a6"].view() if i == 6 else c[6 - i].view(coff=UP_FILTERS[5 - i])
            if i == 1:
                self._norm_fwd(P, B, res, f, P["rd"][i], rk, None, None, L.ACT_LEAKY, None, out_view, None)
            else:
                self._norm_fwd(P, B, res, f, P["rd"][i], rk, self.G.p(f"down{i}.gamma"), self.G.p(f"down{i}.beta"),
                               L.ACT_LEAKY, None, out_view, P["sd"][i])
            src_view, cin = out_view, f
        # up path
        lo_view, cin = P["a6"].view(), 512
        for i, f in enumerate(UP_FILTERS, start=1):
            lh = S // 64 * 2 ** (i - 1)
            rk = self._conv(P, L.OP_P, "G", f"up{i}", B, lh, f, cin, P["ru"][i].view(), lo_view, P["ru"][i])
            P["rk_u"][i] = rk
            mask = None
            if UP_DROPOUT[i - 1]:
                mask = P["mask"][i]
                if masks is not None:
                    mask.copy_(torch.as_tensor(masks[i - 1]).reshape(mask.shape).to(torch.uint8))
                else:       # Bernoulli(0.5) keep mask (networks.py:31-32), counter-based device RNG
                    self.mask_counter += 1
                    L.call("p2p_dropout_mask", _p(mask), mask.numel(), self.seed, self.mask_counter, _stream())
            self._norm_fwd(P, B, 2 * lh, f, P["ru"][i], rk, self.G.p(f"up{i}.gamma"), self.G.p(f"up{i}.beta"),
                           L.ACT_RELU, mask, c[i].view(coff=0), P["su"][i])
            lo_view, cin = c[i].view(), c[i].c
        # head: Conv2D(out, 4, stride 1, SAME, bias) (networks.py:75-78)
        self._conv(P, L.OP_G, "G", "last", B, S, cin, self.out_ch, c[6].view(), P["z"].view(), P["z"], stride=1,
                   bias=self.G.p("last.bias"))

    def discriminator_forward(self, P, N2):
        """PatchDiscriminator on the first N2 images of dcat (networks.py:45-48)."""
        S, ic = self.S, self.in_ch
        self._conv(P, L.OP_G, "D", "down", N2, S // 2, 2 * ic, 64, P["dcat"].view(), P["d_raw"].view(), P["d_raw"])
        self._norm_fwd(P, N2, S // 2, 64, P["d_raw"], (1, 1), None, None, L.ACT_LEAKY, None, P["d_act"].view(), None)
        self._conv(P, L.OP_G, "D", "last", N2, S // 2, 64, 1, P["d_act"].view(), P["logits"].view(), P["logits"],
                   stride=1, bias=self.D.p("last.bias"))

    # ------------------------------------------------------------------ train step (RGBA models)
    def train_step_rgba(self, source, real, lambda_l1, lambda_hist=None, masks=None, global_batch=None,
                        apply_update=True, allreduce=None):
        """Pix2PixModel.train_step / Pix2PixHistogramModel (pix2pix_model.py:62-89,242-250).
        Returns a device tensor [g_total, g_adv, g_l1, g_hist, d_total, d_real, d_fake] (f32)."""
        B = int(source.shape[0])
        P = self.plan(B)
        S, ic = self.S, self.in_ch
        Bg = global_batch or B
        c = P["c"]
        src_t, real_t = self._to_device(source, ic, B), self._to_device(real, ic, B)
        self._pack(P, src_t, c[6].view(coff=UP_FILTERS[5]), ic)
        self._pack(P, src_t, P["dcat"].view(coff=ic), ic)
        self._pack(P, src_t, P["dcat"].view(coff=ic, n0=B), ic)
        self._pack(P, real_t, P["dcat"].view(coff=0), ic)
        self.generator_forward(P, masks)
        real_view, fake_view = P["dcat"].view(coff=0), P["dcat"].view(coff=0, n0=B)
        inv_l1 = 1.0 / (Bg * S * S * self.out_ch)
        L.call("p2p_tanh_l1_fwd", self.dtype, B, S, S, self.out_ch, C.byref(P["z"].view()), C.byref(real_view),
               C.byref(fake_view), inv_l1, _p(self.losses, 3), _stream())
        self.discriminator_forward(P, 2 * B)
        inv_bce = 1.0 / (Bg * (S // 2) * (S // 2))
        L.call("p2p_bce_logits", self.dtype, 2 * B, B, S // 2, S // 2, C.byref(P["logits"].view()), inv_bce,
               C.byref(P["dld"].view()), C.byref(P["dlg"].view()), _p(self.losses, 0), _stream())
        g_extra = None
        if lambda_hist is not None:
            g_extra = self._histogram_loss(P, B, Bg, lambda_hist, allreduce)
        # ---- discriminator gradients (pix2pix_model.py:79)
        h2 = S // 2
        self._wgrad(P, "D", "last", 2 * B, h2, 64, 1, P["d_act"].view(), P["dld"].view(), stride=1,
                    dbias=self.D.g("last.bias"))
        self._conv(P, L.OP_P, "D", "last", 2 * B, h2, 64, 1, P["g_dact"].view(), P["dld"].view(), P["g_dact"], stride=1)
        L.call("p2p_norm_act_bwd", self.dtype, 2 * B, h2, h2, 64, P["d_raw"].ptr(), NULL, NULL, NULL, L.ACT_LEAKY,
               LEAKY_ALPHA, NULL, C.byref(P["g_dact"].gsrc()), None, C.byref(P["d_draw"].view()), NULL, NULL, _stream())
        self._wgrad(P, "D", "down", 2 * B, h2, 2 * ic, 64, P["dcat"].view(), P["d_draw"].view())
        # ---- generator gradients through D (pix2pix_model.py:78; D weights pre-update)
        self._conv(P, L.OP_P, "D", "last", B, h2, 64, 1, P["g_dact"].view(), P["dlg"].view(), P["g_dact"], stride=1)
        L.call("p2p_norm_act_bwd", self.dtype, B, h2, h2, 64, P["d_raw"].ptr(n0=B), NULL, NULL, NULL, L.ACT_LEAKY,
               LEAKY_ALPHA, NULL, C.byref(P["g_dact"].gsrc()), None, C.byref(P["d_draw"].view()), NULL, NULL, _stream())
        self._conv(P, L.OP_P, "D", "down", B, h2, 2 * ic, 64, P["g_dcat"].view(), P["d_draw"].view(), P["g_dcat"])
        L.call("p2p_tanh_l1_bwd", self.dtype, B, S, S, self.out_ch, C.byref(fake_view), C.byref(real_view),
               C.byref(P["g_dcat"].gsrc()), C.byref(g_extra) if g_extra is not None else None,
               float(lambda_l1) * inv_l1, C.byref(P["dz"].view()), _stream())
        self.generator_backward(P)
        return self._finish_step(P, lambda_l1, lambda_hist, apply_update, allreduce)

    def generator_backward(self, P):
        """Backward of UnetGenerator from dz (the gradient at the head's pre-activation)."""
        B, S = P["B"], self.S
        c, gc, ga = P["c"], P["gc"], P["ga"]
        cin6 = c[6].c
        self._wgrad(P, "G", "last", B, S, cin6, self.out_ch, c[6].view(), P["dz"].view(), stride=1,
                    dbias=self.G.g("last.bias"))
        self._conv(P, L.OP_P, "G", "last", B, S, cin6, self.out_ch, gc[6].view(), P["dz"].view(), gc[6], stride=1)
        rk_gc = {6: (1, 1)}
        # up path, last to first
        for i in range(6, 0, -1):
            f = UP_FILTERS[i - 1]
            lh = S // 64 * 2 ** (i - 1)
            lo_buf = c[i - 1] if i > 1 else P["a6"]
            cin = lo_buf.c
            mask = P["mask"].get(i)
            self._norm_bwd(P, f"up{i}", B, 2 * lh, f, P["ru"][i], P["su"][i], L.ACT_RELU, mask,
                           self._gs(P, gc[i], rk_gc[i], 0), None, P["du"][i].view())
            self._wgrad(P, "G", f"up{i}", B, lh, f, cin, P["du"][i].view(), lo_buf.view())
            if i > 1:
                # d(concat_{i-1}) must outlive the split-K slabs: finalise into the dense buffer when needed
                rk = self._conv(P, L.OP_G, "G", f"up{i}", B, lh, f, cin, P["du"][i].view(), gc[i - 1].view(), gc[i - 1])
                rk_gc[i - 1] = self._materialise(P, gc[i - 1], rk)
            else:
                rk = self._conv(P, L.OP_G, "G", "up1", B, lh, f, cin, P["du"][1].view(), ga[6].view(), ga[6])
                rk_ga6 = self._materialise(P, ga[6], rk)
        # down path, last to first
        g_from_down = self._gs(P, ga[6], rk_ga6, 0)
        for i in range(6, 0, -1):
            f = DOWN_FILTERS[i - 1]
            res = S // 2 ** i
            cin = DOWN_FILTERS[i - 2] if i > 1 else self.in_ch
            if i == 6:
                g1, g2 = g_from_down, None
            else:
                g1 = self._gs(P, gc[6 - i], rk_gc[6 - i], UP_FILTERS[5 - i])     # skip-connection slice
                g2 = g_from_down
            if i > 1:
                self._norm_bwd(P, f"down{i}", B, res, f, P["rd"][i], P["sd"][i], L.ACT_LEAKY, None, g1, g2,
                               P["dd"][i].view())
            else:
                L.call("p2p_norm_act_bwd", self.dtype, B, res, res, f, P["rd"][1].ptr(), NULL, NULL, NULL, L.ACT_LEAKY,
                       LEAKY_ALPHA, NULL, C.byref(g1), C.byref(g2), C.byref(P["dd"][1].view()), NULL, NULL, _stream())
            hi_view = c[6].view(coff=UP_FILTERS[5]) if i == 1 else (c[7 - i].view(coff=UP_FILTERS[6 - i]))
            self._wgrad(P, "G", f"down{i}", B, res, cin, f, hi_view, P["dd"][i].view())
            if i > 1:
                rk = self._conv(P, L.OP_P, "G", f"down{i}", B, res, cin, f, ga[i - 1].view(), P["dd"][i].view(), ga[i - 1])
                g_from_down = self._gs(P, ga[i - 1], self._materialise(P, ga[i - 1], rk), 0)

    def _materialise(self, P, buf, rk):
        """Split-K slabs are a single shared workspace: sum them into `buf` (activation dtype) right away so
        a later conv may reuse the workspace.  (Uses the norm kernel in pass-through mode.)"""
        if rk[0] == 1:
            return rk
        L.call("p2p_norm_act_fwd", self.dtype, buf.n, buf.h, buf.w, buf.c, _p(P["slabs"]), 2, rk[1],
               buf.n * buf.h * buf.w * buf.c, NULL, NULL, IN_EPS, L.ACT_NONE, 0.0, NULL, C.byref(buf.view()),
               NULL, NULL, _stream())
        return (1, 1)

    def _finish_step(self, P, lambda_l1, lambda_hist, apply_update, allreduce):
        if allreduce is not None:
            allreduce(self.G.grads, self.D.grads, self.losses)
        if apply_update:
            self.apply_adam()
        l = self.losses
        hist = l[4] if lambda_hist is not None else torch.zeros((), device=self.device)
        g_adv, g_l1 = l[2], l[3]
        g_total = g_adv + float(lambda_l1) * g_l1 + (float(lambda_hist) * hist if lambda_hist is not None else 0.0)
        d_real, d_fake = l[0], l[1]
        self.step_count += 1
        return torch.stack([g_total, g_adv, g_l1, hist, d_real + d_fake, d_real, d_fake])

    def apply_adam(self):
        """Both optimizers step with gradients taken at the same pre-update weights (pix2pix_model.py:81-83)."""
        for store in (self.G, self.D):
            store.t += 1
            L.call("p2p_adam_flat", _p(store.params), _p(store.grads), _p(store.m), _p(store.v), store.numel,
                   store.t, self.lr, self.beta1, self.beta2, self.adam_eps, 1.0, _stream())
        self.refresh_weight_copies()

    def _histogram_loss(self, P, B, Bg, lambda_hist, allreduce):
        raise NotImplementedError("histogram loss kernels are wired in engine_hist")

    # ------------------------------------------------------------------ inference-style helpers
    def generate(self, source, masks=None):
        """generator(source, training=True) (pix2pix_model.py:58-60): f32 device tensor (B,S,S,out) for the tanh
        head; dropout stays on, as in the reference."""
        if self.head != "tanh":
            raise NotImplementedError("use generate_indexed for the softmax head")
        B = int(source.shape[0])
        P = self.plan(B)
        S = self.S
        src_t = self._to_device(source, self.in_ch, B)
        self._pack(P, src_t, P["c"][6].view(coff=UP_FILTERS[5]), self.in_ch)
        self.generator_forward(P, masks)
        fake_view = P["dcat"].view(coff=0, n0=B)
        # tanh through the loss kernel (its L1 output lands in a scratch slot and is ignored)
        L.call("p2p_tanh_l1_fwd", self.dtype, B, S, S, self.out_ch, C.byref(P["z"].view()), C.byref(fake_view),
               C.byref(fake_view), 0.0, _p(self.losses, 15), _stream())
        out = torch.empty((B, S, S, self.out_ch), dtype=torch.float32, device=self.device)
        L.call("p2p_unpack", self.dtype, B, S, S, self.out_ch, C.byref(fake_view), _p(out), _stream())
        return out
