"""Algorithmic work of the train step: what bench.py prices the kernels against."""
from .engine import DOWN_FILTERS, UP_FILTERS

__all__ = ["layer_macs", "train_step_flops_per_image", "call_work", "entry_roofline", "roofline_for_dominant", "rooflines_top"]


def layer_macs(S, in_ch, out_ch):
    """Forward MACs per image of every conv layer: dict name -> MACs (networks.py:39-98 shapes)."""
    macs = {}
    c = in_ch
    for i, f in enumerate(DOWN_FILTERS, start=1):
        res = S // 2 ** i
        macs[f"down{i}"] = res * res * 16 * c * f
        c = f
    skips = list(reversed(DOWN_FILTERS[:-1])) + [in_ch]
    for i, (f, s) in enumerate(zip(UP_FILTERS, skips), start=1):
        lh = S // 64 * 2 ** (i - 1)
        macs[f"up{i}"] = lh * lh * 16 * c * f
        c = f + s
    macs["last"] = S * S * 16 * c * out_ch
    macs["D.down"] = (S // 2) ** 2 * 16 * (2 * in_ch) * 64
    macs["D.last"] = (S // 2) ** 2 * 16 * 64
    return macs


def train_step_flops_per_image(S, in_ch=4, out_ch=4, indexed=False):
    """2 x [3*G_fwd - down1 + 5*D.down + 7*D.last] (RGBA) or 2 x [3*G_fwd - down1 + 4*D.down + 6*D.last] (indexed):
    every executed GEMM counted once (G: fwd + wgrad + dgrad, no dgrad for down1; D: see SURVEY.md 8a A8)."""
    m = layer_macs(S, in_ch, out_ch)
    g_fwd = sum(v for k, v in m.items() if not k.startswith("D."))
    if indexed:
        total = 3 * g_fwd - m["down1"] + 4 * m["D.down"] + 6 * m["D.last"]
    else:
        total = 3 * g_fwd - m["down1"] + 5 * m["D.down"] + 7 * m["D.last"]
    return 2 * total


HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md chip-level parameters (spec; 6.3 TB/s measured copy)
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}


def _ints(args, n):
    return [int(a) for a in args[:n]]


def _null(a):
    """a NULL pointer argument (None, or a ctypes pointer holding 0)"""
    return a is None or (hasattr(a, "value") and not a.value)


def _view_ld(byref_obj, default):
    """channel count of a pixel of a p2p_tensor passed with ctypes.byref"""
    t = getattr(byref_obj, "_obj", None)
    return int(getattr(t, "ld", default)) if t is not None else default


def _gsrc_bytes(byref_obj, elems, esz):
    """bytes read from a gradient source: activation-dtype tensor (kind 1) or `nslabs` f32 split-K slabs (kind 2)"""
    g = getattr(byref_obj, "_obj", None)
    if g is None:
        return 0.0
    return elems * (esz if int(g.kind) == 1 else 4 * int(g.nslabs))


def call_work(name, args, dtype):
    """Algorithmic work of ONE C-ABI call from its arguments (include/p2pgan.h): dict with
         flops      multiply-add work (2 x MACs, SURVEY.md 8a / 8d D2 conventions: every tap counted, zero padding included)
         mfma       "bf16" | "f32" | None: the matrix pipe those FLOPs run on (None: no dense contraction)
         bytes      algorithmic HBM bytes: every operand view read once, every result written once, in the stored layout
       or None for calls that move a few KB (loss finishing, counters)."""
    esz = 2 if dtype == "bf16" else 4
    mf = dtype
    if name in ("p2p_igemm", "p2p_igemm_norm_act", "p2p_conv_strip"):
        _, _, n, lh, lw, cg, cd = _ints(args, 7)
        fl = 2.0 * n * lh * lw * 16 * cg * cd
        by = (n * 4 * lh * lw * cg + n * lh * lw * cd + 16 * cg * cd) * esz
        if name == "p2p_igemm_norm_act":          # + the normalised activation written into its concat slice
            by += n * (lh * lw * cd if args[0] == 0 else 4 * lh * lw * cg) * esz
        return {"flops": fl, "mfma": mf, "bytes": by}
    if name in ("p2p_igemm_edge", "p2p_conv_fewin", "p2p_conv_fewout", "p2p_conv_fewin_actbwd"):
        op, stride, _, n, lh, lw, cin, nc = _ints(args, 8)
        fl = 2.0 * n * lh * lw * 16 * cin * nc
        hi_px, lo_px = n * stride * stride * lh * lw, n * lh * lw
        in_px, out_px = (hi_px, lo_px) if op == 0 else (lo_px, hi_px)
        by = (in_px * cin + out_px * nc + 16 * cin * nc) * esz
        if name == "p2p_conv_fewin_actbwd":
            by += out_px * nc * esz                 # the forward activation that gates the gradient
        return {"flops": fl, "mfma": mf, "bytes": by}
    if name in ("p2p_wgemm", "p2p_wgemm_edge", "p2p_wgrad_small"):
        if name == "p2p_wgemm":
            _, n, lh, lw, cg, cd = _ints(args, 6)
            stride, hi_i = 2, 6
        else:
            a = _ints(args, 7)
            stride = a[1]
            n, lh, lw, cg, cd = a[2:7]
            hi_i = 7
        fl = 2.0 * n * lh * lw * 16 * cg * cd
        hi_ld, lo_ld = _view_ld(args[hi_i], cg), _view_ld(args[hi_i + 1], cd)
        # a view that is a channel slice of a wider concat buffer is read as its own channels only
        by = (n * stride * stride * lh * lw * min(hi_ld, max(cg, 8)) + n * lh * lw * min(lo_ld, max(cd, 8))) * esz + 16 * cg * cd * 4
        return {"flops": fl, "mfma": mf, "bytes": by}
    if name == "p2p_conv_direct":
        _, _, _, n, lh, lw, cg, cd = _ints(args, 8)
        return {"flops": 2.0 * n * lh * lw * 16 * cg * cd, "mfma": None, "bytes": 0.0}
    if name in ("p2p_norm_act_fwd", "p2p_norm_act_fwd_tail"):
        _, n, h, w, c = _ints(args, 5)
        raw_kind, nslabs = int(args[6]), int(args[7])
        t = n * h * w * c
        by = t * (esz if raw_kind == 1 else 4 * nslabs) + t * esz
        if raw_kind == 2:
            by += t * esz                           # the folded raw tensor is written for the backward pass
        if name == "p2p_norm_act_fwd_tail":
            by += 2.0 * n * h * w * int(args[22]) * esz          # the tail channels: read + written
        return {"flops": 0.0, "mfma": None, "bytes": by}
    if name == "p2p_norm_act_bwd":
        _, n, h, w, c = _ints(args, 5)
        t = n * h * w * c
        by = t * esz + _gsrc_bytes(args[12], t, esz) + (_gsrc_bytes(args[13], t, esz) if args[13] is not None else 0) + t * esz
        return {"flops": 0.0, "mfma": None, "bytes": by}
    if name == "p2p_act_bwd":
        _, n, h, w, c = _ints(args, 5)
        t = n * h * w * c
        by = 2 * t * esz + _gsrc_bytes(args[6], t, esz) + (_gsrc_bytes(args[7], t, esz) if args[7] is not None else 0)
        return {"flops": 0.0, "mfma": None, "bytes": by}
    if name == "p2p_rgbuv_hist_fwd":
        # histogram.py:29-30: three (64 x HW) . (HW x 64) contractions per image, exact-f32 MFMA (201.3 MFLOP
        # per image PAIR at S = 64); bytes: the RGBA image in, three 64 x 64 planes out
        _, n, h, w = _ints(args, 4)
        return {"flops": 3 * 2.0 * 64 * 64 * h * w * n, "mfma": "f32", "bytes": n * (h * w * 4 + 3 * 64 * 64) * 4.0}
    if name == "p2p_rgbuv_hist_fwd3":
        # the same contraction, all three components in one workgroup; with a colour-point list (args[5]) the contraction
        # runs over the image's distinct colours: no longer matrix work, priced by its bytes
        _, n, h, w = _ints(args, 4)
        listed = getattr(args[5], "value", None) not in (None, 0)
        by = n * (h * w * 4 + 3 * 64 * 64) * 4.0
        if listed:
            return {"flops": 0.0, "mfma": None, "bytes": by}
        # round 4: every f32 product is taken as SIX bf16 partial products (three-way split operands, csrc/hist.hip).  `flops` stays
        # ALGORITHMIC (what `achieved` / `frac` are computed from, one definition for every entry and every round -- ADVICE r04);
        # `executed_flops` = 6 x that is what the bf16 pipe really issues: it prices the launch's bound time and is reported
        # separately as `frac_executed`
        alg = 3 * 2.0 * 64 * 64 * h * w * n
        return {"flops": alg, "executed_flops": 6 * alg, "mfma": "bf16", "bytes": by}
    if name == "p2p_rgbuv_points":
        _, n, h, w = _ints(args, 4)
        return {"flops": 0.0, "mfma": None, "bytes": n * h * w * 4 * 4.0}
    if name == "p2p_rgbuv_hist_hellinger_bwd3":
        _, n, h, w = _ints(args, 4)
        alg = 2 * 3 * 2.0 * 64 * 64 * h * w * n          # (six bf16 partial products per f32 product, as in the forward)
        return {"flops": alg, "executed_flops": 6 * alg, "mfma": "bf16", "bytes": n * (h * w * 4 + 3 * 3 * 64 * 64 + h * w * 4) * 4.0}
    if name == "p2p_rgbuv_hist_hellinger_bwd":
        # closed-form backward: A = GH . kv and Bm = GH^T . ku per colour component = twice the forward
        _, n, h, w = _ints(args, 4)
        return {"flops": 2 * 3 * 2.0 * 64 * 64 * h * w * n, "mfma": "f32",
                "bytes": n * (h * w * 4 + 3 * 3 * 64 * 64 + 3 * h * w * 4) * 4.0}
    if name == "p2p_head_softmax_cce":
        _, n, h, w, cin, ncls = _ints(args, 6)
        return {"flops": 2.0 * n * h * w * 16 * cin * ncls, "mfma": mf,
                "bytes": n * h * w * (cin + ncls + 16.0) * esz + 16 * cin * ncls * esz}
    if name == "p2p_head_dgrad":
        _, n, h, w, ncls, cout = _ints(args, 6)
        return {"flops": 2.0 * n * h * w * 16 * ncls * cout, "mfma": mf, "bytes": n * h * w * (ncls + cout) * esz + 16.0 * ncls * cout * esz}
    if name == "p2p_softmax_cce_argmax":
        _, n, h, w, c = _ints(args, 5)
        return {"flops": 0.0, "mfma": None, "bytes": n * h * w * (2.0 * c * esz + 2 * 8 * esz)}
    if name == "p2p_adam_flat_dev":
        return {"flops": 0.0, "mfma": None, "bytes": 7 * 4.0 * int(args[4])}        # g, m, v, p in; m, v, p out
    if name == "p2p_adam_prep_batched":       # g, m, v, theta in; m, v, theta out; two operand copies out
        return {"flops": 0.0, "mfma": None, "bytes": (7 * 4.0 + 2 * esz) * int(args[1])}
    if name == "p2p_pack_pair":
        _, n, h, w = _ints(args, 4)
        ch = 8 + 8 + (0 if _null(args[7]) else 8) + (0 if _null(args[9]) else 4)
        return {"flops": 0.0, "mfma": None, "bytes": n * h * w * (2 * 4 * 4.0 + ch * esz)}
    if name == "p2p_pack_pair_idx":
        _, n, h, w = _ints(args, 4)
        return {"flops": 0.0, "mfma": None, "bytes": n * h * w * (2 * 4.0 + (24 + (0 if _null(args[7]) else 8)) * esz)}
    if name == "p2p_tanh_l1_fwd_pair":          # z (4 ch) + [target | source] in, [fake | source] out (+ the f32 copy of fake)
        _, n, h, w = _ints(args, 4)
        return {"flops": 0.0, "mfma": None, "bytes": n * h * w * ((4 + 8 + 8) * esz + (0.0 if _null(args[9]) else 16.0))}
    if name == "p2p_tanh_l1_bwd_pad8":
        _, n, h, w = _ints(args, 4)
        return {"flops": 0.0, "mfma": None, "bytes": n * h * w * 3.0 * 8 * esz}
    if name in ("p2p_tanh_l1_fwd", "p2p_tanh_l1_bwd"):
        _, n, h, w, c = _ints(args, 5)
        return {"flops": 0.0, "mfma": None, "bytes": n * h * w * 3.0 * max(c, 8) * esz}
    if name == "p2p_dropout_mask_dev":
        return {"flops": 0.0, "mfma": None, "bytes": float(int(args[1]))}
    return None


def _call_flops(name, args, dtype="bf16"):
    w = call_work(name, args, dtype)
    return w["flops"] if w and w["mfma"] else None


def entry_roofline(name, records, dtype):
    """Roofline of one entry point over the profiled launches.  Every LAUNCH is classified by the larger of (FLOPs / MFMA peak
    of the pipe it uses) and (algorithmic bytes / HBM peak) -- one entry point can hold both kinds (p2p_wgrad_small: 64 x 256
    channel layers are MFMA work, 36 -> 4 channel layers are streams).  Reported: the class that takes more of the entry's
    measured time (`bound`, `achieved` over the launches of that class, `peak`, `frac`), and `frac_all` = sum over all launches
    of their bound time / measured time.  An MFMA kernel is priced against HBM only where its bytes really bound it."""
    cls = {"mfma": [0.0, 0.0, 0.0, 0, None], "hbm": [0.0, 0.0, 0.0, 0, None]}      # work, bound seconds, measured ms, launches, pipe
    fl_all = ex_all = by_all = ms_all = 0.0
    n = 0
    for rname, args, a, b in records:
        if rname != name:
            continue
        w = call_work(rname, args, dtype)
        if w is None:
            continue
        ms = a.elapsed_time(b)
        t_mfma = w.get("executed_flops", w["flops"]) / (MFMA_PEAK_TFLOPS[w["mfma"]] * 1e12) if w["mfma"] else 0.0
        t_hbm = w["bytes"] / (HBM_PEAK_GBS * 1e9)
        c = cls["mfma"] if (t_mfma >= t_hbm and w["mfma"]) else cls["hbm"]
        c[0] += w["flops"] if c is cls["mfma"] else w["bytes"]
        c[1] += max(t_mfma, t_hbm)
        c[2] += ms
        c[3] += 1
        c[4] = c[4] or w["mfma"]
        fl_all += w["flops"] if w["mfma"] else 0.0
        ex_all += w.get("executed_flops", w["flops"]) if w["mfma"] else 0.0
        by_all += w["bytes"]
        ms_all += ms
        n += 1
    if n == 0 or ms_all <= 0:
        return None
    bound = "mfma" if cls["mfma"][2] >= cls["hbm"][2] else "hbm"
    work, tb, ms, k, pipe = cls[bound]
    out = {"kernel": name, "launches": n, "avg_launch_ms": round(ms_all / n, 5), "flops_per_launch_avg": fl_all / n,
           "algorithmic_bytes_per_launch_avg": by_all / n, "traffic": None,
           "frac_all": round((cls["mfma"][1] + cls["hbm"][1]) / (ms_all * 1e-3), 5), "launches_in_bound_class": k}
    # headline `achieved` / `frac`: ALL launches of the entry (algorithmic FLOPs or bytes of every launch over the entry's measured
    # time) -- the definition of rounds 1-2, comparable across rounds.  `frac_bound_class` is the r03 figure (launches of the
    # dominant class only); `frac_all` prices every launch against the roofline that bounds IT.
    if bound == "mfma":
        ach = fl_all / (ms_all * 1e-3) / 1e12
        ach_c = work / (ms * 1e-3) / 1e12
        out.update({"bound": "mfma", "mfma_dtype": pipe, "achieved": round(ach, 3), "peak": MFMA_PEAK_TFLOPS[pipe],
                    "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK_TFLOPS[pipe], 5),
                    "frac_bound_class": round(ach_c / MFMA_PEAK_TFLOPS[pipe], 5)})
        if ex_all > fl_all:     # split-operand kernels: the pipe's own utilisation (instructions issued), beside the algorithmic figure
            out["frac_executed"] = round(ex_all / (ms_all * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS[pipe], 5)
    else:
        ach = by_all / (ms_all * 1e-3) / 1e9
        ach_c = work / (ms * 1e-3) / 1e9
        out.update({"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 5), "frac_bound_class": round(ach_c / HBM_PEAK_GBS, 5)})
    return out


def roofline_for_dominant(prof, records, B, S, dtype, peak_tflops=None):
    """Roofline object of the entry point that takes the most device time per step."""
    for name, _ in sorted(prof.items(), key=lambda kv: -kv[1]["ms_per_step"]):
        r = entry_roofline(name, records, dtype)
        if r is not None:
            return r
    return {"kernel": None, "bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None}


def rooflines_top(prof, records, dtype, k=8):
    """the k entry points with the most device time, each with its own roofline (bench.py `rooflines`)"""
    out = []
    for name, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms_per_step"]):
        r = entry_roofline(name, records, dtype)
        if r is not None:
            r["ms_per_step"] = round(v["ms_per_step"], 4)
            for key in ("traffic", "flops_per_launch_avg", "algorithmic_bytes_per_launch_avg"):
                r.pop(key, None)
            out.append(r)
        if len(out) >= k:
            break
    return out


def write_detail(records, path, n_steps, dtype="bf16"):
    """Per-call table: entry point, leading integer arguments (shape), ms per launch, algorithmic TFLOP/s and GB/s."""
    agg, work = {}, {}
    for name, args, a, b in records:
        ints = tuple(x for x in args[:8] if isinstance(x, int))
        key = (name, ints)
        if key not in work:
            work[key] = call_work(name, args, dtype)
        d = agg.setdefault(key, [0.0, 0])
        d[0] += a.elapsed_time(b)
        d[1] += 1
    rows = sorted(agg.items(), key=lambda kv: -kv[1][0])
    with open(path, "w") as f:
        f.write("ms_per_step  launches  ms_each  TFLOP/s     GB/s  entry  int-args\n")
        for (name, ints), (ms, n) in rows:
            w = work.get((name, ints))
            tf = (w["flops"] * n / (ms * 1e-3) / 1e12) if w and w["mfma"] else 0.0
            gbs = (w["bytes"] * n / (ms * 1e-3) / 1e9) if w else 0.0
            f.write(f"{ms / n_steps:10.4f} {n / n_steps:8.1f} {ms / n:9.4f} {tf:8.1f} {gbs:8.0f}  {name} {ints}\n")
