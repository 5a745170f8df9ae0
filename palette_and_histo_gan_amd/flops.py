"""Algorithmic work of the train step: what bench.py prices the kernels against."""
from .engine import DOWN_FILTERS, UP_FILTERS

__all__ = ["layer_macs", "train_step_flops_per_image", "roofline_for_dominant"]


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


def _call_flops(name, args):
    """Algorithmic FLOPs of one C-ABI conv call from its leading integer arguments."""
    if name == "p2p_igemm":
        _, _, n, lh, lw, cg, cd = args[:7]
    elif name == "p2p_wgemm":
        _, n, lh, lw, cg, cd = args[:6]
    elif name == "p2p_conv_direct":
        _, _, _, n, lh, lw, cg, cd = args[:8]
    else:
        return None
    return 2.0 * n * lh * lw * 16 * cg * cd


def roofline_for_dominant(prof, records, B, S, dtype, peak_tflops):
    """Roofline object for the entry point that takes the most device time per step.  For the conv entry
    points the bound is the MFMA peak and `achieved` = sum of algorithmic FLOPs of its launches / sum of their
    event-measured durations; `traffic` (HBM bytes from PMC counters) is filled from profiles/ by hand, not here."""
    dominant = max(prof.items(), key=lambda kv: kv[1]["ms_per_step"])[0]
    fl, ms, n = 0.0, 0.0, 0
    for name, args, a, b in records:
        if name != dominant:
            continue
        f = _call_flops(name, args)
        if f is None:
            continue
        fl += f
        ms += a.elapsed_time(b)
        n += 1
    if n == 0:
        return {"kernel": dominant, "bound": "hbm", "achieved": None, "peak": 8000.0, "unit": "GB/s", "frac": None,
                "traffic": None}
    achieved = fl / (ms * 1e-3) / 1e12
    return {"kernel": dominant, "bound": "mfma", "achieved": round(achieved, 3), "peak": peak_tflops, "unit": "TFLOP/s",
            "frac": round(achieved / peak_tflops, 5), "traffic": None, "launches": n,
            "avg_launch_ms": round(ms / n, 5), "flops_per_launch_avg": fl / n,
            "algorithmic_bytes_per_launch_avg": _dominant_bytes(dominant, records, dtype) / n}


def _dominant_bytes(dominant, records, dtype):
    """Algorithmic HBM bytes of the conv launches: input view + weights + output, each touched once."""
    esz = 2 if dtype == "bf16" else 4
    tot = 0.0
    for name, args, a, b in records:
        if name != dominant:
            continue
        if name == "p2p_igemm":
            _, _, n, lh, lw, cg, cd = args[:7]
        elif name == "p2p_wgemm":
            _, n, lh, lw, cg, cd = args[:6]
        else:
            continue
        tot += (n * 4 * lh * lw * cg + n * lh * lw * cd + 16 * cg * cd) * esz
    return tot


def write_detail(records, path, n_steps):
    """Per-call table: entry point, leading integer arguments (shape), ms per launch, algorithmic TFLOP/s."""
    agg = {}
    for name, args, a, b in records:
        ints = tuple(x for x in args if isinstance(x, int))
        key = (name, ints)
        d = agg.setdefault(key, [0.0, 0])
        d[0] += a.elapsed_time(b)
        d[1] += 1
    rows = sorted(agg.items(), key=lambda kv: -kv[1][0])
    with open(path, "w") as f:
        f.write("ms_per_step  launches  ms_each  TFLOP/s  entry  int-args\n")
        for (name, ints), (ms, n) in rows:
            fl = _call_flops(name, ints)
            tf = (fl * n / (ms * 1e-3) / 1e12) if fl else 0.0
            f.write(f"{ms / n_steps:10.4f} {n / n_steps:8.1f} {ms / n:9.4f} {tf:8.1f}  {name} {ints}\n")
