"""Weight interchange with the TensorFlow/Keras reference (pix2pix_model.py:30-36, side2side_model.py:178-200).

FILE FORMAT (`*.p2pw.npz`, a plain numpy .npz -- writable from a TF process with numpy alone):

    format                   "p2pgan-keras-weights-1"
    generator/NNN:<name>     f32 array, NNN = position in `generator.get_weights()` (Keras variable order), <name> as below
    discriminator/NNN:<name>
    generator_optimizer/iterations            int64 scalar   (Adam step count t)           } optional: present when the
    generator_optimizer/m/NNN:<name>, .../v/NNN:<name>    first / second moment, same shape   } optimizer state is exported
    discriminator_optimizer/...               likewise

Variable ORDER = `keras.Model.get_weights()` of the reference's functional models (networks.py:39-98): the blocks in creation
order, inside a block the Conv kernel, then tfa InstanceNormalization's gamma, then beta; the heads contribute kernel, bias:

    generator      down1.kernel, down2.kernel, down2.gamma, down2.beta, ..., down6.beta,
                   up1.kernel, up1.gamma, up1.beta, ..., up6.beta, last.kernel, last.bias          (36 arrays)
    discriminator  down.kernel, last.kernel, last.bias                                               (3 arrays)

LAYOUTS = Keras' own, unchanged:
    Conv2D kernel           (kh, kw, Cin, Cout)   "HWIO"            down*.kernel, last.kernel, D down/last
    Conv2DTranspose kernel  (kh, kw, Cout, Cin)                     up*.kernel
    gamma / beta / bias     (C,)
Both kernel layouts are the engine's [tap][Cg][Cd] array (include/p2pgan.h: Cg = channels of the high-resolution side),
so no transposition happens on import or export.

From the reference side (TensorFlow process):
    np.savez("front2right.p2pw.npz", format="p2pgan-keras-weights-1",
             **{f"generator/{i:03d}:{n}": w for i, (n, w) in enumerate(zip(GENERATOR_NAMES, model.generator.get_weights()))},
             **{f"discriminator/{i:03d}:{n}": w for i, (n, w) in enumerate(zip(DISCRIMINATOR_NAMES, model.discriminator.get_weights()))})
and back: `model.generator.set_weights(load_weight_list(path, "generator"))`.
Keras Adam slot variables map as m <-> optimizer.get_slot(var, "m"), v <-> get_slot(var, "v"), iterations <-> optimizer.iterations.
"""
from collections import OrderedDict

import numpy as np

FORMAT = "p2pgan-keras-weights-1"


def variable_names(store):
    """names in Keras variable order (engine.ParamStore.shapes keeps that order)"""
    return list(store.shapes)


def _check(name, arr, shape):
    if tuple(arr.shape) != tuple(shape):
        raise ValueError(f"{name}: shape {tuple(arr.shape)} in the file, the model expects {tuple(shape)}")


GROUPS = ("generator", "discriminator")


def _groups(eng, which):
    which = GROUPS if which is None else tuple(which)
    for tag in which:
        if tag not in GROUPS:
            raise ValueError(f"unknown network '{tag}' (expected a subset of {GROUPS})")
    return [(tag, eng.G if tag == "generator" else eng.D) for tag in which]


def export_model(model, path, with_optimizer=True, which=None):
    """Writes the networks named in `which` (default: generator and discriminator) and, with `with_optimizer`, their Adam
    states of a Pix2Pix*Model / engine to `path` (.npz)."""
    eng = getattr(model, "engine", model)
    out = {"format": np.array(FORMAT)}
    for tag, store in _groups(eng, which):
        for i, (name, w) in enumerate(store.export().items()):
            out[f"{tag}/{i:03d}:{name}"] = w.astype(np.float32)
        if with_optimizer:
            out[f"{tag}_optimizer/iterations"] = np.array(store.t, np.int64)
            for slot, buf in (("m", store.m), ("v", store.v)):
                for i, (name, w) in enumerate(store.export(buf).items()):
                    out[f"{tag}_optimizer/{slot}/{i:03d}:{name}"] = w.astype(np.float32)
    np.savez(path, **out)
    return path


def _read_group(z, prefix, store):
    keys = sorted(k for k in z.files if k.startswith(prefix + "/") and k[len(prefix) + 1:len(prefix) + 4].isdigit())
    if not keys:
        return None
    names = variable_names(store)
    if len(keys) != len(names):
        raise ValueError(f"{prefix}: {len(keys)} arrays in the file, the model has {len(names)} variables")
    vals = OrderedDict()
    for k, name in zip(keys, names):
        fname = k.split(":", 1)[1] if ":" in k else name
        if fname != name:
            raise ValueError(f"{prefix}: position {k.split('/')[1][:3]} holds '{fname}', expected '{name}' (Keras variable order)")
        arr = np.asarray(z[k], np.float32)
        _check(f"{prefix}/{name}", arr, store.shapes[name])
        vals[name] = arr
    return vals


def import_model(model, path, with_optimizer=True, which=None):
    """Loads a file written by export_model (or by the reference-side snippet in the module docstring).  Only the networks
    named in `which` (default: both) are touched -- their weights and, with `with_optimizer`, their Adam state; a file may
    hold a single network as long as it holds every network asked for."""
    import torch
    eng = getattr(model, "engine", model)
    z = np.load(path, allow_pickle=False)
    if "format" in z.files and str(z["format"]) != FORMAT:
        raise ValueError(f"unknown weight file format {z['format']}")
    groups = _groups(eng, which)
    loaded = [(tag, store, _read_group(z, tag, store)) for tag, store in groups]
    for tag, _, vals in loaded:          # validate everything before the first byte of the model changes
        if vals is None:
            raise ValueError(f"{path}: no '{tag}/NNN:name' arrays")
    for tag, store, vals in loaded:
        store.load(vals)
        if with_optimizer and f"{tag}_optimizer/iterations" in z.files:
            store.t = int(z[f"{tag}_optimizer/iterations"])
            store.t_dev.fill_(store.t)
            for slot, buf in (("m", store.m), ("v", store.v)):
                sv = _read_group(z, f"{tag}_optimizer/{slot}", store)
                if sv is not None:
                    for k, a in sv.items():
                        store.view(buf, k).copy_(torch.as_tensor(a))
    eng.refresh_weight_copies()


def load_weight_list(path, tag):
    """list of arrays in Keras variable order for `keras.Model.set_weights` on the reference side"""
    z = np.load(path, allow_pickle=False)
    keys = sorted(k for k in z.files if k.startswith(tag + "/") and k[len(tag) + 1:len(tag) + 4].isdigit())
    return [np.asarray(z[k]) for k in keys]
