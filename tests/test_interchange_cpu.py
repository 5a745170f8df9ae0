"""Not-gpu: host-side pieces around the hot path -- the Keras-layout weight file, the palette helpers of
io_utils.py (bit-exact integer work, F1/F3) and the TensorBoard event writer (F4)."""
import numpy as np
import pytest
import torch

from palette_and_histo_gan_amd import engine as E
from palette_and_histo_gan_amd import io_utils
from palette_and_histo_gan_amd import keras_weights as KW
from palette_and_histo_gan_amd import tb_events
from palette_and_histo_gan_amd.configuration import INVALID_INDEX_COLOR, MAX_PALETTE_SIZE


class _HostEngine:
    """the two parameter stores of an engine, on the CPU (keras_weights only touches G, D and refresh_weight_copies)"""

    def __init__(self, in_ch=4, out_ch=4, seed=0):
        self.G = E.ParamStore(E.generator_param_shapes(in_ch, out_ch), "cpu")
        self.D = E.ParamStore(E.discriminator_param_shapes(in_ch), "cpu")
        rng = np.random.default_rng(seed)
        for st in (self.G, self.D):
            st.params.copy_(torch.as_tensor(rng.normal(size=st.numel).astype(np.float32)))
            st.m.copy_(torch.as_tensor(rng.normal(size=st.numel).astype(np.float32)))
            st.v.copy_(torch.as_tensor(rng.random(size=st.numel).astype(np.float32)))
            st.t = int(rng.integers(1, 1000))
        self.refreshed = 0

    def refresh_weight_copies(self):
        self.refreshed += 1


def test_keras_layout_weight_file_round_trip(tmp_path):
    a, b = _HostEngine(seed=1), _HostEngine(seed=2)
    path = KW.export_model(a, str(tmp_path / "front2right.p2pw.npz"))
    z = np.load(path)
    # Keras variable order and layouts (networks.py:39-98): 36 generator arrays, 3 discriminator arrays
    gen = sorted(k for k in z.files if k.startswith("generator/"))
    assert len(gen) == 36 and gen[0] == "generator/000:down1.kernel" and gen[-1] == "generator/035:last.bias"
    assert z["generator/000:down1.kernel"].shape == (4, 4, 4, 64)            # Conv2D: HWIO
    assert z["generator/016:up1.kernel"].shape == (4, 4, 512, 512)           # Conv2DTranspose: (kh, kw, Cout, Cin)
    assert z["generator/031:up6.kernel"].shape == (4, 4, 32, 128)
    assert z["generator/034:last.kernel"].shape == (4, 4, 36, 4)
    assert sorted(k for k in z.files if k.startswith("discriminator/")) == [
        "discriminator/000:down.kernel", "discriminator/001:last.kernel", "discriminator/002:last.bias"]
    assert sum(int(np.prod(z[k].shape)) for k in gen) == 29_307_844
    KW.import_model(b, path)
    for sa, sb in ((a.G, b.G), (a.D, b.D)):
        for k in sa.shapes:         # per-variable: the flat buffers also hold alignment padding
            for buf in ("params", "m", "v"):
                assert torch.equal(sa.view(getattr(sa, buf), k), sb.view(getattr(sb, buf), k)), (k, buf)
        assert sa.t == sb.t and int(sb.t_dev[0]) == sb.t
    assert b.refreshed == 1
    # the reference side reads the same file as a plain list for keras.Model.set_weights
    lst = KW.load_weight_list(path, "generator")
    assert len(lst) == 36 and np.array_equal(lst[1], a.G.export()["down2.kernel"])
    # a file whose variables are out of order is refused
    bad = {k: z[k] for k in z.files}
    bad["generator/001:down2.gamma"] = bad.pop("generator/001:down2.kernel")
    np.savez(str(tmp_path / "bad.npz"), **bad)
    with pytest.raises(ValueError):
        KW.import_model(b, str(tmp_path / "bad.npz"))


def test_loading_one_network_leaves_the_other_and_both_optimizers_untouched(tmp_path):
    """side2side_model.py:186-200 replace ONE Keras model; ADVICE r02: load_generator must not clobber D or the Adam states"""
    a, b = _HostEngine(seed=3), _HostEngine(seed=4)
    before = {(sid, buf): getattr(st, buf).clone() for sid, st in (("G", b.G), ("D", b.D)) for buf in ("params", "m", "v")}
    t_before = (b.G.t, b.D.t)
    path = KW.export_model(a, str(tmp_path / "gen.p2pw.npz"), with_optimizer=False, which=("generator",))
    z = np.load(path)
    assert not any(k.startswith(("discriminator", "generator_optimizer")) for k in z.files)
    KW.import_model(b, path, with_optimizer=False, which=("generator",))
    for k in a.G.shapes:
        assert torch.equal(a.G.view(a.G.params, k), b.G.view(b.G.params, k)), k
    assert torch.equal(b.D.params, before[("D", "params")])
    for sid, st in (("G", b.G), ("D", b.D)):
        for buf in ("m", "v"):
            assert torch.equal(getattr(st, buf), before[(sid, buf)]), (sid, buf)
    assert (b.G.t, b.D.t) == t_before
    # a generator-only file cannot serve a request for both networks, and nothing changes when it is refused
    snap = b.G.params.clone()
    with pytest.raises(ValueError):
        KW.import_model(b, path)
    assert torch.equal(b.G.params, snap)
    # ... and a two-network file can serve a one-network request
    both = KW.export_model(a, str(tmp_path / "both.p2pw.npz"))
    KW.import_model(b, both, with_optimizer=False, which=("discriminator",))
    assert torch.equal(a.D.view(a.D.params, "down.kernel"), b.D.view(b.D.params, "down.kernel"))
    assert torch.equal(b.D.m, before[("D", "m")])
    with pytest.raises(ValueError):
        KW.import_model(b, both, which=("critic",))


def test_event_records_parse_with_an_independent_protobuf_runtime(tmp_path):
    """the event file was only ever read by its own reader.  Here the serialized `Event` messages are
    parsed by google.protobuf from a schema declared field by field (tensorflow/core/util/event.proto and
    framework/summary.proto: Event.wall_time = 1 double, step = 2 int64, file_version = 3 string, summary = 5; Summary.value =
    1 repeated; Value.tag = 1 string, simple_value = 2 float), the TFRecord framing is taken apart by hand and its
    CRC-32C (Castagnoli) is checked against the algorithm's published check values."""
    pytest.importorskip("google.protobuf")
    import struct
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    fd = descriptor_pb2.FileDescriptorProto(name="p2p_event.proto", package="p2ptest", syntax="proto3")
    val = fd.message_type.add(name="Value")
    val.field.add(name="tag", number=1, type=descriptor_pb2.FieldDescriptorProto.TYPE_STRING, label=1)
    val.field.add(name="simple_value", number=2, type=descriptor_pb2.FieldDescriptorProto.TYPE_FLOAT, label=1)
    summ = fd.message_type.add(name="Summary")
    summ.field.add(name="value", number=1, type=descriptor_pb2.FieldDescriptorProto.TYPE_MESSAGE, label=3, type_name=".p2ptest.Value")
    ev = fd.message_type.add(name="Event")
    ev.field.add(name="wall_time", number=1, type=descriptor_pb2.FieldDescriptorProto.TYPE_DOUBLE, label=1)
    ev.field.add(name="step", number=2, type=descriptor_pb2.FieldDescriptorProto.TYPE_INT64, label=1)
    ev.field.add(name="file_version", number=3, type=descriptor_pb2.FieldDescriptorProto.TYPE_STRING, label=1)
    ev.field.add(name="summary", number=5, type=descriptor_pb2.FieldDescriptorProto.TYPE_MESSAGE, label=1, type_name=".p2ptest.Summary")
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    Event = message_factory.GetMessageClass(pool.FindMessageTypeByName("p2ptest.Event"))
    # CRC-32C known answers (RFC 3720 B.4 / the iSCSI check value)
    assert tb_events.crc32c(b"123456789") == 0xE3069283
    assert tb_events.crc32c(bytes(32)) == 0x8A9136AA and tb_events.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43
    w = tb_events.EventFileWriter(str(tmp_path))
    rows = [("generator/total_loss", 1.5, 0, 100.25), ("l1-evaluation/test", 0.0625, 7, 101.5), ("discriminator/fake_loss", -2.0, 2 ** 40, 102.0)]
    w.add_scalars(rows)
    data = open(w.path, "rb").read()
    pos, events = 0, []
    while pos < len(data):
        (ln,) = struct.unpack_from("<Q", data, pos)
        body = data[pos + 12:pos + 12 + ln]
        for blob, at in ((data[pos:pos + 8], pos + 8), (body, pos + 12 + ln)):       # masked crc: rotate right 15, add the constant
            c = tb_events.crc32c(blob)
            assert struct.unpack_from("<I", data, at)[0] == ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF
        m = Event()
        m.ParseFromString(body)
        events.append(m)
        pos += 16 + ln
    assert events[0].file_version == "brain.Event:2" and len(events) == 1 + len(rows)
    for m, (tag, value, step, wall) in zip(events[1:], rows):
        assert m.step == step and m.wall_time == wall and len(m.summary.value) == 1
        assert m.summary.value[0].tag == tag and m.summary.value[0].simple_value == value
        if step:        # (proto3 omits a zero step on re-serialisation; the writer emits it explicitly -- both decode to 0)
            assert m.SerializeToString() == tb_events.encode_event(wall, step=step, scalars=[(tag, value)])     # canonical encoding


def test_image_and_layout_records_parse_with_an_independent_protobuf_runtime(tmp_path):
    """F4: the per-update image summary and the custom-scalar layout (side2side_model.py:58-61,86-93,240-273) as google.protobuf
    reads them from a schema declared field by field -- tensorflow/core/framework/{summary,tensor,tensor_shape}.proto and
    tensorboard/plugins/custom_scalar/layout.proto: Value.tensor = 8, Value.metadata = 9, SummaryMetadata.plugin_data = 1
    {plugin_name = 1, content = 2}, data_class = 4, TensorProto.dtype = 1, tensor_shape = 2 {dim = 2 {size = 1}}, string_val = 8;
    Layout.category = 2 {title = 1, chart = 2 {title = 1, multiline = 2 {tag = 1}}}.  The re-serialized messages must equal the
    writer's bytes (canonical field order), the PNG inside must decode to the sheet."""
    pytest.importorskip("google.protobuf")
    import struct
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    from palette_and_histo_gan_amd import png
    from palette_and_histo_gan_amd.side2side_model import S2SModel, ScalarLog
    F = descriptor_pb2.FieldDescriptorProto
    fd = descriptor_pb2.FileDescriptorProto(name="p2p_event2.proto", package="p2pt2", syntax="proto3")

    def msg(name, *fields):
        m = fd.message_type.add(name=name)
        for fname, num, typ, rep, tname in fields:
            m.field.add(name=fname, number=num, type=typ, label=3 if rep else 1, **({"type_name": ".p2pt2." + tname} if tname else {}))
    msg("Dim", ("size", 1, F.TYPE_INT64, False, None))
    msg("Shape", ("dim", 2, F.TYPE_MESSAGE, True, "Dim"))
    msg("Tensor", ("dtype", 1, F.TYPE_INT32, False, None), ("tensor_shape", 2, F.TYPE_MESSAGE, False, "Shape"), ("string_val", 8, F.TYPE_BYTES, True, None))
    msg("PluginData", ("plugin_name", 1, F.TYPE_STRING, False, None), ("content", 2, F.TYPE_BYTES, False, None))
    msg("Meta", ("plugin_data", 1, F.TYPE_MESSAGE, False, "PluginData"), ("data_class", 4, F.TYPE_INT32, False, None))
    msg("Value", ("tag", 1, F.TYPE_STRING, False, None), ("simple_value", 2, F.TYPE_FLOAT, False, None),
        ("tensor", 8, F.TYPE_MESSAGE, False, "Tensor"), ("metadata", 9, F.TYPE_MESSAGE, False, "Meta"))
    msg("Summary", ("value", 1, F.TYPE_MESSAGE, True, "Value"))
    msg("Event", ("wall_time", 1, F.TYPE_DOUBLE, False, None), ("step", 2, F.TYPE_INT64, False, None), ("file_version", 3, F.TYPE_STRING, False, None),
        ("summary", 5, F.TYPE_MESSAGE, False, "Summary"))
    msg("Multiline", ("tag", 1, F.TYPE_STRING, True, None))
    msg("Chart", ("title", 1, F.TYPE_STRING, False, None), ("multiline", 2, F.TYPE_MESSAGE, False, "Multiline"))
    msg("Category", ("title", 1, F.TYPE_STRING, False, None), ("chart", 2, F.TYPE_MESSAGE, True, "Chart"))
    msg("Layout", ("version", 1, F.TYPE_INT32, False, None), ("category", 2, F.TYPE_MESSAGE, True, "Category"))
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    Event = message_factory.GetMessageClass(pool.FindMessageTypeByName("p2pt2.Event"))
    Layout = message_factory.GetMessageClass(pool.FindMessageTypeByName("p2pt2.Layout"))
    log = ScalarLog(str(tmp_path))
    log.write_raw_pb(S2SModel.create_layout_summary(), step=0)
    sheet = np.random.default_rng(3).integers(0, 256, size=(70, 196, 4), dtype=np.uint8)
    log.scalar("generator/l1_loss", 0.25, 4)
    log.image("temp-side2side/logs/a/b/now/step_000005.png", sheet, step=5)
    log.flush()
    data, pos, events = open(log.events.path, "rb").read(), 0, []
    while pos < len(data):
        (ln,) = struct.unpack_from("<Q", data, pos)
        m = Event()
        m.ParseFromString(data[pos + 12:pos + 12 + ln])
        assert m.SerializeToString() == data[pos + 12:pos + 12 + ln] or m.step == 0        # canonical encoding (an explicit zero step aside)
        events.append(m)
        pos += 16 + ln
    assert [len(e.summary.value) for e in events] == [0, 1, 1, 1] and [e.step for e in events] == [0, 0, 4, 5]
    lay = events[1].summary.value[0]
    assert lay.tag == "custom_scalars__config__" and lay.metadata.plugin_data.plugin_name == "custom_scalars"
    assert lay.tensor.dtype == 7 and len(lay.tensor.tensor_shape.dim) == 0 and len(lay.tensor.string_val) == 1
    layout = Layout()
    layout.ParseFromString(lay.tensor.string_val[0])
    assert [(c.title, [(ch.title, list(ch.multiline.tag)) for ch in c.chart]) for c in layout.category] == [
        ("Fréchet Inception Distance", [("FID for train and test", [r"^fid\/"])]),
        ("L1 Evaluation", [("L1 for train and test", [r"^l1\-evaluation\/"])])]                 # side2side_model.py:244-271
    assert layout.SerializeToString() == lay.tensor.string_val[0]
    assert events[2].summary.value[0].simple_value == 0.25
    img = events[3].summary.value[0]
    assert img.tag.endswith("step_000005.png") and img.metadata.plugin_data.plugin_name == "images" and img.metadata.data_class == 3
    assert img.tensor.dtype == 7 and [d.size for d in img.tensor.tensor_shape.dim] == [3]
    w, h, blob = img.tensor.string_val
    assert (w, h) == (b"196", b"70") and blob[:8] == b"\x89PNG\r\n\x1a\n" and np.array_equal(png.decode_png(blob), sheet)
    # and the module's own reader sees the same three values
    vals = list(tb_events.read_events(log.events.path))
    assert [t for _, t, _ in vals] == ["custom_scalars__config__", "generator/l1_loss", img.tag] and vals[2][2][2] == blob


def test_palette_extraction_and_index_round_trip_are_bit_exact():
    rng = np.random.default_rng(7)
    colours = np.array([[0, 0, 0, 0], [10, 20, 30, 255], [200, 10, 10, 255], [10, 200, 10, 255], [250, 250, 250, 255],
                        [30, 30, 30, 255], [31, 29, 30, 255]], np.int32)
    img = colours[rng.integers(0, len(colours), size=(64, 64))]
    img[0, 0] = colours[4]                          # first appearance order: the brightest colour comes first
    pal = io_utils.extract_palette(img, "grayness")
    assert pal.shape == (MAX_PALETTE_SIZE, 4) and pal.dtype == np.int32
    n = len(colours)
    gray = (colours[:, :3].astype(np.float32) * np.array([0.2989, 0.5870, 0.1140], np.float32)).sum(1)
    assert np.array_equal(pal[:n], colours[np.argsort(gray, kind="stable")])        # dark to light (io_utils.py:45-50)
    assert np.array_equal(pal[0], [0, 0, 0, 0])                                       # transparent black is index 0
    assert (pal[n:] == np.array(INVALID_INDEX_COLOR)).all()                           # padded with the invalid colour
    top = io_utils.extract_palette(img, "top2bottom")
    assert np.array_equal(top[0], colours[4])                                         # UniqueWithCounts: order of first appearance
    idx = io_utils.rgba_to_indexed(img, pal)
    assert idx.shape == (64, 64, 1) and idx.dtype == np.int32 and idx.max() < n
    back = io_utils.indexed_to_rgba(idx, pal)
    assert np.array_equal(back, img)
    back_t = io_utils.indexed_to_rgba(torch.as_tensor(idx), torch.as_tensor(pal))
    assert np.array_equal(back_t.numpy(), img)
    # equal grayness: the stable sort keeps the order of first appearance
    tie = np.array([[[100, 0, 0, 255], [0, 0, 0, 255]], [[100, 0, 0, 255], [100, 0, 0, 255]]], np.int32)
    assert np.array_equal(io_utils.extract_palette(tie, "grayness")[:2], [[0, 0, 0, 255], [100, 0, 0, 255]])
    with pytest.raises(ValueError):
        io_utils.extract_palette(rng.integers(0, 256, size=(64, 64, 4)), "grayness")    # > 256 colours


def test_tensorboard_event_file_round_trip(tmp_path):
    assert tb_events.crc32c(b"123456789") == 0xE3069283                               # CRC-32C check value
    w = tb_events.EventFileWriter(str(tmp_path))
    rows = [("generator/total_loss", 34.5, 0, 1.0), ("generator/l1_loss", 0.25, 0, 1.5), ("discriminator/real_loss", 1e-7, 40, 2.0)]
    w.add_scalars(rows)
    got = list(tb_events.read_events(w.path))
    assert [(t, s) for s, t, _ in got] == [(r[0], r[2]) for r in rows]
    np.testing.assert_allclose([v for _, _, v in got], [r[1] for r in rows], rtol=1e-6)
