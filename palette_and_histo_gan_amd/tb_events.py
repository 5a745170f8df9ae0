"""Minimal TensorBoard event-file writer: what the reference's `tf.summary.scalar` calls (pix2pix_model.py:85-101,255-258,327-330),
its per-update `tf.summary.image` of the preview sheet (side2side_model.py:86-93) and the custom-scalar layout it writes at step 0
with `write_raw_pb` (side2side_model.py:58-61,240-273) leave on disk, without TensorFlow.

File format (public, stable): a TFRecord stream -- per record  uint64 length | uint32 masked crc32c(length) | data |
uint32 masked crc32c(data) -- of serialized `tensorflow.Event` protos:
    Event  { double wall_time = 1; int64 step = 2; string file_version = 3; Summary summary = 5; }
    Summary{ repeated Value value = 1; }    Value { string tag = 1; float simple_value = 2; }
The first record carries file_version "brain.Event:2".  `tensorboard --logdir` reads the result.

Tensor-valued summaries (TF2 `tf.summary.image`, the custom-scalar layout) use two more fields of Value:
    Value { SummaryMetadata metadata = 9; TensorProto tensor = 8; }
    SummaryMetadata { PluginData plugin_data = 1 { string plugin_name = 1; bytes content = 2; }  DataClass data_class = 4; }
    TensorProto { DataType dtype = 1 (DT_STRING = 7); TensorShapeProto tensor_shape = 2 { repeated Dim dim = 2 { int64 size = 1; } }
                  repeated bytes string_val = 8; }
An image summary is a rank-1 string tensor [width, height, png, ...] under plugin "images" (data class BLOB_SEQUENCE = 3); the
layout is a scalar string tensor holding a serialized tensorboard `Layout` under plugin "custom_scalars", tag
"custom_scalars__config__":
    Layout { int32 version = 1; repeated Category category = 2; }     Category { string title = 1; repeated Chart chart = 2; }
    Chart { string title = 1; MultilineChartContent multiline = 2; }  MultilineChartContent { repeated string tag = 1; }"""
import os
import socket
import struct
import time

_CRC_TABLE = []


def _crc_table():
    if not _CRC_TABLE:
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            _CRC_TABLE.append(c)
    return _CRC_TABLE


def crc32c(data):
    t = _crc_table()
    c = 0xFFFFFFFF
    for b in data:
        c = t[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _masked(data):
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(n):
    out = bytearray()
    n &= (1 << 64) - 1
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _field_bytes(num, payload):
    return _varint((num << 3) | 2) + _varint(len(payload)) + payload


def _field_varint(num, n):
    return _varint((num << 3) | 0) + _varint(n)


def encode_string_tensor(strings, scalar=False):
    """TensorProto of dtype DT_STRING: rank 1 with len(strings) entries, or rank 0 (scalar=True, one entry)"""
    shape = b"" if scalar else _field_bytes(2, _field_varint(1, len(strings)))
    out = _field_varint(1, 7) + _field_bytes(2, shape)
    for s in strings:
        out += _field_bytes(8, bytes(s))
    return out


def encode_tensor_value(tag, plugin_name, tensor, data_class=0, plugin_content=b""):
    """Summary.Value with a tensor and the metadata that routes it to a TensorBoard plugin (field order as protobuf serializes it)"""
    plugin = _field_bytes(1, plugin_name.encode()) + (_field_bytes(2, plugin_content) if plugin_content else b"")
    meta = _field_bytes(1, plugin) + (_field_varint(4, data_class) if data_class else b"")
    return _field_bytes(1, tag.encode()) + _field_bytes(8, tensor) + _field_bytes(9, meta)


def encode_image_value(tag, png_list, width, height):
    """what tf.summary.image(tag, data, max_outputs=len(png_list)) serializes (TF 2.9 summary/_tf/summary image op):
    [str(width), str(height), png, ...] under the "images" plugin"""
    tensor = encode_string_tensor([str(int(width)).encode(), str(int(height)).encode()] + list(png_list))
    return encode_tensor_value(tag, "images", tensor, data_class=3)


def encode_layout(categories):
    """tensorboard.plugins.custom_scalar layout_pb2.Layout from [(category title, [(chart title, [tag regex, ...]), ...]), ...]"""
    out = b""
    for title, charts in categories:
        cat = _field_bytes(1, title.encode())
        for ctitle, tags in charts:
            multiline = b"".join(_field_bytes(1, t.encode()) for t in tags)
            cat += _field_bytes(2, _field_bytes(1, ctitle.encode()) + _field_bytes(2, multiline))
        out += _field_bytes(2, cat)
    return out


def encode_layout_value(layout):
    """custom_scalar.summary.pb(layout) (side2side_model.py:240-273): the value tf.summary.experimental.write_raw_pb puts into the file"""
    return encode_tensor_value("custom_scalars__config__", "custom_scalars", encode_string_tensor([layout], scalar=True))


def encode_event(wall_time, step=None, file_version=None, scalars=(), values=()):
    """values: already encoded Summary.Value messages (encode_image_value, encode_layout_value)"""
    ev = struct.pack("<Bd", (1 << 3) | 1, wall_time)
    if step is not None:
        ev += _varint((2 << 3) | 0) + _varint(int(step))
    if file_version is not None:
        ev += _field_bytes(3, file_version.encode())
    if scalars or values:
        summary = b""
        for tag, value in scalars:
            val = _field_bytes(1, tag.encode()) + struct.pack("<Bf", (2 << 3) | 5, float(value))
            summary += _field_bytes(1, val)
        for val in values:
            summary += _field_bytes(1, val)
        ev += _field_bytes(5, summary)
    return ev


def record(data):
    head = struct.pack("<Q", len(data))
    return head + struct.pack("<I", _masked(head)) + data + struct.pack("<I", _masked(data))


class EventFileWriter:
    def __init__(self, folder):
        os.makedirs(folder, exist_ok=True)
        self.path = os.path.join(folder, f"events.out.tfevents.{int(time.time())}.{socket.gethostname()}.{os.getpid()}.p2pgan")
        with open(self.path, "wb") as f:
            f.write(record(encode_event(time.time(), file_version="brain.Event:2")))

    def add_scalars(self, rows):
        """rows: iterable of (tag, value, step, wall_time)"""
        with open(self.path, "ab") as f:
            for tag, value, step, wall in rows:
                f.write(record(encode_event(wall, step=step, scalars=[(tag, value)])))

    def add_values(self, values, step, wall_time=None):
        """encoded Summary.Value messages (images, the custom-scalar layout) as one event"""
        with open(self.path, "ab") as f:
            f.write(record(encode_event(time.time() if wall_time is None else wall_time, step=step, values=list(values))))


def read_events(path):
    """Decoder for tests: yields (step, tag, value) of every summary value in an event file (checks both checksums); value = the
    float of a scalar, or the list of byte strings of a string tensor (image: [width, height, png...], layout: [Layout])."""
    data = open(path, "rb").read()
    pos = 0

    def varint(buf, p):
        n = shift = 0
        while True:
            b = buf[p]
            p += 1
            n |= (b & 0x7F) << shift
            shift += 7
            if not b & 0x80:
                return n, p

    def fields(buf):
        p = 0
        while p < len(buf):
            key, p = varint(buf, p)
            num, wt = key >> 3, key & 7
            if wt == 0:
                v, p = varint(buf, p)
            elif wt == 1:
                v, p = buf[p:p + 8], p + 8
            elif wt == 5:
                v, p = buf[p:p + 4], p + 4
            else:
                ln, p = varint(buf, p)
                v, p = buf[p:p + ln], p + ln
            yield num, wt, v

    while pos < len(data):
        (ln,) = struct.unpack_from("<Q", data, pos)
        (c1,) = struct.unpack_from("<I", data, pos + 8)
        body = data[pos + 12:pos + 12 + ln]
        (c2,) = struct.unpack_from("<I", data, pos + 12 + ln)
        assert c1 == _masked(data[pos:pos + 8]) and c2 == _masked(body), "corrupt record"
        pos += 16 + ln
        step = 0
        for num, wt, v in fields(body):
            if num == 2:
                step = v
            elif num == 5:
                for n2, _, val in fields(v):
                    if n2 == 1:
                        tag, x = None, None
                        for n3, w3, v3 in fields(val):
                            if n3 == 1:
                                tag = v3.decode()
                            elif n3 == 2 and w3 == 5:
                                (x,) = struct.unpack("<f", v3)
                            elif n3 == 8:       # tensor: the list of its string values
                                x = [bytes(v4) for n4, _, v4 in fields(v3) if n4 == 8]
                        yield step, tag, x
