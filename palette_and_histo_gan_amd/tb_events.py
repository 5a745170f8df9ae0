"""Minimal TensorBoard event-file writer (scalar summaries only): what the reference's `tf.summary.scalar` calls
(pix2pix_model.py:85-101,255-258,327-330; side2side_model.py:58-61) leave on disk, without TensorFlow.

File format (public, stable): a TFRecord stream -- per record  uint64 length | uint32 masked crc32c(length) | data |
uint32 masked crc32c(data) -- of serialized `tensorflow.Event` protos:
    Event  { double wall_time = 1; int64 step = 2; string file_version = 3; Summary summary = 5; }
    Summary{ repeated Value value = 1; }    Value { string tag = 1; float simple_value = 2; }
The first record carries file_version "brain.Event:2".  `tensorboard --logdir` reads the result."""
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


def encode_event(wall_time, step=None, file_version=None, scalars=()):
    ev = struct.pack("<Bd", (1 << 3) | 1, wall_time)
    if step is not None:
        ev += _varint((2 << 3) | 0) + _varint(int(step))
    if file_version is not None:
        ev += _field_bytes(3, file_version.encode())
    if scalars:
        summary = b""
        for tag, value in scalars:
            val = _field_bytes(1, tag.encode()) + struct.pack("<Bf", (2 << 3) | 5, float(value))
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


def read_events(path):
    """Decoder for tests: yields (step, tag, value) of every scalar in an event file (checks both checksums)."""
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
                        yield step, tag, x
