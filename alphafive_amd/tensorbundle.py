"""Reader for TensorFlow TensorBundle checkpoints (the format tf.train.Saver writes and the
reference restores at genData/network.py:113-122), without TensorFlow.

A bundle is `<prefix>.index` — an SSTable (LevelDB table format: prefix-compressed key/value
blocks, an index block of block handles, a 48-byte footer) mapping tensor names to serialized
BundleEntryProto {dtype, shape, shard_id, offset, size, crc32c} — plus `<prefix>.data-XXXXX-of-YYYYY`
shards of raw little-endian tensor bytes.  Only what the reference's checkpoints use is
supported: uncompressed blocks (snappy is rejected loudly), DT_FLOAT/DT_INT32/DT_INT64 tensors.
"""
import os
import struct

import numpy as np

_MAGIC = 0xDB4775248B80FB57
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64}


def _varint(buf, pos):
    result = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _read_block(data, offset, size):
    body = data[offset:offset + size]
    ctype = data[offset + size]
    if ctype != 0:
        raise ValueError("compressed SSTable block (type %d) not supported" % ctype)
    n_restarts = struct.unpack_from("<I", body, len(body) - 4)[0]
    limit = len(body) - 4 - 4 * n_restarts
    pos, key, out = 0, b"", []
    while pos < limit:
        shared, pos = _varint(body, pos)
        non_shared, pos = _varint(body, pos)
        vlen, pos = _varint(body, pos)
        key = key[:shared] + body[pos:pos + non_shared]
        pos += non_shared
        out.append((key, body[pos:pos + vlen]))
        pos += vlen
    return out


def _parse_proto(buf):
    """Minimal protobuf wire parser -> {field: [values]} (varint ints, fixed32 ints, bytes)."""
    pos, out = 0, {}
    while pos < len(buf):
        tag, pos = _varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise ValueError("unsupported wire type %d" % wt)
        out.setdefault(field, []).append(v)
    return out


def _signed(v):
    return v - (1 << 64) if v >= 1 << 63 else v


def read_index(prefix):
    """-> {name: dict(dtype, shape, shard_id, offset, size)} for every tensor in the bundle."""
    with open(prefix + ".index", "rb") as f:
        data = f.read()
    footer = data[-48:]
    if struct.unpack_from("<Q", footer, 40)[0] != _MAGIC:
        raise ValueError("not an SSTable: bad magic in %s.index" % prefix)
    pos = 0
    _, pos = _varint(footer, pos)      # metaindex handle
    _, pos = _varint(footer, pos)
    ioff, pos = _varint(footer, pos)   # index handle
    isize, pos = _varint(footer, pos)
    entries = {}
    for _, handle in _read_block(data, ioff, isize):
        boff, p = _varint(handle, 0)
        bsize, p = _varint(handle, p)
        for key, val in _read_block(data, boff, bsize):
            if key == b"":
                continue                  # BundleHeaderProto
            msg = _parse_proto(val)
            dtype = msg.get(1, [0])[0]
            shape = []
            if 2 in msg:
                for dim in _parse_proto(msg[2][0]).get(2, []):
                    shape.append(_signed(_parse_proto(dim).get(1, [0])[0]))
            entries[key.decode()] = dict(dtype=dtype, shape=tuple(shape), shard_id=msg.get(3, [0])[0],
                                         offset=msg.get(4, [0])[0], size=msg.get(5, [0])[0])
    return entries


def load_bundle(prefix):
    """-> {name: ndarray} (copies)."""
    index = read_index(prefix)
    d = os.path.dirname(prefix) or "."
    base = os.path.basename(prefix)
    shards = sorted(f for f in os.listdir(d) if f.startswith(base + ".data-"))
    blobs = {}
    out = {}
    for name, e in index.items():
        if e["dtype"] not in _DTYPES:
            raise ValueError("tensor %s: unsupported dtype enum %d" % (name, e["dtype"]))
        sid = e["shard_id"]
        if sid not in blobs:
            with open(os.path.join(d, shards[sid]), "rb") as f:
                blobs[sid] = f.read()
        dt = np.dtype(_DTYPES[e["dtype"]]).newbyteorder("<")
        arr = np.frombuffer(blobs[sid], dtype=dt, count=int(np.prod(e["shape"], dtype=np.int64)) if e["shape"] else 1,
                            offset=e["offset"])
        out[name] = arr.reshape(e["shape"]).copy()
    return out


def resolve_checkpoint(path):
    """tf.train.get_checkpoint_state semantics used by network.py:114-121: `path` is a directory
    holding a `checkpoint` file, or a checkpoint prefix."""
    if os.path.isdir(path):
        ck = os.path.join(path, "checkpoint")
        if os.path.exists(ck):
            with open(ck) as f:
                for line in f:
                    if line.startswith("model_checkpoint_path:"):
                        name = line.split(":", 1)[1].strip().strip('"')
                        return name if os.path.isabs(name) else os.path.join(path, name)
        raise FileNotFoundError("Could not find old network weights")
    if os.path.exists(path + ".index"):
        return path
    raise FileNotFoundError("Could not find old network weights")


# ---------------------------------------------------------------------------------------------
# writer — lets the untouched TF trainer / GUI (network.py:113-122) restore weights produced here
# ---------------------------------------------------------------------------------------------
_CRC_TABLE = None


def crc32c(data, crc=0):
    """CRC-32C (Castagnoli), as used by TensorBundle entries and SSTable block trailers."""
    global _CRC_TABLE
    if _CRC_TABLE is None:
        tab = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            tab.append(c)
        _CRC_TABLE = np.array(tab, np.uint32)
    crc ^= 0xFFFFFFFF
    tab = _CRC_TABLE
    for b in bytes(data):
        crc = int(tab[(crc ^ b) & 0xFF]) ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def masked_crc32c(data):
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _enc_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _field(num, wt, payload):
    return _enc_varint((num << 3) | wt) + payload


def _entry_proto(dtype_enum, shape, offset, size, crc):
    dims = b"".join(_field(2, 2, _enc_varint(len(d)) + d) for d in (_field(1, 0, _enc_varint(int(s))) for s in shape))
    msg = _field(1, 0, _enc_varint(dtype_enum)) + _field(2, 2, _enc_varint(len(dims)) + dims)
    if offset:
        msg += _field(4, 0, _enc_varint(offset))
    msg += _field(5, 0, _enc_varint(size)) + _field(6, 5, struct.pack("<I", crc))
    return msg


def _build_block(entries, restart_interval=16):
    body, restarts, prev = bytearray(), [], b""
    for i, (key, val) in enumerate(entries):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(body))
        else:
            while shared < min(len(prev), len(key)) and prev[shared] == key[shared]:
                shared += 1
        body += _enc_varint(shared) + _enc_varint(len(key) - shared) + _enc_varint(len(val)) + key[shared:] + val
        prev = key
    if not restarts:
        restarts = [0]
    for r in restarts:
        body += struct.pack("<I", r)
    body += struct.pack("<I", len(restarts))
    return bytes(body)


def _with_trailer(block):
    return block + b"\x00" + struct.pack("<I", masked_crc32c(block + b"\x00"))


def _short_successor(key):
    """LevelDB BytewiseComparator::FindShortSuccessor — the index key TF's table builder emits."""
    for i, b in enumerate(key):
        if b != 0xFF:
            return key[:i] + bytes([b + 1])
    return key


def save_bundle(prefix, variables):
    """Write `<prefix>.index` and `<prefix>.data-00000-of-00001` (single shard, fp32/int tensors) in the
    layout tf.train.Saver produces: tensors in name order, no padding, masked crc32c per tensor."""
    inv = {np.dtype(v): k for k, v in _DTYPES.items()}
    names = sorted(variables)
    data = bytearray()
    entries = [(b"", _field(1, 0, _enc_varint(1)) + _field(3, 2, _enc_varint(2) + _field(1, 0, _enc_varint(1))))]
    for name in names:
        arr = np.ascontiguousarray(variables[name])
        arr = arr.astype(arr.dtype.newbyteorder("<"), copy=False)
        raw = arr.tobytes()
        entries.append((name.encode(), _entry_proto(inv[np.dtype(arr.dtype.name)], arr.shape, len(data), len(raw),
                                                    masked_crc32c(raw))))
        data += raw
    block = _build_block(entries)
    out = bytearray(_with_trailer(block))
    meta_off = len(out)
    meta = _build_block([])
    out += _with_trailer(meta)
    index_off = len(out)
    handle = _enc_varint(0) + _enc_varint(len(block))
    index = _build_block([(_short_successor(entries[-1][0]), handle)])
    out += _with_trailer(index)
    footer = _enc_varint(meta_off) + _enc_varint(len(meta)) + _enc_varint(index_off) + _enc_varint(len(index))
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", _MAGIC)
    out += footer
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        f.write(bytes(data))
    with open(prefix + ".index", "wb") as f:
        f.write(bytes(out))


def write_checkpoint_state(directory, name):
    """The `checkpoint` text file tf.train.get_checkpoint_state reads (network.py:114)."""
    with open(os.path.join(directory, "checkpoint"), "w") as f:
        f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (name, name))
