"""TensorFlow checkpoint (tensor-bundle) reader / writer and the reference -> manifest name mapping.

The reference saves ``tf.train.Checkpoint(optimizer=..., model=model_Dict['Train'])`` (reference Model.py:186-189,
284-291) and restores the latest one (Model.py:267-276).  TensorFlow is not installable here, so this module reads the
format directly:

* ``<prefix>.index`` -- a LevelDB-format table (prefix-compressed key blocks with restart arrays, a block index, a
  48-byte footer ending in the magic 0xdb4775248b80fb57).  The key "" holds a BundleHeaderProto, every other key a
  BundleEntryProto (dtype, shape, shard_id, offset, size, masked crc32c).
* ``<prefix>.data-NNNNN-of-MMMMM`` -- raw little-endian tensor bytes.
* object-based checkpoints name a variable by its path in the Python object graph:
  ``model/layer_with_weights-2/layer_Dict/Decoder_Step/.../kernel/.ATTRIBUTES/VARIABLE_VALUE``; dictionary attributes
  (the reference's ``layer_Dict``) appear as path components, Sequential members as ``layer_with_weights-N``.

UNVERIFIED AGAINST TENSORFLOW: no TensorFlow and no reference checkpoint exist in this environment (the published
38 k-step checkpoint, reference README.md:287-292, is a Google-Drive link).  What is tested is (a) the table / proto /
crc32c layers against their published known answers and against the writer below, (b) that a checkpoint written with
the reference's object paths converts back to the exact manifest.  ``reference_paths()`` is the single place that
encodes the reference's attribute names (cited per group) -- if a real checkpoint disagrees, ``convert()`` reports the
unmatched keys on both sides instead of guessing.
"""
import os
import re
import struct

import numpy as np

from .hparams import Dims
from .weights import manifest

MAGIC = 0xDB4775248B80FB57
SUFFIX = "/.ATTRIBUTES/VARIABLE_VALUE"
OBJECT_GRAPH_KEY = "_CHECKPOINTABLE_OBJECT_GRAPH"
# tensorflow/core/framework/types.proto
DT_FLOAT, DT_DOUBLE, DT_INT32, DT_STRING, DT_INT64, DT_BOOL, DT_BFLOAT16, DT_HALF = 1, 2, 3, 7, 9, 10, 14, 19
_NP = {DT_FLOAT: np.dtype("<f4"), DT_DOUBLE: np.dtype("<f8"), DT_INT32: np.dtype("<i4"), DT_INT64: np.dtype("<i8"),
       DT_BOOL: np.dtype("bool"), DT_HALF: np.dtype("<f2")}
_DT = {np.dtype("float32"): DT_FLOAT, np.dtype("float64"): DT_DOUBLE, np.dtype("int32"): DT_INT32,
       np.dtype("int64"): DT_INT64, np.dtype("bool"): DT_BOOL, np.dtype("float16"): DT_HALF}


# ------------------------------------------------------------------------------------------------ crc32c (Castagnoli)
def _make_table():
    t = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        t.append(c)
    return t


_TABLE = _make_table()
_native = None


def _native_crc():
    """The C-ABI library carries a host-side crc32c (gsttaco_crc32c) so that 100 MB of weights check in milliseconds."""
    global _native
    if _native is None:
        try:
            import ctypes
            from . import capi
            lib = capi.load_library()
            fn = lib.gsttaco_crc32c
            fn.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint32]
            fn.restype = ctypes.c_uint32
            _native = fn
        except Exception:
            _native = False
    return _native


def crc32c(data, crc=0, native=True):
    data = bytes(data) if not isinstance(data, (bytes, bytearray)) else data
    fn = _native_crc() if native and len(data) > 4096 else None
    if fn:
        return int(fn(bytes(data), len(data), crc))
    c = crc ^ 0xFFFFFFFF
    for b in data:
        c = _TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def mask_crc(crc):
    """leveldb / TF 'masked' crc: rotate right by 15 and add a constant, so a crc of crcs stays well distributed."""
    return (((crc >> 15) | (crc << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def unmask_crc(m):
    rot = (m - 0xA282EAD8) & 0xFFFFFFFF
    return ((rot >> 17) | (rot << 15)) & 0xFFFFFFFF


# ------------------------------------------------------------------------------------------------ varint / protobuf
def _get_varint(buf, pos):
    out = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def _put_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def parse_proto(buf):
    """Generic protobuf wire parser -> list of (field_number, wire_type, value); nested messages stay bytes."""
    pos, out = 0, []
    while pos < len(buf):
        tag, pos = _get_varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            n, pos = _get_varint(buf, pos)
            v = bytes(buf[pos:pos + n])
            pos += n
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise ValueError("unsupported protobuf wire type {}".format(wt))
        out.append((field, wt, v))
    return out


def _field(field, wt, payload):
    tag = _put_varint((field << 3) | wt)
    if wt == 0:
        return tag + _put_varint(payload)
    if wt == 2:
        return tag + _put_varint(len(payload)) + payload
    if wt == 5:
        return tag + struct.pack("<I", payload)
    raise ValueError(wt)


def _shape_proto(shape):
    return b"".join(_field(2, 2, _field(1, 0, int(s))) for s in shape)          # TensorShapeProto.dim[].size


def _parse_shape(buf):
    dims = []
    for f, _, v in parse_proto(buf):
        if f == 2:
            size = 0
            for f2, _, v2 in parse_proto(v):
                if f2 == 1:
                    size = v2
            dims.append(size)
    return tuple(dims)


def _entry_proto(dtype, shape, shard, offset, size, crc):
    """BundleEntryProto (tensor_bundle.proto): dtype=1, shape=2, shard_id=3, offset=4, size=5, crc32c=6 (fixed32)."""
    out = _field(1, 0, dtype) + _field(2, 2, _shape_proto(shape))
    if shard:
        out += _field(3, 0, shard)
    if offset:
        out += _field(4, 0, offset)
    out += _field(5, 0, size) + _field(6, 5, crc)
    return out


def _parse_entry(buf):
    e = {"dtype": 0, "shape": (), "shard": 0, "offset": 0, "size": 0, "crc": None, "slices": 0}
    for f, _, v in parse_proto(buf):
        if f == 1: e["dtype"] = v
        elif f == 2: e["shape"] = _parse_shape(v)
        elif f == 3: e["shard"] = v
        elif f == 4: e["offset"] = v
        elif f == 5: e["size"] = v
        elif f == 6: e["crc"] = v
        elif f == 7: e["slices"] += 1
    return e


# ------------------------------------------------------------------------------------------------ LevelDB table
def _read_block(buf, offset, size, verify):
    block = buf[offset:offset + size]
    ctype = buf[offset + size]
    if verify:
        stored = struct.unpack_from("<I", buf, offset + size + 1)[0]
        if unmask_crc(stored) != crc32c(bytes(block) + bytes([ctype]), native=False):
            raise ValueError("checkpoint index: block checksum mismatch at offset {}".format(offset))
    if ctype != 0:
        raise NotImplementedError("compressed table block (type {}): TF writes checkpoint indexes uncompressed".format(ctype))
    return block


def _block_entries(block):
    n_restarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * n_restarts
    pos, key, out = 0, b"", []
    while pos < end:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        out.append((key, bytes(block[pos:pos + vlen])))
        pos += vlen
    return out


def read_table(path, verify=True):
    """All (key, value) pairs of a LevelDB-format table file, in key order."""
    with open(path, "rb") as f:
        buf = f.read()
    if len(buf) < 48 or struct.unpack_from("<Q", buf, len(buf) - 8)[0] != MAGIC:
        raise ValueError("{}: not a TensorFlow checkpoint index (bad table magic)".format(path))
    footer = buf[len(buf) - 48:]
    pos = 0
    _, pos = _get_varint(footer, pos)           # metaindex handle
    _, pos = _get_varint(footer, pos)
    ioff, pos = _get_varint(footer, pos)        # index handle
    isize, pos = _get_varint(footer, pos)
    out = []
    for _, handle in _block_entries(_read_block(buf, ioff, isize, verify)):
        boff, p = _get_varint(handle, 0)
        bsize, p = _get_varint(handle, p)
        out.extend(_block_entries(_read_block(buf, boff, bsize, verify)))
    return out


def _build_block(items, restart_interval=16):
    out, restarts, prev = bytearray(), [], b""
    for i, (k, v) in enumerate(items):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
                shared += 1
        out += _put_varint(shared) + _put_varint(len(k) - shared) + _put_varint(len(v)) + k[shared:] + v
        prev = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def write_table(path, items, block_size=4096):
    """Writes sorted (key, value) pairs as an uncompressed LevelDB-format table."""
    items = sorted(items)
    out = bytearray()
    index = []

    def emit(block):
        off = len(out)
        out.extend(block)
        out.append(0)                                                           # kNoCompression
        out.extend(struct.pack("<I", mask_crc(crc32c(block + b"\x00", native=False))))
        return _put_varint(off) + _put_varint(len(block))

    cur, cur_bytes = [], 0
    for k, v in items:
        cur.append((k, v))
        cur_bytes += len(k) + len(v) + 3
        if cur_bytes >= block_size:
            index.append((cur[-1][0], emit(_build_block(cur))))
            cur, cur_bytes = [], 0
    if cur or not index:
        index.append((cur[-1][0] if cur else b"", emit(_build_block(cur))))
    meta_handle = emit(_build_block([]))
    index_handle = emit(_build_block(index, restart_interval=1))
    footer = meta_handle + index_handle
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", MAGIC)
    out.extend(footer)
    with open(path, "wb") as f:
        f.write(bytes(out))


# ------------------------------------------------------------------------------------------------ tensor bundle
def _shard_name(prefix, shard, num):
    return "{}.data-{:05d}-of-{:05d}".format(prefix, shard, num)


def read_bundle(prefix, verify=False, keys=None):
    """{key: ndarray} for every numeric tensor of the bundle (string tensors such as the object graph are returned as
    bytes).  ``verify`` also checks every tensor's crc32c."""
    pairs = read_table(prefix + ".index")
    header = dict((f, v) for f, _, v in parse_proto(pairs[0][1])) if pairs and pairs[0][0] == b"" else {}
    num_shards = header.get(1, 1)
    if header.get(2, 0) != 0:
        raise NotImplementedError("big-endian tensor bundle")
    shards, out = {}, {}
    for k, v in pairs:
        if k == b"":
            continue
        name = k.decode()
        if keys is not None and name not in keys:
            continue
        e = _parse_entry(v)
        if e["slices"]:
            raise NotImplementedError("partitioned variable '{}'".format(name))
        if e["shard"] not in shards:
            with open(_shard_name(prefix, e["shard"], num_shards), "rb") as f:
                shards[e["shard"]] = f.read()
        raw = shards[e["shard"]][e["offset"]:e["offset"] + e["size"]]
        if len(raw) != e["size"]:
            raise ValueError("'{}': data shard is truncated".format(name))
        if e["dtype"] == DT_STRING:
            n = int(np.prod(e["shape"])) if e["shape"] else 1
            pos, lens = 0, []
            for _ in range(n):
                ln, pos = _get_varint(raw, pos)
                lens.append(ln)
            pos += 4                                                            # checksum of the lengths
            vals = []
            for ln in lens:
                vals.append(bytes(raw[pos:pos + ln]))
                pos += ln
            out[name] = vals[0] if not e["shape"] else vals
            continue
        if e["dtype"] not in _NP:
            raise NotImplementedError("'{}': dtype enum {}".format(name, e["dtype"]))
        if verify and e["crc"] is not None and unmask_crc(e["crc"]) != crc32c(raw):
            raise ValueError("'{}': tensor checksum mismatch".format(name))
        a = np.frombuffer(raw, dtype=_NP[e["dtype"]])
        if a.size != int(np.prod(e["shape"], dtype=np.int64)):
            raise ValueError("'{}': {} bytes do not match shape {}".format(name, e["size"], e["shape"]))
        out[name] = a.reshape(e["shape"])
    return out


def write_bundle(prefix, tensors, strings=None):
    """One-shard bundle: ``tensors`` {key: ndarray}, ``strings`` {key: bytes} (scalar DT_STRING entries)."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    data = bytearray()
    # BundleHeaderProto: num_shards=1, endianness=LITTLE(0, default), version {producer=1}
    items = [(b"", _field(1, 0, 1) + _field(3, 2, _field(1, 0, 1)))]
    for key in sorted(set(tensors) | set(strings or {})):
        off = len(data)
        if strings and key in strings:
            s = strings[key]
            lens = _put_varint(len(s))
            # tensor_bundle.cc WriteStringTensor: lengths, masked crc32c of the lengths (as fixed-width ints), bytes;
            # the entry crc runs over the lengths (fixed-width), that checksum, and the bytes
            c = crc32c(struct.pack("<I", len(s)), native=False)
            cks = struct.pack("<I", mask_crc(c))
            raw = lens + cks + s
            c = crc32c(cks, c, native=False)
            c = crc32c(s, c)
            items.append((key.encode(), _entry_proto(DT_STRING, (), 0, off, len(raw), mask_crc(c))))
        else:
            a = np.asarray(tensors[key], order="C")        # (ascontiguousarray would turn a scalar into shape (1,))
            if a.dtype not in _DT:
                raise TypeError("'{}': dtype {}".format(key, a.dtype))
            raw = a.astype(a.dtype.newbyteorder("<"), copy=False).tobytes()
            items.append((key.encode(), _entry_proto(_DT[a.dtype], a.shape, 0, off, len(raw), mask_crc(crc32c(raw)))))
        data.extend(raw)
    write_table(prefix + ".index", items)
    with open(_shard_name(prefix, 0, 1), "wb") as f:
        f.write(bytes(data))


def latest_checkpoint(directory):
    """tf.train.latest_checkpoint: the prefix named by ``model_checkpoint_path`` in ``<directory>/checkpoint``
    (reference Model.py:268-270), or None."""
    state = os.path.join(directory, "checkpoint")
    if not os.path.exists(state):
        return None
    with open(state) as f:
        m = re.search(r'^model_checkpoint_path:\s*"(.*)"\s*$', f.read(), re.M)
    if not m:
        return None
    p = m.group(1)
    p = p if os.path.isabs(p) else os.path.join(directory, p)
    return p if os.path.exists(p + ".index") else None


# ------------------------------------------------------------------------------------------------ name mapping
BN = (("gamma", "gamma"), ("beta", "beta"), ("moving_mean", "moving_mean"), ("moving_variance", "moving_variance"))


def reference_paths(hp, vocab=None):
    """manifest name -> (top-level layer tag, object path below it) in the reference's checkpoints.

    Top-level layers are the weight-bearing layers of the functional Train model (reference Model.py:77-82, 133-150):
    Encoder, Style_Token_Layer, Decoder, Vocoder_Taco1; their ``layer_with_weights-K`` index depends on Keras' graph
    ordering, so ``convert`` recognises them by content instead (``_TOP_MARKERS``)."""
    d = Dims(hp, vocab)
    p = {}

    def bn(name, tag, path):
        for mine, theirs in BN:
            p["{}.bn.{}".format(name, mine)] = (tag, "{}/{}".format(path, theirs))

    def lstm(name, tag, path):
        for leaf in ("kernel", "recurrent_kernel", "bias"):
            p["{}.{}".format(name, leaf)] = (tag, "{}/{}".format(path, leaf))

    def dense(name, tag, path):
        p[name + ".kernel"] = (tag, path + "/kernel")
        p[name + ".bias"] = (tag, path + "/bias")

    # Encoder: self.layer = Sequential[Embedding, (Conv1D, BN, ReLU, Dropout) x n, Bidirectional(LSTM)]  (Taco2.py:16-43)
    p["encoder.embedding"] = ("encoder", "layer/layer_with_weights-0/embeddings")
    n = len(d.enc_filters)
    for i in range(n):
        p["encoder.conv%d.kernel" % i] = ("encoder", "layer/layer_with_weights-%d/kernel" % (1 + 2 * i))
        bn("encoder.conv%d" % i, "encoder", "layer/layer_with_weights-%d" % (2 + 2 * i))
    for mine, theirs in (("fwd", "forward_layer"), ("bwd", "backward_layer")):
        lstm("encoder.bilstm." + mine, "encoder", "layer/layer_with_weights-%d/%s/cell" % (1 + 2 * n, theirs))
    # Style_Token_Layer (GST.py:72-90) -> Reference_Encoder (GST.py:12-45) + MultiHeadAttention (Layers.py:162-168, 254-276)
    if d.gst:
        ref = "layer_Dict/Reference_Encoder/layer_Dict"
        for i in range(len(d.ref_filters)):
            p["gst.ref.conv%d.kernel" % i] = ("gst", "%s/Conv2D_%d/layer_with_weights-0/kernel" % (ref, i))
            bn("gst.ref.conv%d" % i, "gst", "%s/Conv2D_%d/layer_with_weights-1" % (ref, i))
        lstm("gst.ref.gru", "gst", ref + "/RNN/cell")
        dense("gst.ref.dense", "gst", ref + "/Dense")
        p["gst.tokens"] = ("gst", "gst_tokens")
        dense("gst.mha.query", "gst", "layer_Dict/Attention/layer_Dict/Query")
        dense("gst.mha.value", "gst", "layer_Dict/Attention/layer_Dict/Value")
        p["gst.mha.ln.gamma"] = ("gst", "layer_Dict/Attention/layer_Dict/Layer_Normalization/gamma")
        p["gst.mha.ln.beta"] = ("gst", "layer_Dict/Attention/layer_Dict/Layer_Normalization/beta")
    # Decoder (Taco2.py:126-151) -> Decoder_Step (Taco2.py:59-89): Prenet (:269-281), attention (Steps.py:65-84),
    # StackedRNNCells, Projection; Postnet Sequential[(Conv1D, BN, [tanh], Dropout) x n]
    step = "layer_Dict/Decoder_Step/layer_Dict"
    for i in range(len(d.prenet)):
        dense("decoder.prenet%d" % i, "decoder", "%s/Prenet/layer/layer_with_weights-%d" % (step, i))
    dense("decoder.attention.query", "decoder", step + "/Attention/layer_Dict/Query")
    dense("decoder.attention.value", "decoder", step + "/Attention/layer_Dict/Value")
    if d.att_type != "LSA":
        p["decoder.attention.v"] = ("decoder", step + "/Attention/attention_v")
        p["decoder.attention.score_bias"] = ("decoder", step + "/Attention/attention_score_bias")
    for i in range(len(d.dec_rnn)):
        lstm("decoder.lstm%d" % i, "decoder", "%s/RNN/cells/%d" % (step, i))
    dense("decoder.projection", "decoder", step + "/Projection")
    for i in range(len(d.post_filters)):
        p["postnet.conv%d.kernel" % i] = ("decoder", "layer_Dict/Postnet/layer_with_weights-%d/kernel" % (2 * i))
        bn("postnet.conv%d" % i, "decoder", "layer_Dict/Postnet/layer_with_weights-%d" % (2 * i + 1))
    # Vocoder_Taco1 (Taco2.py:238-256) -> CBHG (:313-361): ConvBank (:387-397), Conv1D_Projection, Highwaynet (:410-419), RNN
    if d.vocoder:
        cb = "layer_Dict/CBHG/layer_Dict"
        for i in range(d.bank_count):
            p["vocoder.convbank%d.kernel" % i] = ("vocoder", "%s/ConvBank/layer_Dict/ConvBank_%d/layer_with_weights-0/kernel" % (cb, i))
            bn("vocoder.convbank%d" % i, "vocoder", "%s/ConvBank/layer_Dict/ConvBank_%d/layer_with_weights-1" % (cb, i))
        n = len(d.voc_proj_filters)
        for i in range(n):
            p["vocoder.proj%d.kernel" % i] = ("vocoder", "%s/Conv1D_Projection/layer_with_weights-%d/kernel" % (cb, 2 * i))
            bn("vocoder.proj%d" % i, "vocoder", "%s/Conv1D_Projection/layer_with_weights-%d" % (cb, 2 * i + 1))
        man = manifest(hp, vocab)
        if "vocoder.proj_dense.kernel" in man:
            dense("vocoder.proj_dense", "vocoder", "%s/Conv1D_Projection/layer_with_weights-%d" % (cb, 2 * n))
        off = 0
        if "vocoder.highway_in.kernel" in man:
            dense("vocoder.highway_in", "vocoder", cb + "/Highwaynet/layer_with_weights-0")
            off = 1
        for i in range(d.highway_count):
            dense("vocoder.highway%d.relu" % i, "vocoder", "%s/Highwaynet/layer_with_weights-%d/layer_Dict/Dense_Relu" % (cb, off + i))
            dense("vocoder.highway%d.sigmoid" % i, "vocoder", "%s/Highwaynet/layer_with_weights-%d/layer_Dict/Dense_Sigmoid" % (cb, off + i))
        for mine, theirs in (("fwd", "forward_layer"), ("bwd", "backward_layer")):
            lstm("vocoder.bilstm." + mine, "vocoder", "%s/RNN/%s/cell" % (cb, theirs))
        dense("vocoder.dense", "vocoder", "layer_Dict/Dense")
    return p


# how each top-level layer of the Train model is recognised among model/layer_with_weights-K
_TOP_MARKERS = {"encoder": "/layer/layer_with_weights-0/embeddings", "gst": "/gst_tokens",
                "decoder": "/layer_Dict/Decoder_Step/", "vocoder": "/layer_Dict/CBHG/"}
_TOP_ORDER = ("encoder", "gst", "decoder", "vocoder")           # used by the writer only


def _variable_keys(keys):
    """object-graph variable keys without optimizer slots / bookkeeping -> path without the attribute suffix"""
    out = {}
    for k in keys:
        if not k.endswith(SUFFIX) or ".OPTIMIZER_SLOT" in k or k.startswith("optimizer/") or k.startswith("save_counter"):
            continue
        out[k[:-len(SUFFIX)]] = k
    return out


def convert(bundle, hp, vocab=None, root="model"):
    """{checkpoint key: ndarray} of a reference checkpoint -> weight dict in gst_tacotron_amd.weights.manifest names.
    Raises KeyError listing what could not be matched on either side."""
    var = _variable_keys(bundle.keys())
    tops = {}
    for path in var:
        m = re.match(r"^(%s/layer_with_weights-\d+)(/.*)$" % re.escape(root), path)
        if not m:
            continue
        for tag, marker in _TOP_MARKERS.items():
            if (m.group(2) + "/").startswith(marker.rstrip("/") + "/"):
                if tops.setdefault(tag, m.group(1)) != m.group(1):
                    raise KeyError("two top-level layers look like the {}: {} and {}".format(tag, tops[tag], m.group(1)))
    man = manifest(hp, vocab)
    out, missing, used = {}, [], set()
    for name, (tag, rel) in reference_paths(hp, vocab).items():
        path = "{}/{}".format(tops.get(tag, "<no {} layer found>".format(tag)), rel)
        if path not in var:
            missing.append("{} <- {}".format(name, path))
            continue
        a = np.asarray(bundle[var[path]])
        if a.dtype != np.float32:
            a = a.astype(np.float32)                       # mixed_float16 keeps float32 variables; be lenient anyway
        if tuple(a.shape) != tuple(man[name]):
            raise ValueError("'{}' ({}): shape {} but the hyper-parameters expect {}".format(name, path, a.shape, man[name]))
        out[name] = a
        used.add(path)
    extra = sorted(p for p in var if p not in used and p.startswith(root + "/"))
    if missing or set(out) != set(man):
        raise KeyError("checkpoint does not match the reference layout for these hyper-parameters.\n  missing:\n    "
                       + "\n    ".join(missing or ["-"]) + "\n  unused checkpoint variables:\n    " + "\n    ".join(extra[:40] or ["-"]))
    return out


def load_reference_checkpoint(prefix, hp, vocab=None, verify=False):
    """Reads ``<prefix>.index`` / ``.data-*`` and returns the weight dict (reference Model.py:267-276)."""
    return convert(read_bundle(prefix, verify=verify), hp, vocab)


def _object_graph(paths):
    """Serialized TrackableObjectGraph (trackable_object_graph.proto) for variable object paths: nodes[0] is the root;
    every path component is a child edge (node_id=1, local_name=2); a variable node carries one attribute
    (name=1 'VARIABLE_VALUE', full_name=2, checkpoint_key=3)."""
    nodes = [{"children": {}, "attr": None}]
    for path in sorted(paths):
        cur = 0
        for comp in path.split("/"):
            nxt = nodes[cur]["children"].get(comp)
            if nxt is None:
                nodes.append({"children": {}, "attr": None})
                nxt = len(nodes) - 1
                nodes[cur]["children"][comp] = nxt
            cur = nxt
        nodes[cur]["attr"] = path
    out = b""
    for n in nodes:
        body = b""
        for name, nid in n["children"].items():
            body += _field(1, 2, _field(1, 0, nid) + _field(2, 2, name.encode()))
        if n["attr"] is not None:
            body += _field(2, 2, _field(1, 2, b"VARIABLE_VALUE") + _field(2, 2, n["attr"].encode())
                           + _field(3, 2, (n["attr"] + SUFFIX).encode()))
        out += _field(1, 2, body)
    return out


def save_reference_checkpoint(prefix, hp, weights, vocab=None, root="model"):
    """Writes ``weights`` (manifest names) as a checkpoint in the reference's object layout, with an object graph and
    the ``checkpoint`` state file next to it.  Used by the round-trip tests; TensorFlow-side loading is unverified."""
    tops = {}
    k = 0
    paths = reference_paths(hp, vocab)
    for tag in _TOP_ORDER:
        if any(t == tag for t, _ in paths.values()):
            tops[tag] = "{}/layer_with_weights-{}".format(root, k)
            k += 1
    tensors = {}
    for name, (tag, rel) in paths.items():
        tensors["{}/{}{}".format(tops[tag], rel, SUFFIX)] = np.asarray(weights[name], dtype=np.float32)
    tensors["save_counter" + SUFFIX] = np.asarray(1, dtype=np.int64)
    graph = _object_graph([k_[:-len(SUFFIX)] for k_ in tensors])
    write_bundle(prefix, tensors, strings={OBJECT_GRAPH_KEY: graph})
    with open(os.path.join(os.path.dirname(os.path.abspath(prefix)), "checkpoint"), "w") as f:
        base = os.path.basename(prefix)
        f.write('model_checkpoint_path: "{}"\nall_model_checkpoint_paths: "{}"\n'.format(base, base))
    return prefix


def object_graph_paths(graph_bytes):
    """Variable checkpoint keys listed by a serialized TrackableObjectGraph (for inspection / tests)."""
    keys = []
    for f, _, node in parse_proto(graph_bytes):
        if f != 1:
            continue
        for f2, _, v in parse_proto(node):
            if f2 == 2:
                for f3, _, v3 in parse_proto(v):
                    if f3 == 3:
                        keys.append(v3.decode())
    return keys
