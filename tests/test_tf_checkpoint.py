"""the TensorFlow checkpoint reader / reference-name mapping, without TensorFlow.  Known answers for the
format layers (crc32c, masking, varints, table layout) + round trips through the writer."""
import os
import struct

import numpy as np
import pytest

from gst_tacotron_amd import hparams, synthetic, tf_checkpoint as T, weights


def test_crc32c_known_answers():
    # RFC 3720 (iSCSI) appendix B.4 test vectors for CRC-32C
    assert T.crc32c(b"123456789", native=False) == 0xE3069283
    assert T.crc32c(bytes(32), native=False) == 0x8A9136AA
    assert T.crc32c(bytes([0xFF] * 32), native=False) == 0x62A8AB43
    assert T.crc32c(bytes(range(32)), native=False) == 0x46DD794E
    data = np.random.default_rng(0).integers(0, 256, 100003, dtype=np.uint8).tobytes()
    slow = T.crc32c(data, native=False)
    assert T.crc32c(data) == slow                                           # native slicing-by-8 in libgsttaco.so
    assert T.crc32c(data[5000:], T.crc32c(data[:5000])) == slow             # continuation
    # leveldb's masking: rotate right 15, add 0xa282ead8; must invert
    for c in (0, 1, 0xE3069283, 0xFFFFFFFF):
        assert T.unmask_crc(T.mask_crc(c)) == c
    assert T.mask_crc(0) == 0xA282EAD8


def test_varint_and_proto_roundtrip():
    for v in (0, 1, 127, 128, 300, 2 ** 32 + 5):
        b = T._put_varint(v)
        assert T._get_varint(b, 0) == (v, len(b))
    assert T._put_varint(300) == b"\xac\x02"                                 # protobuf documentation example
    e = T._parse_entry(T._entry_proto(T.DT_FLOAT, (3, 5), 0, 64, 60, 0xDEADBEEF))
    assert (e["dtype"], e["shape"], e["offset"], e["size"], e["crc"]) == (1, (3, 5), 64, 60, 0xDEADBEEF)


def test_table_layout_and_roundtrip(tmp_path):
    p = str(tmp_path / "t.index")
    items = [(("key%05d" % i).encode(), os.urandom(i % 50)) for i in range(700)] + [(b"", b"hdr")]
    T.write_table(p, items, block_size=512)
    raw = open(p, "rb").read()
    assert struct.unpack("<Q", raw[-8:])[0] == 0xDB4775248B80FB57 and len(raw) > 48      # leveldb footer magic
    assert T.read_table(p) == sorted(items)
    corrupted = bytearray(raw)
    corrupted[10] ^= 0x40
    open(p, "wb").write(bytes(corrupted))
    with pytest.raises(ValueError, match="checksum"):
        T.read_table(p)
    open(p, "wb").write(raw[:-3])
    with pytest.raises(ValueError, match="magic"):
        T.read_table(p)


def test_bundle_roundtrip_dtypes(tmp_path):
    rng = np.random.default_rng(1)
    tensors = {"a/b" + T.SUFFIX: rng.normal(size=(7, 3)).astype(np.float32), "scalar": np.asarray(3.5, np.float32),
               "i64": np.arange(5, dtype=np.int64), "half": rng.normal(size=(4,)).astype(np.float16)}
    prefix = str(tmp_path / "ck" / "S_10.CHECKPOINT.H5-1")
    T.write_bundle(prefix, tensors, strings={T.OBJECT_GRAPH_KEY: b"\x0a\x00"})
    assert os.path.exists(prefix + ".index") and os.path.exists(prefix + ".data-00000-of-00001")
    back = T.read_bundle(prefix, verify=True)
    for k, v in tensors.items():
        assert back[k].dtype == v.dtype and back[k].shape == v.shape and np.array_equal(back[k], v)
    assert back[T.OBJECT_GRAPH_KEY] == b"\x0a\x00"
    data = bytearray(open(prefix + ".data-00000-of-00001", "rb").read())
    data[-1] ^= 1                                   # inside the last numeric tensor
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(data))
    with pytest.raises(ValueError, match="checksum"):
        T.read_bundle(prefix, verify=True)


@pytest.mark.parametrize("cfg", ["tiny_gst", "tiny_nogst", "full"])
def test_reference_layout_roundtrip(cfg, tmp_path):
    """weights -> checkpoint in the reference's object paths -> weights: every manifest tensor comes back bit-identical,
    optimizer slots and bookkeeping entries are ignored, the object graph lists exactly the written variables."""
    hp = {"tiny_gst": synthetic.tiny_hp(), "tiny_nogst": synthetic.tiny_hp(gst=False), "full": hparams.load_hp()}[cfg]
    w = weights.synthetic_weights(hp, seed=3)
    d = str(tmp_path / "Checkpoint")
    prefix = T.save_reference_checkpoint(os.path.join(d, "S_38000.CHECKPOINT.H5-38"), hp, w)
    assert T.latest_checkpoint(d) == prefix and T.latest_checkpoint(str(tmp_path)) is None
    bundle = T.read_bundle(prefix, verify=(cfg != "full"))
    # what an optimizer adds (reference Model.py:186-189 checkpoints Adam too): must be skipped
    some = next(k for k in bundle if k.endswith("/kernel" + T.SUFFIX))
    bundle[some[:-len(T.SUFFIX)] + "/.OPTIMIZER_SLOT/optimizer/m" + T.SUFFIX] = np.zeros(3, np.float32)
    bundle["optimizer/iter" + T.SUFFIX] = np.asarray(7, np.int64)
    back = T.convert(bundle, hp)
    assert list(back) and set(back) == set(weights.manifest(hp))
    for k in back:
        assert np.array_equal(back[k], w[k]), k
    keys = T.object_graph_paths(bundle[T.OBJECT_GRAPH_KEY])
    assert set(keys) == {k for k in bundle if k.endswith(T.SUFFIX) and "OPTIMIZER" not in k and not k.startswith("optimizer/")}
    # the layout is the reference's: spot-check paths against the reference source (Taco2.py:16-43, 59-89; GST.py:77-90)
    paths = {k[:-len(T.SUFFIX)] for k in keys}
    assert "model/layer_with_weights-0/layer/layer_with_weights-0/embeddings" in paths
    assert any(p.endswith("layer_Dict/Decoder_Step/layer_Dict/RNN/cells/1/recurrent_kernel") for p in paths)
    assert any(p.endswith("layer_Dict/Decoder_Step/layer_Dict/Attention/attention_score_bias") for p in paths)
    assert any(p.endswith("/forward_layer/cell/kernel") for p in paths)
    if hp["GST"]["Use"]:
        assert any(p.endswith("layer_Dict/Reference_Encoder/layer_Dict/Conv2D_5/layer_with_weights-1/moving_variance") for p in paths)
        assert any(p.endswith("/gst_tokens") for p in paths)


def test_top_level_layers_are_found_in_any_order_and_mismatches_are_reported(tmp_path):
    hp = synthetic.tiny_hp()
    w = weights.synthetic_weights(hp, seed=4)
    prefix = T.save_reference_checkpoint(str(tmp_path / "c" / "ck-1"), hp, w)
    bundle = T.read_bundle(prefix)
    # Keras may number Encoder / Style_Token_Layer / Decoder / Vocoder differently: permute the top-level indices
    perm = {"0": "2", "1": "0", "2": "3", "3": "1"}
    shuffled = {}
    for k, v in bundle.items():
        if k.startswith("model/layer_with_weights-"):
            head, rest = k[len("model/layer_with_weights-"):].split("/", 1)
            k = "model/layer_with_weights-{}/{}".format(perm[head], rest)
        shuffled[k] = v
    back = T.convert(shuffled, hp)
    assert all(np.array_equal(back[k], w[k]) for k in w)
    broken = {k: v for k, v in bundle.items() if "attention_v" not in k}
    with pytest.raises(KeyError, match="decoder.attention.v"):
        T.convert(broken, hp)
    hp2 = synthetic.tiny_hp()
    hp2["Tacotron2"]["Decoder"]["RNN"]["Size"] = [32, 32]
    with pytest.raises(ValueError, match="shape"):
        T.convert(bundle, hp2)


def test_restore_accepts_a_reference_checkpoint(tmp_path, capsys):
    import torch
    from gst_tacotron_amd.model import GST_Tacotron
    hp = synthetic.tiny_hp()
    hp["Checkpoint_Path"] = str(tmp_path / "Checkpoint")
    m = GST_Tacotron(hyper_parameters=hp, max_batch=2, max_tokens=8, max_ref_frames=9)
    m.Restore()
    assert "There is no checkpoint." in capsys.readouterr().out                                 # Model.py:271-273
    w = weights.synthetic_weights(hp, seed=5)
    T.save_reference_checkpoint(os.path.join(hp["Checkpoint_Path"], "S_100.CHECKPOINT.H5-1"), hp, w)
    if torch.cuda.is_available():
        m.Restore()                                                                             # latest checkpoint, Model.py:268-276
        assert "is loaded" in capsys.readouterr().out
    else:
        from gst_tacotron_amd import capi
        with pytest.raises(capi.GstTacoError, match="no CPU fallback"):                       # parsed fine, then needs the GPU
            m.Restore()
