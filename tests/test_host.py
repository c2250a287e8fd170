"""CPU: host logic, the C-ABI surface (load + exported symbols + error behaviour, no compute calls), the Feeder
conventions, and the world-size-2 utterance sharding / gather over gloo."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from gst_tacotron_amd import capi, hparams, synthetic, weights
from gst_tacotron_amd.feeder import Feeder


@pytest.fixture(scope="module")
def lib():
    from gst_tacotron_amd import build
    build.build()
    return capi.load_library()


def test_header_symbols_are_all_exported(lib):
    header = open(os.path.join(ROOT, "include", "gsttaco.h")).read()
    declared = set(re.findall(r"\b(gsttaco_[a-z_0-9]+)\s*\(", header))
    assert declared == set(capi.EXPORTED_SYMBOLS)
    for sym in declared:
        assert getattr(lib, sym) is not None
    assert lib.gsttaco_abi_version() == capi.ABI_VERSION


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "gst_tacotron_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f
                assert "torch_ref" not in src and "oracle_np" not in src, f


def test_c_manifest_matches_python_manifest(lib):
    lsa = synthetic.tiny_hp()
    lsa["Tacotron2"]["Decoder"]["Attention"] = {"Type": "LSA", "Size": 16, "Conv": {"Filters": 8, "Kernel_Size": 7}}
    for hp in (hparams.load_hp(), synthetic.tiny_hp(gst=False), synthetic.config_hp("cfg2"), lsa):
        ctx = capi.Context(hp, max_batch=2, max_tokens=8, max_ref_frames=9)
        assert list(ctx.manifest().items()) == [(k, tuple(v)) for k, v in weights.manifest(hp).items()]
        ctx.close()


def test_create_rejects_bad_configs(lib):
    hp = hparams.load_hp()
    cfg = capi.make_config(hp)
    h = ctypes.c_void_p()
    cfg.abi_version = 99
    assert lib.gsttaco_create(ctypes.byref(cfg), ctypes.byref(h)) == -1
    cfg = capi.make_config(hp)
    cfg.att_type = 7
    assert lib.gsttaco_create(ctypes.byref(cfg), ctypes.byref(h)) == -1
    assert b"Unsupported attention type" in lib.gsttaco_last_error(None)       # reference Taco2.py:74-75
    cfg = capi.make_config(hp)
    cfg.heads = 3
    assert lib.gsttaco_create(ctypes.byref(cfg), ctypes.byref(h)) == -1
    assert b"divisible by num_heads" in lib.gsttaco_last_error(None)           # reference Layers.py:155-156
    cfg = capi.make_config(hp)
    cfg.mel_dim = 81
    assert lib.gsttaco_create(ctypes.byref(cfg), ctypes.byref(h)) == -1
    bad = hparams.load_hp()
    bad["Tacotron2"]["Decoder"]["Attention"]["Type"] = "DCA"
    with pytest.raises(ValueError, match="Unsupported attention type: DCA"):
        hparams.Dims(bad)


def test_weight_loading_errors_and_no_cpu_fallback(lib):
    import torch
    hp = synthetic.tiny_hp()
    ctx = capi.Context(hp, max_batch=2, max_tokens=8, max_ref_frames=9)
    w = weights.synthetic_weights(hp, 0)
    f32p = ctypes.POINTER(ctypes.c_float)
    a = w["decoder.attention.v"]
    shape = (ctypes.c_int64 * 1)(a.shape[0])
    assert lib.gsttaco_load_weight(ctx.handle, b"no.such.tensor", a.ctypes.data_as(f32p), shape, 1) == -4
    bad = (ctypes.c_int64 * 1)(a.shape[0] + 1)
    assert lib.gsttaco_load_weight(ctx.handle, b"decoder.attention.v", a.ctypes.data_as(f32p), bad, 1) == -4
    # compute before finalize -> E_WEIGHTS; finalize with a weight missing -> E_WEIGHTS
    assert lib.gsttaco_postnet(ctx.handle, None, 1, 1, None, None) == -4
    assert lib.gsttaco_finalize_weights(ctx.handle) == -4
    assert b"missing weight" in lib.gsttaco_last_error(ctx.handle)
    ctx.load_weights(w)
    if not torch.cuda.is_available():
        # the product path must fail loudly without a GPU: there is no CPU fallback
        assert lib.gsttaco_finalize_weights(ctx.handle) == -2
        assert b"no CPU fallback" in lib.gsttaco_last_error(ctx.handle)
        from gst_tacotron_amd.model import GST_Tacotron
        m = GST_Tacotron(hyper_parameters=hp, max_batch=2, max_tokens=8, max_ref_frames=9)
        with pytest.raises(capi.GstTacoError):
            m.Restore(weights=w)
        with pytest.raises(capi.GstTacoError):
            m.Inference_Step(np.zeros((1, 4), np.int32))
    ctx.close()


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(ImportError, match="no CPU fallback"):
        capi.load_library(str(tmp_path / "libgsttaco.so"))


def test_dims_of_the_reference_defaults():
    d = hparams.Dims(hparams.load_hp())
    assert (d.vocab, d.emb, d.enc_out, d.mem, d.proj_out, d.steps) == (34, 512, 512, 640, 81, 1000)
    assert d.sigmoid_noise == 2.0 and d.att_type == "SMA" and d.post_tanh == 3 and d.gru_in == 256
    d2 = hparams.Dims(synthetic.config_hp("cfg2"))
    assert d2.proj_out == 161 and d2.steps == 500
    man = weights.manifest(synthetic.config_hp("cfg2"))
    n = sum(int(np.prod(s)) for k, s in man.items() if not k.startswith("vocoder."))
    assert 24.5e6 < n < 25.5e6            # ~24.9 M parameters (vocoder excluded)
    nv = sum(int(np.prod(s)) for k, s in man.items() if k.startswith("vocoder."))
    assert 2.5e6 < nv < 3.2e6             # CBHG Vocoder_Taco1 (row N1)
    assert (d2.spec, d2.bank_count, d2.bank_filters, d2.highway_count, d2.voc_rnn) == (513, 8, 256, 4, 256)


def test_feeder_inference_pattern_conventions():
    hp = hparams.load_hp()
    f = Feeder(hp)
    sents = ["Strike while the iron is hot. ", "birds, of a feather?"]
    mel = np.full((7, 80), 0.5, np.float32)
    pat = f.Get_Inference_Pattern(sents, [mel])
    tok = pat["tokens"]
    assert tok.dtype == np.int32 and tok.shape == (2, len("STRIKE WHILE THE IRON IS HOT.") + 2)
    assert tok[0, 0] == 0 and tok[0, -1] == 1                       # <S> ... <E>
    assert tok[0, 1] == f.token_Index_Dict["S"] and tok[0, 2] == f.token_Index_Dict["T"]
    n1 = len("BIRDS, OF A FEATHER?") + 2
    assert list(pat["token_lengths"]) == [tok.shape[1], n1]
    assert np.all(tok[1, n1:] == 1)                                 # padded with <E> = 1 (Feeder.py:177-180)
    assert pat["initial_mels"].shape == (2, 1, 80) and not pat["initial_mels"].any()
    g = pat["mels_for_gst"]
    assert g.shape == (2, 8, 80) and not g[:, 0].any() and np.all(g[:, 1:] == 0.5)   # zero frame prepended
    assert list(pat["mel_lengths_for_gst"]) == [7, 7]
    with pytest.raises(KeyError):
        f.Get_Inference_Pattern(["naïve"], [mel])                   # out-of-vocabulary (Feeder.py:169)
    assert f.Get_Inference_Pattern(sents, None) is None             # "GST is enabled, but no wav information."
    assert f.Get_Inference_Pattern(sents, [mel, mel, mel]) is None
    ragged = f.Get_Inference_GST_Pattern([np.ones((3, 80)), np.ones((5, 80))])
    assert ragged["mels_for_gst"].shape == (2, 6, 80) and not ragged["mels_for_gst"][0, 4:].any()


def test_feeder_matches_reference_generated_fixture():
    """SURVEY row A0 against the REFERENCE's own Feeder.Get_Inference_Pattern / Get_Inference_GST_Pattern
    (Feeder.py:161-252), executed in the build container by oracle/gen_golden_feeder.py on the 8 sentences of the
    reference's Inference_Sentence_for_Training.txt -- the first fixture in tests/golden that the reference, not the
    oracle, produced.  Bit-exact (integer / copy work)."""
    import io
    import contextlib
    import json
    import torch
    z = np.load(os.path.join(ROOT, "tests", "golden", "feeder_tokens.npz"))
    sentences = [str(s) for s in z["sentences"]]
    assert len(sentences) == 8
    bank = {k[4:]: z[k] for k in z.files if k.startswith("mel_ref")}
    wavs = ["ref{}.wav".format(i) for i in range(8)]
    calls = []

    def frontend(wav_List, top_db):
        # what GST_Tacotron.Mel_Generate hands back: the mels_for_gst layout (zero frame 0, zero padding) + lengths
        calls.extend((w, int(top_db)) for w in wav_List)
        mels = [bank[w] for w in wav_List]
        out = np.zeros((len(mels), max(m.shape[0] for m in mels) + 1, 80), np.float32)
        for i, m in enumerate(mels):
            out[i, 1:m.shape[0] + 1] = m
        return torch.from_numpy(out), torch.tensor([m.shape[0] for m in mels], dtype=torch.int32)

    def same(pat, prefix, keys):
        for k in keys:
            got = pat[k].numpy() if hasattr(pat[k], "numpy") else pat[k]
            want = z[prefix + "." + k]
            assert got.dtype == want.dtype and got.shape == want.shape, (k, got.dtype, got.shape, want.dtype, want.shape)
            assert np.array_equal(got, want), k

    base = ("tokens", "token_lengths", "initial_mels")
    gst = base + ("mels_for_gst", "mel_lengths_for_gst")
    hp_off = hparams.load_hp()
    hp_off["GST"]["Use"] = False
    assert json.loads(str(z["token_dict_json"])) == Feeder(hp_off).token_Index_Dict
    same(Feeder(hp_off).Get_Inference_Pattern(sentences), "nogst", base)

    f = Feeder(hparams.load_hp(), mel_frontend=frontend)
    same(f.Get_Inference_Pattern(sentences, [wavs[1]]), "gst_one", gst)
    assert calls == [(w, db) for w, db, _ in json.loads(str(z["gst_one.calls"]))]          # top_db 60 (Feeder.py:205)
    calls.clear()
    same(f.Get_Inference_Pattern(sentences, wavs), "gst_many", gst)
    assert calls == [(w, db) for w, db, _ in json.loads(str(z["gst_many.calls"]))]         # top_db 15 (Feeder.py:209)
    calls.clear()
    same(f.Get_Inference_GST_Pattern(wavs[2:6]), "gst_only", ("mels_for_gst", "mel_lengths_for_gst"))
    assert calls == [(w, db) for w, db, _ in json.loads(str(z["gst_only.calls"]))]         # top_db 60 (Feeder.py:232)
    # the precomputed-mel entry (no GPU): same layout from the arrays themselves
    same(f.Get_Inference_Pattern(sentences, [bank[w] for w in wavs]), "gst_many", gst)
    same(f.Get_Inference_Pattern(sentences, [bank[wavs[1]]]), "gst_one", gst)
    # error behaviour: the reference's messages and None (Feeder.py:197-202), KeyError on OOV (:169)
    for args, key in (((sentences, None), "err.no_wav_message"), ((sentences, wavs[:3]), "err.bad_count_message")):
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            assert f.Get_Inference_Pattern(*args) is None
        assert buf.getvalue() == str(z[key])
    assert bool(z["err.oov_raises_keyerror"])


def test_shard_bounds_cover_everything():
    from gst_tacotron_amd.distributed import shard_bounds
    for n in (1, 7, 32, 256, 257):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


_GLOO_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from gst_tacotron_amd import distributed as gd
rank, local_rank, world = gd.init_process_group(backend="gloo")
assert world == 2
n_total = 5                                   # ragged: 3 + 2 utterances
lo, hi = gd.shard_bounds(n_total, rank, world)
full = torch.arange(n_total * 4 * 3, dtype=torch.float32).reshape(n_total, 4, 3)
inputs = gd.shard_inputs({"tokens": full, "none": None}, rank, world)
assert inputs["none"] is None and inputs["tokens"].shape[0] == hi - lo
local = inputs["tokens"] * 2.0                # stands in for the per-rank Inference_Step output
out = gd.gather_to_root(local, n_total=n_total)
out2 = gd.gather_to_root(local)               # size discovered with an all_reduce
pend = gd.gather_to_root(local, n_total=n_total, async_op=True)     # next batch's work would be enqueued here
out3 = pend.result()
if rank == 0:
    assert torch.equal(out, full * 2.0) and torch.equal(out2, full * 2.0) and torch.equal(out3, full * 2.0)
    print("GLOO_OK")
else:
    assert out is None and out2 is None and out3 is None
dist.barrier()
dist.destroy_process_group()
'''


def test_two_rank_gloo_shard_and_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_GLOO_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29611")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2")
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=e, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=120)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "GLOO_OK" in outs[0]


def test_spawn_local_ranks_runs_the_gloo_worker_and_forwards_failures(tmp_path):
    """bench.py --gpus N without a launcher starts its own rank processes (gst_tacotron_amd.distributed.spawn_local_ranks):
    the same 2-rank shard + gather as above through the spawner, and a failing rank ends the job with its status."""
    from gst_tacotron_amd.distributed import spawn_local_ranks
    script = tmp_path / "worker.py"
    script.write_text(_GLOO_WORKER)
    assert spawn_local_ranks(2, [sys.executable, str(script), ROOT]) == 0
    bad = tmp_path / "bad.py"
    bad.write_text("import os, sys, time\nr = int(os.environ['RANK'])\ntime.sleep(0 if r else 60)\nsys.exit(7 if r else 0)\n")
    assert spawn_local_ranks(2, [sys.executable, str(bad)]) == 7


def test_bench_refuses_a_world_size_that_disagrees_with_gpus():
    env = dict(os.environ, WORLD_SIZE="4", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE is 4" in r.stderr and not r.stdout.strip()


def test_bench_prices_the_decode_step_at_the_surveys_bytes():
    """bench.py's roofline arithmetic is SURVEY.md section 8(d)'s, not its own: 57.88 MB of decoder weights + 3.2 MB of activation rows =
    61.1 MB per decode step at configs[1], 5.9 ms ideal for the whole Inference_Step; bf16 weights halve the weight share; what a
    persistent launch MUST move is far less (weights once per launch)."""
    import bench
    from gst_tacotron_amd import synthetic
    from gst_tacotron_amd.hparams import Dims, load_hp, load_token_dict
    hp = load_hp(synthetic.config_hp("cfg2"))
    d = Dims(hp, vocab=len(load_token_dict(hp)))
    total, w, a = bench.survey_step_bytes(d, 32, 128, False)
    assert w == 4 * (86528 + 32896 + 129 + 5771264 + 8392704 + 1152 * 161 + 161)      # SURVEY 8(d)'s own sum (prenet, query, v + bias, LSTM 1, LSTM 2, projection)
    assert abs(w / 1e6 - 57.88) < 0.01 and abs(a / 1e6 - 3.2) < 0.05 and abs(total / 1e6 - 61.1) < 0.05
    assert bench.survey_step_bytes(d, 32, 128, True)[1] * 2 == w
    assert abs(bench.ideal_ms(d, 32, 128, 256, False, 500) - 5.9) < 0.05
    comp = bench.persistent_compulsory_bytes(d, 32, 128, False, 500)
    assert w < comp < 500 * total / 50


def test_run_steps_claims_each_gather_before_the_next_batch_unless_told_to_overlap():
    """N > 1 default of bench.py (gst_tacotron_amd.distributed.run_steps): batch i's gather is claimed BEFORE batch i + 1 is enqueued,
    so a persistent decode launch (256 workgroups that must all be resident) is stream-ordered behind the previous gather and can
    never meet a resident receive kernel; --overlap-gather 1 keeps the round-5 form (claimed one batch later)."""
    from gst_tacotron_amd.distributed import run_steps
    for overlap, want in ((False, "e0 c0 e1 c1 e2 c2"), (True, "e0 e1 c0 e2 c1 c2")):
        log = []

        class P:
            def __init__(self, i):
                self.i = i

            def result(self):
                log.append("c%d" % self.i)
                return self.i

        def one_step(i):
            log.append("e%d" % i)
            return P(i)
        assert run_steps(3, one_step, overlap_gather=overlap) == 2
        assert " ".join(log) == want


def test_bench_dry_run_on_two_cpu_ranks():
    """`bench.py --gpus 2 --dry-run`: the self-spawned ranks, a gloo process group, the step / gather ordering, the all-reduced
    give-up flag and rank 0's line -- without a GPU.  The default ordering is the safe one; the line says which was used."""
    import json
    for extra, gather, gave_up in (([], "claimed before the next batch is enqueued", False),
                                   (["--overlap-gather", "1", "--inject-give-up"], "overlapped", True)):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1"] + extra,
                           capture_output=True, text=True, timeout=300, env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE")})
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout            # ONE line, from rank 0
        d = json.loads(lines[0])
        assert d["dry_run"] is True and d["value"] is None and d["n_gpus"] == 2 and d["config"]["gather"] == gather
        assert d["fallback_taken"] is gave_up       # (raised on the LAST rank only: rank 0 knows through the all-reduce)
