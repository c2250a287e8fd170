"""GPU parity: the HIP path (through the C-ABI) against the committed golden vectors and the oracle."""
import numpy as np
import pytest

from conftest import GOLDEN_CASES, load_golden

pytestmark = pytest.mark.gpu

# fp32 bar from BASELINE.json north_star: mel max-abs diff <= 1e-3 vs the reference on identical inputs.
# The fp32-vs-fp64 noise floor of the oracle itself is ~2e-6 over 500 full-size steps, so the tests hold the
# HIP path to a 20x tighter bound than the bar.
TOL = 5e-5
BAR = 1e-3


def _model(hp, w, B, Tv, Tref1):
    from gst_tacotron_amd.model import GST_Tacotron
    m = GST_Tacotron(hyper_parameters=hp, max_batch=max(B, 1), max_tokens=Tv, max_ref_frames=max(Tref1, 2))
    m.Restore(weights=w)
    return m


def _sole_context():
    """The persistent decode launch is taken only while the process has ONE live context: drop what earlier tests left behind."""
    import gc
    gc.collect()


def _assert_persistent_decode(m, calls=1):
    """The test means the persistent decode launch (csrc/persist_decode.hip): fail, do not fall back silently to the launch path."""
    n, on = m.decode_counters()
    assert on == 1 and n >= calls, ("the persistent decode launch was not taken", n, on, m.last_message())
    assert m.handoff_error() == 0


def _run_case(name, **kw):
    import torch
    hp, w, g = load_golden(name)
    B, Tv = g["tokens"].shape
    gst = bool(hp["GST"]["Use"])
    Tref1 = g["mels_for_gst"].shape[1] if gst else 0
    _sole_context()
    m = _model(hp, w, B, Tv, Tref1)
    out = m.Inference_Step(
        g["tokens"], g["token_lengths"], None,
        g["mels_for_gst"] if gst else None, g["mel_lengths_for_gst"] if gst else None,
        prenet_masks=g["prenet_masks"], attn_noise=g["attn_noise"], steps=int(g["steps"]), return_pre_mel=True, **kw)
    torch.cuda.synchronize()
    mel, stop, spec, align, pre = out
    assert (spec is None) == (not kw.get("with_vocoder"))
    if name.startswith("full_"):        # the reference's decoder sizes, <= 32 utterances, <= 256 tokens: the persistent decode launch
        _assert_persistent_decode(m)
    if spec is not None:
        g = dict(g); g["_spec"] = spec.cpu().numpy()
    return g, mel.cpu().numpy(), stop.cpu().numpy(), align.cpu().numpy(), pre.cpu().numpy(), m


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_inference_step_matches_golden(name):
    g, mel, stop, align, pre, _ = _run_case(name)
    for what, got, exp in (("alignments", align, g["alignments"]), ("pre_mel", pre, g["pre_mel"]),
                           ("stops", stop, g["stops"]), ("mels", mel, g["mels"])):
        err = np.abs(got - exp).max()
        print(name, what, "max abs err", err)
        assert got.shape == exp.shape
        assert np.isfinite(got).all()
        assert err <= TOL, (name, what, err)
        assert err <= BAR


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_inference_step_with_vocoder_matches_golden(name):
    """spectrogram_Logits of Inference_Step (CBHG Vocoder_Taco1 on the post-net mels, Model.py:126-129)."""
    g, mel, stop, align, pre, _ = _run_case(name, with_vocoder=True)
    assert g["_spec"].shape == g["spectrograms"].shape
    err = np.abs(g["_spec"] - g["spectrograms"]).max()
    print(name, "spectrograms max abs err", err)
    assert np.isfinite(g["_spec"]).all() and err <= TOL
    assert np.abs(mel - g["mels"]).max() <= TOL


@pytest.mark.parametrize("B,T", [(1, 1), (3, 7), (2, 130), (17, 33)])
def test_vocoder_alone_matches_oracle(B, T):
    """gsttaco_vocoder on caller-supplied mels at the reference's full CBHG dimensions (8 banks x 256, 513 bins):
    1 frame, fewer frames than the widest bank kernel, more than one 128-row GEMM tile, a ragged batch."""
    import torch
    from gst_tacotron_amd import synthetic, weights
    from oracle import oracle_np
    hp = synthetic.config_hp("cfg2")
    w = weights.synthetic_weights(hp, seed=3)
    rng = np.random.default_rng(B * 100 + T)
    mel = np.clip(rng.normal(0, 1.5, (B, T, 80)), -4, 4).astype(np.float32)
    m = _model(hp, w, B, 8, 2)
    spec = m.vocoder(mel)
    torch.cuda.synchronize()
    ref = oracle_np.vocoder_taco1(hp, oracle_np.cast_weights(w, np.float64), mel.astype(np.float64), np.float64)
    err = np.abs(spec.cpu().numpy() - ref).max()
    print("vocoder", B, T, "max abs err", err, "scale", np.abs(ref).max())
    assert spec.shape == (B, T, 513) and err <= TOL


# ------------------------------------------------------------------ per-module parity against the oracle
def _full_case(B, Tv, Tref, steps, seed, att="SMA", lens=None, rate=0.5):
    from gst_tacotron_amd import synthetic, weights
    hp = synthetic.config_hp("cfg2")
    if att.startswith("LSA"):           # "LSA" or "LSA/filters/kernel[/s]" (s: the smoothing normalisation)
        p = att.split("/")
        hp["Tacotron2"]["Decoder"]["Attention"] = {"Type": "LSA", "Size": 128, "Conv": {"Filters": int(p[1]) if len(p) > 1 else 32,
                                                   "Kernel_Size": int(p[2]) if len(p) > 2 else 31}, "Smoothing": len(p) > 3}
    else:
        hp["Tacotron2"]["Decoder"]["Attention"]["Type"] = att
    hp["Tacotron2"]["Decoder"]["Prenet"]["Dropout_Rate"] = rate
    w = weights.synthetic_weights(hp, seed=0)
    rng = np.random.default_rng(seed)
    tokens, tl = synthetic.make_tokens(rng, B, Tv)
    mels, ml = synthetic.make_ref_mels(rng, B, Tref, lengths=lens)
    masks, noise = synthetic.make_randomness(rng, steps, B, Tv, [256, 256], rate=max(rate, 1e-9))
    return hp, w, tokens, tl, mels, ml, masks, noise


def test_encoder_gst_decode_postnet_modules_match_oracle():
    import torch
    from oracle import oracle_np
    B, Tv, Tref, steps = 3, 21, 150, 6
    hp, w, tokens, tl, mels, ml, masks, noise = _full_case(B, Tv, Tref, steps, seed=21, lens=np.array([150, 64, 65]))
    m = _model(hp, w, B, Tv, Tref + 1)
    w64 = oracle_np.cast_weights(w, np.float64)
    enc = m.encode(tokens)
    gst = m.Inference_GST_Step(mels, ml)
    torch.cuda.synchronize()
    enc_ref = oracle_np.encoder(hp, w64, tokens, np.float64)
    gst_ref = oracle_np.style_token_layer(hp, w64, mels, ml, np.float64)
    assert np.abs(enc.cpu().numpy() - enc_ref).max() <= TOL
    assert np.abs(gst.cpu().numpy() - gst_ref).max() <= TOL
    # decode from the ORACLE's encoder/gst so each module is checked at its own scale
    pre, stop, align = m.decode(enc_ref.astype(np.float32), gst_ref.astype(np.float32), masks, noise, steps=steps)
    torch.cuda.synchronize()
    mem = oracle_np.gst_concat(enc_ref, gst_ref)
    pre_ref, stop_ref, align_ref = oracle_np.decoder(hp, w64, mem, np.float64, masks.astype(np.float64),
                                                     noise.astype(np.float64), steps=steps)
    assert np.abs(pre.cpu().numpy() - pre_ref).max() <= TOL
    assert np.abs(stop.cpu().numpy() - stop_ref).max() <= TOL
    assert np.abs(align.cpu().numpy() - align_ref).max() <= TOL
    post = m.postnet(pre_ref.astype(np.float32))
    torch.cuda.synchronize()
    assert np.abs(post.cpu().numpy() - oracle_np.postnet(hp, w64, pre_ref, np.float64)).max() <= TOL


@pytest.mark.parametrize("B,Tv,Tref,att", [(1, 32, 3, "SMA"), (17, 40, 70, "BMA"), (33, 24, 64, "SMA"),
                                            (2, 200, 40, "BMA"), (2, 300, 40, "SMA")])
def test_edge_shapes_match_oracle(B, Tv, Tref, att):
    """Batch 1 / not a multiple of 16 / more than one 32-row chunk; token counts that need 2 and 3 passes of the
    attention rows (beyond the rows kept in registers); a 3-frame reference mel (one compressed GRU step)."""
    import torch
    from oracle import oracle_np
    steps = 5
    lens = np.maximum(1, np.minimum(Tref, np.arange(B) * 7 + Tref // 2)).astype(np.int32)
    hp, w, tokens, tl, mels, ml, masks, noise = _full_case(B, Tv, Tref, steps, seed=B * 1000 + Tv, att=att, lens=lens)
    m = _model(hp, w, B, Tv, Tref + 1)
    mel, stop, _, align = m.Inference_Step(tokens, tl, None, mels, ml, prenet_masks=masks, attn_noise=noise, steps=steps)
    torch.cuda.synchronize()
    ref = oracle_np.inference_step(hp, w, tokens, mels, ml, masks, noise, steps=steps, dt=np.float64)
    assert np.abs(mel.cpu().numpy() - ref[0]).max() <= TOL
    assert np.abs(stop.cpu().numpy() - ref[1]).max() <= TOL
    assert np.abs(align.cpu().numpy() - ref[3]).max() <= TOL


def test_eager_graph_and_unfused_paths_agree(monkeypatch):
    """The hipGraph replay, eager launches and the unfused (4-kernel) decoder front end are the same arithmetic."""
    import torch
    hp, w, tokens, tl, mels, ml, masks, noise = _full_case(4, 33, 90, 8, seed=5)
    outs = []
    for env in ({}, {"GSTTACO_GRAPH": "0"}, {"GSTTACO_FUSED_FRONT": "0"}):      # (FUSED_FRONT: 0 = four kernels, 1 = general fused, 2 = lean)
        for k in ("GSTTACO_GRAPH", "GSTTACO_FUSED_FRONT"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        m = _model(hp, w, 4, 33, 91)
        mel, stop, _, align = m.Inference_Step(tokens, tl, None, mels, ml, prenet_masks=masks, attn_noise=noise, steps=8)
        torch.cuda.synchronize()
        outs.append((mel.cpu().numpy(), align.cpu().numpy()))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])   # graph == eager, bitwise
    assert np.abs(outs[0][0] - outs[2][0]).max() <= TOL and np.abs(outs[0][1] - outs[2][1]).max() <= TOL


@pytest.mark.parametrize("env", [{"GSTTACO_FUSED_FRONT": "1"}, {"GSTTACO_FUSED_FRONT": "2", "GSTTACO_PERSIST_DECODE": "0"}, {"GSTTACO_FUSED_FRONT": "2"}])
@pytest.mark.parametrize("att", ["SMA", "BMA"])
def test_front_end_variants_match_oracle(monkeypatch, env, att):
    """The decode step's front end exists as the general fused kernel (GSTTACO_FUSED_FRONT=1), as the fused kernel with the lean
    utterance path (front_lean.h, buffer loads with counted waits; 2 with GSTTACO_PERSIST_DECODE=0: the launch path's default) and
    inside the persistent decode launch (persist_decode.hip, the default at these shapes): the same function.  Each against the
    float64 oracle over 40 steps at full dimensions, injected and hashed (throughput-mode) dropout, 5 and 32 utterances."""
    import torch
    from oracle import oracle_np
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    persistent = env == {"GSTTACO_FUSED_FRONT": "2"}
    for B in (5, 32):
        steps, Tv, Tref = 40, 48, 80
        hp, w, tokens, tl, mels, ml, masks, noise = _full_case(B, Tv, Tref, steps, seed=71 + B, att=att)
        m = None
        _sole_context()
        m = _model(hp, w, B, Tv, Tref + 1)
        mel, stop, _, align = m.Inference_Step(tokens, tl, None, mels, ml, prenet_masks=masks, attn_noise=noise, steps=steps)
        torch.cuda.synchronize()
        if persistent:
            _assert_persistent_decode(m)
        else:
            assert m.decode_counters()[0] == 0
        ref = oracle_np.inference_step(hp, w, tokens, mels, ml, masks, noise, steps=steps, dt=np.float64)
        assert np.abs(mel.cpu().numpy() - ref[0]).max() <= TOL
        assert np.abs(stop.cpu().numpy() - ref[1]).max() <= TOL
        assert np.abs(align.cpu().numpy() - ref[3]).max() <= TOL
        # throughput mode: masks / noise from the seed, read back and fed to the oracle
        mel2, _, _, align2 = m.Inference_Step(tokens, tl, None, mels, ml, seed=99, steps=steps)
        torch.cuda.synchronize()
        mk, nz = m.debug_randomness(steps, B, Tv)
        from gst_tacotron_amd import hparams
        ref2 = oracle_np.inference_step(hp, w, tokens, mels, ml, mk, nz if hparams.attention_sigmoid_noise(hp) > 0 else None,
                                        steps=steps, dt=np.float64)
        assert np.abs(mel2.cpu().numpy() - ref2[0]).max() <= TOL
        assert np.abs(align2.cpu().numpy() - ref2[3]).max() <= TOL
        if persistent:
            _assert_persistent_decode(m, 2)


@pytest.mark.parametrize("mixed,B", [(False, 5), (True, 5), (False, 37), (True, 37), (False, 65), (True, 65), (False, 128), (True, 128)])
def test_lean_and_general_decode_kernels_are_the_same_arithmetic(monkeypatch, mixed, B):
    """csrc/lean_body.h restates the general skinny GEMM body for the decode shapes with K fixed at compile time: same
    k-block-to-wave assignment, same summation order, so the decode loop (LSTM input halves, projection, recurrent-half
    workers) is BITWISE the general kernels' result at full dimensions; the lean encoder BiLSTM hoists its input halves
    into one GEMM (different summation order) and agrees within the parity tolerance."""
    import torch
    # (B = 37: two 32-row chunks per tile, a partial 16-row M-tile, worker jobs looping over the chunks; B = 65 / 128: the
    # multi-chunk bodies that keep a tile's weights in registers over 3 / 4 chunks, an odd number of M-tiles, at 128 in fp32 the
    # utterance workgroups taking recurrent-half jobs after their chain, and two slabs of the persistent encoder BiLSTM)
    hp, w, tokens, tl, mels, ml, masks, noise = _full_case(B, 40, 90, 12, seed=9)
    hp = dict(hp); hp["Use_Mixed_Precision"] = bool(mixed)        # bf16 operands: the lean bf16 bodies, same claim
    w64 = None
    outs = []
    for lean in ("1", "0"):
        monkeypatch.setenv("GSTTACO_LEAN", lean)
        m = _model(hp, w, B, 40, 91)
        enc = m.encode(tokens)
        gst = m.Inference_GST_Step(mels, ml)
        torch.cuda.synchronize()
        if w64 is None:
            w64 = (enc.cpu().numpy().copy(), gst.cpu().numpy().copy())     # decode both variants from the SAME memory
        pre, stop, align = m.decode(w64[0], w64[1], masks, noise, steps=12)
        torch.cuda.synchronize()
        outs.append((enc.cpu().numpy(), pre.cpu().numpy(), stop.cpu().numpy(), align.cpu().numpy()))
    monkeypatch.delenv("GSTTACO_LEAN")
    # encoder: hoisted vs fused input half (another summation order; under bf16 operands a rounding can flip: mixed tolerance)
    assert np.abs(outs[0][0] - outs[1][0]).max() <= (MIXED_TOL if mixed else TOL)
    for a, b in zip(outs[0][1:], outs[1][1:]):
        assert np.array_equal(a, b)                                          # decode loop: bitwise


@pytest.mark.parametrize("B,mixed", [(5, False), (32, False), (70, False), (128, False), (64, True)])
def test_fused_lstm_launch_is_bitwise_the_two_launches(monkeypatch, B, mixed):
    """Both decode LSTM cells run as ONE launch (skinny_gemm.hip gt_lstm12_kernel; above 32 rows gt_lstm12_mc_kernel on the
    multi-chunk bodies, fp32 and bf16): layer 2's weights are requested before the
    workgroup waits for the other tiles' h1, which is handed over in-kernel (write-through stores, sharded arrival counter,
    sc1 loads).  Same arithmetic in the same order as the two launches (GSTTACO_FUSED_LSTM=0): bitwise equal over 60 steps at
    full dimensions, repeated calls (the per-step counters are re-zeroed), no give-up; and against the float64 oracle."""
    import torch
    from oracle import oracle_np
    steps, Tv, Tref = (60, 40, 70) if B <= 32 else (12, 24, 40)
    hp, w, tokens, tl, mels, ml, masks, noise = _full_case(B, Tv, Tref, steps, seed=17 + B)
    hp = dict(hp); hp["Use_Mixed_Precision"] = bool(mixed)      # (above 32 rows the fused launch also exists on bf16 operands)
    outs = {}
    monkeypatch.setenv("GSTTACO_PERSIST_DECODE", "0")           # (the launch path: at <= 32 rows the default is the persistent launch)
    for flag in ("1", "0"):
        monkeypatch.setenv("GSTTACO_FUSED_LSTM", flag)
        m = _model(hp, w, B, Tv, Tref + 1)
        for rep in range(2):
            mel, stop, _, align = m.Inference_Step(tokens, tl, None, mels, ml, prenet_masks=masks, attn_noise=noise, steps=steps)
        m.synchronize()
        assert m.handoff_error() == 0
        outs[flag] = (mel.cpu().numpy(), stop.cpu().numpy(), align.cpu().numpy())
        del m
    monkeypatch.delenv("GSTTACO_FUSED_LSTM")
    for a, b in zip(outs["1"], outs["0"]):
        assert np.array_equal(a, b)
    if not mixed and B <= 70:
        ref = oracle_np.inference_step(hp, w, tokens, mels, ml, masks, noise, steps=steps, dt=np.float64)
        assert np.abs(outs["1"][0] - ref[0]).max() <= TOL and np.abs(outs["1"][2] - ref[3]).max() <= TOL


@pytest.mark.parametrize("persist,B,mixed,Tv,att", [("0", 4, False, 24, "SMA"), ("1", 4, False, 24, "SMA"), ("1", 40, False, 24, "SMA"), ("1", 40, True, 24, "SMA"),
                                                    ("1", 128, False, 256, "SMA"), ("1", 7, False, 40, "LSA/20/9")])
def test_fused_lstm_give_up_is_reported_and_the_next_call_recovers(monkeypatch, persist, B, mixed, Tv, att):
    """The fused LSTM launch's wait (persist = 0) and the persistent decode launch's waits (1) are bounded: with one arrival too
    many expected (fault injection) every workgroup runs into the bound, the call's outputs are invalid and ``synchronize``
    says so; the next call uses the next launch form down (persistent -> fused -> two launches) and is correct."""
    import gc
    import time
    import torch
    from gst_tacotron_amd.capi import GstTacoError
    gc.collect()                                    # (both launch forms are taken only while the process has ONE live context)
    monkeypatch.setenv("GSTTACO_PERSIST_DECODE", persist)
    # (B = 40: the group kernel, two groups of rows; 128 x 256 tokens: four groups, the compact LDS layout; LSA: the LSA chain -- the give-up
    # leaves a failed step's remaining phases running on whatever the wait left: no address or loop bound may depend on it)
    hp, w, tokens, tl, mels, ml, masks, noise = _full_case(B, Tv, 40, 3, seed=8, att=att)
    hp = dict(hp); hp["Use_Mixed_Precision"] = bool(mixed)                              # (mixed: the bf16 kernel and its helpers' hand-back)
    m = _model(hp, w, B, Tv, 41)
    assert m.decode_counters()[1] == int(persist)
    ref = m.Inference_Step(tokens, tl, None, mels, ml, prenet_masks=masks, attn_noise=noise, steps=3)[0].cpu().numpy()
    m.synchronize()
    assert m.decode_counters()[0] == int(persist)
    m.ctx.check(m.ctx.lib.gsttaco_debug_raise_handoff_error(m.ctx.handle, 1 << 16))
    t0 = time.perf_counter()
    m.Inference_Step(tokens, tl, None, mels, ml, prenet_masks=masks, attn_noise=noise, steps=3)
    with pytest.raises(GstTacoError, match="gave up"):
        m.synchronize()
    assert time.perf_counter() - t0 < 10.0
    out = m.Inference_Step(tokens, tl, None, mels, ml, prenet_masks=masks, attn_noise=noise, steps=3)[0].cpu().numpy()
    m.synchronize()
    if mixed:       # (the injected fault also drops a member of the encoder's persistent BiLSTM; its per-step bf16 fallback sums in another order)
        assert np.abs(out - ref).max() <= MIXED_TOL
    else:
        assert np.array_equal(out, ref)
    assert "warning" in m.last_message() and m.handoff_error() == 0
    assert m.decode_counters()[1] == 0


PERSIST_CASES = [
    # the one-group kernel (<= 32 rows): 256 workgroups, helpers for the chain workgroups' recurrent halves
    (32, 128, "SMA", "hashed", 150, {}), (32, 128, "BMA", "hashed", 150, {}), (5, 40, "SMA", "injected", 150, {}),
    (17, 100, "BMA", "injected", 150, {}), (9, 77, "SMA", "masked", 150, {}), (16, 128, "SMA", "rate25", 150, {}), (3, 20, "SMA", "nodrop", 150, {}),
    # more than 128 tokens (round 5): the utterance's whole processed memory in LDS up to 256 rows, the context summed in the launch
    # path's chunk order; 187 = the reference's own 8-sentence inference batch (Inference_Sentence_for_Training.txt, Feeder.py:161-180)
    (32, 129, "BMA", "injected", 100, {}), (8, 187, "SMA", "hashed", 150, {}), (20, 256, "BMA", "hashed", 100, {}), (7, 256, "SMA", "masked", 100, {}),
    # groups of 32 rows through one set of resident weights (round 5): 2 groups up to 64 rows, 4 up to 128; partial last groups and M-tiles
    (37, 60, "SMA", "hashed", 100, {}), (64, 128, "BMA", "injected", 60, {}), (65, 50, "SMA", "masked", 100, {}), (128, 128, "SMA", "hashed", 100, {}),
    (128, 256, "BMA", "masked", 60, {}), (100, 187, "SMA", "rate25", 60, {}),
    # mixed precision (round 5): the bf16 kernel -- one group of up to 64 rows, activations as bf16 mirrors only -- against the bf16 launch path
    (64, 128, "SMA", "hashed", 100, {"MIXED": "1"}), (40, 60, "BMA", "injected", 100, {"MIXED": "1"}), (20, 150, "SMA", "masked", 100, {"MIXED": "1"}),
    (57, 100, "SMA", "rate25", 60, {"MIXED": "1"}), (5, 33, "BMA", "nodrop", 100, {"MIXED": "1"}),
    # the step-wise LSA extension in the chain (round 5: the one-group kernel, up to 128 tokens; both location products on the fp32 matrix
    # pipe as in dec_front_lsa.hip, the score's channel halves summed as that kernel's two waves sum them): against the launch path's
    # fused front end; filter / tap counts off the MFMA granules; smoothing; masked
    (32, 128, "LSA", "hashed", 100, {}), (11, 90, "LSA", "injected", 100, {}), (20, 128, "LSA/20/9/s", "masked", 60, {}), (7, 33, "LSA/8/7", "nodrop", 100, {}),
    # 17..32 rows as two groups of 16 (GSTTACO_PERSIST_SPLIT16=1: the measured alternative to the helpers of the one-group kernel)
    (32, 128, "SMA", "hashed", 100, {"GSTTACO_PERSIST_SPLIT16": "1"}), (23, 70, "BMA", "injected", 100, {"GSTTACO_PERSIST_SPLIT16": "1"}),
]


@pytest.mark.parametrize("B,Tv,att,mode,steps,env", PERSIST_CASES)
def test_persistent_decode_launch_is_bitwise_the_launches(monkeypatch, B, Tv, att, mode, steps, env):
    """The whole decoder loop as ONE persistent launch (csrc/persist_decode.hip: every GEMM weight resident in registers, the
    utterances' processed memory in LDS, in-kernel hand-offs with bounded waits) against the launch path
    (GSTTACO_PERSIST_DECODE=0: fused front launch + fused LSTM launch + projection launch per step): the SAME arithmetic in
    the same order -- mel, stop and alignment outputs bitwise equal over 60-150 steps at full dimensions, which is also the test
    that no in-kernel hand-off ever delivers a stale word.  Hashed (throughput-mode) and injected dropout, a rate the hash
    does not cover (masks from the buffer), no dropout, SMA and BMA, batches of one and two M-tiles, ragged batches in masked
    mode, repeated calls, up to 256 tokens, up to 128 utterances (the group kernels); and the persistent form against the float64
    oracle (injected cases of up to 32 utterances)."""
    import gc
    import torch
    from oracle import oracle_np
    gc.collect()                                    # (the persistent launch is taken only while the process has ONE live context)
    Tref = 60
    rate = {"rate25": 0.25, "nodrop": 0.0}.get(mode, 0.5)
    hp, w, tokens, tl, mels, ml, masks, noise = _full_case(B, Tv, Tref, steps, seed=40 + B, att=att, rate=rate)
    env = dict(env)
    mixed = env.pop("MIXED", None) is not None
    hp = dict(hp); hp["Use_Mixed_Precision"] = mixed
    kw = dict(steps=steps)
    if mode == "injected" or mode == "masked":
        kw.update(prenet_masks=masks, attn_noise=noise)
    else:
        kw.update(seed=77)
    if mode == "masked":
        tl = np.random.default_rng(3).integers(Tv // 3, Tv + 1, B).astype(np.int32)
        tl[0] = Tv
        kw.update(masked=True)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    outs = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("GSTTACO_PERSIST_DECODE", flag)
        m = _model(hp, w, B, Tv, Tref + 1)
        for rep in range(2):
            mel, stop, _, align = m.Inference_Step(tokens, tl, None, mels, ml, **kw)
        m.synchronize()
        assert m.handoff_error() == 0
        n_persist, on = m.decode_counters()            # (enqueued once: the second call replays the captured graph)
        assert (n_persist >= 1 and on == 1) if flag == "1" else (n_persist == 0 and on == 0)
        outs[flag] = (mel.cpu().numpy(), stop.cpu().numpy(), align.cpu().numpy())
        if flag == "1" and mode == "injected" and B <= 32 and not mixed:
            ref = oracle_np.inference_step(hp, w, tokens, mels, ml, masks, noise, steps=steps, dt=np.float64)
            assert np.abs(outs[flag][0] - ref[0]).max() <= TOL and np.abs(outs[flag][2] - ref[3]).max() <= TOL
        del m, mel, stop, align
        gc.collect()
    monkeypatch.delenv("GSTTACO_PERSIST_DECODE")
    assert np.isfinite(outs["1"][0]).all()
    for a, b in zip(outs["1"], outs["0"]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("B,Tv,mixed", [(40, 70, False), (100, 33, False), (40, 70, True)])
def test_group_and_bf16_kernels_match_the_oracle_directly(B, Tv, mixed):
    """The group kernels (33..128 rows, fp32) and the bf16 kernel (<= 64 rows) are held bitwise to the launch path elsewhere; here
    they meet the float64 oracle themselves -- 12 steps at full dimensions, injected randomness, the persistent path asserted (bf16:
    the oracle that emulates the operand roundings, decode-loop outputs only)."""
    import torch
    from oracle import oracle_np
    steps, Tref = 12, 50
    hp, w, tokens, tl, mels, ml, masks, noise = _full_case(B, Tv, Tref, steps, seed=300 + B)
    hp = dict(hp); hp["Use_Mixed_Precision"] = bool(mixed)
    _sole_context()
    m = _model(hp, w, B, Tv, Tref + 1)
    mel, stop, _, align, pre = m.Inference_Step(tokens, tl, None, mels, ml, prenet_masks=masks, attn_noise=noise, steps=steps, return_pre_mel=True)
    m.synchronize()
    _assert_persistent_decode(m)
    if mixed:
        assert m.decode_plan(Tv)[1] is True
        ref = oracle_np.inference_step(hp, w, tokens, mels, ml, masks, noise, steps=steps, dt=np.float64, mixed=True, fused_prenet0=True)
        e_pre, e_al = np.abs(pre.cpu().numpy() - ref[-1]["pre_mel"]).max(), np.abs(align.cpu().numpy() - ref[3]).max()
        print("bf16 kernel vs the emulating oracle: pre-net mel %.3g alignment %.3g" % (e_pre, e_al))
        assert e_pre <= 5e-3 and e_al <= 2e-3
    else:
        ref = oracle_np.inference_step(hp, w, tokens, mels, ml, masks, noise, steps=steps, dt=np.float64)
        errs = (np.abs(mel.cpu().numpy() - ref[0]).max(), np.abs(stop.cpu().numpy() - ref[1]).max(), np.abs(align.cpu().numpy() - ref[3]).max())
        print("group kernel vs oracle: mel %.3g stop %.3g alignment %.3g" % errs)
        assert max(errs) <= TOL


def test_the_reference_inference_sentences_take_the_persistent_launch():
    """The reference's own inference example: its 8 sentences (Inference_Sentence_for_Training.txt, the longest 185 characters ->
    a batch padded to 187 tokens by Feeder.py:161-180) through ``Inference`` -- tokens from the committed fixture the reference's
    Feeder produced -- run on the persistent decode launch (T_v <= 256 since round 5), not silently on the launch path."""
    import os
    import torch
    from gst_tacotron_amd import synthetic, weights
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "feeder_tokens.npz"), allow_pickle=False)
    sentences = [str(x) for x in fx["sentences"]]
    hp = synthetic.config_hp("cfg2")
    w = weights.synthetic_weights(hp, seed=0)
    rng = np.random.default_rng(5)
    ref_mels = [np.clip(rng.normal(0, 1.5, (n, 80)), -4, 4).astype(np.float32) for n in rng.integers(100, 240, len(sentences))]
    _sole_context()
    m = _model(hp, w, len(sentences), 256, 260)
    out = m.Inference(sentences, ref_mels, steps=40)
    assert out is not None and out[3].shape[2] == int(fx["nogst.tokens"].shape[1]) == 187
    _assert_persistent_decode(m)
    assert np.isfinite(out[0].cpu().numpy()).all()


def test_on_device_randomness_is_seeded():
    """Throughput mode (no injected tensors): Philox dropout / SMA noise are a pure function of the seed."""
    import torch
    hp, w, tokens, tl, mels, ml, _, _ = _full_case(4, 20, 64, 10, seed=6)
    m = _model(hp, w, 4, 20, 65)
    a = m.Inference_Step(tokens, tl, None, mels, ml, seed=11, steps=10)[0].cpu().numpy()
    b = m.Inference_Step(tokens, tl, None, mels, ml, seed=11, steps=10)[0].cpu().numpy()
    c = m.Inference_Step(tokens, tl, None, mels, ml, seed=12, steps=10)[0].cpu().numpy()
    assert np.isfinite(a).all() and np.array_equal(a, b) and not np.array_equal(a, c)


@pytest.mark.parametrize("att,rate", [("SMA", 0.5), ("BMA", 0.5), ("SMA", 0.25)])
def test_throughput_mode_randomness_matches_oracle(att, rate, monkeypatch):
    """Throughput mode (nothing injected): the prenet keep decisions and the SMA noise come from the seed on the device.  The
    front kernel derives the decisions itself (a counter hash at rate 0.5, Philox otherwise) and -- at rate 0.5 and the
    reference's sizes -- never requests the prenet-1 / query weight rows they zero.  The tensors it used are read back
    (gsttaco_debug_randomness) and fed to the oracle: same mels; and the four-kernel front end (GSTTACO_FUSED_FRONT=0), which reads
    the pre-generated masks from HBM and skips nothing, uses the same tensors and gives the same result."""
    import torch
    from oracle import oracle_np
    B, Tv, Tref, steps = 3, 24, 70, 9
    hp, w, tokens, tl, mels, ml, _, _ = _full_case(B, Tv, Tref, steps, seed=31, att=att, rate=rate)
    outs = []
    for flag in ("2", "0"):
        monkeypatch.setenv("GSTTACO_FUSED_FRONT", flag)
        m = _model(hp, w, B, Tv, Tref + 1)
        mel, stop, _, align = m.Inference_Step(tokens, tl, None, mels, ml, seed=1234, steps=steps)
        torch.cuda.synchronize()
        masks, noise = m.debug_randomness(steps, B, Tv)
        outs.append((mel.cpu().numpy(), align.cpu().numpy(), masks, noise))
    monkeypatch.delenv("GSTTACO_FUSED_FRONT")
    mel, align, masks, noise = outs[0]
    assert set(np.unique(masks)) <= {0.0, 1.0} and abs(masks.mean() - (1.0 - rate)) < 0.03
    from gst_tacotron_amd import hparams
    if hparams.attention_sigmoid_noise(hp) > 0:             # SMA draws N(0,1) noise (Steps.py:212,220-221), BMA none
        assert abs(noise.mean()) < 0.1 and abs(noise.std() - 1.0) < 0.1 and np.array_equal(noise, outs[1][3])
    else:
        noise = None
    assert np.array_equal(masks, outs[1][2])
    assert np.abs(mel - outs[1][0]).max() <= TOL and np.abs(align - outs[1][1]).max() <= TOL
    ref = oracle_np.inference_step(hp, w, tokens, mels, ml, masks, noise, steps=steps, dt=np.float64)
    assert np.abs(mel - ref[0]).max() <= TOL
    assert np.abs(align - ref[3]).max() <= TOL


def test_hashed_keep_decisions_look_random():
    """The rate-0.5 keep decisions are words of a counter hash (device_utils.h gt_keep_word), not Philox: check the bits the
    decode actually used for balance and for independence across steps, rows, layers and neighbouring columns."""
    import torch
    B, Tv, Tref, steps = 4, 16, 40, 64
    hp, w, tokens, tl, mels, ml, _, _ = _full_case(B, Tv, Tref, steps, seed=41)
    m = _model(hp, w, B, Tv, Tref + 1)
    m.Inference_Step(tokens, tl, None, mels, ml, seed=987654321, steps=steps)
    torch.cuda.synchronize()
    k = m.debug_randomness(steps, B, Tv)[0].astype(np.int64)          # [steps, 2, B, 256]
    n = k.size
    tol = 4.0 / np.sqrt(n)                                            # 4 sigma of a fair coin over n bits
    assert abs(k.mean() - 0.5) < tol
    agree = lambda a, b: (a == b).mean()
    assert abs(agree(k[1:], k[:-1]) - 0.5) < 2 * tol                  # consecutive steps
    assert abs(agree(k[:, 0], k[:, 1]) - 0.5) < 2 * tol               # the two prenet layers
    assert abs(agree(k[:, :, 1:], k[:, :, :-1]) - 0.5) < 2 * tol      # neighbouring utterances
    assert abs(agree(k[..., 1:], k[..., :-1]) - 0.5) < 2 * tol        # neighbouring columns
    assert abs(agree(k[..., 32:], k[..., :-32]) - 0.5) < 2 * tol      # same bit of neighbouring words
    assert np.abs(k.mean(axis=(0, 1, 2)) - 0.5).max() < 6.0 / np.sqrt(steps * 2 * B)     # no stuck column
    m.Inference_Step(tokens, tl, None, mels, ml, seed=987654322, steps=steps)
    torch.cuda.synchronize()
    k2 = m.debug_randomness(steps, B, Tv)[0].astype(np.int64)
    assert abs(agree(k, k2) - 0.5) < 2 * tol                           # adjacent seeds


def test_long_trajectory_matches_oracle():
    """All Max_Step // r = 500 decode steps at the reference's dimensions against the float64 oracle (2 utterances, so the
    oracle finishes in seconds): rounding differences must not grow along the recurrence."""
    import time
    import torch
    from oracle import oracle_np
    B, Tv, Tref, steps = 2, 48, 100, 500
    hp, w, tokens, tl, mels, ml, masks, noise = _full_case(B, Tv, Tref, steps, seed=51)
    _sole_context()
    m = _model(hp, w, B, Tv, Tref + 1)
    mel, stop, _, align = m.Inference_Step(tokens, tl, None, mels, ml, prenet_masks=masks, attn_noise=noise)
    m.synchronize()
    _assert_persistent_decode(m)        # (ONE launch runs all 500 steps: the form the headline number is measured on)
    t0 = time.time()
    ref = oracle_np.inference_step(hp, w, tokens, mels, ml, masks, noise, dt=np.float64)
    print("oracle: %.1f s" % (time.time() - t0))
    assert mel.shape == (B, 1000, 80)
    err = np.abs(mel.cpu().numpy() - ref[0]).max(axis=(0, 2))
    print("mel max-abs error, first / last 100 frames:", err[:100].max(), err[-100:].max())
    assert err.max() <= TOL
    assert np.abs(stop.cpu().numpy() - ref[1]).max() <= TOL
    assert np.abs(align.cpu().numpy() - ref[3]).max() <= TOL


def test_full_size_batch_independence_and_invariants():
    """BASELINE configs[1] at full size (batch 32 x 128 tokens x 500 steps), too big for the oracle in seconds, through
    size-independent properties: (1) utterances are independent -- utterance b decoded inside the batch of 32 equals the
    same utterance decoded alone at batch 1 (different tiling / M-tile count); (2) SMA alignments are non-negative and
    never gain mass; (3) the decoder always runs Max_Step // r iterations."""
    import torch
    from gst_tacotron_amd import synthetic, weights
    hp, inputs = synthetic.config_inputs("cfg2", batch=32)
    w = weights.synthetic_weights(hp, seed=0)
    rng = np.random.default_rng(77)
    masks, noise = synthetic.make_randomness(rng, 500, 32, 128, [256, 256])
    _sole_context()
    m = _model(hp, w, 32, 128, 257)
    mel, stop, _, align = m.Inference_Step(inputs["tokens"], None, None, inputs["mels_for_gst"], inputs["mel_lengths_for_gst"],
                                           prenet_masks=masks, attn_noise=noise)
    m.synchronize()
    _assert_persistent_decode(m)
    mel, align = mel.cpu().numpy(), align.cpu().numpy()
    assert mel.shape == (32, 1000, 80) and stop.shape == (32, 500) and align.shape == (32, 500, 128)
    assert np.isfinite(mel).all()
    assert align.min() >= 0.0
    mass = align.sum(-1)
    assert np.all(mass[:, 0] <= 1.0 + 1e-5) and np.all(np.diff(mass, axis=1) <= 1e-5)
    for b in (0, 13, 31):
        sl = slice(b, b + 1)
        one = m.Inference_Step(inputs["tokens"][sl], None, None, inputs["mels_for_gst"][sl], inputs["mel_lengths_for_gst"][sl],
                               prenet_masks=masks[:, :, sl], attn_noise=noise[:, sl])
        torch.cuda.synchronize()
        assert np.abs(one[0].cpu().numpy()[0] - mel[b]).max() <= TOL
        assert np.abs(one[3].cpu().numpy()[0] - align[b]).max() <= TOL
    m.synchronize()
    _assert_persistent_decode(m, 2)      # (batch 32 and batch 1: two shapes, one persistent launch each)


@pytest.mark.parametrize("att", ["SMA", "BMA"])
def test_headline_tile_shape_on_the_persistent_launch_matches_oracle(att):
    """The persistent decode launch at the tile shape the headline number is measured on -- 32 utterances (two M-tiles), 128 tokens
    (the whole processed-memory tile in LDS) -- against the float64 oracle over 64 steps with injected keep masks and noise, the
    persistent path asserted (the bitwise test against the launch path covers the same shape; this is the oracle's word on it)."""
    import time
    import torch
    from oracle import oracle_np
    B, Tv, Tref, steps = 32, 128, 90, 64
    hp, w, tokens, tl, mels, ml, masks, noise = _full_case(B, Tv, Tref, steps, seed=88, att=att)
    _sole_context()
    m = _model(hp, w, B, Tv, Tref + 1)
    mel, stop, _, align = m.Inference_Step(tokens, tl, None, mels, ml, prenet_masks=masks, attn_noise=noise, steps=steps)
    m.synchronize()
    _assert_persistent_decode(m)
    t0 = time.time()
    ref = oracle_np.inference_step(hp, w, tokens, mels, ml, masks, noise, steps=steps, dt=np.float64)
    print("oracle: %.1f s" % (time.time() - t0))
    errs = (np.abs(mel.cpu().numpy() - ref[0]).max(), np.abs(stop.cpu().numpy() - ref[1]).max(), np.abs(align.cpu().numpy() - ref[3]).max())
    print("B 32 x T_v 128 x 64 steps on the persistent launch vs oracle: mel %.3g stop %.3g alignment %.3g" % errs)
    assert max(errs) <= TOL


def test_error_behaviour_on_gpu():
    from gst_tacotron_amd import capi
    hp, w, tokens, tl, mels, ml, masks, noise = _full_case(2, 16, 32, 2, seed=8)
    m = _model(hp, w, 2, 16, 33)
    with pytest.raises(capi.GstTacoError) as e:
        m.Inference_Step(np.concatenate([tokens, tokens]), None, None, np.concatenate([mels, mels]), np.concatenate([ml, ml]), steps=2)
    assert e.value.code == -5                                     # capacity given at create
    with pytest.raises(ValueError, match="GST is enabled"):
        m.Inference_Step(tokens, steps=2)
    with pytest.raises(capi.GstTacoError):
        m.Inference_Step(tokens, None, None, mels, ml, steps=10_000)


@pytest.mark.parametrize("att", ["SMA", "BMA"])
def test_masked_mode_ragged_batch(att):
    """Masked-mode extension (BASELINE configs[2]): a ragged padded batch with token_lengths honoured matches
    the masked oracle, and every utterance equals the same utterance decoded alone (unmasked, at its own length)."""
    import torch
    from oracle import oracle_np
    B, Tv, Tref, steps = 5, 150, 80, 6
    lens = np.array([150, 33, 128, 64, 97], np.int32)
    hp, w, _, _, mels, ml, masks, noise = _full_case(B, Tv, Tref, steps, seed=31, att=att, lens=np.array([80, 20, 64, 65, 33]))
    from gst_tacotron_amd import synthetic
    tokens, _ = synthetic.make_tokens(np.random.default_rng(32), B, Tv, lengths=lens)
    m = _model(hp, w, B, Tv, Tref + 1)
    mel, stop, _, align = m.Inference_Step(tokens, lens, None, mels, ml, prenet_masks=masks, attn_noise=noise, steps=steps,
                                           masked=True)
    torch.cuda.synchronize()
    mel, align = mel.cpu().numpy(), align.cpu().numpy()
    ref = oracle_np.inference_step(hp, w, tokens, mels, ml, masks, noise, steps=steps, dt=np.float64, token_lengths=lens)
    assert np.abs(mel - ref[0]).max() <= TOL and np.abs(align - ref[3]).max() <= TOL
    enc = m.encode(tokens, lens).cpu().numpy()
    assert np.abs(enc - ref[4]["encoder"]).max() <= TOL
    for b in (1, 3):
        n = int(lens[b])
        assert not align[b][:, n:].any() and not enc[b, n:].any()
        one = m.Inference_Step(tokens[b:b + 1, :n], None, None, mels[b:b + 1], ml[b:b + 1], prenet_masks=masks[:, :, b:b + 1],
                               attn_noise=np.ascontiguousarray(noise[:, b:b + 1, :n]), steps=steps)
        torch.cuda.synchronize()
        assert np.abs(one[0].cpu().numpy()[0] - mel[b]).max() <= TOL
        assert np.abs(one[3].cpu().numpy()[0] - align[b][:, :n]).max() <= TOL
    # the default (reference behaviour) still ignores token_lengths
    un = m.Inference_Step(tokens, lens, None, mels, ml, prenet_masks=masks, attn_noise=noise, steps=steps)
    torch.cuda.synchronize()
    ref_un = oracle_np.inference_step(hp, w, tokens, mels, ml, masks, noise, steps=steps, dt=np.float64)
    assert np.abs(un[0].cpu().numpy() - ref_un[0]).max() <= TOL


@pytest.mark.parametrize("front", ["2", "0"], ids=["fused-front", "four-kernel"])
def test_lsa_extension_full_dims_and_masked(monkeypatch, front):
    """Step-wise location-sensitive attention (extension A13) at full dimensions, k=31 / 32 filters, with and without
    the smoothing normalisation, unmasked and masked (140 positions: two passes of the fused kernel's 128-row tile) -- on the fused
    front end (dec_front_lsa.hip: the two location GEMMs on the fp32 matrix pipe) and on the four-kernel path (attention.hip)."""
    import torch
    from oracle import oracle_np
    monkeypatch.setenv("GSTTACO_FUSED_FRONT", front)
    B, Tv, Tref, steps = 3, 140, 64, 5
    lens = np.array([140, 60, 101], np.int32)
    for smoothing in (False, True):
        hp, w, _, _, mels, ml, masks, noise = _full_case(B, Tv, Tref, steps, seed=41)
        hp["Tacotron2"]["Decoder"]["Attention"] = {"Type": "LSA", "Size": 128, "Conv": {"Filters": 32, "Kernel_Size": 31},
                                                   "Smoothing": smoothing}
        from gst_tacotron_amd import synthetic, weights
        w = weights.synthetic_weights(hp, seed=0)
        tokens, _ = synthetic.make_tokens(np.random.default_rng(42), B, Tv, lengths=lens)
        m = _model(hp, w, B, Tv, Tref + 1)
        assert m.decode_plan(Tv)[0] is (front != "0")
        for tl in (None, lens):
            out = m.Inference_Step(tokens, tl, None, mels, ml, prenet_masks=masks, attn_noise=noise, steps=steps, masked=tl is not None)
            torch.cuda.synchronize()
            ref = oracle_np.inference_step(hp, w, tokens, mels, ml, masks, noise, steps=steps, dt=np.float64, token_lengths=tl)
            assert np.abs(out[0].cpu().numpy() - ref[0]).max() <= TOL
            assert np.abs(out[3].cpu().numpy() - ref[3]).max() <= TOL
            assert np.allclose(out[3].cpu().numpy().sum(-1), 1.0, atol=1e-5)
        del m


@pytest.mark.parametrize("filters,kernel,att,fits", [(32, 31, 128, True), (20, 9, 128, True), (8, 7, 64, True), (40, 15, 128, True),
                                                     (64, 63, 128, False)])
def test_lsa_fused_front_matches_the_four_kernel_path(monkeypatch, filters, kernel, att, fits):
    """The LSA extension's two implementations against each other in throughput mode (device-generated dropout), at filter counts and
    kernel sizes that are not multiples of the matrix instruction's granules, 24 utterances x 96 positions x 40 steps: the location
    sums are the same fmaf chains in the same order, so the outputs differ only through the order of the score's sum over channels.
    (Last case: operands that do not fit the fused kernel's LDS beside the memory tile stay on the four-kernel path.)"""
    import torch
    from gst_tacotron_amd import synthetic, weights
    B, Tv, Tref, steps = 24, 96, 64, 40
    hp = synthetic.config_hp("cfg2")
    hp["Tacotron2"]["Decoder"]["Attention"] = {"Type": "LSA", "Size": att, "Conv": {"Filters": filters, "Kernel_Size": kernel}}
    w = weights.synthetic_weights(hp, seed=3)
    rng = np.random.default_rng(5)
    tokens, _ = synthetic.make_tokens(rng, B, Tv)
    mels, ml = synthetic.make_ref_mels(rng, B, Tref)
    outs = []
    for front in ("2", "0"):
        monkeypatch.setenv("GSTTACO_FUSED_FRONT", front)
        m = _model(hp, w, B, Tv, Tref + 1)
        assert m.decode_plan(Tv)[0] is (front != "0" and fits)
        o = m.Inference_Step(tokens, None, None, mels, ml, seed=11, steps=steps)
        torch.cuda.synchronize()
        outs.append([x.cpu().numpy() for x in (o[0], o[3])])
        del m
    assert np.isfinite(outs[0][0]).all()
    assert np.abs(outs[0][0] - outs[1][0]).max() <= TOL
    assert np.abs(outs[0][1] - outs[1][1]).max() <= TOL
    assert np.allclose(outs[0][1].sum(-1), 1.0, atol=1e-5)


def test_config3_batch128_variable_length_with_padding_masks():
    """BASELINE configs[2]: batch 128, 32-256 tokens, padding masks.  Too big for the oracle, so through the property the
    masks exist for: every utterance of the padded batch equals that utterance decoded alone at its own length (a batch
    of 1, 4 M-chunks vs 1, one vs several attention row passes), alignments are zero beyond each length, and bucketing
    by length (sorting the batch) permutes the outputs and changes nothing else."""
    import torch
    from gst_tacotron_amd import synthetic, weights
    B, Tv, Tref, steps = 128, 256, 120, 24
    hp = synthetic.config_hp("cfg2")
    w = weights.synthetic_weights(hp, seed=0)
    rng = np.random.default_rng(123)
    lens = rng.integers(32, 257, B).astype(np.int32)
    lens[0], lens[1] = 256, 32
    tokens, _ = synthetic.make_tokens(rng, B, Tv, lengths=lens)
    mels, ml = synthetic.make_ref_mels(rng, B, Tref, lengths=rng.integers(40, Tref + 1, B))
    masks, noise = synthetic.make_randomness(rng, steps, B, Tv, [256, 256])
    m = _model(hp, w, B, Tv, Tref + 1)
    mel, stop, _, align = m.Inference_Step(tokens, lens, None, mels, ml, prenet_masks=masks, attn_noise=noise, steps=steps,
                                           masked=True)
    torch.cuda.synchronize()
    mel, stop, align = mel.cpu().numpy(), stop.cpu().numpy(), align.cpu().numpy()
    assert np.isfinite(mel).all() and mel.shape == (B, steps * 2, 80)
    for b in range(B):
        assert not align[b][:, lens[b]:].any()
    for b in (0, 1, 77, 127):
        n = int(lens[b])
        # the reference encoder's convolutions see the batch's zero padding of the reference mel (reference behaviour, F5):
        # keep the padded width when decoding alone
        one = m.Inference_Step(tokens[b:b + 1, :n], None, None, mels[b:b + 1], ml[b:b + 1], prenet_masks=masks[:, :, b:b + 1],
                               attn_noise=np.ascontiguousarray(noise[:, b:b + 1, :n]), steps=steps)
        torch.cuda.synchronize()
        assert np.abs(one[0].cpu().numpy()[0] - mel[b]).max() <= TOL, b
        assert np.abs(one[3].cpu().numpy()[0] - align[b][:, :n]).max() <= TOL, b
    order = np.argsort(lens, kind="stable")                      # a length bucket = a permutation of the batch
    srt = m.Inference_Step(tokens[order], lens[order], None, mels[order], ml[order], prenet_masks=masks[:, :, order],
                           attn_noise=noise[:, order], steps=steps, masked=True)
    torch.cuda.synchronize()
    assert np.abs(srt[0].cpu().numpy() - mel[order]).max() <= TOL
    assert np.abs(srt[1].cpu().numpy() - stop[order]).max() <= TOL


# ------------------------------------------------------------------ mixed precision (BASELINE configs[4])
# bf16 operands / fp32 accumulation in the MFMA GEMMs; the oracle emulates the same roundings (oracle_np.mm).  Where no
# rounded activation feeds another rounded GEMM (one GEMM deep), or at the tiny dimensions, the HIP path reproduces the
# emulation to fp32 noise (MIXED_EXACT).  Through deep stacks the comparison is ill-conditioned by construction: an fp32
# activation that differs from the float64 emulation in its last bits occasionally rounds to the neighbouring bf16 value
# (0.4 % of that element), and the synthetic-weight postnet amplifies such a flip ~10x per layer (measured: 1 layer
# 9e-7, 2 layers 1e-4, 3 layers 1e-3, 5 layers 4e-3 max-abs; two CPU emulations in float64 / float32 differ by 1e-3 the
# same way).  MIXED_TOL / MIXED_MEAN bound that; MIXED_VS_FP32 states how far the mode sits from fp32 on these inputs.
MIXED_EXACT = 1e-5
MIXED_TOL = 2e-2
MIXED_MEAN = 2e-3
MIXED_VS_FP32 = 0.25


def _mixed_case(kind, seed=9):
    from gst_tacotron_amd import synthetic, weights
    if kind == "tiny":
        hp = synthetic.tiny_hp("SMA", r=2, gst=True, max_step=24)
        B, Tv, Tref, steps = 3, 12, 70, 12
    else:
        hp = synthetic.config_hp("cfg2")
        B, Tv, Tref, steps = 5, 40, 90, 10
    hp["Use_Mixed_Precision"] = True
    w = weights.synthetic_weights(hp, seed=seed)
    rng = np.random.default_rng(seed + 1)
    tokens, tl = synthetic.make_tokens(rng, B, Tv)
    mels, ml = synthetic.make_ref_mels(rng, B, Tref, mel=hp["Sound"]["Mel_Dim"])
    masks, noise = synthetic.make_randomness(rng, steps, B, Tv, hp["Tacotron2"]["Decoder"]["Prenet"]["Size"])
    return hp, w, tokens, tl, mels, ml, masks, noise, steps


@pytest.mark.parametrize("kind", ["tiny", "full"])
def test_mixed_precision_matches_the_bf16_emulating_oracle(kind):
    import torch
    from oracle import oracle_np
    hp, w, tokens, tl, mels, ml, masks, noise, steps = _mixed_case(kind)
    B, Tv = tokens.shape
    m = _model(hp, w, B, Tv, mels.shape[1])
    mel, stop, spec, align, pre = m.Inference_Step(tokens, tl, None, mels, ml, prenet_masks=masks, attn_noise=noise, steps=steps,
                                                   with_vocoder=True, return_pre_mel=True)
    torch.cuda.synchronize()
    # Which operands get rounded to bf16 depends on the algebraic form: DESIGN 3.1b folds prenet-0 into the projection for every BMA / SMA
    # model (the projection and the first prenet Dense are both linear).  The oracle is told that form a priori -- not by the product --
    # and the product is held to it.
    assert m.decode_plan(Tv)[1] is True
    ref = oracle_np.inference_step(hp, w, tokens, mels, ml, masks, noise, steps=steps, dt=np.float64, with_vocoder=True, mixed=True,
                                   fused_prenet0=True)
    fp32 = oracle_np.inference_step(hp, w, tokens, mels, ml, masks, noise, steps=steps, dt=np.float64, with_vocoder=True)
    errs = {"mel": np.abs(mel.cpu().numpy() - ref[0]).max(), "stop": np.abs(stop.cpu().numpy() - ref[1]).max(),
            "spec": np.abs(spec.cpu().numpy() - ref[2]).max(), "align": np.abs(align.cpu().numpy() - ref[3]).max()}
    drift = np.abs(ref[0] - fp32[0]).max()
    print(kind, "mixed vs emulating oracle", errs, " emulated-mixed vs fp32 oracle (mel)", drift)
    if kind == "tiny":
        # the decode loop itself (alignments, stop tokens, pre-net mels) reproduces the emulation to fp32 noise; behind it
        # the postnet and the vocoder round those mels to bf16 again and amplify a flipped rounding (see above)
        loop = {"pre": np.abs(pre.cpu().numpy() - ref[-1]["pre_mel"]).max(), "stop": errs["stop"], "align": errs["align"]}
        assert max(loop.values()) <= MIXED_EXACT, loop
        assert max(errs.values()) <= MIXED_TOL, errs
    else:
        assert max(errs.values()) <= MIXED_TOL, errs
        assert np.abs(mel.cpu().numpy() - ref[0]).mean() <= MIXED_MEAN
    assert 0.0 < drift <= MIXED_VS_FP32
    hp32 = dict(hp); hp32["Use_Mixed_Precision"] = False
    m32 = _model(hp32, w, B, Tv, mels.shape[1])
    mel32 = m32.Inference_Step(tokens, tl, None, mels, ml, prenet_masks=masks, attn_noise=noise, steps=steps)[0]
    torch.cuda.synchronize()
    assert np.abs(mel32.cpu().numpy() - fp32[0]).max() <= TOL            # the flag off is still the fp32 parity path


def test_mixed_precision_single_gemm_depth_is_exact_at_full_dimensions():
    """Full-size kernels (128x128 bf16 conv tiles, 8- and 16-wave skinny GEMMs, co-scheduled workers) one GEMM deep:
    a 1-layer postnet, the hoisted Value projection, and decoder step 0 from the oracle's memory."""
    import torch
    from gst_tacotron_amd import synthetic, weights
    from oracle import oracle_np
    hp, w, tokens, tl, mels, ml, masks, noise, steps = _mixed_case("full", seed=11)
    B, Tv = tokens.shape
    m = _model(hp, w, B, Tv, mels.shape[1])
    w64 = oracle_np.cast_weights(w, np.float64)
    oracle_np.MIXED = True
    try:
        enc_ref = oracle_np.encoder(hp, w64, tokens, np.float64)
        gst_ref = oracle_np.style_token_layer(hp, w64, mels, ml, np.float64)
        mem = oracle_np.gst_concat(enc_ref, gst_ref)
        pre_ref, stop_ref, align_ref = oracle_np.decoder(hp, w64, mem, np.float64, masks.astype(np.float64),
                                                         noise.astype(np.float64), steps=1)
    finally:
        oracle_np.MIXED = False
    pre, stop, align = m.decode(enc_ref.astype(np.float32), gst_ref.astype(np.float32), masks, noise, steps=1)
    torch.cuda.synchronize()
    e = max(np.abs(pre.cpu().numpy() - pre_ref).max(), np.abs(align.cpu().numpy() - align_ref).max())
    print("decoder step 0 (value projection, LSTM x2, projection)", e)
    assert e <= 1e-4                      # three GEMMs deep (h1, h2 are rounded again): a flip is possible, rarely
    hp1 = synthetic.config_hp("cfg2"); hp1["Use_Mixed_Precision"] = True
    hp1["Tacotron2"]["Decoder"]["Conv"].update({"Filters": [], "Kernel_Size": [], "Strides": []})
    w1 = weights.synthetic_weights(hp1, seed=12)
    m1 = _model(hp1, w1, 5, 8, 4)
    x = np.clip(np.random.default_rng(1).normal(0, 1.5, (5, 150, 80)), -4, 4).astype(np.float32)
    post = m1.postnet(x).cpu().numpy()
    oracle_np.MIXED = True
    try:
        ref = oracle_np.postnet(hp1, oracle_np.cast_weights(w1, np.float64), x.astype(np.float64), np.float64)
    finally:
        oracle_np.MIXED = False
    print("1-layer postnet", np.abs(post - ref).max())
    assert np.abs(post - ref).max() <= MIXED_EXACT
