"""Audio front end: oracle vs the committed fixtures and independent cross-checks on CPU;
the HIP front end (through the C-ABI) vs the oracle on the GPU."""
import ctypes
import json
import os

import numpy as np
import pytest

from gst_tacotron_amd import capi, hparams, synthetic
from oracle import audio_np as A

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
F32_STORE = 1e-6
# GPU: float32 FFT / log10 against the float64 oracle.  Mel values span [-4, 4] (8 units per 100 dB); a 1e-6 relative
# amplitude error is 7e-7 units, but bands at the 1e-5 amplitude floor amplify absolute FFT noise, hence 2e-3.
MEL_TOL = 2e-3


def _sound(name):
    g = np.load(os.path.join(GOLD, "audio_synth_%s.npz" % name))
    return g, json.loads(str(g["sound_json"]))


# ------------------------------------------------------------------------------------------------ CPU
def test_oracle_reproduces_real_speech_fixture():
    g = np.load(os.path.join(GOLD, "audio_fv_ksp.npz"))
    y = g["pcm"].astype(np.float32) / 32768.0
    snd = hparams.load_hp()["Sound"]
    for top_db in (60, 15):
        assert tuple(A.trim_bounds(A.preemphasis(y), top_db, 32, 16)) == tuple(g["bounds_%d" % top_db])
        mel = A.mel_generate(y, snd, top_db)
        assert mel.dtype == np.float32 and mel.shape == g["mel_%d" % top_db].shape
        np.testing.assert_allclose(mel, g["mel_%d" % top_db], atol=F32_STORE, rtol=0)
    assert g["mel_60"].shape == (115, 80) and np.abs(g["mel_60"]).max() <= 4.0      # 115 frames for this wav


def test_oracle_reproduces_the_seven_fastvox_references():
    """tests/golden/audio_fv_all.npz (oracle/gen_golden_audio_fv.py): the reference repo's seven FastVox reference wavs as int16
    PCM, the multi-speaker references BASELINE configs[4] names (Inference_Wav_for_Training.txt:1-7), with the oracle's
    top_db = 15 mels (Feeder.py:204-209).  The KSP wav is also the older single-wav fixture: both must agree."""
    g = np.load(os.path.join(GOLD, "audio_fv_all.npz"))
    ksp = np.load(os.path.join(GOLD, "audio_fv_ksp.npz"))
    snd = hparams.load_hp()["Sound"]
    assert int(g["n"]) == 7 and int(g["top_db"]) == 15 and int(g["sample_rate"]) == snd["Sample_Rate"]
    names = [str(g["name%d" % i]) for i in range(7)]
    assert [n.split(".")[1] for n in names] == ["AWB", "BDL", "CLB", "JMK", "KSP", "RMS", "SLT"]
    for i in range(7):
        y = g["pcm%d" % i].astype(np.float32) / 32768.0
        assert tuple(A.trim_bounds(A.preemphasis(y), 15, 32, 16)) == tuple(g["bounds%d" % i])
        mel = A.mel_generate(y, snd, 15)
        np.testing.assert_allclose(mel, g["mel%d" % i], atol=F32_STORE, rtol=0)
    assert np.array_equal(g["pcm4"], ksp["pcm"]) and np.array_equal(g["mel4"], ksp["mel_15"])


@pytest.mark.parametrize("name", ["tiny", "full"])
def test_oracle_reproduces_synthetic_fixtures(name):
    g, snd = _sound(name)
    for i in range(int(g["n"])):
        for top_db in (60, 15):
            np.testing.assert_allclose(A.mel_generate(g["sig%d" % i], snd, top_db), g["mel%d_%d" % (i, top_db)],
                                       atol=F32_STORE, rtol=0)


def test_oracle_matches_reference_executed_pure_functions():
    """The librosa-free functions of the reference's Audio.py (:11-15, :86-102), EXECUTED from the reference source in the build
    container (oracle/gen_golden_audio_ref.py) -- reference-generated vectors, so these eight restatements are pinned."""
    z = np.load(os.path.join(GOLD, "audio_ref_pure.npz"))
    sig, mag, db, nrm = z["sig"], z["mag"], z["db"], z["nrm"]
    eq = lambda got, key: np.testing.assert_allclose(got, z[key], rtol=1e-13, atol=1e-13)
    eq(A.preemphasis(sig), "preemphasis")
    eq(A.inv_preemphasis(sig), "inv_preemphasis")
    eq(A.inv_preemphasis(A.preemphasis(sig)), "roundtrip")
    eq(A.amp_to_db(mag), "amp_to_db")
    eq(np.power(10.0, db * 0.05), "db_to_amp")                    # inlined in A.inv_spectrogram (Audio.py:26,89-90)
    eq(A.normalize(db), "normalize")
    eq(A.symmetric_normalize(db, max_abs_value=4), "symmetric_normalize")
    eq(A.denormalize(nrm), "denormalize")
    eq(A.symmetric_denormalize(nrm, max_abs_value=4), "symmetric_denormalize")


def test_stft_matches_scipy():
    """Independent check of the restated librosa.stft: scipy.signal.stft on the reflect-padded signal, rescaled."""
    from scipy import signal
    y = np.random.default_rng(0).normal(size=5000)
    for n_fft, hop, win in ((1024, 256, 1024), (64, 16, 64), (256, 64, 200)):
        D = A.stft(y, n_fft, hop, win)
        w = A.pad_center(A.hann_periodic(win), n_fft)
        _, _, Z = signal.stft(np.pad(y, n_fft // 2, mode="reflect"), window=w, nperseg=n_fft, noverlap=n_fft - hop,
                              nfft=n_fft, boundary=None, padded=False)
        Z = Z * w.sum()
        assert D.shape == Z.shape == (n_fft // 2 + 1, 1 + len(y) // hop)
        assert np.abs(D - Z).max() < 1e-4 * np.abs(Z).max()          # complex64 storage


def test_istft_inverts_stft():
    y = np.random.default_rng(1).normal(size=4096).astype(np.float32)
    z = A.istft(A.stft(y, 1024, 256, 1024), 256, 1024)
    assert z.shape == y.shape and np.abs(z - y).max() < 1e-5


def test_preemphasis_pair_is_the_identity():
    """Why the HIP front end has no IIR pass: inv_preemphasis then preemphasis from the same first sample is exact."""
    x = np.random.default_rng(2).normal(size=3000)
    np.testing.assert_allclose(A.preemphasis(A.inv_preemphasis(x)), x, atol=1e-12)
    np.testing.assert_allclose(A.inv_preemphasis(A.preemphasis(x)), x, atol=1e-10)
    assert A.preemphasis(np.array([1.0, 1.0, 1.0])).tolist() == [1.0, 1.0 - 0.97, 1.0 - 0.97]


def test_mel_basis_is_slaney_area_normalised_triangles():
    B = A.mel_basis(16000, 1024, 80)
    assert B.shape == (80, 513) and B.dtype == np.float32 and (B >= 0).all()
    assert ((B > 0).sum(0) <= 2).all()                       # adjacent triangles: a bin feeds at most two bands
    peaks = B.argmax(1)
    assert (np.diff(peaks) > 0).all()
    for m in range(80):                                       # unimodal rows
        nz = np.flatnonzero(B[m])
        assert (np.diff(nz) == 1).all()
        k = peaks[m]
        assert (np.diff(B[m, nz[0]:k + 1]) >= 0).all() and (np.diff(B[m, k:nz[-1] + 1]) <= 0).all()
    df = 16000 / 1024
    assert np.abs(B[40:].sum(1) * df - 1.0).max() < 0.02      # area 1 in Hz (wide bands: sampling error is small)
    # below 1 kHz the Slaney scale is linear: equally spaced centres (up to bin rounding)
    assert np.ptp(np.diff(peaks[:15])) <= 1


def test_trim_bounds_known_answer():
    sr = 16000
    y = np.zeros(sr, dtype=np.float64)
    y[4000:8000] = np.sin(2 * np.pi * 440 * np.arange(4000) / sr)
    s, e = A.trim_bounds(y, 60, 32, 16)
    assert 4000 - 32 <= s <= 4000 and 8000 <= e <= 8000 + 32 and s % 16 == 0
    assert A.trim_bounds(np.zeros(1000), 60, 32, 16) == (0, 1000)        # all frames tie with the max: nothing is trimmed


def test_library_mel_basis_equals_the_oracle():
    """gsttaco_mel_basis is host code (C++ restatement of librosa.filters.mel): checkable without a GPU."""
    for hp in (hparams.load_hp(), synthetic.tiny_hp()):
        ctx = capi.Context(hp, max_batch=1, max_tokens=4, max_ref_frames=4, max_wav_seconds=1.0)
        d = hparams.Dims(hp)
        out = np.zeros((d.mel, d.spec), dtype=np.float32)
        ctx.check(ctx.lib.gsttaco_mel_basis(ctx.handle, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float))))
        ref = A.mel_basis(d.sample_rate, 2 * (d.spec - 1), d.mel)
        assert np.abs(out - ref).max() <= 1e-9 and ((out != 0) == (ref != 0)).all()
        ctx.close()


def test_audio_config_validation_and_no_cpu_path(tmp_path):
    hp = synthetic.tiny_hp()
    hp["Sound"]["Spectrogram_Dim"] = 21                     # n_fft 40 is not a power of two
    with pytest.raises(capi.GstTacoError, match="Spectrogram_Dim"):
        capi.Context(hp, max_batch=1, max_tokens=4, max_ref_frames=4, max_wav_seconds=1.0)
    capi.Context(hp, max_batch=1, max_tokens=4, max_ref_frames=4).close()     # fine while the audio entry points are off
    from gst_tacotron_amd.feeder import Feeder
    f = Feeder(hparams.load_hp())
    with pytest.raises(RuntimeError, match="no CPU path"):
        f.Get_Inference_GST_Pattern([np.zeros(4000, np.float32)])
    import torch
    if not torch.cuda.is_available():
        from gst_tacotron_amd.model import GST_Tacotron
        m = GST_Tacotron(hyper_parameters=synthetic.tiny_hp(), max_batch=2, max_tokens=8, max_ref_frames=9)
        with pytest.raises(capi.GstTacoError, match="no CPU fallback"):
            m.Mel_Generate([np.zeros(4000, np.float32)])


def test_load_wav_matches_librosa_load_conventions(tmp_path):
    from scipy.io import wavfile
    from gst_tacotron_amd.audio import load_wav
    pcm = (np.random.default_rng(3).integers(-2000, 2000, 4000)).astype(np.int16)
    p = str(tmp_path / "a.wav")
    wavfile.write(p, 16000, pcm)
    y = load_wav(p, 16000)
    assert y.dtype == np.float32 and np.array_equal(y, pcm.astype(np.float32) / 32768.0)
    assert np.array_equal(y, A.load_wav(p, 16000))
    wavfile.write(p, 16000, np.stack([pcm, -pcm], 1))       # stereo -> channel mean
    assert np.abs(load_wav(p, 16000)).max() == 0.0
    wavfile.write(p, 8000, pcm)                              # other rate: resampled to Sound.Sample_Rate
    y2 = load_wav(p, 16000)
    assert y2.dtype == np.float32 and y2.shape[0] == 8000      # fix_length(ceil(n * ratio))
    np.testing.assert_allclose(y2[:600], A.load_wav(p, 16000, max_out=600), atol=2e-7, rtol=0)


def test_kaiser_best_resampling_of_the_22khz_reference_wav():
    """librosa.core.load resamples a file at another rate with resampy's 'kaiser_best' (Pattern_Generator.py:40-43; the
    reference's LJ wav, Inference_Wav_for_Training.txt:8, is 22.05 kHz).  The product's vectorised restatement
    (gst_tacotron_amd/audio.py) against the fixture the oracle's loop-for-loop restatement produced, and the oracle's loop
    itself on a slice of it; the filter's published design numbers; a pure tone comes out as the same tone."""
    from gst_tacotron_amd.audio import resample_kaiser_best
    g = np.load(os.path.join(GOLD, "audio_lj_excerpt.npz"))
    x = g["pcm"].astype(np.float32) / 32768.0
    y = resample_kaiser_best(x, int(g["sample_rate"]), int(g["target_rate"]))
    assert y.dtype == np.float32 and y.shape == g["resampled"].shape == (24000,)
    np.testing.assert_allclose(y, g["resampled"], atol=1e-6, rtol=0)        # float64 sums vs resampy's per-tap float32 rounding
    sl = A.resample_kaiser_best(x, int(g["sample_rate"]), int(g["target_rate"]), 11000, 11400)
    assert np.array_equal(sl, g["resampled"][11000:11400])
    win, bits = A.kaiser_best_filter()
    assert bits == 512 and win.shape == (64 * 512 + 1,) and abs(win[0] - A.KAISER_BEST["rolloff"]) < 1e-12 and abs(win[-1]) < 1e-7
    t = np.arange(22050) / 22050.0
    tone = resample_kaiser_best((0.5 * np.sin(2 * np.pi * 3000.0 * t)).astype(np.float32), 22050, 16000)
    ref = 0.5 * np.sin(2 * np.pi * 3000.0 * np.arange(tone.shape[0]) / 16000.0)
    assert np.abs(tone - ref)[1000:-1000].max() < 2e-3
    above = resample_kaiser_best((0.5 * np.sin(2 * np.pi * 9000.0 * t)).astype(np.float32), 22050, 16000)   # beyond the new Nyquist
    assert np.abs(above)[1000:-1000].max() < 2e-3


# ------------------------------------------------------------------------------------------------ GPU
def _model(hp, B):
    from gst_tacotron_amd.model import GST_Tacotron
    return GST_Tacotron(hyper_parameters=hp, max_batch=B, max_tokens=8, max_ref_frames=4, max_wav_seconds=4.0)


@pytest.mark.gpu
@pytest.mark.parametrize("top_db", [60, 15])
def test_gpu_front_end_matches_oracle_on_real_speech(top_db):
    import torch
    g = np.load(os.path.join(GOLD, "audio_fv_ksp.npz"))
    y = g["pcm"].astype(np.float32) / 32768.0
    m = _model(hparams.load_hp(), 2)
    mels, lens = m.Mel_Generate([y, y[:20000]], top_db)          # no Restore needed: no weights on this path
    torch.cuda.synchronize()
    exp = g["mel_%d" % top_db]
    assert int(lens[0]) == exp.shape[0]
    got = mels[0, 1:1 + exp.shape[0]].cpu().numpy()
    err = np.abs(got - exp).max()
    print("front end top_db", top_db, "max abs err", err)
    assert err <= MEL_TOL
    assert float(mels[:, 0].abs().max()) == 0.0                  # prepended zero frame (Feeder.py:219-223)
    n1 = int(lens[1])
    assert float(mels[1, 1 + n1:].abs().max()) == 0.0           # zero padding after the shorter utterance
    ref1 = A.mel_generate(y[:20000], hparams.load_hp()["Sound"], top_db)
    assert n1 == ref1.shape[0] and np.abs(mels[1, 1:1 + n1].cpu().numpy() - ref1).max() <= MEL_TOL


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["tiny", "full"])
def test_gpu_front_end_matches_oracle_on_ragged_batches(name):
    import torch
    g, snd = _sound(name)
    hp = synthetic.tiny_hp() if name == "tiny" else hparams.load_hp()
    assert hp["Sound"]["Spectrogram_Dim"] == snd["Spectrogram_Dim"]
    n = int(g["n"])
    m = _model(hp, n)
    sigs = [g["sig%d" % i] for i in range(n)]
    for top_db in (60, 15):
        mels, lens = m.Mel_Generate(sigs, top_db)
        torch.cuda.synchronize()
        worst = 0.0
        for i in range(n):
            exp = g["mel%d_%d" % (i, top_db)]
            assert int(lens[i]) == exp.shape[0], (i, top_db)     # the trim decision is taken in float64 like the reference
            worst = max(worst, float(np.abs(mels[i, 1:1 + exp.shape[0]].cpu().numpy() - exp).max()))
            assert float(mels[i, 1 + exp.shape[0]:].abs().max() if mels.shape[1] > 1 + exp.shape[0] else 0.0) == 0.0
        print(name, top_db, "max abs err", worst)
        assert worst <= MEL_TOL


@pytest.mark.gpu
def test_gpu_wav_to_style_embedding_and_inference_with_wav_references(tmp_path):
    """Feeder conventions with wav inputs (Feeder.py:197-225, 229-250): one wav -> top_db 60 and repeated; several ->
    top_db 15; Inference_GST on wavs == the oracle's style_token_layer on the oracle's mels."""
    import torch
    from scipy.io import wavfile
    from gst_tacotron_amd import weights
    from gst_tacotron_amd.model import GST_Tacotron
    from oracle import oracle_np
    hp = hparams.load_hp()
    hp["Max_Step"] = 8
    g = np.load(os.path.join(GOLD, "audio_fv_ksp.npz"))
    paths = []
    for i, cut in enumerate((29360, 24000)):
        p = str(tmp_path / ("r%d.wav" % i))
        wavfile.write(p, 16000, g["pcm"][:cut])
        paths.append(p)
    w = weights.synthetic_weights(hp, seed=1)
    m = GST_Tacotron(hyper_parameters=hp, max_batch=2, max_tokens=32, max_ref_frames=260, max_wav_seconds=4.0)
    m.Restore(weights=w)
    gst = m.Inference_GST(paths)
    torch.cuda.synchronize()
    w64 = oracle_np.cast_weights(w, np.float64)
    ref_mels = [A.mel_generate(g["pcm"][:cut].astype(np.float32) / 32768.0, hp["Sound"], 60) for cut in (29360, 24000)]
    batch = np.zeros((2, 1 + max(x.shape[0] for x in ref_mels), 80))           # Feeder.py:234-248: zero pad + prepended frame
    for i, x in enumerate(ref_mels):
        batch[i, 1:1 + x.shape[0]] = x
    # (the zero padding is seen by the reference encoder's convolutions: an utterance inside a batch is NOT the
    #  utterance alone -- reference behaviour, SURVEY F5 -- so the oracle runs on the same padded batch)
    ref = oracle_np.style_token_layer(hp, w64, batch, np.array([x.shape[0] for x in ref_mels]), np.float64)
    assert np.abs(gst.cpu().numpy() - ref).max() <= 1e-3
    pat = m.feeder.Get_Inference_Pattern(["Hello there.", "Second one."], [paths[0]])
    assert pat["mels_for_gst"].shape[0] == 2 and int(pat["mel_lengths_for_gst"][0]) == 115
    assert torch.equal(pat["mels_for_gst"][0], pat["mels_for_gst"][1])
    pat2 = m.feeder.Get_Inference_Pattern(["Hello there.", "Second one."], paths)
    assert int(pat2["mel_lengths_for_gst"][0]) == 88                    # top_db 15 for several references
    out = m.Inference(["Hello there.", "Second one."], paths)
    torch.cuda.synchronize()
    assert out[0].shape == (2, 8, 80) and torch.isfinite(out[0]).all()


# ================================================================================================ N4: back end
def _realistic_spectrogram(snd, seed, n_samples):
    """Normalised linear spectrogram [T, bins] of a synthetic voiced-like signal (Audio.spectrogram, Audio.py:18-21)."""
    rng = np.random.default_rng(seed)
    sr = snd["Sample_Rate"]
    t = np.arange(n_samples) / sr
    y = 0.3 * np.sin(2 * np.pi * 180 * t) * (1 + 0.5 * np.sin(2 * np.pi * 3 * t)) + 0.1 * np.sin(2 * np.pi * 1200 * t) \
        + 0.02 * rng.standard_normal(n_samples)
    n_fft = 2 * (snd["Spectrogram_Dim"] - 1)
    M = A.magnitude(y, n_fft, snd["Frame_Shift"], snd["Frame_Length"])
    return np.transpose(A.symmetric_normalize(A.amp_to_db(M) - 20, max_abs_value=snd["Max_Abs_Mel"])).astype(np.float32)


def _spectral_error(y, S_target, snd):
    n_fft = 2 * (snd["Spectrogram_Dim"] - 1)
    M = np.abs(A.stft(y, n_fft, snd["Frame_Shift"], snd["Frame_Length"]))
    return np.linalg.norm(M - S_target) / np.linalg.norm(S_target)


def test_griffin_lim_oracle_converges():
    snd = synthetic.tiny_hp()["Sound"]
    spec = _realistic_spectrogram(snd, 0, 3000)
    S = np.power(10.0, (A.symmetric_denormalize(spec.T, max_abs_value=4) + 20) * 0.05) ** 1.5
    ph = np.random.default_rng(0).random(S.shape)
    e = [_spectral_error(A.griffin_lim(S, snd["Frame_Shift"], snd["Frame_Length"], it, ph), S, snd) for it in (0, 5, 30)]
    assert e[0] > e[1] > e[2] and e[2] < 0.5 * e[0]
    y = A.inv_spectrogram(spec.T, snd, max_abs_value=4, iters=2, angles0=ph)
    assert y.shape == (snd["Frame_Shift"] * (spec.shape[0] - 1),)


def test_export_helpers(tmp_path):
    from gst_tacotron_amd import export
    assert export.stop_slice_index(np.array([1.0, 0.5, -0.1, 2.0, -3.0])) == 2        # Model.py:380
    assert export.stop_slice_index(np.array([1.0, 0.5])) == 2
    p = str(tmp_path / "x.wav")
    sig = np.array([0.0, 0.5, -0.5, 1.5, -1.5, 0.25])
    export.write_wav(p, sig, 16000)
    from scipy.io import wavfile
    sr, pcm = wavfile.read(p)
    assert sr == 16000 and pcm.tolist() == [0, 16384, -16384, 32767, -32768, 8192]
    t = str(tmp_path / "GST" / "l.GST.TXT")
    export.export_gst(t, ["a.wav", "b.wav"], ["A", "B"], np.array([[1.0, 2.0], [3.0, 4.5]]))
    rows = open(t).read().split("\n")
    assert rows[0] == "Wav\tTag\tUnit_0\tUnit_1" and rows[2] == "b.wav\tB\t3.0\t4.5"


@pytest.mark.gpu
@pytest.mark.parametrize("name,iters", [("tiny", 0), ("tiny", 1), ("tiny", 4), ("full", 0), ("full", 3)])
def test_gpu_griffin_lim_matches_oracle(name, iters):
    """Injected initial phases; pointwise comparison for a few iterations (float32 FFTs vs the oracle's float64/complex64
    mix: the tolerance is relative to the waveform's peak)."""
    import torch
    hp = synthetic.tiny_hp() if name == "tiny" else hparams.load_hp()
    snd = hp["Sound"]
    specs = [_realistic_spectrogram(snd, s, n) for s, n in ((1, 9000 if name == "full" else 1500), (2, 6000 if name == "full" else 1000))]
    T = max(s.shape[0] for s in specs)
    nb = snd["Spectrogram_Dim"]
    batch = np.zeros((2, T, nb), np.float32)
    for i, s in enumerate(specs):
        batch[i, :s.shape[0]] = s
    frames = np.array([s.shape[0] for s in specs], np.int32)
    ph = np.random.default_rng(5).random((2, T, nb)).astype(np.float32)
    m = _model(hp, 2)
    wav, lens = m.Inv_Spectrogram(batch, frames=frames, iters=iters, init_phase=ph)
    torch.cuda.synchronize()
    wav, lens = wav.cpu().numpy(), lens.cpu().numpy()
    for i, s in enumerate(specs):
        ref = A.inv_spectrogram(s.T.astype(np.float64), snd, max_abs_value=snd["Max_Abs_Mel"], iters=iters,
                                angles0=ph[i, :s.shape[0]].T.astype(np.float64))
        assert lens[i] == ref.shape[0] == snd["Frame_Shift"] * (s.shape[0] - 1)
        err = np.abs(wav[i, :lens[i]] - ref).max() / np.abs(ref).max()
        print(name, iters, i, "rel err", err)
        assert err <= 2e-3
        assert np.abs(wav[i, lens[i]:]).max(initial=0.0) == 0.0


@pytest.mark.gpu
def test_gpu_griffin_lim_60_iterations_converge_like_the_oracle():
    import torch
    hp = hparams.load_hp()
    snd = hp["Sound"]
    spec = _realistic_spectrogram(snd, 3, 12000)
    S = np.power(10.0, (A.symmetric_denormalize(spec.T, max_abs_value=4) + 20) * 0.05) ** 1.5
    ph = np.random.default_rng(6).random(spec.shape).astype(np.float32)
    m = _model(hp, 1)
    wav, lens = m.Inv_Spectrogram(spec[None], iters=60, init_phase=ph[None])
    wav0, _ = m.Inv_Spectrogram(spec[None], iters=0, init_phase=ph[None])
    wav_rng, _ = m.Inv_Spectrogram(spec[None], iters=60, seed=7)               # Philox initial phases
    torch.cuda.synchronize()
    pre = lambda w: A.preemphasis(w.cpu().numpy()[0].astype(np.float64))      # undo the inverse pre-emphasis
    e60, e0, er = (_spectral_error(pre(w), S, snd) for w in (wav, wav0, wav_rng))
    ref = A.griffin_lim(S, snd["Frame_Shift"], snd["Frame_Length"], 60, ph.T.astype(np.float64))
    e_ref = _spectral_error(ref, S, snd)
    print("spectral error: gpu 60 it", e60, "gpu 0 it", e0, "oracle 60 it", e_ref, "gpu philox", er)
    assert e60 < 0.5 * e0 and abs(e60 - e_ref) <= 0.05 * e_ref + 1e-3 and er < 0.5 * e0


@pytest.mark.gpu
def test_gpu_export_inference_writes_wavs_and_tables(tmp_path):
    import torch
    from scipy.io import wavfile
    from gst_tacotron_amd import weights
    from gst_tacotron_amd.model import GST_Tacotron
    hp = synthetic.tiny_hp(max_step=24)
    hp["Inference_Path"] = str(tmp_path / "out")
    hp["Vocoder_Taco1"]["Griffin-Lim_Iter"] = 3
    m = GST_Tacotron(hyper_parameters=hp, max_batch=2, max_tokens=16, max_ref_frames=200, max_wav_seconds=1.0)
    m.Restore(weights=weights.synthetic_weights(hp, seed=2))
    rng = np.random.default_rng(0)
    ref_mel = np.clip(rng.normal(0, 1.5, (40, 16)), -4, 4).astype(np.float32)
    sents = ["Hi there.", "Yes."]
    out = m.Inference(sents, [ref_mel], label="T", export=True)
    torch.cuda.synchronize()
    assert out[2] is not None and out[2].shape == (2, 24, 33)
    for i in range(2):
        sr, pcm = wavfile.read(os.path.join(hp["Inference_Path"], "Wav", "T.IDX_%d.WAV" % i))
        from gst_tacotron_amd.export import stop_slice_index
        frames = max(1, stop_slice_index(out[1][i].cpu().numpy())) * 2
        assert sr == 16000 and pcm.dtype == np.int16 and pcm.shape[0] == (16 * (frames - 1) if 16 * (frames - 1) > 32 else 0)
        assert os.path.exists(os.path.join(hp["Inference_Path"], "Plot", "T.IDX_%d.PNG" % i))
    wavs = []
    for i, n in enumerate((3000, 2000)):
        wavs.append(str(tmp_path / ("ref%d.wav" % i)))
        wavfile.write(wavs[-1], 16000, (3000 * np.sin(np.arange(n) * 0.05 * (i + 1))).astype(np.int16))
    gst = m.Inference_GST(wavs, tag_List=["a", "b"], label="T")
    rows = open(os.path.join(hp["Inference_Path"], "GST", "T.GST.TXT")).read().split("\n")
    assert len(rows) == 3 and rows[0].startswith("Wav\tTag\tUnit_0") and gst.shape == (2, 16)
