"""Pins the oracle to the reference EXECUTED under TensorFlow -- the one-command job for whoever has a TF-equipped build container:

    python -m oracle.gen_golden_tf [--reference /root/reference]          # writes tests/golden/tf_*.npz
    python -m pytest tests/test_oracle.py -k tensorflow                   # then holds oracle_np to <= 1e-5 of them

TEST INFRASTRUCTURE, not product.  PARITY UNPINNED until this has been run: TensorFlow (reference Requirements.txt:3) is not
installed in the build container and cannot be installed (no network), so this script has never executed past its import check
here; it is written against the reference sources and Keras' documented behaviour, and it fails loudly (never guesses) where a
real TensorFlow disagrees with what it expects.  Nothing of the reference is copied: its modules are IMPORTED from where they
lie, and only arrays (inputs, expected outputs, checksums) are written.

What it does, per case, in a fresh interpreter (the reference's modules read ``Hyper_Parameters.json`` from the CURRENT DIRECTORY
at import time -- Modules/Taco2.py:6-10, Modules/GST.py:6-10 -- so a case = a scratch directory holding that case's JSON):
  1. builds the layers and the functional Inference model exactly as ``GST_Tacotron.Model_Generate`` does (reference Model.py:45-72 the
     Inputs, :76-82 the layers, :108-129 the inference tensors, :145-156 ``model_Dict['Inference']``) -- without importing Model.py itself,
     which drags in librosa / matplotlib / the Feeder;
  2. assigns ``gst_tacotron_amd.weights.synthetic_weights(hp, seed)`` to the Keras variables.  Each manifest name is resolved to its
     variable by walking the Python object graph along ``gst_tacotron_amd.tf_checkpoint.reference_paths`` -- the same paths a
     ``tf.train.Checkpoint`` of the reference uses -- so a successful run also validates that table against real
     Keras objects; every shape is checked, every variable of the model must receive a value (the unused attention ``Key`` Dense,
     SURVEY F12, is never built and has none);
  3. runs ``model_Dict['Inference'](inputs=[initial_mels, tokens, mels_for_gst, mel_lengths_for_gst], training=False)`` -- the body
     of ``Inference_Step`` (Model.py:249-255) -- in the reference's DETERMINISTIC setting: ``Prenet.Dropout_Rate = 0``
     (the prenet's dropout is live at inference, Taco2.py:283) and ``Attention.Type = "BMA"`` (sigmoid_noise 0.0, Steps.py:58;
     SMA always adds noise, Steps.py:212);
  4. writes inputs + (mel, stop, spectrogram, alignment) + weight checksums + the TensorFlow version to ``tests/golden/tf_<case>.npz``.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")


def cases():
    """name -> (hp, weight seed, input seed, B, T_v, T_ref, ref_lengths).  Deterministic settings only (see the module docstring)."""
    from gst_tacotron_amd import synthetic
    c = {}
    c["tf_tiny_bma_r2_gst"] = (synthetic.tiny_hp("BMA", r=2, gst=True, max_step=24, prenet_rate=0.0), 3, 5, 3, 12, 70, [70, 33, 64])
    c["tf_tiny_bma_r1_nogst"] = (synthetic.tiny_hp("BMA", r=1, gst=False, max_step=16, prenet_rate=0.0), 5, 7, 2, 9, 0, None)
    hp = synthetic.config_hp("cfg2")                      # LJSpeech dimensions, a short trajectory
    hp["Tacotron2"]["Decoder"]["Attention"]["Type"] = "BMA"
    hp["Tacotron2"]["Decoder"]["Prenet"]["Dropout_Rate"] = 0.0
    hp["Max_Step"] = 40
    c["tf_full_bma_r2_short"] = (hp, 0, 11, 2, 16, 96, [96, 50])
    return c


def resolve(obj, path):
    """The Keras variable at object-graph path `path` below `obj` (the naming tf.train.Checkpoint uses: attribute names, dictionary
    keys, ``layer_with_weights-N`` = the N-th layer of a Sequential / Model that has weights, list indices)."""
    for part in path.split("/"):
        if part.startswith("layer_with_weights-"):
            n = int(part.split("-")[1])
            with_w = [l for l in obj.layers if l.weights]
            if n >= len(with_w):
                raise KeyError("{}: only {} layers with weights under {!r}".format(part, len(with_w), obj))
            obj = with_w[n]
        elif isinstance(obj, dict):
            obj = obj[part]
        elif isinstance(obj, (list, tuple)):
            obj = obj[int(part)]
        else:
            obj = getattr(obj, part)
    return obj


def run_case(name, reference):
    import tensorflow as tf                                     # noqa: F401  (checked by main() before any case is spawned)
    sys.path.insert(0, ROOT)
    from gst_tacotron_amd import synthetic, weights
    from gst_tacotron_amd.hparams import Dims
    from gst_tacotron_amd.tf_checkpoint import reference_paths
    hp, wseed, iseed, B, Tv, Tref, ref_lengths = cases()[name]
    d = Dims(hp)
    # the scratch directory the reference's modules read their configuration from (cwd-relative opens at import time)
    work = tempfile.mkdtemp(prefix="gsttaco_tf_")
    hp_file = dict(hp)
    hp_file["Token_JSON_Path"] = "Token_Index_Dict.ENG.json"
    hp_file.setdefault("Use_Mixed_Precision", False)
    json.dump(hp_file, open(os.path.join(work, "Hyper_Parameters.json"), "w"))
    token_dict = json.load(open(os.path.join(ROOT, "gst_tacotron_amd", "Token_Index_Dict.ENG.json")))
    json.dump(token_dict, open(os.path.join(work, "Token_Index_Dict.ENG.json"), "w"))
    os.chdir(work)
    sys.path.insert(0, reference)
    from Modules import Taco2 as Modules                        # reference Model.py:25
    from Modules.GST import GST_Concated_Encoder, Style_Token_Layer

    # ---- Model_Generate, inference half (reference Model.py:45-72, 76-82, 108-129, 145-156)
    inp = {"Mel": tf.keras.layers.Input(shape=[None, d.mel], dtype=tf.float32),
           "Token": tf.keras.layers.Input(shape=[None], dtype=tf.int32)}
    layer = {"encoder": Modules.Encoder(), "decoder": Modules.Decoder(), "vocoder": Modules.Vocoder_Taco1()}
    enc = layer["encoder"](inp["Token"], training=False)
    model_inputs = [inp["Mel"], inp["Token"]]
    if d.gst:
        inp["GST_Mel"] = tf.keras.layers.Input(shape=[None, d.mel], dtype=tf.float32)
        inp["Mel_Length"] = tf.keras.layers.Input(shape=[], dtype=tf.int32)
        layer["gst"] = Style_Token_Layer()
        gst = layer["gst"]([inp["GST_Mel"], inp["Mel_Length"]])
        enc = GST_Concated_Encoder()([enc, gst])
        model_inputs += [inp["GST_Mel"], inp["Mel_Length"]]
    _, mel, stop, align = layer["decoder"]([enc, inp["Mel"]], training=False)
    spec = layer["vocoder"](mel, training=False)
    model = tf.keras.Model(inputs=model_inputs, outputs=[mel, stop, spec, align])

    # ---- weights: manifest name -> Keras variable through the checkpoint object paths
    w = weights.synthetic_weights(hp, seed=wseed)
    assigned = set()
    for mname, (tag, path) in reference_paths(hp).items():
        if tag not in layer:
            raise KeyError("reference_paths names top-level layer {!r} for {}; built: {}".format(tag, mname, sorted(layer)))
        var = resolve(layer[tag], path)
        want = tuple(np.shape(w[mname]))
        got = tuple(var.shape)
        if got != want:
            raise ValueError("{} -> {}/{}: Keras variable has shape {}, manifest {}".format(mname, tag, path, got, want))
        var.assign(w[mname])
        assigned.add(var.ref() if hasattr(var, "ref") else id(var))
    missing = [v.name for v in model.variables if (v.ref() if hasattr(v, "ref") else id(v)) not in assigned]
    if missing:
        raise KeyError("Keras variables the manifest does not cover: {}".format(missing))
    if len(assigned) != len(w):
        raise KeyError("{} manifest tensors, {} distinct Keras variables assigned".format(len(w), len(assigned)))

    # ---- Inference_Step (reference Model.py:249-255); inputs as Feeder.Get_Inference_Pattern shapes them (Feeder.py:161-227)
    rng = np.random.default_rng(iseed)
    tokens, tl = synthetic.make_tokens(rng, B, Tv)
    initial_mels = np.zeros((B, 1, d.mel), np.float32)
    feed = [initial_mels, tokens]
    mels = ml = None
    if d.gst:
        mels, ml = synthetic.make_ref_mels(rng, B, Tref, mel=d.mel, lengths=None if ref_lengths is None else np.array(ref_lengths))
        feed += [mels, ml]
    outs = model(inputs=feed, training=False)
    mel_o, stop_o, spec_o, align_o = [np.asarray(o, dtype=np.float32) for o in outs]
    names = sorted(w)
    out = {"hp_json": np.array(json.dumps(hp)), "wseed": np.array(wseed), "steps": np.array(d.steps),
           "weight_checksums": np.array([[float(np.sum(w[n], dtype=np.float64)), float(np.sum(np.abs(w[n]), dtype=np.float64))] for n in names]),
           "tokens": tokens, "token_lengths": tl, "mels": mel_o, "stops": stop_o, "spectrograms": spec_o, "alignments": align_o,
           "tensorflow_version": np.array(tf.__version__)}
    if d.gst:
        out["mels_for_gst"], out["mel_lengths_for_gst"] = mels, ml
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB (TensorFlow {})".format(tf.__version__))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--case", default=None, help="internal: run ONE case in this interpreter")
    args = ap.parse_args()
    try:
        import tensorflow  # noqa: F401
    except ImportError:
        print("oracle.gen_golden_tf: TensorFlow is not importable here (the reference needs tensorflow>=2.1.2, Requirements.txt:3); "
              "nothing written -- parity stays UNPINNED for the model arithmetic", file=sys.stderr)
        return 3
    if not os.path.isdir(os.path.join(args.reference, "Modules")):
        print("oracle.gen_golden_tf: no reference tree at " + args.reference, file=sys.stderr)
        return 4
    if args.case:
        run_case(args.case, os.path.abspath(args.reference))
        return 0
    sys.path.insert(0, ROOT)
    for name in cases():                    # one interpreter per case: the reference caches its configuration at import
        rc = subprocess.call([sys.executable, "-m", "oracle.gen_golden_tf", "--reference", args.reference, "--case", name], cwd=ROOT)
        if rc:
            return rc
    return 0


if __name__ == "__main__":
    sys.exit(main())
