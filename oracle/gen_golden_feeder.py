#!/usr/bin/env python3
"""Reference-generated fixture for SURVEY row A0 (TEST INFRASTRUCTURE; runs in the build container only).

Executes the REFERENCE's own ``Feeder.Get_Inference_Pattern`` / ``Get_Inference_GST_Pattern``
(/root/reference/Feeder.py:161-252) and stores their inputs and outputs as tests/golden/feeder_tokens.npz.

``import Feeder`` is impossible here: the module imports librosa (absent, not installable) and reads
Hyper_Parameters.json from the CWD at import time.  The two methods themselves are plain NumPy, so this script parses
the reference source with ``ast``, takes exactly those two FunctionDef nodes, and compiles them -- unmodified -- in a
namespace that holds ``np``, the reference's own ``hp_Dict`` (its Hyper_Parameters.json) and its own token dictionary
(Token_Index_Dict.ENG.json).  Nothing of the reference's text is copied into the repository; only the arrays it
produces are.

Cases:
  * ``nogst``   : GST off, the 8 sentences of /root/reference/Inference_Sentence_for_Training.txt:1-8 (the reference's own
                  inference sentences) -> tokens, token_lengths, initial_mels.
  * ``gst_one`` : GST on, one reference mel for all sentences   (Feeder.py:204-207, tiled).
  * ``gst_many``: GST on, one reference mel per sentence, ragged (Feeder.py:208-218, zero padded) and the prepended zero
                  frame (:220-225).
  * ``gst_only``: Get_Inference_GST_Pattern (:229-252).
The wav -> mel front end the GST branches call (``Mel_Generate``, librosa) is NOT part of row A0 (it is row N2): here
it is a lookup that hands back the seeded mel arrays stored in the fixture, so that the reference's batching code runs
on known data.  Error returns (``None`` + message) are recorded as flags.

    python oracle/gen_golden_feeder.py          # needs /root/reference
"""
import ast
import contextlib
import io
import json
import os
import sys

import numpy as np

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "feeder_tokens.npz")
METHODS = ("Get_Inference_Pattern", "Get_Inference_GST_Pattern")


def extract_methods():
    src = open(os.path.join(REF, "Feeder.py")).read()
    tree = ast.parse(src)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "Feeder")
    fns = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in METHODS]
    assert [f.name for f in fns] == list(METHODS), [f.name for f in fns]
    lines = {f.name: (f.lineno, f.end_lineno) for f in fns}
    mod = ast.Module(body=fns, type_ignores=[])
    return compile(mod, os.path.join(REF, "Feeder.py"), "exec"), lines


def main():
    code, lines = extract_methods()
    hp = json.load(open(os.path.join(REF, "Hyper_Parameters.json")))
    token_dict = json.load(open(os.path.join(REF, hp["Token_JSON_Path"])))
    sentences = [s.rstrip("\n") for s in open(os.path.join(REF, "Inference_Sentence_for_Training.txt"))][:8]
    assert len(sentences) == 8

    rng = np.random.default_rng(20260)
    mel_dim = hp["Sound"]["Mel_Dim"]
    mel_lens = [37, 60, 64, 5, 101, 88, 1, 43]
    mel_bank = {"ref{}.wav".format(i): np.clip(rng.normal(0, 1.5, (n, mel_dim)), -4, 4).astype(np.float32)
                for i, n in enumerate(mel_lens)}
    calls = []

    def Mel_Generate(path, top_db, range_Ignore=False):       # stand-in DATA SOURCE for row N2, see the module docstring
        calls.append((path, int(top_db), bool(range_Ignore)))
        return mel_bank[path]

    class Self:
        token_Index_Dict = token_dict

    out = {"sentences": np.array(sentences), "method_lines": json.dumps(lines), "token_dict_json": json.dumps(token_dict),
           "mel_dim": np.int32(mel_dim)}
    for name, arr in mel_bank.items():
        out["mel_" + name] = arr

    def run(gst_on, fn, *args):
        hp_case = json.loads(json.dumps(hp))
        hp_case["GST"]["Use"] = gst_on
        ns = {"np": np, "hp_Dict": hp_case, "Mel_Generate": Mel_Generate}
        exec(code, ns)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            res = ns[fn](Self(), *args)
        return res, buf.getvalue()

    def store(prefix, pat):
        for k, v in pat.items():
            out[prefix + "." + k] = v

    pat, _ = run(False, "Get_Inference_Pattern", sentences)
    store("nogst", pat)

    wavs = ["ref{}.wav".format(i) for i in range(8)]
    calls.clear()
    pat, _ = run(True, "Get_Inference_Pattern", sentences, [wavs[1]])
    store("gst_one", pat)
    out["gst_one.calls"] = json.dumps(calls)
    calls.clear()
    pat, _ = run(True, "Get_Inference_Pattern", sentences, wavs)
    store("gst_many", pat)
    out["gst_many.calls"] = json.dumps(calls)
    calls.clear()
    pat, _ = run(True, "Get_Inference_GST_Pattern", wavs[2:6])
    store("gst_only", pat)
    out["gst_only.calls"] = json.dumps(calls)

    # error behaviour (Feeder.py:197-202): None + a message
    r1, m1 = run(True, "Get_Inference_Pattern", sentences, None)
    r2, m2 = run(True, "Get_Inference_Pattern", sentences, wavs[:3])
    assert r1 is None and r2 is None
    out["err.no_wav_message"] = np.array(m1)
    out["err.bad_count_message"] = np.array(m2)
    # out-of-vocabulary characters raise KeyError (:169)
    try:
        run(False, "Get_Inference_Pattern", ["naïve"])
        raised = False
    except KeyError:
        raised = True
    out["err.oov_raises_keyerror"] = np.bool_(raised)

    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: getattr(v, "shape", None) for k, v in out.items() if k.startswith(("nogst", "gst_many"))})


if __name__ == "__main__":
    sys.exit(main())
