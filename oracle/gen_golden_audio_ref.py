#!/usr/bin/env python3
"""Reference-generated fixture for the librosa-free part of the audio path (rows N2 / N4; TEST INFRASTRUCTURE, build
container only).

/root/reference/Audio.py cannot be imported (it imports librosa, absent).  Eight of its functions are plain NumPy / SciPy:
``preemphasis``, ``inv_preemphasis`` (:11-15), ``_amp_to_db``, ``_db_to_amp`` (:86-90), ``_normalize``,
``_symmetric_normalize``, ``_denormalize``, ``_symmetric_denormalize`` (:92-102).  This script takes exactly those
FunctionDef nodes out of the parsed source (``ast``), compiles them unmodified with ``np`` and ``scipy.signal`` in scope,
runs them on seeded inputs and stores inputs + outputs in tests/golden/audio_ref_pure.npz.  tests/test_oracle.py holds
oracle/audio_np.py to these vectors (the functions that DO need librosa -- STFT, mel basis, trim, Griffin-Lim -- stay
unpinned and say so).

    python oracle/gen_golden_audio_ref.py       # needs /root/reference
"""
import ast
import os

import numpy as np
from scipy import signal

REF = "/root/reference/Audio.py"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "audio_ref_pure.npz")
NAMES = ["preemphasis", "inv_preemphasis", "_amp_to_db", "_db_to_amp", "_normalize", "_symmetric_normalize", "_denormalize",
         "_symmetric_denormalize"]


def main():
    tree = ast.parse(open(REF).read())
    fns = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in NAMES]
    assert sorted(f.name for f in fns) == sorted(NAMES)
    ns = {"np": np, "signal": signal}
    exec(compile(ast.Module(body=fns, type_ignores=[]), REF, "exec"), ns)
    rng = np.random.default_rng(7)
    sig = (rng.normal(0, 0.2, 1500) * np.hanning(1500)).astype(np.float32)
    mag = np.abs(rng.normal(0, 1, (40, 20))) ** 3          # spans 1e-7 .. 30: exercises the 1e-5 floor
    mag[0, :5] = 0.0
    db = rng.uniform(-140, 20, (40, 20))
    nrm = rng.uniform(-5, 5, (40, 20))
    out = {"sig": sig, "mag": mag, "db": db, "nrm": nrm,
           "preemphasis": ns["preemphasis"](sig), "inv_preemphasis": ns["inv_preemphasis"](sig),
           "roundtrip": ns["inv_preemphasis"](ns["preemphasis"](sig)),
           "amp_to_db": ns["_amp_to_db"](mag), "db_to_amp": ns["_db_to_amp"](db),
           "normalize": ns["_normalize"](db), "symmetric_normalize": ns["_symmetric_normalize"](db, max_abs_value=4),
           "denormalize": ns["_denormalize"](nrm), "symmetric_denormalize": ns["_symmetric_denormalize"](nrm, max_abs_value=4)}
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: (v.dtype, v.shape) for k, v in out.items()})


if __name__ == "__main__":
    main()
