"""Export side of Inference: the artefacts reference Model.py:369-425 and :448-459 write.

The heavy part -- Griffin-Lim on the predicted linear spectrograms (Audio.py:23-27, 57-68) -- runs on the GPU for the
whole batch at once (``gsttaco_griffin_lim``); this module slices by the stop token, writes 16-bit wavs, the optional
matplotlib figure per utterance and the GST embedding table."""
import os
import wave

import numpy as np


def stop_slice_index(stop):
    """reference Model.py:380: first step whose stop logit is negative, else the number of steps."""
    neg = np.flatnonzero(np.asarray(stop) < 0)
    return int(neg[0]) if neg.size else int(len(stop))


def write_wav(path, signal, sample_rate):
    """16-bit PCM like reference Model.py:423-427 ((sig * 32768).astype(int16)); out-of-range samples are clipped
    instead of wrapping."""
    pcm = np.clip(np.asarray(signal, dtype=np.float64) * 32768.0, -32768, 32767).astype("<i2")
    with wave.open(path, "wb") as f:
        f.setnchannels(1)
        f.setsampwidth(2)
        f.setframerate(int(sample_rate))
        f.writeframes(pcm.tobytes())


def plot_inference(path, sentence, mel, spect, alignment, stop, slice_index):
    """Mel / spectrogram / alignment / stop-token figure (reference Model.py:382-411).  Needs matplotlib."""
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    fig, axes = plt.subplots(4, 1, figsize=(24, 30), dpi=100, gridspec_kw={"height_ratios": [1, 1, 2, 1]})
    panels = (("Mel", mel), ("Spectrogram", spect), ("Alignment", alignment))
    for ax, (title, data) in zip(axes, panels):
        if data is None:
            ax.set_axis_off()
            continue
        im = ax.imshow(np.transpose(data), aspect="auto", origin="lower")
        ax.set_title("{}    Sentence: {}".format(title, sentence))
        fig.colorbar(im, ax=ax)
    labels = ["<S>"] + list(sentence) + ["<E>"]
    if alignment is not None and alignment.shape[1] == len(labels):
        axes[2].set_yticks(range(len(labels)))
        axes[2].set_yticklabels(labels, fontsize=10)
    axes[3].plot(stop)
    axes[3].axvline(x=slice_index, linestyle="--", linewidth=1)
    axes[3].set_title("Stop token    Sentence: {}".format(sentence))
    fig.tight_layout()
    fig.savefig(path)
    plt.close(fig)


def export_gst(path, wav_List, tag_List, gst_List):
    """reference Model.py:448-459: tab-separated Wav, Tag, Unit_0..Unit_{n-1}."""
    gst_List = np.asarray(gst_List)
    lines = ["\t".join(["Wav", "Tag"] + ["Unit_{}".format(i) for i in range(gst_List.shape[1])])]
    for wav_path, tag, gst in zip(wav_List, tag_List, gst_List):
        lines.append("\t".join(["{}".format(x) for x in [wav_path, tag] + list(gst)]))
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    with open(path, "w") as f:
        f.write("\n".join(lines))
