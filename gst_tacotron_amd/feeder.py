"""Inference-side input conventions of the reference Feeder (reference Feeder.py:161-252).

Only what the hot path's caller needs: tokenise / pad, zero ``initial_mels``, and the
``mels_for_gst`` batch layout (a zero frame is prepended, lengths exclude it).  The wav ->
mel front-end (Pattern_Generator.Mel_Generate, librosa) is a later row, so
reference audio is accepted as precomputed mel arrays ``[T, Mel_Dim]`` in [-Max_Abs_Mel, Max_Abs_Mel].
"""
import numpy as np

from .hparams import load_token_dict


class Feeder:
    def __init__(self, hp, token_index_dict=None):
        self.hp = hp
        self.token_Index_Dict = token_index_dict or load_token_dict(hp)

    def Get_Inference_Pattern(self, sentence_List, mel_List_for_GST=None):
        """reference Feeder.py:161-227.  Out-of-vocabulary characters raise KeyError like the
        reference (Feeder.py:169); bad GST inputs print the reference's message and return None."""
        pattern_Count = len(sentence_List)
        sentence_List = [sentence.upper().strip() for sentence in sentence_List]          # :164
        token_List = [
            np.array([self.token_Index_Dict["<S>"]] + [self.token_Index_Dict[ch] for ch in s]
                     + [self.token_Index_Dict["<E>"]], dtype=np.int32)                     # :166-174
            for s in sentence_List]
        max_len = max(t.shape[0] for t in token_List)
        tokens = np.full((pattern_Count, max_len), self.token_Index_Dict["<E>"], dtype=np.int32)   # :177-180
        for i, t in enumerate(token_List):
            tokens[i, :t.shape[0]] = t
        mel_dim = self.hp["Sound"]["Mel_Dim"]
        pattern = {
            "tokens": tokens,
            "token_lengths": np.array([t.shape[0] for t in token_List], dtype=np.int32),
            "initial_mels": np.zeros((pattern_Count, 1, mel_dim), dtype=np.float32),        # :182-185
        }
        if self.hp["GST"]["Use"]:
            if mel_List_for_GST is None:
                print("GST is enabled, but no wav information.")                           # :197-199
                return None
            if len(mel_List_for_GST) not in (1, pattern_Count):
                print("The length of wav_List_for_GST must be 1 or same to the length of sentence_List "
                      "and wav_List_for_GST must be same.")                                 # :200-202
                return None
            pattern.update(self.Get_Inference_GST_Pattern(
                list(mel_List_for_GST) * (pattern_Count if len(mel_List_for_GST) == 1 else 1)))
        return pattern

    def Get_Inference_GST_Pattern(self, mel_List):
        """reference Feeder.py:229-252: zero-pad to the longest mel, PREPEND one zero frame."""
        mel_dim = self.hp["Sound"]["Mel_Dim"]
        mel_List = [np.asarray(m, dtype=np.float32) for m in mel_List]
        for m in mel_List:
            if m.ndim != 2 or m.shape[1] != mel_dim or m.shape[0] < 1:
                raise ValueError("reference mels must be [T>=1, {}] arrays".format(mel_dim))
        max_len = max(m.shape[0] for m in mel_List)
        mels = np.zeros((len(mel_List), max_len + 1, mel_dim), dtype=np.float32)
        for i, m in enumerate(mel_List):
            mels[i, 1:m.shape[0] + 1] = m
        return {"mels_for_gst": mels,
                "mel_lengths_for_gst": np.array([m.shape[0] for m in mel_List], dtype=np.int32)}
