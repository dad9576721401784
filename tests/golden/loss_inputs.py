"""Deterministic inputs of the loss fixtures, shared by make_golden_loss.py (which feeds them to the genuine reference) and
the tests (which regenerate them instead of storing 768 KB of waveforms)."""
import numpy as np


def make_loss_inputs():
    from speech_enhancement_mi_amd import synth
    B, L = 4, 24000
    mix, clean = synth.synth_utterances(B, L, 3, seed=101)
    noise = synth.hash_tensor("loss.noise", (B, L)) * 0.05
    pred = (0.8 * clean + 0.3 * (mix[:, 0] - clean) + noise).astype(np.float32)
    clean = clean.astype(np.float32).copy()
    clean[3, 6000:15000] = 0.0           # a long silent stretch: exercises removeSilentFrames' compaction
    pred[3, 6000:15000] *= 0.1
    lens = np.array([24000, 20000, 16001, 24000], np.int64)
    return clean, pred, lens
