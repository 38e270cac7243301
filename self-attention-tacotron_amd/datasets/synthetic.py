"""Synthetic LJSpeech-/VCTK-shaped padded batches.

Follows the reference's input contract: `_prepare_target` (reference datasets/ljspeech/dataset.py:127-167) and
`group_by_batch` padding values (:264-281): source padded with 0, mel with silence_mel_level_db=-3.0, done with 1,
loss masks with 0; target lengths are multiples of outputs_per_step.
"""
import numpy as np


def synthetic_batch(batch_size=32, max_source_length=160, max_target_length=800, num_mels=80, r=2,
                    min_source_length=60, min_target_steps=150, seed=1234, num_symbols=67,
                    silence_mel_level_db=-3.0, num_speakers=0, speaker_offset=0):
    g = np.random.default_rng(seed)
    B, Ti, Tm = batch_size, max_source_length, max_target_length
    Td = Tm // r
    slen = g.integers(min(min_source_length, Ti), Ti + 1, size=B)
    slen[g.integers(0, B)] = Ti
    tsteps = g.integers(min(min_target_steps, Td), Td + 1, size=B)
    tsteps[g.integers(0, B)] = Td
    tlen = tsteps * r
    source = np.zeros((B, Ti), dtype=np.int64)
    mel = np.full((B, Tm, num_mels), silence_mel_level_db, dtype=np.float32)
    done = np.ones((B, Td), dtype=np.float32)
    spec_mask = np.zeros((B, Tm), dtype=np.float32)
    bin_mask = np.zeros((B, Td), dtype=np.float32)
    for b in range(B):
        L = int(slen[b])
        s = g.integers(1, num_symbols + 1, size=L)
        s[0] = 0
        s[-1] = 0
        source[b, :L] = s
        n = int(tlen[b])
        mel[b, :n] = np.clip(g.normal(0, 1, size=(n, num_mels)), -4, 4)
        done[b, :n // r - 1] = 0.0
        spec_mask[b, :n] = 1.0
        bin_mask[b, :n // r] = 1.0
    batch = dict(source=source, source_length=slen.astype(np.int64), mel=mel, target_length=tlen.astype(np.int64),
                 done=done, spec_loss_mask=spec_mask, binary_loss_mask=bin_mask)
    if num_speakers > 0:
        batch["speaker_id"] = (g.integers(0, num_speakers, size=B) + speaker_offset).astype(np.int64)
    return batch
