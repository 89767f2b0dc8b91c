"""The two audioBasicIO functions the feature path touches (reference: pyAudioAnalysis/audioBasicIO.py).

Host-side file I/O only -- nothing here runs on the GPU.  WAV goes through scipy.io.wavfile exactly like the
reference (:99); AIFF through the stdlib `aifc`; mp3/au/ogg need pydub/ffmpeg in the reference (:100-101) and
report a decoding failure here.
"""
import os

import numpy as np


def read_audio_file(input_file):
    """Returns (sampling_rate, signal) like audioBasicIO.read_audio_file (:86-110); (-1, []) on decode failure."""
    sampling_rate = 0
    signal = np.array([])
    extension = os.path.splitext(str(input_file))[1].lower()
    if extension == ".wav":
        from scipy.io import wavfile
        try:
            sampling_rate, signal = wavfile.read(input_file)
        except Exception:         # the reference lets scipy's error escape (:99); a directory walk skips the file
            sampling_rate, signal = -1, np.array([])
            print("Error: read wav file. (DECODING FAILED)")
    elif extension in (".aif", ".aiff"):
        sampling_rate = -1
        try:
            import aifc
            with aifc.open(input_file, "r") as s:
                raw = s.readframes(s.getnframes())
                signal = np.frombuffer(raw, np.short).byteswap()
                sampling_rate = s.getframerate()
        except Exception:
            print("Error: read aif file. (DECODING FAILED)")
    elif extension in (".mp3", ".au", ".ogg"):
        sampling_rate = -1
        print("Error: file not found or other I/O error. (DECODING FAILED)")
    else:
        print("Error: unknown file type {extension}")
    if signal.ndim == 2 and signal.shape[1] == 1:
        signal = signal.flatten()
    return sampling_rate, signal


def stereo_to_mono(signal):
    """(:156-168): two channels -> (R/2) + (L/2) as float64; one-column input is flattened."""
    if signal.ndim == 2:
        if signal.shape[1] == 1:
            signal = signal.flatten()
        else:
            if signal.shape[1] == 2:
                signal = (signal[:, 1] / 2) + (signal[:, 0] / 2)
    return signal
