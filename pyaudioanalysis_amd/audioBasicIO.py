"""Host-side audio file access used by the directory walkers.

Only what the feature path needs from the reference's audioBasicIO: opening a file as (sampling rate, samples)
and collapsing two channels into one.  Nothing here touches the GPU.
  * WAV  -> scipy.io.wavfile, as the reference does (audioBasicIO.py:99)
  * AIFF -> stdlib `aifc` (the reference's read_aif, :113-127)
  * mp3 / au / ogg need pydub + ffmpeg in the reference (:130-153); they are reported as undecodable here.
A file that cannot be decoded yields sampling rate -1 and an empty array, like the reference's readers.
"""
import os

import numpy as np

_PCM_CONTAINERS = (".aif", ".aiff")
_NEEDS_FFMPEG = (".mp3", ".au", ".ogg")


def _undecodable(message):
    print(message)
    return -1, np.array([])


def read_audio_file(input_file):
    """(sampling_rate, signal) of an audio file; signal is 1-D for mono and (n, channels) otherwise."""
    suffix = os.path.splitext(str(input_file))[1].lower()
    if suffix == ".wav":
        from scipy.io import wavfile
        try:
            rate, samples = wavfile.read(input_file)
        except Exception:      # the reference lets scipy's error escape; a directory walk skips the file instead
            return _undecodable("Error: read wav file. (DECODING FAILED)")
    elif suffix in _PCM_CONTAINERS:
        try:
            import aifc
            with aifc.open(input_file, "r") as handle:
                payload = handle.readframes(handle.getnframes())
                rate = handle.getframerate()
            samples = np.frombuffer(payload, np.short).byteswap()
        except Exception:
            return _undecodable("Error: read aif file. (DECODING FAILED)")
    elif suffix in _NEEDS_FFMPEG:
        return _undecodable("Error: file not found or other I/O error. (DECODING FAILED)")
    else:
        print("Error: unknown file type {0:s}".format(suffix))
        return 0, np.array([])
    samples = np.asarray(samples)
    if samples.ndim == 2 and samples.shape[1] == 1:
        samples = samples.reshape(-1)
    return rate, samples


def stereo_to_mono(signal):
    """Two channels -> half the right plus half the left channel (float64, audioBasicIO.py:156-168);
    a single column is flattened; anything else is returned untouched."""
    data = np.asarray(signal)
    if data.ndim != 2:
        return signal
    channels = data.shape[1]
    if channels == 1:
        return data.reshape(-1)
    if channels == 2:
        right, left = data[:, 1], data[:, 0]
        return (right / 2) + (left / 2)
    return signal
