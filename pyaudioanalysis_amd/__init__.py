"""pyaudioanalysis_amd -- MI355X-native drop-in for ONE path of tyiannak/pyAudioAnalysis:
ShortTermFeatures.feature_extraction / spectrogram / chromagram and
MidTermFeatures.mid_feature_extraction (+ a batched many-clip form), plus the adjacent rows of SURVEY 8f
(directory walkers, beat extraction, audioSegmentation.self_similarity_matrix / music_thumbnailing).

Python host -> ctypes -> libpaa_hip.so (hand-written gfx950 HIP kernels).  No PyTorch, no CPU
fallback: importing works anywhere, computing needs the built library and a HIP device.

    from pyaudioanalysis_amd import ShortTermFeatures, MidTermFeatures
"""
from . import MidTermFeatures, ShortTermFeatures, audioSegmentation  # noqa: F401

__all__ = ["ShortTermFeatures", "MidTermFeatures", "audioSegmentation"]
__version__ = "0.1.0"
