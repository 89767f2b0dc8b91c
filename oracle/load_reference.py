"""Import the UNMODIFIED reference (tyiannak/pyAudioAnalysis) from /root/reference.

TEST INFRASTRUCTURE ONLY.  Only oracle/make_golden.py (run in the build
container, where /root/reference exists) may call this.  Nothing in the
product package, the -m gpu tests, smoke() or bench.py reads /root/reference.

audioBasicIO.py:5,9 imports `eyed3` and `pydub`, which are not installed and
are only used for mp3 tag reading / non-WAV decoding; two empty stub modules
are injected so that `pyAudioAnalysis.MidTermFeatures` becomes importable.
"""
import os
import sys
import types
import warnings

REFERENCE_ROOT = os.environ.get("PAA_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "pyAudioAnalysis"))


def load():
    """Return (ShortTermFeatures, MidTermFeatures, audioBasicIO) reference modules."""
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    for name in ("eyed3", "pydub"):
        if name not in sys.modules:
            stub = types.ModuleType(name)
            if name == "pydub":
                stub.AudioSegment = object
            sys.modules[name] = stub
    os.environ.setdefault("MPLBACKEND", "Agg")
    import numpy
    # utilities.peakdet (utilities.py:74-75) still spells numpy.Inf / numpy.NaN, removed in NumPy 2.0: restore the
    # aliases so that the UNMODIFIED reference runs under the installed numpy 2.2 (environment shim only)
    if not hasattr(numpy, "Inf"):
        numpy.Inf = numpy.inf
    if not hasattr(numpy, "NaN"):
        numpy.NaN = numpy.nan
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from pyAudioAnalysis import ShortTermFeatures as ref_st
        from pyAudioAnalysis import MidTermFeatures as ref_mt
        from pyAudioAnalysis import audioBasicIO as ref_io
    return ref_st, ref_mt, ref_io


def load_segmentation():
    """The reference's audioSegmentation module (for the self-similarity / thumbnailing goldens).  Its imports of
    hmmlearn, imblearn, plotly (via audioTrainTest) are not installed and are not used by
    self_similarity_matrix / music_thumbnailing: empty stub modules stand in for them."""
    load()
    stubs = ["hmmlearn", "hmmlearn.hmm", "imblearn", "imblearn.under_sampling", "imblearn.over_sampling",
             "plotly", "plotly.subplots", "plotly.graph_objs", "simplejson"]
    for name in stubs:
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["imblearn.under_sampling"].RandomUnderSampler = object
    sys.modules["imblearn.over_sampling"].SMOTE = object
    sys.modules["hmmlearn"].hmm = sys.modules["hmmlearn.hmm"]
    sys.modules["plotly"].subplots = sys.modules["plotly.subplots"]
    sys.modules["plotly"].graph_objs = sys.modules["plotly.graph_objs"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from pyAudioAnalysis import audioSegmentation as ref_seg
    return ref_seg
