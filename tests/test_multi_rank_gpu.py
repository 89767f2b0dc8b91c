"""The RCCL path of the sharded many-clip extraction with TWO ranks (-m gpu).

With >= 2 visible devices: two processes, one per GPU, run distributed.extract_sharded() through RcclGather (grouped
ncclSend / ncclRecv) and rank 0 checks every clip against the single-GPU result.
With ONE device (the usual gpurun box): both ranks would have to share device 0, on which ncclCommInitRank never
returns; paa_comm_init refuses that before calling RCCL.  The test asserts that it fails on both ranks with PAA_ERR_COMM
and a clear message -- promptly, without a hang or a crash -- and that the library keeps working afterwards."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _rank_main(rank, world, n_dev, comm_id, res_q):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        from synth import synth_clip
        from pyaudioanalysis_amd import ShortTermFeatures, _ffi
        from pyaudioanalysis_amd import distributed as D
        _ffi.lib()
        _ffi.init(rank % n_dev)

        def bcast(payload):          # the parent process created the unique id (it hosts the RCCL bootstrap root)
            return comm_id

        lens = [4000, 16000, 9000, 800, 5200, 2500, 24000, 1199]
        clips = [synth_clip(4000 + i, n) for i, n in enumerate(lens)]
        try:
            comm = D.RcclGather(world, rank, bcast)
        except _ffi.HipLibraryError as exc:
            # the library must stay usable after a failed communicator init
            single, _ = ShortTermFeatures.feature_extraction(clips[1], 16000, 800, 400)
            res_q.put((rank, "comm_error", str(exc), bool(np.all(np.isfinite(single)))))
            return
        try:
            res = D.extract_sharded(clips, 16000, 800, 400, True, world, rank, comm)
            comm.barrier()
            ok = True
            if rank == 0:
                ok = res is not None and len(res) == len(clips)
                for c, r in zip(clips, res or []):
                    single, _ = ShortTermFeatures.feature_extraction(c, 16000, 800, 400)
                    ok &= bool(np.array_equal(single, r))
            res_q.put((rank, "ok" if ok else "mismatch", "", True))
        finally:
            comm.close()
    except Exception as exc:  # anything else is a test failure, reported by the parent
        res_q.put((rank, "exception", repr(exc), False))


def test_two_ranks_rccl_gather_or_clean_failure(gpu_lib):
    import multiprocessing as mp
    from pyaudioanalysis_amd import _ffi
    n_dev = _ffi.device_count()
    ctx = mp.get_context("spawn")
    import ctypes
    res_q = ctx.Queue()
    buf = ctypes.create_string_buffer(_ffi.COMM_ID_BYTES)
    _ffi.check(gpu_lib.paa_comm_unique_id(buf))
    comm_id = bytes(buf.raw)
    procs = [ctx.Process(target=_rank_main, args=(r, 2, n_dev, comm_id, res_q)) for r in range(2)]
    for p in procs:
        p.start()
    results = {}
    try:
        for _ in range(2):
            rank, status, msg, alive = res_q.get(timeout=90)
            results[rank] = (status, msg, alive)
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():          # never leave a rank behind on the GPU box
                p.terminate()
                p.join(timeout=10)
    assert set(results) == {0, 1}, results
    if n_dev >= 2:
        assert all(v[0] == "ok" for v in results.values()), results
    else:
        for status, msg, alive in results.values():
            assert status == "comm_error", results
            assert "one process per GPU" in msg, msg                            # PAA_ERR_COMM says what is wrong
            assert alive                                                        # single-GPU extraction still works
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
