"""The RCCL path of the sharded many-clip extraction with TWO ranks (-m gpu).

With >= 2 visible devices: two processes, one per GPU, run distributed.extract_sharded() through RcclGather (grouped
ncclSend / ncclRecv) and rank 0 checks every clip against the single-GPU result.
With ONE device (the usual gpurun box): both ranks would have to share device 0, on which ncclCommInitRank never
returns.  RcclGather compares the PCI bus ids of all ranks over the control plane before RCCL is touched, so BOTH ranks
refuse with a clear message -- promptly, without a hang or a crash -- and the library keeps working afterwards.  The
library's own backstop for callers without a control plane (node-local marker files keyed on job id + bus id,
paa_comm_init) is exercised by planting the marker of a live "other rank"."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _rank_main(rank, world, n_dev, comm_id, res_q, slots, gate):
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

        def exchange(payload):       # all-gather over shared memory: slot per rank, then a barrier
            slots[64 * rank:64 * rank + 64] = payload[:63].ljust(64, b"\0")
            gate.wait(timeout=60)
            return [bytes(slots[64 * r:64 * r + 64]).rstrip(b"\0") for r in range(world)]

        lens = [4000, 16000, 9000, 800, 5200, 2500, 24000, 1199]
        clips = [synth_clip(4000 + i, n) for i, n in enumerate(lens)]
        try:
            comm = D.RcclGather(world, rank, bcast, exchange)
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
    slots, gate = ctx.Array("c", 2 * 64, lock=False), ctx.Barrier(2)
    procs = [ctx.Process(target=_rank_main, args=(r, 2, n_dev, comm_id, res_q, slots, gate)) for r in range(2)]
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


def _marker_refusal_child(marker_dir, queue):
    """Child of the test below (own process: should the guard ever fail, ncclCommInitRank waits for a rank that does not
    exist -- the parent kills this process instead of hanging the suite)."""
    import ctypes
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["PAA_COMM_MARKER_DIR"] = marker_dir
    from pyaudioanalysis_amd import _ffi
    lib = _ffi.lib()
    _ffi.init(0)
    buf = ctypes.create_string_buffer(_ffi.COMM_ID_BYTES)
    _ffi.check(lib.paa_comm_unique_id(buf))
    name = ctypes.create_string_buffer(256)
    _ffi.check(lib.paa_debug_comm_marker_name(buf, 1, name, 256))
    other = name.value.decode()
    with open(os.path.join(marker_dir, other), "w") as f:          # "rank 1" = the parent of this process: alive
        f.write("%d\n" % os.getppid())
    rc = lib.paa_comm_init(2, 0, buf)
    msg = _ffi.last_error()
    lib.paa_comm_destroy()                                           # removes rank 0's own marker
    queue.put((rc, msg, other, sorted(os.listdir(marker_dir))))


def test_comm_init_refuses_a_device_another_live_rank_of_the_job_holds(gpu_lib, tmp_path):
    """Backstop inside paa_comm_init (no control plane): a marker for this job id and this device that names a live
    process as "rank 1" makes rank 0 fail with PAA_ERR_COMM before RCCL is called; paa_comm_destroy removes rank 0's
    own marker again.  The marker's name comes from the library (paa_debug_comm_marker_name: host, pid namespace and PCI
    bus id are part of it)."""
    import multiprocessing as mp
    from pyaudioanalysis_amd import _ffi
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_marker_refusal_child, args=(str(tmp_path), q))
    p.start()
    p.join(120)
    if p.is_alive():
        p.kill()
        p.join()
        pytest.fail("paa_comm_init did not refuse the device within 120 s (it is waiting inside ncclCommInitRank)")
    assert p.exitcode == 0, p.exitcode
    rc, msg, other, left = q.get(timeout=10)
    assert rc == _ffi.ERR_COMM, (rc, msg)
    assert "one process per GPU" in msg, msg
    assert other.startswith("paa_comm_") and other.endswith(".1")
    assert left == [other]


def test_bench_n2_socket_control_plane_without_rccl(gpu_lib, tmp_path):
    """`bench.py --gpus 2 --no-gather` as the driver launches it (RANK / WORLD_SIZE / MASTER_* in the environment, one
    process per rank): on this one-GPU box both ranks share device 0, so RCCL stays out (--no-gather) and what runs is the
    N > 1 code path itself -- partition_by_frames, the socket control plane (barrier, max over ranks), the per-rank plans,
    the in-line parity check -- ending in one parseable JSON line with ranks = 2."""
    import json
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen(
            [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-gather", "--clips", "2000", "--steps", "5",
             "--warmup", "2", "--prewarm-seconds", "0", "--no-cpu-baseline"],
            env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT))
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=300))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("bench.py --gpus 2 --no-gather did not finish within 300 s")
    assert all(p.returncode == 0 for p in procs), [(p.returncode, o[1][-800:]) for p, o in zip(procs, outs)]
    lines = [ln for ln in outs[0][0].splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and not [ln for ln in outs[1][0].splitlines() if ln.startswith("{")]      # rank 0 prints, once
    line = json.loads(lines[0])
    # n_gpus counts DEVICES (advisor, round 5): two ranks on this box's one GPU are `ranks: 2, n_gpus: 1, scaling: null`
    distinct = line["config"]["distinct_devices"]
    assert line["ranks"] == 2 and line["n_gpus"] == distinct and line["value"] > 0
    assert line["scaling"] == ("strong" if distinct == 2 else None)
    assert line["config"]["clips_in_job"] == 2000 and line["config"]["frames_per_step_job"] == 2000 * 399
    assert line["config"]["frames_per_step_rank0"] == 1000 * 399
    assert line["parity_check"]["status"] == "ok", line["parity_check"]


def _run_bare_bench(extra, timeout=420):
    """`python bench.py --gpus 2 ...` with NO launcher environment (RANK / WORLD_SIZE / MASTER_* removed): bench.py must spawn its
    own ranks.  -> (returncode, stdout, stderr)"""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--clips", "2000", "--steps", "5", "--warmup", "2",
           "--prewarm-seconds", "0", "--no-cpu-baseline"] + extra
    try:
        res = subprocess.run(cmd, env=env, capture_output=True, text=True, cwd=ROOT, timeout=timeout)
    except subprocess.TimeoutExpired:
        pytest.fail("bare bench.py --gpus 2 %s did not finish within %d s" % (" ".join(extra), timeout))
    return res.returncode, res.stdout, res.stderr


def test_bare_bench_gpus_2_launches_its_own_ranks(gpu_lib):
    """`python bench.py --gpus 2 --no-gather` as the driver runs N = 1 (no RANK / WORLD_SIZE in the
    environment) becomes a two-rank run by itself and says so: ranks = 2 (n_gpus = the devices that ran), one JSON line, the device of every rank in
    config.devices, config.rccl_ranks = null (no exchange asked for), the compact `configs` object last in the line."""
    import json
    rc, out, err = _run_bare_bench(["--no-gather"])
    assert rc == 0, err[-1500:]
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out[-1500:]
    line = json.loads(lines[0])
    distinct = line["config"]["distinct_devices"]
    assert line["ranks"] == 2 and line["n_gpus"] == distinct == len(set(line["config"]["devices"])) and line["value"] > 0
    assert line["scaling"] == ("strong" if distinct == 2 else None)
    assert len(lines[0]) < 8000
    assert line["config"]["frames_per_step_job"] == 2000 * 399 and line["config"]["frames_per_step_rank0"] == 1000 * 399
    assert len(line["config"]["devices"]) == 2 and line["config"]["rccl_ranks"] is None
    assert "bench.py itself" in line["config"]["launched_by"]
    assert line["parity_check"]["status"] == "ok", line["parity_check"]
    assert list(line)[-1] == "configs" and "cfg4_job" in line["configs"]


def test_bare_bench_gpus_2_with_the_gather_runs_rccl_or_refuses_loudly(gpu_lib):
    """The same with the RCCL gather.  Two devices: a real two-rank job (config.rccl_ranks = 2, two distinct devices).  One
    device (the usual test box): refused before any rank starts -- non-zero exit, a message that names the device count, and NO
    JSON line (never `n_gpus: 1` under a `--gpus 2` command)."""
    import json
    from pyaudioanalysis_amd import _ffi
    rc, out, err = _run_bare_bench([])
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    if _ffi.device_count() >= 2:
        assert rc == 0, err[-1500:]
        line = json.loads(lines[0])
        assert line["n_gpus"] == 2 and line["config"]["rccl_ranks"] == 2 and line["config"]["distinct_devices"] == 2
        assert line["parity_check"]["status"] == "ok"
    else:
        assert rc != 0 and not lines, (rc, out[-800:])
        assert "--gpus 2" in err and "1 HIP device(s) visible" in err, err[-800:]


def test_two_processes_on_one_device_cold_starts(gpu_lib, tmp_path):
    """the two-ranks-on-one-device path every shared-device test rides on once died with a GPU memory-access
    fault on a cold box (first of six identical runs).  Three cold starts of the bare two-rank bench here (the 20-run record of
    the round is profiles/r06_two_proc_stress.txt, scripts/two_proc_stress.py): every run must end in a line with parity ok."""
    import subprocess
    out = tmp_path / "stress.txt"
    res = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "two_proc_stress.py"), "--runs", "3", "--clips", "20000",
                          "--out", str(out)], capture_output=True, text=True, cwd=ROOT, timeout=900)
    text = out.read_text() if out.exists() else ""
    assert res.returncode == 0, (res.stdout[-1500:], text[-3000:])
    assert "3 runs, 0 failures, 0 GPU memory-access faults" in text
