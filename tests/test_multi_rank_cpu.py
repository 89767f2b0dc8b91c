"""World-size-2 test of the sharded many-clip path on CPU (gloo): the workers call the product's own
distributed.extract_sharded() -- partitioning by frames, per-rank blocks, gather to rank 0, re-assembly -- with a gloo
communicator and an oracle-backed engine injected in place of RcclGather / HipEngine (no GPU here); the exchanged bytes,
counts and layout are exactly those of the RCCL path."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _HostBuf:
    def __init__(self, arr):
        self.arr = arr


class _OracleEngine:
    """Stands in for distributed.HipEngine on a host without a GPU: same four methods, NumPy buffers, the CPU oracle as
    the per-rank compute (test infrastructure -- the product module never imports it)."""

    def __init__(self, oracle):
        self.O = oracle
        self.extract_calls = 0

    def upload(self, flat):
        return _HostBuf(np.array(flat, dtype=np.float64))

    def extract(self, clips, sampling_rate, window, step, deltas):
        self.extract_calls += 1
        return _HostBuf(np.concatenate([self.O.feature_extraction(c, sampling_rate, window, step, deltas)[0].reshape(-1)
                                        for c in clips]))

    def extract_mid(self, clips, sampling_rate, window, step, mid_ratio, mid_step_ratio):
        self.extract_calls += 1
        out = []
        for c in clips:
            st, _ = self.O.feature_extraction(c, sampling_rate, window, step, True)
            # the reference's loop (MidTermFeatures.py:116-124) on the frame ratios extract_sharded derived
            cols = [np.concatenate([st[:, p:p + mid_ratio].mean(axis=1), st[:, p:p + mid_ratio].std(axis=1)])
                    for p in range(0, st.shape[1], mid_step_ratio)]
            out.append(np.stack(cols, axis=1).reshape(-1))
        return _HostBuf(np.concatenate(out))

    def alloc(self, n_doubles):
        return _HostBuf(np.zeros(max(int(n_doubles), 1)))

    def expand_deltas(self, d_base, frames, d_out):
        """what paa_dev_expand_deltas does on the GPU: [34][T] base slabs -> [68][T] slabs, delta column 0 = zeros"""
        self.expand_calls = getattr(self, "expand_calls", 0) + 1
        pos_b = pos_o = 0
        for t in np.asarray(frames, dtype=np.int64):
            t = int(t)
            b = d_base.arr[pos_b:pos_b + 34 * t].reshape(34, t)
            o = d_out.arr[pos_o:pos_o + 68 * t].reshape(68, t)
            o[:34] = b
            o[34:, 0] = 0.0
            o[34:, 1:] = b[:, 1:] - b[:, :-1]
            pos_b += 34 * t
            pos_o += 68 * t

    def view(self, buf, offset_doubles, n_doubles):
        return _HostBuf(buf.arr[int(offset_doubles):int(offset_doubles) + int(n_doubles)])

    def to_host(self, buf, n_doubles):
        return buf.arr[:int(n_doubles)]

    def sync(self):
        pass


class _GlooGather:
    """Stands in for distributed.RcclGather: the same variable-size gather to the root (what paa_comm_gather_f64 does with
    grouped ncclSend / ncclRecv), over torch.distributed point-to-point on gloo."""

    def __init__(self, dist, torch, world_size, rank):
        self.dist, self.torch, self.world_size, self.rank = dist, torch, world_size, rank

    def gather(self, send, counts, root, recv, displs=None):
        counts = [int(c) for c in counts]
        if displs is None:
            displs = np.concatenate(([0], np.cumsum(counts)[:-1]))
        if self.rank == root:
            reqs, parts = [], {}
            for r in range(self.world_size):
                off = int(displs[r])
                if r != root and counts[r] > 0:
                    parts[r] = (off, self.torch.empty(counts[r], dtype=self.torch.float64))
                    reqs.append(self.dist.irecv(parts[r][1], src=r))
                elif r == root:
                    recv.arr[off:off + counts[r]] = send.arr[:counts[r]]
            for q in reqs:
                q.wait()
            for r, (o, t) in parts.items():
                recv.arr[o:o + counts[r]] = t.numpy()
        elif counts[self.rank] > 0:
            self.dist.send(self.torch.from_numpy(send.arr[:counts[self.rank]].copy()), dst=root)

    def barrier(self):
        self.dist.barrier()

    def close(self):
        pass


def _worker(rank, world, port, q, rdir):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    import paa_oracle as O
    from synth import synth_clip
    from pyaudioanalysis_amd import distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lens = [4000, 1600, 9000, 800, 5200, 2500, 7000]
        clips = [synth_clip(900 + i, n) for i, n in enumerate(lens)]
        W, S, F = 800, 400, 34          # rows on the wire: with deltas the ranks ship the 34 base rows, the root re-forms the rest
        frames = D.frames_per_clip(lens, W, S)
        ranges = D.partition_by_frames(frames, world)
        counts = D.block_counts(frames, ranges, F)
        comm = _GlooGather(dist, torch, world, rank)
        # the product's own sharding function, with the communicator and the engine injected
        per_clip = D.extract_sharded(clips, 16000, W, S, True, world, rank, comm, root=0, engine=_OracleEngine(O))
        comm.barrier()
        # restart files: the first run writes one block per rank, the rerun loads them (no extraction), a changed clip
        # invalidates only the rank that owns it
        eng1, eng2, eng3 = _OracleEngine(O), _OracleEngine(O), _OracleEngine(O)
        first = D.extract_sharded(clips, 16000, W, S, True, world, rank, comm, root=0, engine=eng1, restart_dir=rdir)
        again = D.extract_sharded(clips, 16000, W, S, True, world, rank, comm, root=0, engine=eng2, restart_dir=rdir)
        changed = list(clips)
        changed[0] = synth_clip(12345, lens[0])
        third = D.extract_sharded(changed, 16000, W, S, True, world, rank, comm, root=0, engine=eng3, restart_dir=rdir)
        restart_ok = (eng1.extract_calls, eng2.extract_calls) == (1, 0) and eng3.extract_calls == (1 if rank == 0 else 0)
        # chunked pipeline: 3 pieces per rank, each gathered to its final place; with restart files a rerun sends
        # slices of the loaded block piece by piece
        eng4, eng5 = _OracleEngine(O), _OracleEngine(O)
        rdir3 = os.path.join(rdir, "k3")
        chunked = D.extract_sharded(clips, 16000, W, S, True, world, rank, comm, root=0, engine=eng4, chunks=3,
                                    restart_dir=rdir3)
        chunked_again = D.extract_sharded(clips, 16000, W, S, True, world, rank, comm, root=0, engine=eng5, chunks=3,
                                          restart_dir=rdir3)
        restart_ok = restart_ok and eng4.extract_calls >= 2 and eng5.extract_calls == 0
        # the cheap gather: (136, M) mid-term matrices of 1.0 s / 0.5 s over 50 ms / 25 ms, 2 pieces per rank
        mids = D.extract_sharded(clips, 16000, W, S, True, world, rank, comm, root=0, engine=_OracleEngine(O), chunks=2,
                                 gather="mid", mid_window=16000, mid_step=8000)
        flags = [None] * world
        dist.all_gather_object(flags, bool(restart_ok))
        comm.barrier()
        if rank == 0:
            ok = per_clip is not None and len(per_clip) == len(clips) and all(flags)
            ok &= all(np.array_equal(x, y) and np.array_equal(x, z) for x, y, z in zip(per_clip, first, again))
            ok &= bool(np.array_equal(third[0], O.feature_extraction(changed[0], 16000, W, S)[0]))
            ok &= all(np.array_equal(x, y) for x, y in zip(per_clip[1:], third[1:]))
            ok &= all(np.array_equal(x, y) and np.array_equal(x, z) for x, y, z in zip(per_clip, chunked, chunked_again))
            for c, got in zip(clips, mids):
                ref_mid = O.mid_feature_extraction(c, 16000, 16000, 8000, W, S)[0]
                ok &= bool(got.shape == ref_mid.shape and np.allclose(got, ref_mid, rtol=1e-12, atol=1e-12))
            for c, got in zip(clips, per_clip or []):
                ref, _ = O.feature_extraction(c, 16000, W, S)
                ok &= bool(np.array_equal(got, ref))
            q.put(("ok" if ok else "mismatch", ranges, counts.tolist()))
        else:
            assert per_clip is None
    finally:
        dist.destroy_process_group()


def test_world_size_2_gather_roundtrip(tmp_path):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    status, ranges, counts = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert status == "ok"
    assert ranges[0][0] == 0 and ranges[0][1] == ranges[1][0] and ranges[1][1] == 7
    assert sum(counts) == 34 * int(sum((n - 800) // 400 + 1 for n in [4000, 1600, 9000, 800, 5200, 2500, 7000]))


def test_partition_properties():
    sys.path.insert(0, ROOT)
    from pyaudioanalysis_amd import distributed as D
    rng = np.random.default_rng(0)
    for world in (1, 2, 3, 4, 8):
        for n in (1, 5, 8, 100, 1000):
            frames = rng.integers(1, 2000, n)
            ranges = D.partition_by_frames(frames, world)
            assert len(ranges) == world
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            for (a, b), (c, d) in zip(ranges[:-1], ranges[1:]):
                assert b == c and a <= b
            loads = [int(frames[a:b].sum()) for a, b in ranges]
            if n >= 50 * world:
                assert max(loads) <= 1.1 * sum(loads) / world + frames.max()
    # equal clips split evenly: BASELINE config 4 on 8 GPUs
    ranges = D.partition_by_frames(np.full(100000, 399), 8)
    assert [b - a for a, b in ranges] == [12500] * 8
    assert list(D.frames_per_clip([799, 800, 1199, 1200, 160000], 800, 400)) == [0, 1, 1, 2, 399]


def _socket_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    from pyaudioanalysis_amd._rendezvous import SocketGroup
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    g = SocketGroup(timeout=60)
    try:
        got = g.all_gather({"rank": rank, "blob": bytes([rank]) * 1000})
        ok = [d["rank"] for d in got] == list(range(world)) and all(d["blob"] == bytes([r]) * 1000 for r, d in enumerate(got))
        ok &= g.broadcast(b"id-%d" % rank, 0) == b"id-0"
        ok &= g.all_max(10.0 + rank) == 10.0 + world - 1
        for _ in range(20):
            g.barrier()
        q.put((rank, bool(ok)))
    finally:
        g.close()


def test_socket_control_plane_world_3():
    """_rendezvous.SocketGroup on the launcher's environment variables: all-gather / broadcast / max / barrier among three
    processes, with MASTER_PORT itself occupied by a foreign listener (as torchrun's agent store does) and the first
    candidate port taken by another foreign listener that never answers the handshake."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    foreign = socket.socket()
    foreign.bind(("127.0.0.1", 0))
    port = foreign.getsockname()[1]
    foreign.listen(4)
    squatter = socket.socket()
    try:
        squatter.bind(("", port + 1))
        squatter.listen(4)
    except OSError:
        squatter = None
    procs = [ctx.Process(target=_socket_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(3))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    foreign.close()
    if squatter is not None:
        squatter.close()
    assert got == [(0, True), (1, True), (2, True)]


def test_rank0_listener_interfaces(monkeypatch):
    """Where rank 0 of the control plane listens (advisor, round 4): a loopback LITERAL means a single-node job and binds
    loopback only; the host's own NAME resolving to loopback (Debian's 127.0.1.1 line in /etc/hosts) must NOT pin the listener to
    loopback -- remote ranks resolve that name to the real interface -- so all interfaces are bound; a routable address is
    tried itself first, all interfaces after it; an unresolvable name binds all interfaces."""
    from pyaudioanalysis_amd import _rendezvous as R
    hosts = R.SocketGroup._bind_hosts
    assert hosts("127.0.0.1") == ["127.0.0.1"]
    assert hosts("127.0.1.1") == ["127.0.1.1"]
    assert hosts("localhost") == ["127.0.0.1"]
    monkeypatch.setattr(R.socket, "gethostbyname", lambda name: {"node7": "127.0.1.1", "node8": "10.1.2.3"}.get(name, name))
    assert hosts("node7") == [""]
    assert hosts("node7", single_node=True) == ["127.0.1.1"]      # LOCAL_WORLD_SIZE == WORLD_SIZE: nothing off the node needs it
    assert hosts("node8", single_node=True) == ["10.1.2.3", ""]
    assert hosts("node8") == ["10.1.2.3", ""]
    assert hosts("10.1.2.3") == ["10.1.2.3", ""]

    def boom(name):
        raise OSError("no such host")
    monkeypatch.setattr(R.socket, "gethostbyname", boom)
    assert hosts("elsewhere.example") == [""]


def test_restart_shard_is_on_disk_while_the_exchange_is_still_blocked(tmp_path):
    """A rank whose peer died in the exchange never gets out of the communication stream.  Its finished block must be on
    disk by then: extract_sharded saves the pieces through the engine's compute-stream-only copy (to_host_source ->
    paa_memcpy_d2h_compute) BEFORE it waits for the gathers; the plain to_host would wait for the exchange (advisor, round 3)."""
    import threading
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import paa_oracle as O
    from synth import synth_clip
    from pyaudioanalysis_amd import distributed as D

    exchange_done = threading.Event()           # never set while the assertions run: the "peer" is dead

    class BlockingEngine(_OracleEngine):
        def __init__(self):
            super().__init__(O)
            self.plain_to_host_calls = 0

        def to_host(self, buf, n_doubles):      # = paa_memcpy_d2h: waits for the communication stream
            self.plain_to_host_calls += 1
            exchange_done.wait()
            return super().to_host(buf, n_doubles)

        def to_host_source(self, buf, n_doubles):      # = paa_memcpy_d2h_compute: kernels only
            return _OracleEngine.to_host(self, buf, n_doubles)

        def sync(self):                         # = paa_dev_sync: both streams
            exchange_done.wait()

    class QueuedGather:                         # the gather is only QUEUED (RCCL on its own stream)
        def gather(self, send, counts, root, recv, displs=None):
            pass

        def barrier(self):
            pass

        def close(self):
            pass

    clips = [synth_clip(700 + i, n) for i, n in enumerate([4000, 2400, 5200])]
    engine = BlockingEngine()
    res = {}

    def run():
        res["out"] = D.extract_sharded(clips, 16000, 800, 400, True, 2, 1, QueuedGather(), root=0, engine=engine,
                                       restart_dir=str(tmp_path), chunks=2)
    th = threading.Thread(target=run, daemon=True)
    th.start()
    shard = tmp_path / "shard_001_of_002.npz"
    for _ in range(600):
        if shard.exists() or not th.is_alive():
            break
        th.join(0.05)
    assert th.is_alive(), "extract_sharded must still be waiting for the exchange"
    assert shard.exists(), "the rank's block must be saved before it waits for the exchange"
    assert engine.plain_to_host_calls == 0
    with np.load(shard, allow_pickle=False) as z:
        frames = D.frames_per_clip([len(c) for c in clips], 800, 400)
        a, b = D.partition_by_frames(frames, 2)[1]
        # (with deltas the rank ships -- and saves -- its 34 base rows; the root re-forms rows 34..67)
        want = np.concatenate([O.feature_extraction(c, 16000, 800, 400, False)[0].reshape(-1) for c in clips[a:b]])
        assert np.array_equal(z["block"], want)
    exchange_done.set()
    th.join(30)
    assert not th.is_alive() and res["out"] is None
