"""Many clips across the GPUs of one node: one process per GPU, contiguous clip ranges balanced by frame
count, per-rank plans, and ONE gather of the result blocks to a root rank over RCCL (xGMI).

Clips are independent units (normalisation is per clip, ShortTermFeatures.py:570), so there is no data-path
collective besides that gather.  The reference has no distributed code; this is the batched form of the
per-file loop in MidTermFeatures.directory_feature_extraction (MidTermFeatures.py:167-201).

A rank's range is cut into chunks: while chunk k travels to the root on the communication stream, chunk k+1 is uploaded
and computed (extract_sharded(chunks=K)); gather="mid" ships the (136, M) mid-term matrices -- what the directory
walkers consume, 1/40 of the bytes of the short-term matrices -- instead of the (F, T) slabs.

The control plane (unique-id broadcast, barriers) is supplied by the caller -- _rendezvous.SocketGroup (bench.py, the
standard library only) or any process group with a broadcast (the CPU tests use torch.distributed / gloo); the feature
data itself moves GPU -> GPU through libpaa_hip.so's paa_comm_gatherv_f64 (grouped ncclSend/ncclRecv).
"""
import ctypes
import hashlib
import os
import socket

import numpy as np

from . import _ffi


def frames_per_clip(lengths, window, step):
    """T_c = floor((n_c - window)/step) + 1 (ShortTermFeatures.py:608), 0 when the clip is too short."""
    lengths = np.asarray(lengths, dtype=np.int64)
    t = (lengths - int(window)) // int(step) + 1
    return np.where(lengths >= int(window), t, 0).astype(np.int64)


def partition_by_frames(frames, world_size):
    """Contiguous clip ranges [(start, end)] per rank, balanced by the running sum of frames.

    Rank r takes the clips whose cumulative-frame midpoint falls into the r-th equal slice of the total, so
    equal-length clips split evenly (12 500 each for 100 000 clips on 8 GPUs) and ragged batches stay balanced
    to within one clip.
    """
    frames = np.asarray(frames, dtype=np.int64)
    n = len(frames)
    total = int(frames.sum())
    if world_size < 1:
        raise ValueError("world_size must be >= 1")
    if n == 0 or total == 0:
        base = [(min(r * n // world_size, n), min((r + 1) * n // world_size, n)) for r in range(world_size)]
        return base
    csum = np.cumsum(frames)
    mid = csum - frames / 2.0
    owner = np.minimum((mid * world_size / total).astype(np.int64), world_size - 1)
    ranges = []
    for r in range(world_size):
        idx = np.nonzero(owner == r)[0]
        if len(idx) == 0:
            start = ranges[-1][1] if ranges else 0
            ranges.append((start, start))
        else:
            ranges.append((int(idx[0]), int(idx[-1]) + 1))
    return ranges


def block_counts(frames, ranges, n_rows):
    """float64 elements each rank contributes: n_rows * sum of its clips' frames."""
    frames = np.asarray(frames, dtype=np.int64)
    return np.array([n_rows * int(frames[a:b].sum()) for a, b in ranges], dtype=np.int64)


def split_gathered(flat, frames, n_rows):
    """Cut the rank-ordered concatenation of [n_rows][T_c] slabs back into one array per clip."""
    out, pos = [], 0
    for t in np.asarray(frames, dtype=np.int64):
        cnt = n_rows * int(t)
        out.append(np.asarray(flat[pos:pos + cnt]).reshape(n_rows, int(t)))
        pos += cnt
    if pos != len(flat):
        raise ValueError("gathered buffer has %d elements, expected %d" % (len(flat), pos))
    return out


class RcclGather:
    """The RCCL communicator of libpaa_hip.so for this process (one per GPU).

    The interface extract_sharded() needs from a communicator is gather / barrier / close; the CPU tests inject a
    gloo-backed object with the same three methods."""

    def __init__(self, world_size, rank, broadcast_bytes, exchange_bytes=None):
        """broadcast_bytes(payload_or_None) -> payload: broadcast from rank 0 over any control-plane group.
        exchange_bytes(payload) -> [payload of rank 0, 1, ..]: optional all-gather on the same control plane.  With it
        every rank learns (host, PCI bus id) of all ranks BEFORE RCCL is touched and all of them refuse a job that puts
        two ranks on one physical device (ncclCommInitRank would hang); without it the library's node-local marker
        files catch the case (paa_comm_init)."""
        lib = _ffi.lib()
        if exchange_bytes is not None and int(world_size) > 1:
            bus = ctypes.create_string_buffer(64)
            _ffi.check(lib.paa_device_bus_id(bus, 64))
            me = (socket.gethostname() + "|" + bus.value.decode()).encode()
            seen = {}
            for r, who in enumerate(exchange_bytes(me)):
                if who in seen:
                    raise _ffi.HipLibraryError("ranks %d and %d both use device %s: one process per GPU"
                                               % (seen[who], r, bytes(who).decode(errors="replace")))
                seen[who] = r
        buf = ctypes.create_string_buffer(_ffi.COMM_ID_BYTES)
        if rank == 0:
            _ffi.check(lib.paa_comm_unique_id(buf))
        # (a control plane that already holds an id -- e.g. one created by a launcher process -- may return that one)
        payload = broadcast_bytes(bytes(buf.raw) if rank == 0 else None)
        buf = ctypes.create_string_buffer(payload, _ffi.COMM_ID_BYTES)
        _ffi.check(lib.paa_comm_init(int(world_size), int(rank), buf))
        self.world_size, self.rank = int(world_size), int(rank)

    @classmethod
    def over(cls, group):
        """Communicator on a _rendezvous.SocketGroup (or anything with rank / world_size / broadcast / all_gather)."""
        return cls(group.world_size, group.rank, lambda payload: group.broadcast(payload, 0), group.all_gather)

    def gather(self, d_send, counts, root, d_recv, displs=None):
        """counts[r] doubles of rank r land at d_recv + displs[r] on the root (back to back in rank order when displs is
        None).  Asynchronous: queued on the library's communication stream behind the kernels launched so far."""
        counts = np.ascontiguousarray(counts, dtype=np.int64)
        if displs is None:
            displs = np.concatenate(([0], np.cumsum(counts)[:-1]))
        displs = np.ascontiguousarray(displs, dtype=np.int64)
        recv = d_recv.ptr if d_recv is not None else None
        _ffi.check(_ffi.lib().paa_comm_gatherv_f64(d_send.ptr, _ffi.as_i64p(counts), _ffi.as_i64p(displs), int(root), recv))

    def barrier(self):
        _ffi.check(_ffi.lib().paa_comm_barrier())

    def close(self):
        _ffi.lib().paa_comm_destroy()


class _DeviceView:
    """A range of a DeviceBuffer (keeps the parent alive)."""

    def __init__(self, parent, offset_doubles, n_doubles):
        self.parent = parent
        self.ptr = ctypes.c_void_p(parent.ptr.value + 8 * offset_doubles)
        self.nbytes = 8 * n_doubles

    def to_host(self, dtype, count, offset_bytes=0, wait_comm=True):
        return self.parent.to_host(dtype, count, offset_bytes + self.ptr.value - self.parent.ptr.value, wait_comm)


class HipEngine:
    """The compute / memory side of extract_sharded() on the GPU (libpaa_hip.so).  The CPU tests inject an object with
    the same four methods (extract, alloc, to_host, sync) so that the sharding logic itself runs without a device."""

    def extract(self, clips, sampling_rate, window, step, deltas):
        """int16 clips -> device buffer holding their [F][T_c] slabs back to back (kept in HBM)."""
        offsets = np.zeros(len(clips) + 1, dtype=np.int64)
        np.cumsum([len(c) for c in clips], out=offsets[1:])
        d_in = _ffi.DeviceBuffer.from_host(np.concatenate(clips) if len(clips) > 1 else clips[0])
        plan = _ffi.Plan(offsets, sampling_rate, window, step, deltas=deltas, sample_kind=0)
        d_out = _ffi.DeviceBuffer(max(plan.out_doubles, 1) * 8)
        plan.execute(d_in, d_out)
        d_out._keep = (d_in, plan)          # the launch is asynchronous: inputs live as long as the result
        return d_out

    def extract_mid(self, clips, sampling_rate, window, step, mid_ratio, mid_step_ratio):
        """int16 clips -> device buffer holding their (136, M_c) mid-term matrices back to back (MidTermFeatures.py:87-127:
        68 short-term rows, mean and std over mid_ratio frames every mid_step_ratio frames); the short-term matrices
        stay in HBM and are dropped."""
        offsets = np.zeros(len(clips) + 1, dtype=np.int64)
        np.cumsum([len(c) for c in clips], out=offsets[1:])
        d_in = _ffi.DeviceBuffer.from_host(np.concatenate(clips) if len(clips) > 1 else clips[0])
        plan = _ffi.Plan(offsets, sampling_rate, window, step, deltas=True, sample_kind=0)
        d_st = _ffi.DeviceBuffer(max(plan.out_doubles, 1) * 8)
        plan.execute(d_in, d_st)
        d_mid = _ffi.DeviceBuffer(max(plan.mid_doubles(mid_step_ratio), 1) * 8)
        plan.mid_execute(d_st, mid_ratio, mid_step_ratio, d_mid)
        d_mid._keep = (d_in, d_st, plan)
        return d_mid

    def alloc(self, n_doubles):
        return _ffi.DeviceBuffer(max(int(n_doubles), 1) * 8)

    def expand_deltas(self, d_base, frames, d_out):
        """[34][T_c] base slabs back to back -> [68][T_c] slabs back to back (paa_dev_expand_deltas: the delta rows of
        ShortTermFeatures.py:668-680 re-formed on the device, bit-identical to a 68-row plan's; queued behind the gathers)."""
        frames = np.ascontiguousarray(frames, dtype=np.int64)
        _ffi.check(_ffi.lib().paa_dev_expand_deltas(d_base.ptr, _ffi.as_i64p(frames), len(frames), d_out.ptr))

    def view(self, buf, offset_doubles, n_doubles):
        """The sub-range [offset, offset + n) of a device buffer as something gather() / to_host() accept."""
        return _DeviceView(buf, int(offset_doubles), int(n_doubles))

    def to_host(self, buf, n_doubles):
        return buf.to_host(np.float64, int(n_doubles))

    def to_host_source(self, buf, n_doubles):
        """A buffer that is at most the SOURCE of queued gathers: the copy waits for the kernels, not for the exchange."""
        return buf.to_host(np.float64, int(n_doubles), wait_comm=False)

    def upload(self, flat):
        """float64 host block (a restart file) -> device buffer"""
        return _ffi.DeviceBuffer.from_host(np.ascontiguousarray(flat, dtype=np.float64))

    def sync(self):
        _ffi.sync()

    _build_id = None

    def build_id(self):
        """Digest of the library binary: restart files written by another build of the kernels are not reused."""
        if HipEngine._build_id is None:
            h = hashlib.sha256()
            with open(_ffi.library_path(), "rb") as f:
                for chunk in iter(lambda: f.read(1 << 20), b""):
                    h.update(chunk)
            HipEngine._build_id = h.hexdigest()[:16]
        return HipEngine._build_id


def _shard_key(clips, sampling_rate, window, step, deltas, build_id="", what="short"):
    """Identity of one rank's work for the restart files: the build of the compute engine, the parameters, what is
    gathered, the clip lengths and a digest of the samples."""
    h = hashlib.sha256()
    h.update(str(build_id).encode())
    h.update(repr((float(sampling_rate), int(window), int(step), bool(deltas), what, [len(c) for c in clips])).encode())
    for c in clips:
        h.update(np.ascontiguousarray(c, dtype=np.int16).tobytes())
    return h.hexdigest()


def mid_windows_per_clip(frames, mid_step_ratio):
    """M_c = ceil(T_c / step ratio): one mid-term window per start 0, r, 2r, .. < T_c (MidTermFeatures.py:116-124)."""
    return -(-np.asarray(frames, dtype=np.int64) // int(mid_step_ratio))


def extract_sharded(clips, sampling_rate, window, step, deltas, world_size, rank, comm, root=0, engine=None,
                    restart_dir=None, chunks=1, gather="short", mid_window=None, mid_step=None, ship_base_rows=None):
    """Rank-local part of a sharded batch extraction.

    clips: the FULL list of int16 clips (every rank sees the list; only its own range is uploaded).
    comm: gather(send, counts, root, recv, displs) / barrier() / close() -- RcclGather on the GPU.
    engine: extract / extract_mid / alloc / view / to_host / upload / sync -- HipEngine (default) on the GPU.
    chunks: every rank cuts its range into this many contiguous pieces (balanced by frames); piece k is gathered on the
        communication stream while piece k+1 is uploaded and computed, and lands at its final place in the root's buffer.
    gather: "short" -- the (F, T_c) short-term matrices (F = 68 with deltas, else 34); with deltas only the 34 base rows
                       travel and the root re-forms the delta rows on its device (bit-identical, half the link bytes);
            "mid"   -- the (136, M_c) mid-term matrices of mid_feature_extraction(mid_window, mid_step in samples,
                       MidTermFeatures.py:87-127) and nothing else: 1/40 of the bytes at 1.0 s / 1.0 s over 50 / 25 ms.
    ship_base_rows: gather="short" with deltas -- True: the 34 base rows travel and the root re-forms rows 34..67; False: all 68
        travel; None (default): True.  A function of the ARGUMENTS only, so every rank derives the same gather sizes (it used
        to depend on the rank's engine object: ranks with different engines would have posted mismatched sends / receives).
    restart_dir: when given, every rank leaves its finished block there (shard_<rank>_of_<world>.npz, keyed by the
        parameters and a digest of its clips) and a rerun of the same job loads the block instead of extracting it
        again -- a 100 000-clip job that died in the gather or on another rank restarts without redoing finished shards.
    Returns, on the root, the list of per-clip arrays for all clips (None elsewhere).
    """
    engine = engine or HipEngine()
    base_only = False
    window, step = int(window), int(step)
    lengths = [len(c) for c in clips]
    frames = frames_per_clip(lengths, window, step)
    if np.any(frames < 1):
        raise ValueError("need at least one array to concatenate")
    if gather == "mid":
        if mid_window is None or mid_step is None:
            raise ValueError('gather="mid" needs mid_window and mid_step (samples)')
        mid_ratio = int(round((mid_window - (window - step)) / step))          # MidTermFeatures.py:100-102
        mid_step_ratio = int(round(mid_step / step))
        if mid_step_ratio < 1:
            raise ValueError("mid_step / short_step rounds to 0: the reference never terminates")
        units, n_rows, what = mid_windows_per_clip(frames, mid_step_ratio), 136, ("mid", mid_ratio, mid_step_ratio)
    elif gather == "short":
        # with deltas the ranks compute and ship the 34 BASE rows only: rows 34..67 are exact differences of consecutive columns
        # of rows 0..33 (ShortTermFeatures.py:668-680), so the root re-forms them on its device (engine.expand_deltas) -- half the
        # bytes on the xGMI links, and the peers run the cheaper 34-row kernel (ship_base_rows=False ships all 68).
        base_only = bool(deltas) and (True if ship_base_rows is None else bool(ship_base_rows))
        units, n_rows = frames, (34 if (base_only or not deltas) else 68)
        what = ("short", "base34") if base_only else "short"
    else:
        raise ValueError('gather must be "short" or "mid"')
    chunks = max(1, int(chunks))
    ranges = partition_by_frames(frames, world_size)
    counts = block_counts(units, ranges, n_rows)
    rank_base = np.concatenate(([0], np.cumsum(counts)[:-1]))
    # piece k of every rank: clip range, size and place in the root's buffer (every rank can compute all of it)
    pieces = []                                   # pieces[k][r] = (clip_start, clip_end, count, displ)
    for r, (ra, rb) in enumerate(ranges):
        sub = partition_by_frames(frames[ra:rb], chunks) if rb > ra else [(0, 0)] * chunks
        off = int(rank_base[r])
        row = []
        for (sa, sb) in sub:
            cnt = n_rows * int(units[ra + sa:ra + sb].sum())
            row.append((ra + sa, ra + sb, cnt, off))
            off += cnt
        pieces.append(row)
    a, b = ranges[rank]
    mine = [np.ascontiguousarray(c, dtype=np.int16) for c in clips[a:b]]
    d_block = None                                # a restart file's block, uploaded whole
    shard_file = key = None
    if restart_dir is not None and mine:
        os.makedirs(restart_dir, exist_ok=True)
        shard_file = os.path.join(restart_dir, "shard_%03d_of_%03d.npz" % (rank, world_size))
        build_id = engine.build_id() if hasattr(engine, "build_id") else ""
        key = _shard_key(mine, sampling_rate, window, step, deltas, build_id, what)
        if os.path.exists(shard_file):
            try:
                with np.load(shard_file, allow_pickle=False) as z:
                    if str(z["key"]) == key:
                        block = z["block"]                       # (read once: every access decompresses the member)
                        if block.shape == (int(counts[rank]),):
                            d_block = engine.upload(block)
            except Exception:                    # an unreadable or foreign file is simply recomputed
                d_block = None
    d_all = engine.alloc(int(counts.sum())) if rank == root else None
    sent = []                                     # this rank's pieces, kept until the gathers have run
    for k in range(chunks):
        ca, cb, cnt, displ = pieces[rank][k]
        if cnt == 0:
            d_piece = engine.alloc(1)
        elif d_block is not None:
            d_piece = engine.view(d_block, displ - int(rank_base[rank]), cnt)
        elif gather == "mid":
            d_piece = engine.extract_mid(mine[ca - a:cb - a], sampling_rate, window, step, mid_ratio, mid_step_ratio)
        else:
            d_piece = engine.extract(mine[ca - a:cb - a], sampling_rate, window, step, bool(deltas) and not base_only)
        sent.append((d_piece, cnt))
        comm.gather(d_piece, np.array([pieces[r][k][2] for r in range(world_size)], dtype=np.int64), root, d_all,
                    np.array([pieces[r][k][3] for r in range(world_size)], dtype=np.int64))
    if shard_file is not None and d_block is None:
        # written BEFORE waiting for the gathers: the pieces are only SOURCES of the queued exchange, so they are copied
        # with the compute-stream-only variant (paa_memcpy_d2h_compute) -- a job that dies in the exchange, or on another
        # rank, still leaves this rank's finished block behind (the plain to_host waits for the communication stream)
        to_host_source = getattr(engine, "to_host_source", engine.to_host)
        parts = [to_host_source(d, cnt) for d, cnt in sent if cnt]
        tmp = shard_file + ".tmp.npz"
        np.savez(tmp, key=np.array(key), block=np.concatenate(parts) if len(parts) > 1 else parts[0])
        os.replace(tmp, shard_file)
    d_base = None
    if rank == root and base_only:
        if not hasattr(engine, "expand_deltas"):
            raise TypeError("ship_base_rows needs an engine with expand_deltas() on the root (HipEngine has it); pass "
                            "ship_base_rows=False on EVERY rank for an engine without")
        d_full = engine.alloc(2 * int(counts.sum()))
        engine.expand_deltas(d_all, frames, d_full)          # on the device, behind the gathers
        # the gathered base rows stay referenced until the sync below: the expansion kernel is only QUEUED behind the gathers
        # and still reads them (dropping the last reference here would free a buffer with a reader in flight)
        d_base, d_all = d_all, d_full
        n_rows, total = 68, 2 * int(counts.sum())
    else:
        total = int(counts.sum())
    engine.sync()
    del d_base
    if rank != root:
        return None
    flat = engine.to_host(d_all, total)
    return split_gathered(flat, units, n_rows)
