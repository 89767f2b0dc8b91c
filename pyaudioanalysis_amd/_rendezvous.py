"""Control plane of a multi-rank job: a star of TCP sockets around rank 0, on the launcher's environment variables
(RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT as torchrun, srun wrappers and mpirun shims export them).

It carries only what the data path cannot: the RCCL unique id, per-rank sizes, a barrier and a max over ranks.  Feature
data never goes through it (that is paa_comm_gather_f64 over xGMI).  The reference has no distributed code; this is
plumbing for distributed.extract_sharded and bench.py, with no dependency beyond the standard library.

The port: MASTER_PORT itself usually belongs to the launcher (torchrun's agent keeps its store there), so rank 0 binds
the first free port of MASTER_PORT + 1 .. + 16 and a peer tries those in turn; both sides check a token derived from
the job's environment, so a foreign listener on one of them is skipped, not joined.

Trust model: the token keeps OTHER JOBS and stray listeners apart; it is not a credential -- address, port, world size and
run id are guessable.  On a network where hosts outside the job can reach rank 0's ports, export the same PAA_RDZV_SECRET
on every rank: it is mixed into the token (HMAC-SHA256), and a peer without it cannot claim a rank slot.  Rank 0 listens on
loopback only when MASTER_ADDR is a loopback literal (single-node jobs), on MASTER_ADDR's own interface when that is a
non-loopback address of this host, on the resolved loopback address when a NAME resolves to loopback and the launcher says
the job is single-node (LOCAL_WORLD_SIZE == WORLD_SIZE), and on all interfaces otherwise (_bind_hosts; a warning is issued
when that happens without PAA_RDZV_SECRET); every handshake runs on its own thread
with a 5 s budget, so a half-open connection cannot stall the accept loop.
"""
import base64
import errno
import hashlib
import hmac
import json
import os
import socket
import struct
import threading
import time

_PORT_SPAN = 16


# Wire format: length-prefixed JSON (never pickle: a peer is another process on the network).  Plain numbers, strings,
# booleans, None, lists / tuples (tuples come back as lists) and string-keyed dicts pass as they are, bytes as {"__b64__": ..}.
def _encode(obj):
    if isinstance(obj, (bytes, bytearray, memoryview)):
        return {"__b64__": base64.b64encode(bytes(obj)).decode("ascii")}
    if isinstance(obj, (list, tuple)):
        return [_encode(v) for v in obj]
    if isinstance(obj, dict):
        return {str(k): _encode(v) for k, v in obj.items()}
    if obj is None or isinstance(obj, (bool, int, float, str)):
        return obj
    if hasattr(obj, "item"):                      # NumPy scalars
        return _encode(obj.item())
    raise TypeError("control plane carries numbers, strings, bytes, lists and dicts only, not %r" % type(obj))


def _decode(obj):
    if isinstance(obj, dict):
        if set(obj) == {"__b64__"}:
            return base64.b64decode(obj["__b64__"])
        return {k: _decode(v) for k, v in obj.items()}
    if isinstance(obj, list):
        return [_decode(v) for v in obj]
    return obj


def _send(sock, obj):
    blob = json.dumps(_encode(obj), separators=(",", ":")).encode("utf-8")
    sock.sendall(struct.pack("<Q", len(blob)) + blob)


def _recv_exact(sock, n):
    parts = []
    while n:
        chunk = sock.recv(min(n, 1 << 20))
        if not chunk:
            raise ConnectionError("control-plane peer closed the connection")
        parts.append(chunk)
        n -= len(chunk)
    return b"".join(parts)


def _recv(sock):
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    if n > (64 << 20):
        raise ConnectionError("control-plane message of %d bytes refused" % n)
    return _decode(json.loads(_recv_exact(sock, n).decode("utf-8")))


class SocketGroup:
    """world_size processes; every collective is 'send to rank 0, rank 0 answers' (payloads are a few bytes)."""

    def __init__(self, rank=None, world_size=None, addr=None, port=None, timeout=120.0, job_tag=None):
        env = os.environ
        self.rank = int(env.get("RANK", "0")) if rank is None else int(rank)
        self.world_size = int(env.get("WORLD_SIZE", "1")) if world_size is None else int(world_size)
        addr = addr or env.get("MASTER_ADDR", "127.0.0.1")
        port = int(env.get("MASTER_PORT", "29500")) if port is None else int(port)
        tag = job_tag if job_tag is not None else env.get("TORCHELASTIC_RUN_ID", "")
        ident = ("paa-rdzv|%s|%d|%d|%s" % (addr, port, self.world_size, tag)).encode()
        secret = env.get("PAA_RDZV_SECRET", "")
        self._token = (hmac.new(secret.encode(), ident, hashlib.sha256).digest() if secret
                       else hashlib.sha256(ident).digest())
        self._lock = threading.Lock()
        self._peers = {}          # rank 0: rank -> socket
        self._root = None         # other ranks: socket to rank 0
        self._listener = None
        if self.world_size == 1:
            return
        deadline = time.monotonic() + timeout
        if self.rank == 0:
            self._serve(addr, port, deadline)
        else:
            self._join(addr, port, deadline)

    # ---- connection set-up
    @staticmethod
    def _bind_hosts(addr, single_node=False):
        """Interfaces rank 0 tries to listen on, in order.  MASTER_ADDR given as a loopback LITERAL (127.x.y.z, localhost)
        declares a single-node job: every peer connects to that same literal, so loopback is enough and nothing off the node
        can connect.  A NAME that merely resolves to loopback here (Debian / Ubuntu map the host's own name to 127.0.1.1 in
        /etc/hosts) says nothing about where the peers are -- remote ranks resolve it to the real interface -- so all
        interfaces are bound, as for a name that resolves elsewhere (NAT) -- UNLESS the launcher says the whole job is on this
        node (single_node: LOCAL_WORLD_SIZE == WORLD_SIZE, what torchrun exports for --nnodes=1): then the loopback address the
        name resolved to is enough and nothing off the node can reach the listener (advisor, round 5).  A non-loopback address
        of this host is bound itself, with all interfaces as the fallback when it turns out not to be bindable here."""
        literal = addr.strip().lower()
        try:
            packed = socket.inet_aton(literal)
            is_literal_ip = literal.count(".") == 3
        except OSError:
            packed, is_literal_ip = None, False
        if literal == "localhost" or (is_literal_ip and packed[0] == 127):
            return ["127.0.0.1" if literal == "localhost" else literal]
        try:
            resolved = socket.gethostbyname(addr)
        except OSError:
            return [""]
        if resolved.startswith("127."):
            return [resolved] if single_node else [""]
        return [resolved, ""]

    def _serve(self, addr, port, deadline):
        last = None
        local_world = os.environ.get("LOCAL_WORLD_SIZE", "")
        hosts = self._bind_hosts(addr, single_node=local_world.isdigit() and int(local_world) == self.world_size)
        if "" in hosts and not os.environ.get("PAA_RDZV_SECRET"):
            import warnings
            warnings.warn("control-plane listener of rank 0 binds all interfaces (MASTER_ADDR=%r) without PAA_RDZV_SECRET: "
                          "export the same secret on every rank when hosts outside the job can reach this one" % addr)
        for cand in range(port + 1, port + 1 + _PORT_SPAN):
            srv = None
            for host in hosts:
                srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                try:
                    srv.bind((host, cand))
                    break
                except OSError as exc:
                    last = exc
                    srv.close()
                    srv = None
                    if getattr(exc, "errno", None) == errno.EADDRINUSE:          # the port is taken, try the next one
                        break
            if srv is None:
                continue
            srv.listen(self.world_size)
            self._listener = srv
            break
        if self._listener is None:
            raise OSError("no free control-plane port in %d..%d: %s" % (port + 1, port + _PORT_SPAN, last))

        claimed = set()

        def handshake(conn):
            try:
                conn.settimeout(5.0)
                hello = _recv_exact(conn, 32 + 4)
                peer = struct.unpack("<i", hello[32:])[0]
                with self._lock:
                    ok = (hmac.compare_digest(hello[:32], self._token) and 0 < peer < self.world_size
                          and peer not in claimed)
                    if ok:
                        claimed.add(peer)                 # (claimed before the reply: a second claimant is refused)
                if not ok:
                    conn.close()
                    return
                try:
                    conn.sendall(self._token)
                    conn.settimeout(None)
                    conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    with self._lock:
                        self._peers[peer] = conn          # (joined: the socket is ready for the collectives)
                except OSError:
                    with self._lock:
                        claimed.discard(peer)
                    conn.close()
            except (OSError, struct.error, ConnectionError):
                conn.close()

        while True:
            with self._lock:
                joined = len(self._peers)
            if joined >= self.world_size - 1:
                break
            left = deadline - time.monotonic()
            if left <= 0:
                raise TimeoutError("control plane: %d of %d ranks joined" % (joined + 1, self.world_size))
            self._listener.settimeout(min(left, 0.2))
            try:
                conn, _ = self._listener.accept()
            except socket.timeout:
                continue
            threading.Thread(target=handshake, args=(conn,), daemon=True).start()

    def _join(self, addr, port, deadline):
        while time.monotonic() < deadline:
            for cand in range(port + 1, port + 1 + _PORT_SPAN):
                try:
                    s = socket.create_connection((addr, cand), timeout=2.0)
                except OSError:
                    continue
                try:
                    s.settimeout(5.0)
                    s.sendall(self._token + struct.pack("<i", self.rank))
                    if _recv_exact(s, 32) == self._token:
                        s.settimeout(None)
                        s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        self._root = s
                        return
                except OSError:
                    pass
                s.close()
            time.sleep(0.05)
        raise TimeoutError("control plane: rank %d could not reach rank 0 at %s:%d+" % (self.rank, addr, port + 1))

    # ---- collectives
    def all_gather(self, obj):
        """[obj of rank 0, obj of rank 1, ...] on every rank."""
        if self.world_size == 1:
            return [_decode(json.loads(json.dumps(_encode(obj))))]
        if self.rank == 0:
            out = [obj] + [None] * (self.world_size - 1)
            for r, s in self._peers.items():
                out[r] = _recv(s)
            for s in self._peers.values():
                _send(s, out)
            return _decode(json.loads(json.dumps(_encode(out))))          # the same value every rank sees
        _send(self._root, obj)
        return _recv(self._root)

    def broadcast(self, obj, src=0):
        """obj of rank `src` on every rank."""
        return self.all_gather(obj if self.rank == src else None)[src]

    def barrier(self):
        self.all_gather(None)

    def all_max(self, value):
        return max(self.all_gather(value))

    def close(self):
        for s in list(self._peers.values()) + [self._root, self._listener]:
            if s is not None:
                try:
                    s.close()
                except OSError:
                    pass
        self._peers, self._root, self._listener = {}, None, None
