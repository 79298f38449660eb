"""Camera-per-GPU surround BEV: every rank stitches the cameras it owns, one exchange, the stitch rank adds the parts.

The reference has no multi-GPU code; this module is the scale-out form of ``BevGenerator.__call__``
(surroundBEV.py:312-325) for BASELINE config 5.  It leans on one arithmetic fact: the reference
combines the four masked images and the car with ``cv2.add`` (surroundBEV.py:318-320, 323-324), which saturates, so the
result is ``min(255, sum)`` however the terms are grouped.  A rank therefore adds its own cameras first
(``bevw_shard_run_device``), sends the bounding box of its masks, and the stitch rank adds the boxes
(``bevw_combine_device``) -- never a summing collective, u8 sums would wrap.

    ranks 4g .. 4g+3 form camera group g: rank 4g+c owns camera c (front, back, left, right)      [world 4, 8, 12 ..]
    world 2: rank 0 owns front+back, rank 1 owns left+right;  world 1: one rank owns all four (no exchange)
    groups are replicas: each works on its own frame sets, as in batch sharding.

With balance=True there is one more (tiny) exchange before the warp: the per-frame V sums of every camera
(luminance_balance needs the mean over all four, surroundBEV.py:60-66) are all-gathered inside the group; the white
balance (color_balance, :321-322) and the car run on the stitch rank after the add.

Transport.  The data plane is RCCL over xGMI, called NATIVELY by libbevwarp (``bevw_comm_*``, ``bevw_shard_allgather_vsums``,
``bevw_shard_gather_parts``: csrc/bevw_comm.h) on the engine's own HIP stream -- no PyTorch anywhere in this package, no host
synchronisation between the rank-local stitch, the exchange and the combine.  Only the 128-byte RCCL unique id and the mask
boxes (16 bytes per rank, once) travel out of band, over a plain TCP socket (``SocketGroup``; MASTER_ADDR / MASTER_PORT of the
launcher).  A ``transport`` object with ``all_gather(array)`` / ``gather_parts(part, shapes, root)`` on host arrays can be
injected instead: that is how the CPU tests stand in for RCCL (tests/_shard_common.py: gloo) -- test infrastructure, not a
product path.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

try:
    from .. import _ffi
except ImportError:  # imported as a top-level package (main.py drop-in layout)
    from cameracalibration_amd import _ffi
from . import surroundBEV as _sb

lib, check, ptr, f64 = _ffi.lib, _ffi.check, _ffi.ptr, _ffi.f64

CAMERA_NAMES = ('front', 'back', 'left', 'right')


def camera_assignment(world_size: int):
    """-> [(group, cams)] per rank.  cams: ascending camera ids (0 front, 1 back, 2 left, 3 right)."""
    if world_size == 1:
        return [(0, (0, 1, 2, 3))]
    if world_size == 2:
        return [(0, (0, 1)), (0, (2, 3))]
    if world_size % 4 == 0:
        return [(r // 4, (r % 4,)) for r in range(world_size)]
    raise Exception("camera-per-GPU mode needs 1, 2 or a multiple of 4 ranks, got {}".format(world_size))


def group_ranks(world_size: int, group: int):
    return [r for r, (g, _) in enumerate(camera_assignment(world_size)) if g == group]


class HipShardEngine:
    """One bevw_handle in camera-shard mode (bevw_set_camera_shard).  Host-array methods for the drop-in path and
    ``*_device`` methods for callers that keep frames resident in HBM."""

    def __init__(self, rig, cams, blend, balance, device=0):
        _ffi.require_device()
        self.cams = tuple(int(c) for c in cams)
        self.device = int(device)
        self.cfg = _sb._snapshot_config(blend, balance, device, _ffi.SCHED_TILE_PLAN)
        self.blend, self.balance = bool(blend), bool(balance)
        h = C.c_void_p()
        check(lib().bevw_create(C.byref(self.cfg), C.byref(h)))
        self.h = h
        try:
            ids = np.asarray(self.cams, np.int32)
            check(lib().bevw_set_camera_shard(self.h, ptr(ids), len(self.cams)))
            for c in self.cams:
                K, D, H = rig[c]
                check(lib().bevw_set_camera(self.h, c, ptr(f64(K, 9)), ptr(f64(D, 4)), ptr(f64(H, 9))))
            check(lib().bevw_build(self.h))
            box = np.zeros(4, np.int32)
            check(lib().bevw_shard_box(self.h, ptr(box)))
        except Exception:
            self.close()
            raise
        self.box = tuple(int(v) for v in box)
        self.fw, self.fh = self.cfg.frame_width, self.cfg.frame_height
        self.bw, self.bh = self.cfg.bev_width, self.cfg.bev_height
        self._bufs = {}

    # ---- sizes -------------------------------------------------------------------------------------------------
    @property
    def frame_set_bytes(self):
        return len(self.cams) * self.fw * self.fh * 3

    @property
    def bev_bytes(self):
        return self.bw * self.bh * 3

    @staticmethod
    def box_bytes(box):
        return (box[2] - box[0]) * (box[3] - box[1]) * 3

    def _buf(self, name, nbytes) -> _ffi.DeviceBuffer:
        b = self._bufs.get(name)
        if b is None or b.nbytes < nbytes:
            if b is not None:
                b.free()
            b = self._bufs[name] = _ffi.DeviceBuffer(nbytes, self.device)
        return b

    # ---- device-resident calls (pointers are ints) ---------------------------------------------------------------
    def vsums_device(self, d_frames, batch, d_vsums):
        check(lib().bevw_shard_vsums_device(self.h, d_frames, batch, d_vsums))

    def run_device(self, d_frames, batch, d_all_vsums, d_full):
        check(lib().bevw_shard_run_device(self.h, d_frames, batch, d_all_vsums, d_full))

    def pack_device(self, d_full, batch, d_packed):
        check(lib().bevw_shard_pack_device(self.h, d_full, batch, d_packed))

    def combine_device(self, d_parts, boxes, batch, d_car, d_out):
        n = len(d_parts)
        arr = (C.c_void_p * n)(*[C.c_void_p(p) for p in d_parts])
        bx = np.ascontiguousarray(np.asarray(boxes, np.int32).reshape(n, 4))
        check(lib().bevw_combine_device(self.h, arr, ptr(bx), n, batch, d_car, d_out))

    def sync(self):
        check(lib().bevw_sync(self.h))

    # ---- host-array calls ---------------------------------------------------------------------------------------
    def _upload_frames(self, frames):
        f = np.ascontiguousarray(frames)
        if f.dtype != np.uint8 or f.ndim != 5 or f.shape[1:] != (len(self.cams), self.fh, self.fw, 3):
            raise Exception("frames must be uint8 [B, {}, {}, {}, 3], got {} {}".format(
                len(self.cams), self.fh, self.fw, f.dtype, f.shape))
        return self._buf("frames", max(f.nbytes, 4)).upload(f), f.shape[0]

    def vsums(self, frames) -> np.ndarray:
        d, batch = self._upload_frames(frames)
        v = self._buf("vsums", max(8 * batch * len(self.cams), 8))
        self.vsums_device(d.ptr, batch, v.ptr)
        self.sync()
        return v.download((batch, len(self.cams)), np.uint64)

    def partial(self, frames, all_vsums=None) -> np.ndarray:
        """-> packed part uint8 [B, y1-y0, x1-x0, 3] of this rank's cameras."""
        d, batch = self._upload_frames(frames)
        d_all = None
        if self.balance:
            a = np.ascontiguousarray(np.asarray(all_vsums, np.uint64).reshape(batch, 4))
            d_all = self._buf("all_vsums", max(a.nbytes, 8)).upload(a).ptr
        full = self._buf("full", max(batch * self.bev_bytes, 4))
        packed = self._buf("packed", max(batch * self.box_bytes(self.box), 4))
        self.run_device(d.ptr, batch, d_all, full.ptr)
        self.pack_device(full.ptr, batch, packed.ptr)
        self.sync()
        x0, y0, x1, y1 = self.box
        return packed.download((batch, y1 - y0, x1 - x0, 3))

    def combine(self, parts, boxes, car=None) -> np.ndarray:
        batch = parts[0].shape[0]
        ptrs = []
        for k, (p, bx) in enumerate(zip(parts, boxes)):
            p = np.ascontiguousarray(p, np.uint8)
            assert p.shape == (batch, bx[3] - bx[1], bx[2] - bx[0], 3), (p.shape, bx)
            ptrs.append(self._buf("part%d" % k, max(p.nbytes, 4)).upload(p).ptr)
        d_car = None
        if car is not None:
            c = _ffi.as_u8_image(car, "car image")
            if c.shape != (self.bh, self.bw, 3):
                raise Exception("car image must be padded to the BEV size")
            d_car = self._buf("car", c.nbytes).upload(c).ptr
        out = self._buf("out", max(batch * self.bev_bytes, 4))
        self.combine_device(ptrs, boxes, batch, d_car, out.ptr)
        self.sync()
        return out.download((batch, self.bh, self.bw, 3))

    def close(self):
        for b in getattr(self, "_bufs", {}).values():
            b.free()
        self._bufs = {}
        if getattr(self, "h", None):
            lib().bevw_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _recv_exact(c, n):
    buf = b""
    while len(buf) < n:
        chunk = c.recv(n - len(buf))
        if not chunk:
            raise Exception("control channel closed")
        buf += chunk
    return buf


class HubDirectory:
    """Where the hubs of the camera groups listen.  Only GLOBAL rank 0 is known to be reachable at MASTER_ADDR, and a group's hub
    (global rank 4 g) generally runs on another host, so a hub binds an ephemeral port on all interfaces and REGISTERS it here; the
    members of the group ASK here.  Global rank 0 serves the directory from a daemon thread on (all interfaces, port) until every
    group has registered and every member has asked (or the timeout passes: the listening socket is closed either way).
    Wire format: b"R" group:u32 port:u32 -> no reply (the hub's host is the peer address of that connection);
                 b"Q" group:u32          -> u32 port, u16 n, n bytes of host (sent once the group has registered)."""

    def __init__(self, port: int, ngroups: int, nqueries: int, timeout: float = 120.0):
        import socket
        import threading

        self.srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        self.srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        self.srv.bind(("", port))
        self.srv.listen(ngroups + nqueries + 4)
        self.thread = threading.Thread(target=self._serve, args=(int(ngroups), int(nqueries), float(timeout)), daemon=True)
        self.thread.start()

    def _serve(self, ngroups, nqueries, timeout):
        import socket
        import time

        reg, pending, served = {}, [], 0
        deadline = time.time() + timeout
        self.srv.settimeout(0.1)
        try:
            while (len(reg) < ngroups or served < nqueries) and time.time() < deadline:
                try:
                    c, peer = self.srv.accept()
                except socket.timeout:
                    c = None
                if c is not None:
                    try:
                        c.settimeout(1.0)     # connections are served one after the other: a silent one may hold the directory for a second, not five
                        kind = _recv_exact(c, 1)
                        group = int.from_bytes(_recv_exact(c, 4), "little")
                        if kind == b"R" and 0 <= group < ngroups:
                            reg[group] = (peer[0], int.from_bytes(_recv_exact(c, 4), "little"))
                            c.close()
                        elif kind == b"Q" and 0 <= group < ngroups:
                            pending.append((c, group))
                        else:
                            c.close()            # not one of ours
                    except Exception:
                        c.close()
                still = []
                for c, g in pending:
                    if g not in reg:
                        still.append((c, g))
                        continue
                    host = reg[g][0].encode()
                    try:
                        c.sendall(reg[g][1].to_bytes(4, "little") + len(host).to_bytes(2, "little") + host)
                    except OSError:
                        pass
                    c.close()
                    served += 1
                pending = still
        finally:
            for c, _ in pending:
                c.close()
            self.srv.close()

    @staticmethod
    def _connect(addr, port, timeout):
        import socket
        import time

        t0 = time.time()
        while True:
            try:
                return socket.create_connection((addr, port), timeout=timeout)
            except OSError:
                if time.time() - t0 > timeout:
                    raise
                time.sleep(0.05)

    @staticmethod
    def register(addr, port, group, hub_port, timeout=120.0):
        c = HubDirectory._connect(addr, port, timeout)
        c.sendall(b"R" + int(group).to_bytes(4, "little") + int(hub_port).to_bytes(4, "little"))
        c.close()

    @staticmethod
    def lookup(addr, port, group, timeout=120.0):
        c = HubDirectory._connect(addr, port, timeout)
        c.settimeout(timeout)
        c.sendall(b"Q" + int(group).to_bytes(4, "little"))
        hub_port = int.from_bytes(_recv_exact(c, 4), "little")
        n = int.from_bytes(_recv_exact(c, 2), "little")
        host = _recv_exact(c, n).decode()
        c.close()
        return host, hub_port


class SocketGroup:
    """Out-of-band control channel of one camera group (a star over TCP, group rank 0 is the hub): carries the RCCL unique
    id and the mask boxes once, at construction.  Plain sockets -- the data plane is RCCL.

    ``addr`` / ``port``: where the hub listens.  With ``directory=(addr, port, group)`` the hub instead binds an ephemeral port on all
    interfaces and publishes it through the HubDirectory at that address (multi-node worlds: the hub of group g > 0 does not run
    on MASTER_ADDR's host)."""

    def __init__(self, rank: int, world: int, addr: str, port: int, timeout: float = 120.0, directory=None):
        import socket
        import time

        self.rank, self.world = int(rank), int(world)
        self.peers = []
        if self.world == 1:
            return
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            got = {}
            try:
                if directory is not None:
                    srv.bind(("", 0))
                    srv.listen(self.world)
                    HubDirectory.register(directory[0], directory[1], directory[2], srv.getsockname()[1], timeout)
                else:
                    srv.bind((addr, port))
                    srv.listen(self.world)
                deadline = time.time() + timeout
                while len(got) < self.world - 1:
                    srv.settimeout(max(0.05, deadline - time.time()))
                    c, _ = srv.accept()          # socket.timeout when a member never shows up: the overall deadline
                    # the 4-byte rank hello gets a short deadline of its own: a stray connection that sends nothing (a port scanner, another
                    # job) is dropped after 2 s and the hub keeps accepting, instead of eating the whole membership timeout and aborting
                    c.settimeout(min(2.0, max(0.05, deadline - time.time())))
                    try:
                        r = int.from_bytes(_recv_exact(c, 4), "little")
                    except Exception:
                        c.close()
                        continue
                    if not (0 < r < self.world) or r in got:
                        c.close()                # not a member of this group
                        continue
                    c.settimeout(timeout)
                    got[r] = c
            except Exception:
                for c in got.values():
                    c.close()
                raise
            finally:
                srv.close()
            self.peers = [got[r] for r in range(1, self.world)]
        else:
            if directory is not None:
                addr, port = HubDirectory.lookup(directory[0], directory[1], directory[2], timeout)
            c = HubDirectory._connect(addr, port, timeout)
            c.settimeout(timeout)
            c.sendall(self.rank.to_bytes(4, "little"))
            self.peers = [c]

    _recv = staticmethod(_recv_exact)

    def broadcast(self, payload, nbytes: int) -> bytes:
        """bytes of group rank 0 -> every rank."""
        if self.world == 1:
            return bytes(payload)
        if self.rank == 0:
            for c in self.peers:
                c.sendall(bytes(payload))
            return bytes(payload)
        return self._recv(self.peers[0], nbytes)

    def all_gather(self, payload: bytes):
        """-> [bytes of rank 0, rank 1, ...] on every rank (equal sizes)."""
        if self.world == 1:
            return [bytes(payload)]
        n = len(payload)
        if self.rank == 0:
            parts = [bytes(payload)] + [self._recv(c, n) for c in self.peers]
            blob = b"".join(parts)
            for c in self.peers:
                c.sendall(blob)
            return parts
        self.peers[0].sendall(bytes(payload))
        blob = self._recv(self.peers[0], n * self.world)
        return [blob[i * n:(i + 1) * n] for i in range(self.world)]

    def close(self):
        for c in self.peers:
            try:
                c.close()
            except OSError:
                pass
        self.peers = []


def launcher_env():
    """(rank, world_size, local_rank, master_addr, master_port) from the launcher's environment (the one-process-per-GPU launcher bench.py is started by,
    mpirun wrappers, ...): RANK, WORLD_SIZE, LOCAL_RANK, MASTER_ADDR, MASTER_PORT."""
    import os

    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0")),
            os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ.get("MASTER_PORT", "29500")))


class RcclGroup:
    """The RCCL communicator of one camera group (bevw_comm, csrc/bevw_comm.h) plus its control channel."""

    def __init__(self, group_rank, group_world, device, control: SocketGroup):
        self.rank, self.world, self.control = int(group_rank), int(group_world), control
        ident = (C.c_uint8 * 128)()
        if self.rank == 0:
            check(lib().bevw_comm_unique_id(ident))
        raw = control.broadcast(bytes(ident), 128)
        ident = (C.c_uint8 * 128).from_buffer_copy(raw)
        comm = C.c_void_p()
        check(lib().bevw_comm_create(int(device), self.rank, self.world, ident, C.byref(comm)))
        self.comm = comm

    def close(self):
        if getattr(self, "comm", None):
            lib().bevw_comm_destroy(self.comm)
            self.comm = None
        self.control.close()


class CameraShardedBev:
    """BevGenerator(blend, balance) spread one-camera-per-GPU.

    ``rig``: {'front': (K, D, H), ...} or None for the repo's data/ directory; sizes come from the module arguments of
    surroundBEV (``BevGenerator.get_args()``), exactly as for BevGenerator.  ``rank`` / ``world_size`` default to the
    launcher's RANK / WORLD_SIZE.  With more than one rank the exchange runs over RCCL (``RcclGroup``) unless a host-array
    ``transport`` is injected (CPU tests); ``engine_factory(rig_list, cams, blend, balance, device)`` exists so the exchange
    logic can be exercised without a GPU (tests) -- the default is the HIP engine and there is no CPU path.
    """

    def __init__(self, blend=False, balance=False, *, rig=None, rank=None, world_size=None, device=0, engine_factory=None,
                 transport=None, control_port_offset=1):
        env_rank, env_world, _, addr, port = launcher_env()
        self.rank = int(env_rank if rank is None else rank)
        self.world_size = int(env_world if world_size is None else world_size)
        assign = camera_assignment(self.world_size)
        self.group, self.cams = assign[self.rank]
        self.ranks = group_ranks(self.world_size, self.group)
        self.group_rank = self.ranks.index(self.rank)
        self.blend, self.balance = bool(blend), bool(balance)
        _sb.BevGenerator.init_args(None)   # args -> module sizes, as BevGenerator.__init__ does (surroundBEV.py:284)
        if rig is None:
            cams = [_sb.Camera(n) for n in CAMERA_NAMES]
            rig_list = [(c.camera_mat, c.dist_coeff, c.homography) for c in cams]
        else:
            rig_list = [rig[n] for n in CAMERA_NAMES]
        factory = engine_factory or HipShardEngine
        self.engine = factory(rig_list, self.cams, self.blend, self.balance, device)
        self.transport, self.rccl = transport, None
        mine = np.asarray(self.engine.box, np.int32)
        if len(self.ranks) == 1:
            boxes = [mine]
        elif transport is not None:
            boxes = transport.all_gather(mine)
        else:
            if not isinstance(self.engine, HipShardEngine):
                raise Exception("the RCCL exchange needs the HIP engine; inject a host transport for stand-in engines")
            # rendezvous: global rank 0 (the one process known to run at MASTER_ADDR) serves the hub directory on
            # MASTER_PORT + control_port_offset (BEVW_CONTROL_PORT overrides the port); every group's hub binds an ephemeral port
            # on its own host and registers it there
            import os as _os

            dport = int(_os.environ.get("BEVW_CONTROL_PORT", port + control_port_offset))
            ngroups = len({g for g, _ in assign})
            self._directory = HubDirectory(dport, ngroups, self.world_size - ngroups) if self.rank == 0 else None
            control = SocketGroup(self.group_rank, len(self.ranks), addr, 0, directory=(addr, dport, self.group))
            self.rccl = RcclGroup(self.group_rank, len(self.ranks), device, control)
            boxes = [np.frombuffer(b, np.int32) for b in control.all_gather(mine.tobytes())]
        self.boxes = [tuple(int(v) for v in b) for b in boxes]
        self.step = 0
        self._pipe = None

    @property
    def camera_names(self):
        return tuple(CAMERA_NAMES[c] for c in self.cams)

    def next_root(self) -> int:
        """The stitch role rotates over the group so that ingress is spread over the ranks' links."""
        r = self.ranks[self.step % len(self.ranks)]
        self.step += 1
        return r

    def __call__(self, frames, car=None, root=None):
        """frames: uint8 [B, len(cams), FH, FW, 3] -- this rank's cameras of B frame sets (all ranks of a group pass
        the same B).  Returns uint8 [B, BH, BW, 3] on the stitch rank and None on the others."""
        if root is None:
            root = self.next_root()
        if root not in self.ranks:
            raise Exception("stitch rank {} is not in camera group {}".format(root, self.ranks))
        frames = np.ascontiguousarray(frames)
        batch = frames.shape[0]
        if self.rccl is not None:
            return self._call_rccl(frames, car, root)
        all_vsums = None
        if self.balance:
            mine = self.engine.vsums(frames)                      # [B, ncams]
            per_rank = [mine] if len(self.ranks) == 1 else self.transport.all_gather(mine)   # group order == camera order
            all_vsums = np.concatenate(per_rank, axis=1)
            assert all_vsums.shape == (batch, 4)
        part = self.engine.partial(frames, all_vsums)
        shapes = [(batch, b[3] - b[1], b[2] - b[0], 3) for b in self.boxes]
        parts = [part] if len(self.ranks) == 1 else self.transport.gather_parts(part, shapes, root)
        if self.rank != root:
            return None
        return self.engine.combine(parts, self.boxes, car)

    def _call_rccl(self, frames, car, root):
        """host arrays in, host array out, the exchange on the device over RCCL (the resident pipeline with an upload in
        front and a download behind)."""
        e, batch = self.engine, frames.shape[0]
        if self._pipe is None or self._pipe.batch != batch:
            if self._pipe is not None:
                self._pipe.close()
            self._pipe = ResidentShardPipeline(self, batch)
        d_frames, _ = e._upload_frames(frames)
        d_car = None
        if car is not None:
            c = _ffi.as_u8_image(car, "car image")
            if c.shape != (e.bh, e.bw, 3):
                raise Exception("car image must be padded to the BEV size")
            d_car = e._buf("car", c.nbytes).upload(c).ptr
        self._pipe.step(d_frames.ptr, d_car, root=root)
        e.sync()
        if self.rank != root:
            return None
        return self._pipe.out.download((batch, e.bh, e.bw, 3))

    def close(self):
        if getattr(self, "_pipe", None) is not None:
            self._pipe.close()
            self._pipe = None
        if getattr(self, "rccl", None) is not None:
            self.rccl.close()
            self.rccl = None
        if getattr(self, "engine", None) is not None:
            self.engine.close()
            self.engine = None

    # the RCCL communicator, the control sockets and the device buffers are released when the object is dropped or leaves a `with`
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ResidentShardPipeline:
    """The same step with everything resident in HBM (what bench.py times): frames of the owned cameras stay in a device
    buffer, V sums and parts travel device-to-device over RCCL on the engine's stream (bevw_shard_allgather_vsums,
    bevw_shard_gather_parts), the stitch rank writes the BEV batch into its own device buffer.  No host synchronisation inside
    a step.  (With an injected host transport -- CPU-test stand-in -- the exchange is host-staged.)"""

    def __init__(self, gen: CameraShardedBev, batch: int):
        if not isinstance(gen.engine, HipShardEngine):
            raise Exception("the resident pipeline needs the HIP engine")
        self.gen, self.e, self.batch = gen, gen.engine, int(batch)
        e, dev = self.e, gen.engine.device
        D = _ffi.DeviceBuffer
        self.full = D(batch * e.bev_bytes, dev)
        self.packed = D(batch * e.box_bytes(e.box), dev)
        self.out = D(batch * e.bev_bytes, dev)
        self.recv = [None if r == gen.rank else D(batch * e.box_bytes(b), dev) for r, b in zip(gen.ranks, gen.boxes)]
        n = len(e.cams)
        self.vs = D(8 * batch * n, dev)
        self.vs_all = D(8 * batch * 4, dev)
        self.multi = len(gen.ranks) > 1
        self.on_device = self.multi and gen.rccl is not None
        nr = len(gen.ranks)
        self._recv_ptrs = (C.c_void_p * nr)(*[C.c_void_p(b.ptr if b is not None else None) for b in self.recv])
        self._recv_bytes = (C.c_size_t * nr)(*[batch * e.box_bytes(b) for b in gen.boxes])

    def _all_vsums(self, d_frames):
        e, g, B, n = self.e, self.gen, self.batch, len(self.e.cams)
        e.vsums_device(d_frames, B, self.vs.ptr)
        if not self.multi:
            return self.vs.ptr                                   # one rank owns all four: already [B][4]
        if self.on_device:
            check(lib().bevw_shard_allgather_vsums(e.h, g.rccl.comm, self.vs.ptr, B, self.vs_all.ptr))
        else:
            e.sync()
            per_rank = g.transport.all_gather(self.vs.download((B, n), np.uint64))
            self.vs_all.upload(np.ascontiguousarray(np.concatenate(per_rank, axis=1)))
        return self.vs_all.ptr

    def step(self, d_frames: int, d_car=None, root=None):
        """One batch.  Returns the stitch rank (its `out` buffer holds uint8 [batch, BH, BW, 3])."""
        g, e, B = self.gen, self.e, self.batch
        if root is None:
            root = g.next_root()
        d_all = self._all_vsums(d_frames) if g.balance else None
        e.run_device(d_frames, B, d_all, self.full.ptr)
        e.pack_device(self.full.ptr, B, self.packed.ptr)
        parts = [self.packed.ptr if b is None else b.ptr for b in self.recv]
        if self.multi:
            if self.on_device:
                check(lib().bevw_shard_gather_parts(e.h, g.rccl.comm, self.packed.ptr, self.packed.nbytes,
                                                    g.ranks.index(root), self._recv_ptrs, self._recv_bytes))
            else:
                e.sync()
                shapes = [(B, b[3] - b[1], b[2] - b[0], 3) for b in g.boxes]
                mine = self.packed.download(shapes[g.ranks.index(g.rank)])
                got = g.transport.gather_parts(mine, shapes, root)
                if g.rank == root:
                    for b, p in zip(self.recv, got):
                        if b is not None:
                            b.upload(p)
        if g.rank == root:
            e.combine_device(parts, g.boxes, B, d_car, self.out.ptr)
        return root

    def close(self):
        for b in [self.full, self.packed, self.out, self.vs, self.vs_all] + [r for r in self.recv if r is not None]:
            b.free()
