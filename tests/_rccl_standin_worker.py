"""Worker of tests/test_camera_shard.py::test_rccl_standin_mesh_on_host_buffers: one process per rank drives tests/native/rccl_standin.cpp
directly through ctypes with HOST buffers (BEVW_RCCL_STANDIN_HOST=1) -- the mesh, group and all-gather logic of the stand-in without a GPU.

argv: library, unique id (hex), rank, world."""
import ctypes as C
import sys

import numpy as np


class UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


def main():
    lib_path, ident_hex, rank, world = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    L = C.CDLL(lib_path)
    u8 = 1   # ncclUint8
    ident = UniqueId.from_buffer_copy(bytes.fromhex(ident_hex))
    comm = C.c_void_p()
    L.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    L.ncclSend.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.ncclRecv.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
    L.ncclCommDestroy.argtypes = [C.c_void_p]
    assert L.ncclCommInitRank(C.byref(comm), world, ident, rank) == 0

    def payload(r, n):
        return np.random.default_rng(1000 + r).integers(0, 256, n, dtype=np.uint8)

    # all-gather of 4 KB pieces
    mine = payload(rank, 4096)
    gathered = np.zeros(4096 * world, np.uint8)
    assert L.ncclAllGather(mine.ctypes.data, gathered.ctypes.data, 4096, u8, comm, None) == 0
    for r in range(world):
        assert np.array_equal(gathered[4096 * r:4096 * (r + 1)], payload(r, 4096)), "all-gather piece %d" % r
    # the product's gather pattern, every rank once the root: the others send boxes of different sizes (3 MB + rank: far beyond a socket buffer),
    # the root receives them in one group
    for root in range(world):
        n_of = [3 * 1024 * 1024 + 17 * r for r in range(world)]
        if rank != root:
            box = payload(10 + rank + 7 * root, n_of[rank])
            assert L.ncclSend(box.ctypes.data, box.nbytes, u8, root, comm, None) == 0
        else:
            bufs = [np.zeros(n_of[r], np.uint8) for r in range(world)]
            assert L.ncclGroupStart() == 0
            for r in range(world):
                if r != root:
                    assert L.ncclRecv(bufs[r].ctypes.data, bufs[r].nbytes, u8, r, comm, None) == 0
            assert L.ncclGroupEnd() == 0
            for r in range(world):
                if r != root:
                    assert np.array_equal(bufs[r], payload(10 + r + 7 * root, n_of[r])), "box of rank %d at root %d" % (r, root)
    # a grouped exchange with sends AND receives on every rank (ring), 2 MB each, plus a send / receive to itself
    nxt, prv = (rank + 1) % world, (rank - 1) % world
    out, inn, self_in = payload(50 + rank, 2 << 20), np.zeros(2 << 20, np.uint8), np.zeros(1000, np.uint8)
    assert L.ncclGroupStart() == 0
    assert L.ncclSend(out.ctypes.data, out.nbytes, u8, nxt, comm, None) == 0
    assert L.ncclRecv(inn.ctypes.data, inn.nbytes, u8, prv, comm, None) == 0
    assert L.ncclSend(out.ctypes.data, 1000, u8, rank, comm, None) == 0
    assert L.ncclRecv(self_in.ctypes.data, 1000, u8, rank, comm, None) == 0
    assert L.ncclGroupEnd() == 0
    assert np.array_equal(inn, payload(50 + prv, 2 << 20)) and np.array_equal(self_in, out[:1000])
    # a size mismatch is an error, as in NCCL (rank 0 <-> rank 1 only)
    if world >= 2 and rank < 2:
        small = np.zeros(64, np.uint8)
        if rank == 0:
            L.ncclSend(small.ctypes.data, 64, u8, 1, comm, None)   # (may fail too: the receiver hangs up once it has seen the wrong size)
        else:
            assert L.ncclRecv(small.ctypes.data, 32, u8, 0, comm, None) != 0
    L.ncclCommDestroy(comm)
    print("standin rank %d of %d ok" % (rank, world))


if __name__ == "__main__":
    main()
