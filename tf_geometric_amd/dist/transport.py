# coding=utf-8
"""Transports of the sharded path: who moves halo rows (and the plan-time integer lists) between ranks.

  * TfgxDistTransport — THE PRODUCT PATH.  Everything goes through the C ABI of include/tfgx_dist.h
    (lib/libtfgx_dist.so = host code over libtfgx.so + librccl): an ncclComm_t created in-process
    (tfgx_dist_unique_id on rank 0, the 128 bytes broadcast through the existing torch.distributed group, then
    tfgx_dist_comm_init = ncclCommInitRank), grouped ncclSend / ncclRecv per round on an EXPLICIT SECOND HIP STREAM,
    HIP events between that stream and the compute stream (tfgx_halo_exchange_start / _finish, tfgx_halo_reverse_start /
    _finish), tfgx_alltoallv for the plan-time lists, tfgx_allreduce_sum_* for histograms and weight gradients.
    torch.distributed is the control channel only (rendezvous, the id broadcast, barriers in bench.py).
  * TorchDistTransport — torch.distributed collectives, host-staged when the group is gloo.  This is what the
    world_size-2/3 CPU tests drive (tests/test_dist_gloo.py, numpy test backend) and what a 2-rank run on ONE GPU uses
    (RCCL refuses two ranks per device); never chosen for a HIP backend on an NCCL group.

Both consume the SAME exchange description the ShardedGraph plan builds (round-major counts, dense / packed entries,
packed index lists), so the layout logic the gloo tests verify is the layout the C ABI is handed.
The reference has no counterpart: its distributed demos replicate the graph and all-reduce gradients
(demo/demo_distributed_gcn.py:52-57,99) — that all-reduce is all_reduce_sum_f32 here.
"""
import ctypes
import os

import torch
import torch.distributed as dist

from .. import _lib as L

_HERE = os.path.dirname(os.path.abspath(__file__))
DIST_LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libtfgx_dist.so")
UNIQUE_ID_BYTES = 128

_P, _I32, _I64, _SZ = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_size_t
_DIST_SIGNATURES = {
    "tfgx_dist_last_error": (ctypes.c_char_p, []),
    "tfgx_dist_unique_id": (ctypes.c_int, [_P]),
    "tfgx_dist_comm_init": (ctypes.c_int, [_I32, _I32, _P, ctypes.POINTER(_P)]),
    "tfgx_dist_comm_destroy": (ctypes.c_int, [_P]),
    "tfgx_dist_comm_abort": (ctypes.c_int, [_P]),
    "tfgx_dist_comm_info": (ctypes.c_int, [_P, ctypes.POINTER(_I32), ctypes.POINTER(_I32), ctypes.POINTER(_I32)]),
    "tfgx_alltoallv": (ctypes.c_int, [_P, _P, _P, _P, _I64, _I32, _P, _P]),
    "tfgx_allreduce_sum_i64": (ctypes.c_int, [_P, _I64, _P, _P]),
    "tfgx_allreduce_sum_f32": (ctypes.c_int, [_P, _I64, _P, _P]),
    "tfgx_halo_plan_create": (ctypes.c_int, [_I32, _I32, _I32, _P, _P, _P, _P, ctypes.POINTER(_P)]),
    "tfgx_halo_plan_destroy": (ctypes.c_int, [_P]),
    "tfgx_halo_plan_rows_sent": (_I64, [_P]),
    "tfgx_halo_plan_rows_packed": (_I64, [_P]),
    "tfgx_halo_plan_rows_received": (_I64, [_P]),
    "tfgx_halo_exchange_start": (ctypes.c_int, [_P, _P, _I64, _I64, _P, _I64, _P, _SZ, _P, _P, _P]),
    "tfgx_halo_exchange_finish": (ctypes.c_int, [_P, _I32, _P]),
    "tfgx_halo_reverse_start": (ctypes.c_int, [_P, _P, _I64, _P, _SZ, _P, _P, _P]),
    "tfgx_halo_reverse_start_round": (ctypes.c_int, [_P, _I32, _P, _I64, _P, _SZ, _P, _P, _P]),
    "tfgx_halo_reverse_finish": (ctypes.c_int, [_P, _P, _I64, _I64, _P, _P]),
}
_dist_lib = None


def load_dist_library():
    """lib/libtfgx_dist.so with every entry point of include/tfgx_dist.h bound.  Raises when it is not built."""
    global _dist_lib
    if _dist_lib is None:
        L.load_library()                      # libtfgx.so first (libtfgx_dist.so links against it)
        if not os.path.exists(DIST_LIB_PATH):
            raise L.TfgxError("tf_geometric_amd: {} is missing — run __graft_entry__.build()".format(DIST_LIB_PATH))
        lib = ctypes.CDLL(DIST_LIB_PATH)
        for name, (res, args) in _DIST_SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _dist_lib = lib
    return _dist_lib


def _dcheck(rc, what):
    if rc != 0:
        raise L.TfgxError("{} failed with code {}: {}".format(what, rc, (load_dist_library().tfgx_dist_last_error() or b"").decode()))


def _i64_array(values):
    return (ctypes.c_int64 * max(len(values), 1))(*[int(v) for v in values])


def _flat(rows):
    return [int(v) for r in rows for v in r]


def ipc_mode_note():
    """What this process runs with for cross-process device memory (see dist/__init__.py): None when
    HSA_ENABLE_IPC_MODE_LEGACY=0 is in the environment, otherwise a sentence for error messages / the bench line."""
    v = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")
    if v == "0":
        return None
    return ("HSA_ENABLE_IPC_MODE_LEGACY={} in this process (the MI355X host driver supports dmabuf IPC only: export "
            "HSA_ENABLE_IPC_MODE_LEGACY=0 before the first HIP call)".format("unset" if v is None else repr(v)))


class TfgxDistUnavailable(L.TfgxError):
    """Raised ON EVERY RANK of the group when any rank could not bring the C-ABI transport up (the ranks agree over the
    control channel before and after each collective bootstrap step, so nobody is left waiting inside RCCL)."""


class TfgxDistTransport(object):
    name = "tfgx_dist"

    def __init__(self, group=None):
        self.group = group
        inited = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if inited else 1
        self.rank = dist.get_rank(group) if inited else 0
        self.comm = None
        self._bufs = {}
        # local preconditions first; nothing collective has run yet, so a rank that fails here can still tell the others
        err = None
        try:
            self.lib = load_dist_library()
            L.require_gpu()
            self.device = L.device()
            self.comm_stream = torch.cuda.Stream(device=self.device)       # north_star's "second HIP stream"
        except Exception as ex:          # noqa: BLE001 - a missing library, no GPU, ...
            err = "{}: {}".format(type(ex).__name__, ex)
        self._agree(err, "loading lib/libtfgx_dist.so")
        if self.world > 1:
            self._comm()          # collective: every rank of the group constructs its transport at the same point

    def _agree(self, err, stage):
        """Every rank reports its local outcome of `stage` over the CONTROL channel (torch.distributed object collective:
        works on gloo and on nccl groups); if any rank failed, EVERY rank raises TfgxDistUnavailable naming the ranks.
        No RCCL call of this transport is in flight while the ranks agree."""
        if self.world == 1:
            if err:
                raise TfgxDistUnavailable("tfgx_dist: {} failed: {}".format(stage, err))
            return
        box = [None] * self.world
        dist.all_gather_object(box, err, group=self.group)
        bad = ["rank {}: {}".format(r, e) for r, e in enumerate(box) if e]
        if bad:
            raise TfgxDistUnavailable("tfgx_dist: {} failed on {} of {} ranks ({})".format(stage, len(bad), self.world,
                                                                                        "; ".join(bad)))

    def _comm(self):
        if self.comm is None:
            uid, err = ctypes.create_string_buffer(UNIQUE_ID_BYTES), None
            if self.rank == 0:
                try:
                    _dcheck(self.lib.tfgx_dist_unique_id(uid), "tfgx_dist_unique_id")
                except Exception as ex:          # noqa: BLE001
                    err = "{}: {}".format(type(ex).__name__, ex)
            self._agree(err, "ncclGetUniqueId")
            if self.world > 1:
                box = [bytes(uid.raw)]
                src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
                dist.broadcast_object_list(box, src=src, group=self.group)      # control channel: 128 bytes
                uid = ctypes.create_string_buffer(box[0], UNIQUE_ID_BYTES)
            comm = ctypes.c_void_p()
            try:
                _dcheck(self.lib.tfgx_dist_comm_init(self.world, self.rank, uid, ctypes.byref(comm)), "tfgx_dist_comm_init")
            except Exception as ex:              # noqa: BLE001
                err = "{}: {}".format(type(ex).__name__, ex)
                if ipc_mode_note():
                    err += " [" + ipc_mode_note() + "]"
            self.comm = comm if err is None else None
            try:
                self._agree(err, "ncclCommInitRank")
            except TfgxDistUnavailable:
                self.close(abort=True)           # this rank's communicator came up, a peer's did not: never wait for it
                raise
        return self.comm

    def comm_info(self):
        """(ranks, rank, device) as the ncclComm_t itself reports them (ncclCommCount / UserRank / CuDevice); a transport
        of a 1-process job has no communicator: (1, 0, current device)."""
        if self.comm is None:
            return 1, 0, int(torch.cuda.current_device())
        w, r, d = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        _dcheck(self.lib.tfgx_dist_comm_info(self.comm, ctypes.byref(w), ctypes.byref(r), ctypes.byref(d)),
                "tfgx_dist_comm_info")
        return int(w.value), int(r.value), int(d.value)

    def checked_self_check(self):
        """self_check, then the ranks agree on its outcome (a rank that received wrong rows raises alone otherwise)."""
        err = None
        try:
            self.self_check()
        except Exception as ex:                  # noqa: BLE001
            err = "{}: {}".format(type(ex).__name__, ex)
        self._agree(err, "the self-check (rows through tfgx_alltoallv / tfgx_allreduce_sum_i64)")

    def close(self, abort=False):
        """Destroys the communicator (ncclCommDestroy after a device synchronise); abort=True is the failure path's
        ncclCommAbort, which does not wait for peers that may never arrive."""
        if self.comm is not None:
            if abort:
                self.lib.tfgx_dist_comm_abort(self.comm)
            else:
                torch.cuda.synchronize()
                self.lib.tfgx_dist_comm_destroy(self.comm)
            self.comm = None

    def self_check(self):
        """A few hundred bytes through every entry point the plan build uses (tfgx_alltoallv with ragged counts,
        tfgx_allreduce_sum_i64): raises when any peer's rows arrive wrong.  get_transport runs it once per communicator."""
        w, r = self.world, self.rank
        send = torch.cat([torch.full((p + 1, 2), r * 1000 + p, dtype=torch.int32, device=self.device) for p in range(w)])
        got = self.all_to_all_v(send, [p + 1 for p in range(w)], [r + 1] * w)
        want = torch.cat([torch.full((r + 1, 2), q * 1000 + r, dtype=torch.int32, device=self.device) for q in range(w)])
        tot = self.all_reduce_sum_i64(torch.tensor([r + 1, 1], dtype=torch.int64, device=self.device))
        torch.cuda.synchronize()
        if not torch.equal(got, want) or tot.tolist() != [w * (w + 1) // 2, w]:
            raise L.TfgxError("tfgx_dist self-check: rank {} received wrong rows / sums over the RCCL communicator".format(r))

    # ---- plan-time exchanges (device tensors, stream-ordered on the current stream)
    def all_reduce_sum_i64(self, t):
        if self.world > 1:
            assert t.dtype == torch.int64 and t.is_contiguous()
            _dcheck(self.lib.tfgx_allreduce_sum_i64(L.ptr(t), int(t.numel()), self._comm(), L.stream_ptr()),
                    "tfgx_allreduce_sum_i64")
        return t

    def all_to_all_v(self, send, send_counts, recv_counts):
        """Rows of `send` (any dtype, 1-D or 2-D) to their peers; counts in rows, per peer.  -> received rows."""
        send = send.contiguous()
        shape = (int(sum(recv_counts)),) + tuple(send.shape[1:])
        recv = torch.empty(shape, dtype=send.dtype, device=send.device)
        row_bytes = send.element_size()
        for s in send.shape[1:]:
            row_bytes *= int(s)
        _dcheck(self.lib.tfgx_alltoallv(L.ptr(send), _i64_array(send_counts), L.ptr(recv), _i64_array(recv_counts),
                                        row_bytes, self.world, self._comm(), L.stream_ptr()), "tfgx_alltoallv")
        return recv

    def all_reduce_sum_f32(self, flat):
        if self.world > 1:
            assert flat.dtype == torch.float32 and flat.is_contiguous()
            _dcheck(self.lib.tfgx_allreduce_sum_f32(L.ptr(flat), int(flat.numel()), self._comm(), L.stream_ptr()),
                    "tfgx_allreduce_sum_f32")
        return flat

    # ---- halo exchange
    def _plans(self, sg):
        """Three C plan objects per shard: two alternate for forward exchanges (aggregate_chunked keeps chunk c+1's
        exchange in flight while chunk c's rounds are being consumed; a plan owns one set of per-round events), one for
        the reverse exchange."""
        if getattr(sg, "_xplans", None) is None:
            R, W = sg.rounds, sg.world
            sc, rc = _i64_array(_flat(sg.round_send_counts)), _i64_array(_flat(sg.round_recv_counts))
            ds = _i64_array(_flat(sg.round_send_dense))
            plans = []
            for _ in range(3):
                p = ctypes.c_void_p()
                _dcheck(self.lib.tfgx_halo_plan_create(W, sg.rank, R, sc, rc, ds, L.ptr(sg.send_idx_packed),
                                                       ctypes.byref(p)), "tfgx_halo_plan_create")
                plans.append(p)
            assert self.lib.tfgx_halo_plan_rows_received(plans[0]) == sg.n_halo
            sg._xplans, sg._xnext = plans, 0
            sg._xrows_packed = int(self.lib.tfgx_halo_plan_rows_packed(plans[0]))
            sg._xrows_sent = int(self.lib.tfgx_halo_plan_rows_sent(plans[0]))
        return sg._xplans

    def _buf(self, key, floats):
        b = self._bufs.get(key)
        if b is None or b.numel() < floats:
            b = torch.empty(max(int(floats), 1), dtype=torch.float32, device=self.device)
            b.record_stream(self.comm_stream)
            self._bufs[key] = b
        return b

    def exchange_start(self, sg, table):
        if sg.rounds == 0 or sg.n_halo + sum(sg.send_counts) == 0:
            return None
        plans = self._plans(sg)
        k = sg._xnext % 2
        sg._xnext += 1
        if not table.is_contiguous():
            raise L.TfgxError("the [own | halo] source table must be dense (use ShardedGraph.alloc_table)")
        F = int(table.shape[1])
        send_buf = self._buf((id(sg), k, F), sg._xrows_packed * F)
        table.record_stream(self.comm_stream)
        halo = table[sg.n_own:]
        _dcheck(self.lib.tfgx_halo_exchange_start(plans[k], L.ptr(table), F, F, L.ptr(halo), F, L.ptr(send_buf),
                                                  int(send_buf.numel()), self._comm(), L.stream_ptr(),
                                                  ctypes.c_void_p(self.comm_stream.cuda_stream)),
                "tfgx_halo_exchange_start")
        return ["tfgx", plans[k], table, send_buf]

    def exchange_finish(self, sg, handle, j=None):
        """The CURRENT stream waits for round j (None: every round) of that exchange; nothing blocks on the host."""
        if handle is None:
            return
        _dcheck(self.lib.tfgx_halo_exchange_finish(handle[1], -1 if j is None else int(j), L.stream_ptr()),
                "tfgx_halo_exchange_finish")

    def reverse_start(self, sg, d_table):
        """Posts the reverse rounds for the halo-row gradients d_table[n_own:] (asynchronous)."""
        if sg.rounds == 0 or sg.n_halo + sum(sg.send_counts) == 0:
            return None
        plan = self._plans(sg)[2]
        assert d_table.is_contiguous()
        U = int(d_table.shape[1])
        back = torch.empty((max(sg._xrows_sent, 1), U), dtype=torch.float32, device=self.device)
        back.record_stream(self.comm_stream)
        d_table.record_stream(self.comm_stream)
        d_halo = d_table[sg.n_own:]
        _dcheck(self.lib.tfgx_halo_reverse_start(plan, L.ptr(d_halo), U, L.ptr(back), int(back.numel()), self._comm(),
                                                 L.stream_ptr(), ctypes.c_void_p(self.comm_stream.cuda_stream)),
                "tfgx_halo_reverse_start")
        return [plan, back, d_table]

    def reverse_start_round(self, sg, d_table, j, handle=None):
        """Round j (0, 1, ..., R - 1 in order) of the reverse exchange, posted after whatever wrote round j's rows of
        d_table[n_own:] on the current stream — the caller computes the transposed pass window by window.  -> handle (pass it
        to the next round and to reverse_finish)."""
        if sg.rounds == 0 or sg.n_halo + sum(sg.send_counts) == 0:
            return None
        plan = self._plans(sg)[2]
        U = int(d_table.shape[1])
        if handle is None:
            assert d_table.is_contiguous()
            back = torch.empty((max(sg._xrows_sent, 1), U), dtype=torch.float32, device=self.device)
            back.record_stream(self.comm_stream)
            d_table.record_stream(self.comm_stream)
            handle = [plan, back, d_table]
        d_halo = d_table[sg.n_own:]
        _dcheck(self.lib.tfgx_halo_reverse_start_round(plan, int(j), L.ptr(d_halo), U, L.ptr(handle[1]),
                                                       int(handle[1].numel()), self._comm(), L.stream_ptr(),
                                                       ctypes.c_void_p(self.comm_stream.cuda_stream)),
                "tfgx_halo_reverse_start_round")
        return handle

    def reverse_finish(self, sg, handle, d_own):
        if handle is None:
            return d_own
        _, ldd = L.row_major_2d(d_own)
        _dcheck(self.lib.tfgx_halo_reverse_finish(handle[0], L.ptr(d_own), ldd, int(d_own.shape[1]), L.ptr(handle[1]),
                                                  L.stream_ptr()), "tfgx_halo_reverse_finish")
        return d_own


class TorchDistTransport(object):
    """torch.distributed collectives; tensors are staged through the host when the group's backend is gloo (CPU tests, or
    two ranks sharing one GPU).  Dense (round, peer) entries are expanded to index lists on this path."""
    name = "torch"

    def __init__(self, group=None, backend=None):
        self.group = group
        inited = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if inited else 1
        self.rank = dist.get_rank(group) if inited else 0
        self.nccl = inited and dist.get_backend(group) == "nccl"
        self.be = backend

    def _stage(self, t):
        return t if self.nccl else t.cpu()

    def all_reduce_sum_i64(self, t):
        if self.world > 1:
            h = self._stage(t)
            dist.all_reduce(h, group=self.group)
            if h is not t:
                t.copy_(h)
        return t

    def all_to_all_v(self, send, send_counts, recv_counts):
        send = send.contiguous()
        shape = (int(sum(recv_counts)),) + tuple(send.shape[1:])
        if self.world == 1:
            return send.clone()
        s = self._stage(send)
        r = torch.empty(shape, dtype=send.dtype, device=s.device)
        dist.all_to_all_single(r, s, [int(v) for v in recv_counts], [int(v) for v in send_counts], group=self.group)
        return r.to(send.device)

    def all_reduce_sum_f32(self, flat):
        if self.world > 1:
            h = self._stage(flat)
            dist.all_reduce(h, group=self.group)
            if h is not flat:
                flat.copy_(h)
        return flat

    def _full_idx(self, sg, j):
        """Round j's send list with dense entries expanded (this path packs everything)."""
        cache = sg.__dict__.setdefault("_full_send_idx", {})
        if j not in cache:
            parts, off = [], 0
            packed = sg.round_send_idx[j]
            for p in range(sg.world):
                cnt, ds = int(sg.round_send_counts[j][p]), int(sg.round_send_dense[j][p])
                if ds >= 0 and cnt:
                    parts.append(torch.arange(ds, ds + cnt, dtype=torch.int32, device=packed.device))
                elif cnt:
                    parts.append(packed[off:off + cnt])
                    off += cnt
            cache[j] = torch.cat(parts) if parts else packed[:0]
        return cache[j]

    def exchange_start(self, sg, table):
        if sg.rounds == 0:
            return None
        be = sg.backend
        handles = []
        for j in range(sg.rounds):
            send = be.gather_rows(sg.own_rows(table), self._full_idx(sg, j))
            halo = table[sg.n_own + int(sg.round_offset[j]):sg.n_own + int(sg.round_offset[j + 1])]
            out_splits, in_splits = list(sg.round_recv_counts[j]), list(sg.round_send_counts[j])
            if self.world == 1:                                   # self-halo test mode: the rank is its own peer
                handles.append(["host", send, halo])
            elif self.nccl:
                work = dist.all_to_all_single(halo, send, out_splits, in_splits, group=self.group, async_op=True)
                handles.append(["nccl", work, send])
            else:
                recv_h = torch.empty((int(halo.shape[0]), int(table.shape[1])), dtype=torch.float32)
                dist.all_to_all_single(recv_h, send.cpu(), out_splits, in_splits, group=self.group)
                handles.append(["host", recv_h, halo])
        return handles

    def exchange_finish(self, sg, handles, j=None):
        if handles is None:
            return
        for h in (handles if j is None else [handles[j]]):
            if h[0] == "nccl":
                h[1].wait()
            elif h[0] == "host":
                h[2].copy_(h[1])
            h[0] = "done"

    def reverse_start(self, sg, d_table):
        if sg.rounds == 0:
            return None
        backs = None
        for j in range(sg.rounds):
            backs = self.reverse_start_round(sg, d_table, j, backs)
        return backs

    def reverse_start_round(self, sg, d_table, j, handle=None):
        if sg.rounds == 0:
            return None
        be = sg.backend
        U = int(d_table.shape[1])
        d_halo = d_table[sg.n_own:sg.n_table]
        backs = [] if handle is None else handle
        assert len(backs) == j, "reverse rounds are started in order"
        seg = d_halo[int(sg.round_offset[j]):int(sg.round_offset[j + 1])].contiguous()
        n_back = int(sum(sg.round_send_counts[j]))
        out_splits, in_splits = list(sg.round_send_counts[j]), list(sg.round_recv_counts[j])
        if self.world == 1:
            back = seg.clone()
        elif self.nccl:
            back = be.empty((n_back, U))
            dist.all_to_all_single(back, seg, out_splits, in_splits, group=self.group)
        else:
            back_h = torch.empty((n_back, U), dtype=torch.float32)
            dist.all_to_all_single(back_h, seg.cpu(), out_splits, in_splits, group=self.group)
            back = be.empty((n_back, U))
            back.copy_(back_h)
        backs.append(back)
        return backs

    def reverse_finish(self, sg, backs, d_own):
        if backs is None:
            return d_own
        be = sg.backend
        for j, back in enumerate(backs):
            idx, off = self._full_idx(sg, j), 0
            for p in range(sg.world):                 # fixed peer order: one writer per element and call
                cnt = int(sg.round_send_counts[j][p])
                if cnt:
                    be.scatter_add_rows(d_own, idx[off:off + cnt], back[off:off + cnt])
                off += cnt
        return d_own


_TRANSPORTS = {}


def get_transport(group, backend, kind=None):
    """One transport (one communicator, one communication stream) per (group, kind).

    kind "tfgx_dist": the C-ABI RCCL transport, STRICT — if any rank cannot bring it up, every rank raises
                      TfgxDistUnavailable (bench.py and the examples ask for it by name: a scaling line must never be
                      carried by something else).  Works over any control group (nccl or gloo).
    kind "torch":     torch.distributed collectives (host-staged on a gloo group) — CPU tests, two ranks on one GPU.
    kind None:        TFGX_DIST_TRANSPORT if set (strict, as above); otherwise AUTO: tfgx_dist for the HIP backend on
                      an nccl group — falling back, on every rank together and with a warning, to torch collectives if
                      the ranks agree that it is unavailable — and torch for gloo groups / the numpy test backend."""
    kind = kind or os.environ.get("TFGX_DIST_TRANSPORT")
    auto = kind is None
    inited = dist.is_available() and dist.is_initialized()
    if kind is None:
        hip = getattr(backend, "name", "") == "hip"
        gloo = inited and dist.get_backend(group) != "nccl"
        kind = "tfgx_dist" if (hip and not gloo) else "torch"
    if kind not in ("tfgx_dist", "torch"):
        raise L.TfgxError("unknown transport kind {!r} (tfgx_dist | torch)".format(kind))
    key = (id(group) if group is not None else 0, kind, inited)
    t = _TRANSPORTS.get(key)
    if t is None:
        multi = inited and dist.get_world_size(group) > 1
        if kind == "torch":
            t = TorchDistTransport(group, backend)
        elif not multi:
            t = TfgxDistTransport(group)
        else:
            t = _bring_up_tfgx(group, backend, auto)
        _TRANSPORTS[key] = t
    return t


def _bring_up_tfgx(group, backend, auto):
    """The C-ABI transport of a multi-rank group: bootstrap (the ranks agree before and after every collective step),
    then a few rows through every plan-time entry point, agreed again.  Strict (auto=False): TfgxDistUnavailable on every
    rank when any rank failed.  AUTO: every rank takes _torch_fallback instead."""
    t = None
    try:
        t = TfgxDistTransport(group)
        t.checked_self_check()
        return t
    except TfgxDistUnavailable as ex:            # raised on every rank (see TfgxDistTransport._agree)
        # a communicator that came up but failed its self-check must not linger beside whatever carries the rows instead
        # (strict mode: beside nothing — the caller exits): every rank destroys it, best effort, before going on
        if t is not None:
            try:
                t.close(abort=True)
            except Exception:                    # noqa: BLE001 - the communicator is already suspect
                pass
        if not auto:
            raise
        return _torch_fallback(group, backend, str(ex))


def _torch_fallback(group, backend, reason):
    """AUTO mode only, all ranks together: torch.distributed's collectives on the same group — same exchange lists, same
    kernels either side — with a warning and a transport name that says so."""
    import warnings
    msg = "tf_geometric_amd.dist: {}; all ranks fall back to torch.distributed collectives on the same group " \
          "(pass transport='tfgx_dist' or set TFGX_DIST_TRANSPORT=tfgx_dist to fail instead)".format(reason)
    warnings.warn(msg)
    tt = TorchDistTransport(group, backend)
    tt.name = "torch (fallback: tfgx_dist unavailable)"
    tt.fallback_reason = msg
    return tt


def close_transports():
    for t in list(_TRANSPORTS.values()):
        if hasattr(t, "close"):
            t.close()
    _TRANSPORTS.clear()
