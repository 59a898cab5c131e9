"""Test infrastructure: play the W ranks of a frame-sharded job ONE AFTER ANOTHER in one process (on one device).

The GPU boxes of this environment have a single MI355X and RCCL wants one device per rank, so BASELINE configs 4 and 5 (1800 frames,
8 contiguous shards of 225: reference render.py:140-182 slicing, the DataParallel role of generate_audiovisual.py:54-55) cannot run as
eight processes there.  ``PlayedWorld`` replaces ``sharding.dist`` / ``sharding.rank_world`` / ``sharding.grouped`` by an in-process
stand-in with record / replay semantics, so that the product's multi-rank code (generate(): flat weight broadcast, object broadcasts,
scatter of the per-frame inputs; render_shard(): one asynchronous gather per batch-round into rank 0's HBM store, pinned ring, sink
thread) runs unmodified for every rank:

  pass 1   rank 0   its outgoing collectives (broadcast tensors / objects, scatter chunks) are RECORDED in call order; gathers fill its
                    own slot only (its sink's output is thrown away by the caller);
  pass 2   ranks 1 .. W-1, each: incoming collectives are REPLAYED from rank 0's record in the same call order (a mismatch in kind /
                    shape / order between the ranks' collective sequences raises — the property a real job needs not to dead-lock);
                    every gather's contribution is recorded;
  pass 3   rank 0 again, from the same seeds: its outgoing collectives must equal pass 1's bit for bit (determinism of the front end),
                    and every gather now receives the peers' recorded rounds, so its sink delivers the WHOLE video in global order.

Only the transport is stubbed; which rank does what, in which order, on which streams is the product's.
"""
import contextlib

import torch


class _Work:
    """Stand-in for the Work handle of an asynchronous collective: completion = an event behind the copies (device) or immediate."""

    def __init__(self, event=None):
        self._event = event

    def wait(self):
        if self._event is not None:
            self._event.synchronize()
        return True

    def is_completed(self):
        return self._event is None or self._event.query()


class PlayedWorld:
    def __init__(self, world):
        self.world = int(world)
        self.rank = None
        self.sent = []  # rank 0's outgoing collectives in call order: (kind, payload)
        self.sent_before = None  # pass 1's record while pass 3 re-records
        self.cursor = 0  # replay position of the rank being played
        self.rounds = {}  # peer rank -> list of gathered tensors in call order
        self.gathers = 0  # gather calls of the rank being played
        self.final = False
        self.stats = {"broadcast": 0, "broadcast_object_list": 0, "scatter": 0, "gather": 0, "barrier": 0, "bytes_gathered": 0}

    # ---- the torch.distributed surface sharding.py uses -------------------------------------------------------------------------
    def _out(self, kind, payload):
        if self.final:  # pass 3: must repeat pass 1
            want_kind, want = self.sent_before[len(self.sent)]
            assert want_kind == kind, f"rank 0's collective #{len(self.sent)} was {want_kind} in pass 1 and is {kind} now"
            self._assert_same(want, payload, f"rank 0's {kind} #{len(self.sent)} differs between its two passes")
        self.sent.append((kind, payload))

    def _in(self, kind):
        assert self.cursor < len(self.sent), f"rank {self.rank} issues a {kind} rank 0 never issued (collective #{self.cursor})"
        got_kind, payload = self.sent[self.cursor]
        assert got_kind == kind, f"collective #{self.cursor}: rank 0 issued {got_kind}, rank {self.rank} issues {kind}"
        self.cursor += 1
        return payload

    @staticmethod
    def _assert_same(a, b, what):
        if isinstance(a, torch.Tensor):
            assert a.shape == b.shape and torch.equal(a, b), what
        elif isinstance(a, (list, tuple)) and a and isinstance(a[0], torch.Tensor):
            assert len(a) == len(b) and all(torch.equal(x, y) for x, y in zip(a, b)), what
        else:
            assert a == b, what

    def broadcast(self, tensor, src=0, group=None, async_op=False):
        assert src == 0
        self.stats["broadcast"] += 1
        if self.rank == 0:
            self._out("broadcast", tensor.detach().clone())
        else:
            payload = self._in("broadcast")
            assert payload.shape == tensor.shape and payload.dtype == tensor.dtype, (payload.shape, tensor.shape)
            tensor.copy_(payload)
        return _Work()

    def broadcast_object_list(self, box, src=0, group=None, device=None):
        assert src == 0
        self.stats["broadcast_object_list"] += 1
        if self.rank == 0:
            self._out("object", list(box))
        else:
            box[:] = self._in("object")

    def scatter(self, tensor, scatter_list=None, src=0, group=None, async_op=False):
        assert src == 0
        self.stats["scatter"] += 1
        if self.rank == 0:
            assert scatter_list is not None and len(scatter_list) == self.world
            assert all(c.shape == tensor.shape for c in scatter_list), "scatter wants equally sized chunks"
            self._out("scatter", [c.detach().clone() for c in scatter_list])
            tensor.copy_(scatter_list[0])
        else:
            assert scatter_list is None
            chunks = self._in("scatter")
            assert chunks[self.rank].shape == tensor.shape, (chunks[self.rank].shape, tensor.shape)
            tensor.copy_(chunks[self.rank])
        return _Work()

    def gather(self, tensor, gather_list=None, dst=0, group=None, async_op=False):
        assert dst == 0
        self.stats["gather"] += 1
        k = self.gathers
        self.gathers += 1
        if self.rank != 0:
            assert gather_list is None
            self.rounds.setdefault(self.rank, []).append(tensor.detach().clone())
            return _Work()
        assert gather_list is not None and len(gather_list) == self.world
        gather_list[0].copy_(tensor)
        if self.final:
            for p in range(1, self.world):
                assert k < len(self.rounds.get(p, [])), f"rank {p} issued {len(self.rounds.get(p, []))} gathers, rank 0 is at #{k}"
                gather_list[p].copy_(self.rounds[p][k])
                self.stats["bytes_gathered"] += self.rounds[p][k].numel() * self.rounds[p][k].element_size()
        event = None
        if tensor.is_cuda:
            event = torch.cuda.Event()
            event.record(torch.cuda.current_stream(tensor.device))
        return _Work(event)

    def barrier(self, group=None, async_op=False):
        self.stats["barrier"] += 1

    def is_available(self):
        return True

    def is_initialized(self):
        return True

    def get_rank(self):
        return self.rank

    def get_world_size(self):
        return self.world

    # ---- driving the passes ---------------------------------------------------------------------------------------------------------
    @contextlib.contextmanager
    def playing(self, rank, final=False):
        """``with world.playing(r): job()`` — the product sees rank ``r`` of ``world`` ranks under an initialised process group."""
        from maua_stylegan2_amd import sharding

        keep = (sharding.dist, sharding.rank_world, sharding.grouped)
        self.rank, self.cursor, self.gathers, self.final = rank, 0, 0, bool(final)
        if rank == 0:
            if final:
                self.sent_before, self.sent = self.sent, []
            else:
                self.sent, self.rounds = [], {}
        else:
            self.rounds[rank] = []
        sharding.dist = self
        sharding.rank_world = lambda: (self.rank, self.world)
        sharding.grouped = lambda: True
        try:
            yield self
            if rank == 0 and final:
                assert len(self.sent) == len(self.sent_before), "rank 0 issued fewer collectives in its second pass"
            if rank != 0:
                assert self.cursor == len(self.sent), (
                    f"rank {rank} consumed {self.cursor} of rank 0's {len(self.sent)} outgoing collectives: the sequences differ")
        finally:
            sharding.dist, sharding.rank_world, sharding.grouped = keep
            self.rank = None

    def play(self, job, reseed=None):
        """Run ``job(rank)`` for rank 0 (recording), ranks 1..W-1, and rank 0 again (delivering); returns the list of the W + 1 results
        (index W = rank 0's final pass).  ``reseed()`` is called before every pass (same random streams for rank 0's two passes)."""
        results = []
        for rank, final in [(0, False)] + [(r, False) for r in range(1, self.world)] + [(0, True)]:
            if reseed is not None:
                reseed()
            with self.playing(rank, final):
                results.append(job(rank, final))
        return results
