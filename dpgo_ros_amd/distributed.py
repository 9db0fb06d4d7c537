"""One process per GPU: the synchronous RBCD schedule across ranks with RCCL point-to-point.

Replaces the ROS transport of the reference for this path:
  * `PublicPoses` messages (msg/PublicPoses.msg, src/PGOAgentROS.cpp:662-690 send, :1255-1284 receive)
    become packed r x 4 fp64 slabs sent with `torch.distributed` isend/irecv (backend "nccl" = RCCL
    over xGMI; "gloo" in the CPU tests);
  * the UPDATE token (src/PGOAgentROS.cpp:443-504, 1161-1189) needs no message: the schedule is a
    deterministic function of the global iteration counter that every rank evaluates locally.
Exchange is pull-before-use: right before agent `sel` optimizes, each of its neighbours' owners
sends the neighbour's current public poses (X, and the auxiliary Y sequence under acceleration);
this equals the reference's staleness gate with maxDelayedIterations = 0 (:136-149).

The module is transport + schedule only.  The compute backend is any object with
    iterate(agent, do_opt), pack(agent, nbr, seqs, count) -> tensor, recv_buffer(agent, nbr, seqs, count),
    unpack(agent, nbr, seqs, tensor), pull_local(agent), partial_cost() -> float
(`HipBackend` below for the product; the CPU tests plug the oracle in through the same protocol).
"""
import numpy as np


def topology(meas, num_robots):
    """neighbour sets and public-pose counts, derived from the (partitioned) measurement list that
    every rank holds.  npub[(a, b)] = number of distinct poses of a that appear in edges with b."""
    nbrs = {a: set() for a in range(num_robots)}
    pub = {}
    for e in meas:
        a, b = int(e["r1"]), int(e["r2"])
        if a == b:
            continue
        nbrs[a].add(b)
        nbrs[b].add(a)
        pub.setdefault((a, b), set()).add(int(e["p1"]))
        pub.setdefault((b, a), set()).add(int(e["p2"]))
    return {a: sorted(s) for a, s in nbrs.items()}, {k: len(v) for k, v in pub.items()}


def owner_of(agent, world):
    return agent % world


def greedy_coloring(nbrs, num_robots):
    """colour classes of the agent graph (id order, smallest free colour): agents of a class share no edge"""
    color = {}
    for a in range(num_robots):
        used = {color[b] for b in nbrs[a] if b in color}
        c = 0
        while c in used:
            c += 1
        color[a] = c
    ncol = 1 + max(color.values())
    return [[a for a in range(num_robots) if color[a] == c] for c in range(ncol)]


class HipBackend:
    """local agents on this rank's GPU, through the C-ABI (dpgo_ros_amd.capi)."""

    def __init__(self, meas, params, local_ids, device, torch_module, host_staging=False):
        """host_staging: bounce the slabs through pinned host tensors (for the `gloo` transport, which
        cannot move device tensors; used by the 2-process single-GPU test).  RCCL runs use False."""
        from . import capi
        self.torch = torch_module
        self.r = params.r
        self.device = device
        self.host_staging = host_staging
        # one explicit (non-default) stream for kernels AND collectives: torch.distributed orders its RCCL
        # work against the *current* torch stream, so every launch of this rank runs under stream_context()
        self.stream = torch_module.cuda.Stream(device=device)
        stream = self.stream.cuda_stream
        self.team = capi.Team.from_measurements(meas, params, device=device, local_ids=local_ids, stream=stream) \
            if local_ids else None
        self._buf = {}

    def stream_context(self):
        return self.torch.cuda.stream(self.stream)

    def buffer(self, key, count):
        if key not in self._buf:
            self._buf[key] = self.torch.empty(count * 4 * self.r, dtype=self.torch.float64, device="cuda")
        return self._buf[key]

    def iterate(self, agent, do_opt):
        return self.team.agents[agent].iterate(do_opt)

    def pack(self, agent, nbr, seqs, count):
        """one message per neighbour: the public poses of every requested sequence (0 = X, 1 = auxiliary Y)
        packed back to back by the device kernels"""
        n = count * 4 * self.r
        t = self.buffer(("s", agent, nbr, len(seqs)), count * len(seqs))
        for q, aux in enumerate(seqs):
            self.team.agents[agent].pack_public_poses_device(nbr, aux, t.data_ptr() + 8 * n * q)
        if self.host_staging:
            return t.cpu()  # synchronises the (current) stream the pack kernels ran on
        return t

    def recv_buffer(self, agent, nbr, seqs, count):
        if self.host_staging:
            key = ("rh", agent, nbr, len(seqs))
            if key not in self._buf:
                self._buf[key] = self.torch.empty(count * len(seqs) * 4 * self.r, dtype=self.torch.float64)
            return self._buf[key]
        return self.buffer(("r", agent, nbr, len(seqs)), count * len(seqs))

    def unpack(self, agent, nbr, seqs, tensor):
        n = tensor.numel() // len(seqs)
        if self.host_staging:
            d = self.buffer(("r", agent, nbr, len(seqs)), tensor.numel() // (4 * self.r))
            d.copy_(tensor)
            tensor = d
        for q, aux in enumerate(seqs):
            self.team.agents[agent].unpack_neighbor_poses_device(nbr, aux, tensor.data_ptr() + 8 * n * q)

    def pull_local(self, agent):
        self.team.agents[agent].pull_local()

    # batched form of the iteration (one launch for all local agents instead of one call per agent)
    def set_groups(self, groups):
        if self.team is not None:
            self.team.set_groups(groups)

    def run_group(self, g, members):
        if self.team is not None:
            self.team.run_group(g, len(members))

    def step_begin(self, sel):
        if self.team is not None:
            self.team.step_begin(sel)

    def tick_local(self):
        """one simultaneous RGD step of every local agent from the neighbour poses as of now (ASAPP lockstep tick)"""
        if self.team is not None:
            self.team.run_simultaneous(1)

    # ---- peer access: neighbours in other processes read in place (HIP IPC; xGMI loads between GPUs)
    def export_states(self):
        """{agent: (ipc handle, offset of X, offset of Y, poses)} of the local agents"""
        if self.team is None:
            return {}
        return {a: self.team.export_state(a) for a in self.team.agents}

    def import_peer(self, robot, state):
        self.team.import_peer(robot, *state)

    # ---- the UPDATE token on the device (dpgo_team_run_peer): mailboxes written over peer access
    def export_mailbox(self):
        return self.team.export_mailbox() if self.team is not None else None

    def import_mailbox(self, handle, robots):
        self.team.import_mailbox(handle, robots)

    def run_peer(self, sel_ids):
        if self.team is not None:
            self.team.run_peer(sel_ids)

    # ---- the exchange carried by RCCL from inside the library (csrc/rank_exchange.cpp): K iterations per host call
    def attach_comm(self, comm, owner_of_robot, max_delayed_iterations=0):
        if self.team is not None:
            self.team.attach_comm(comm, owner_of_robot, max_delayed_iterations)
            self.team.exchange_all_ranks()

    def run_ranks(self, sel_ids):
        if self.team is not None:
            self.team.run_ranks(sel_ids)

    def run_simultaneous_ranks(self, ticks):
        if self.team is not None:
            self.team.run_simultaneous_ranks(ticks)

    def run_group_ranks(self, g, members):
        if self.team is not None:
            self.team.run_group_ranks(g, len(members))

    def sync(self):
        """drain this rank's stream -- through the team, so that a time-out of an in-kernel exchange (mailbox wait of
        the device-side token, two-level preconditioner) is raised here instead of yielding silently wrong iterates"""
        if self.team is not None:
            self.team.synchronize()
        else:
            self.stream.synchronize()

    def free_run(self, ticks):
        """`ticks` steps of every local agent back to back, each from whatever the neighbours' arrays hold when its
        kernels read them -- no message, no rendezvous with the other ranks (the asynchronous mode proper)"""
        if self.team is not None:
            self.team.run_simultaneous(ticks)

    # ---- robust path across ranks (src/PGOAgentROS.cpp:721-754 publishMeasurementWeights, :1315-1353 callback)
    def local_update_weights(self):
        """every local agent re-weights what it owns; co-resident endpoints are served in the same call"""
        return self.team.update_weights() if self.team is not None else 0

    def owned_weights(self, agent, nbr):
        """(weights, fixed flags) of the shared edges agent <-> nbr in the agent's stored order, as a float64 array
        [w0, f0, w1, f1, ...] (fp32-rounded on request: msg/RelativeMeasurementWeights.msg:8)"""
        ms = self.team.agents[agent].measurements()
        sel = ms[(ms["r1"] != ms["r2"]) & ((ms["r1"] == nbr) | (ms["r2"] == nbr))]
        w = sel["weight"].astype(np.float64)
        if self.team.params.weights_as_float32:
            w = w.astype(np.float32).astype(np.float64)
        out = np.empty(2 * len(sel))
        out[0::2], out[1::2] = w, sel["fixed_weight"]
        return out, sel

    def apply_weights(self, agent, nbr, payload):
        ms = self.team.agents[agent].measurements()
        sel = ms[(ms["r1"] != ms["r2"]) & ((ms["r1"] == nbr) | (ms["r2"] == nbr))]
        changed = 0
        for e, w, f in zip(sel, payload[0::2], payload[1::2]):
            self.team.agents[agent].set_measurement_weight(int(e["r1"]), int(e["p1"]), int(e["r2"]), int(e["p2"]), float(w), bool(f))
            changed += 1
        self.team.agents[agent].clear_data_matrices()
        return changed

    def to_transport(self, arr):
        t = self.torch.from_numpy(np.ascontiguousarray(arr))
        return t if self.host_staging else t.cuda()

    def from_transport(self, t):
        return t.cpu().numpy()

    def transport_empty(self, n):
        return self.torch.empty(n, dtype=self.torch.float64, device="cpu" if self.host_staging else "cuda")

    def step_end(self, sel):
        if self.team is not None:
            self.team.step_end(sel)

    def partial_cost(self):
        return self.team.cost() if self.team is not None else 0.0

    def close(self):
        if self.team is not None:
            self.team.close()


class DistributedRBCD:
    def __init__(self, dist, backend, meas, num_robots, acceleration, rank, world, schedule=None, max_delayed_iterations=0):
        """max_delayed_iterations: the staleness gate of src/PGOAgentROS.cpp:136-149 (struct default 3,
        include/dpgo_ros/PGOAgentROS.h:83; the demos set 0).  Robot `sel` may optimize with a neighbour's public poses
        from iteration >= required - max_delayed_iterations, `required` being the neighbour's latest update (every
        iteration under acceleration).  On this lossless transport the gate decides which messages are SENT: a remote
        neighbour b sends to sel only when the copy sel holds is more than max_delayed_iterations behind b's latest
        change -- and never when b has not changed since it last sent (which already removes the redundant messages of
        the plain schedule at 0).  Co-resident neighbours are always read fresh (device-to-device, free)."""
        self.dist, self.be = dist, backend
        self.N, self.rank, self.world = num_robots, rank, world
        self.accel = bool(acceleration)
        self.nbrs, self.npub = topology(meas, num_robots)
        self.owner = [owner_of(a, world) for a in range(num_robots)]
        self.mine = [a for a in range(num_robots) if self.owner[a] == rank]
        self.schedule = list(range(num_robots)) if schedule is None else list(schedule)
        self.k = 0
        self.max_delay = int(max_delayed_iterations)
        self.version = [0] * num_robots   # iteration at which each agent's public poses last changed
        self.sent = {}                    # (b, sel, sequence) -> version of b's X (0) / Y (1) that sel's rank holds
        self.messages = 0                 # point-to-point operations issued by this rank (for the tests / bench)
        self._imported, self.peer_access, self.peer_error = set(), False, None
        self._mail_imported = set()
        # shared-edge counts per ordered pair, for the weight messages of the robust path
        self.nshared = {}
        for e in meas:
            a, b = int(e["r1"]), int(e["r2"])
            if a != b:
                self.nshared[(a, b)] = self.nshared.get((a, b), 0) + 1
                self.nshared[(b, a)] = self.nshared.get((b, a), 0) + 1

    def _ctx(self):
        import contextlib
        return self.be.stream_context() if hasattr(self.be, "stream_context") else contextlib.nullcontext()

    def exchange_to(self, sel, seqs=(0, 1)):
        """neighbours of `sel` that live on other ranks send their public poses to sel's rank."""
        with self._ctx():
            self._exchange_to(sel, seqs)

    def _exchange_to(self, sel, seqs, pull=True, force=False):
        d = self.dist
        ops, todo = [], []
        rs = self.owner[sel]
        for b in self.nbrs[sel]:
            rb = self.owner[b]
            if rb == rs:
                continue
            # the gate is kept PER SEQUENCE (0 = X, 1 = auxiliary Y): a copy of X that is fresh enough says nothing about
            # Y -- an X-only exchange (colour sweep, lockstep tick) followed by an accelerated step must still deliver Y
            held = [self.sent.get((b, sel, q)) for q in seqs]
            if not force and all(h is not None for h in held):
                behind = max(self.version[b] - h for h in held)
                if behind == 0 or behind <= self.max_delay:
                    continue  # the copies on sel's rank are current, or fresh enough for the staleness gate
            for q in seqs:
                self.sent[(b, sel, q)] = self.version[b]
            cnt = self.npub[(b, sel)]
            if self.rank == rb:
                ops.append(d.P2POp(d.isend, self.be.pack(b, sel, seqs, cnt), rs))
            if self.rank == rs:
                t = self.be.recv_buffer(sel, b, seqs, cnt)
                ops.append(d.P2POp(d.irecv, t, rb))
                todo.append((b, t))
        if ops:
            self.messages += len(ops)
            for w in d.batch_isend_irecv(ops):
                w.wait()
        for b, t in todo:
            self.be.unpack(sel, b, seqs, t)
        if self.rank == rs and pull:  # the batched step pulls co-resident poses inside the G-assembly kernel
            self.be.pull_local(sel)

    def exchange_all(self):
        with self._ctx():
            for a in range(self.N):
                self._exchange_to(a, (0, 1), force=True)

    def step(self):
        """one global RBCD iteration (src/PGOAgentROS.cpp:129-220): everyone but the token holder
        calls iterate(false) first, then the token holder receives its neighbours' poses and optimizes."""
        sel = self.schedule[self.k % len(self.schedule)]
        if self.accel:  # iterate(false) moves X and Y of everyone else BEFORE the token holder pulls them
            self.version = [self.k + 1 if a != sel else self.version[a] for a in range(self.N)]
        with self._ctx():
            if hasattr(self.be, "step_begin"):
                self.be.step_begin(sel)
                self._exchange_to(sel, (0, 1) if self.accel else (0,), pull=False)
                self.be.step_end(sel)
            else:
                for a in self.mine:
                    if a != sel:
                        self.be.iterate(a, False)
                self._exchange_to(sel, (0, 1) if self.accel else (0,))
                if self.owner[sel] == self.rank:
                    self.be.iterate(sel, True)
        self.k += 1
        self.version[sel] = self.k
        return sel

    def tick_simultaneous(self):
        """BASELINE configs[4] across ranks: the deterministic lockstep instance of the asynchronous (ASAPP) mode
        (src/PGOAgentROS.cpp:119-127).  Every agent takes ONE preconditioned RGD step from the neighbour poses of the
        tick's start: all boundary slabs cross the ranks in one batch of point-to-point operations, then every rank
        steps all its agents in the same launches.  Equals dpgo_team_run_simultaneous on one GPU."""
        d = self.dist
        with self._ctx():
            ops, todo = [], []
            for a in range(self.N):  # fixed global order: both ends of a pair enumerate it identically
                for b in self.nbrs[a]:
                    ra, rb = self.owner[a], self.owner[b]
                    if ra == rb:
                        continue
                    cnt = self.npub[(b, a)]  # poses of b that a needs
                    if self.rank == rb:
                        ops.append(d.P2POp(d.isend, self.be.pack(b, a, (0,), cnt), ra))
                    if self.rank == ra:
                        t = self.be.recv_buffer(a, b, (0,), cnt)
                        ops.append(d.P2POp(d.irecv, t, rb))
                        todo.append((a, b, t))
            if ops:
                for w in d.batch_isend_irecv(ops):
                    w.wait()
            for a, b, t in todo:
                self.be.unpack(a, b, (0,), t)
            self.be.tick_local()
        self.k += self.N
        self.version = [self.k] * self.N

    def enable_peer_access(self):
        """Every rank exports the X / Y arrays of its agents and imports those of its agents' remote neighbours
        (dpgo_agent_export_state / dpgo_team_import_peer): from here on remote public poses are read in place, like
        co-resident ones.  Collective (all_gather of the 64-byte handles, then of the outcome); call after the team is
        built.  Returns whether EVERY rank succeeded (peer_error holds the first failure) so that all ranks take the same
        branch afterwards."""
        d = self.dist
        err = None
        mail = None
        try:
            mine = self.be.export_states() if hasattr(self.be, "export_states") else {}
            if not hasattr(self.be, "export_states"):
                err = "backend has no peer access"
            elif hasattr(self.be, "export_mailbox"):
                mail = self.be.export_mailbox()
        except RuntimeError as e:  # agree on the outcome before anyone enters another collective
            mine, err = {}, str(e)
        everyone = [None] * self.world
        d.all_gather_object(everyone, (mine, mail))
        states = {}
        for part, _ in everyone:
            states.update(part)
        if err is None and getattr(self.be, "team", None) is not None:
            try:
                for a in self.mine:
                    for b in self.nbrs[a]:
                        if self.owner[b] != self.rank and b not in self._imported:
                            self.be.import_peer(b, states[b])
                            self._imported.add(b)
                # the mailbox of every rank that holds a neighbour of a local robot (the device-side UPDATE token)
                ranks = sorted({self.owner[b] for a in self.mine for b in self.nbrs[a] if self.owner[b] != self.rank})
                for rk in ranks:
                    if rk in self._mail_imported:
                        continue
                    if everyone[rk][1] is not None and hasattr(self.be, "import_mailbox"):
                        self.be.import_mailbox(everyone[rk][1], [b for b in range(self.N) if self.owner[b] == rk])
                        self._mail_imported.add(rk)
            except (RuntimeError, KeyError) as e:
                err = str(e)
        errs = [None] * self.world
        d.all_gather_object(errs, err)
        self.peer_access = all(e is None for e in errs)
        self.peer_error = next((e for e in errs if e is not None), None)
        return self.peer_access

    def step_peer(self):
        """step() with the neighbours read in place instead of sent: the point-to-point messages become rendezvous
        (stream drained + barrier) -- one before the token holder reads, and under acceleration one before everyone
        moves its Y.  Same iterates as step(); needs enable_peer_access."""
        assert self.peer_access, "step_peer needs enable_peer_access()"
        sel = self.schedule[self.k % len(self.schedule)]
        with self._ctx():
            if self.accel:
                self.be.sync()
                self.dist.barrier()
            self.be.step_begin(sel)
            self.be.sync()
            self.dist.barrier()
            self.be.step_end(sel)
        self.k += 1
        return sel

    def run_peer(self, iters):
        """`iters` iterations of the synchronous schedule with NO host in the loop: neighbours in other processes are
        read in place and the UPDATE token lives on the device (dpgo_team_run_peer: every rank enqueues the same
        schedule once; wait / signal kernels around the launches that read a peer or overwrite what a peer was
        reading order the ranks through mailboxes written over peer access).  Same iterates as step(); returns
        without synchronising.  Needs enable_peer_access."""
        assert self.peer_access, "run_peer needs enable_peer_access()"
        sels = [self.schedule[(self.k + q) % len(self.schedule)] for q in range(iters)]
        with self._ctx():
            self.be.run_peer(sels)
        self.k += iters
        self.version = [self.k] * self.N  # (conservative: a later step() sends everything once)
        self.sent = {}
        return sels

    def enable_library_exchange(self, comm):
        """hand the exchange to the library: from here on run_library() enqueues whole iterations -- the public-pose slabs
        by ncclSend / ncclRecv on the team stream, the staleness gate evaluated in the library -- K per host call.  `comm`:
        capi.Comm created collectively by every rank (also those without robots)."""
        with self._ctx():
            self.be.attach_comm(comm, self.owner, self.max_delay)
        self.library_exchange = True

    def run_library(self, iters):
        """`iters` iterations of the synchronous schedule with NO host language in the loop (dpgo_team_run_ranks): same
        iterates as step(), bit for bit (tests/test_gpu_rank_exchange.py).  Returns without synchronising."""
        assert getattr(self, "library_exchange", False), "run_library needs enable_library_exchange()"
        sels = [self.schedule[(self.k + q) % len(self.schedule)] for q in range(iters)]
        with self._ctx():
            self.be.run_ranks(sels)
        self.k += iters
        self.version = [self.k] * self.N  # (conservative: a later step() sends everything once)
        self.sent = {}
        return sels

    def tick_library(self, ticks=1):
        """tick_simultaneous() with the slabs moved by the library (dpgo_team_run_simultaneous_ranks): `ticks` lockstep
        ticks per host call, one batch of ncclSend / ncclRecv each"""
        assert getattr(self, "library_exchange", False), "tick_library needs enable_library_exchange()"
        with self._ctx():
            self.be.run_simultaneous_ranks(ticks)
        self.k += self.N * ticks
        self.version = [self.k] * self.N
        self.sent = {}

    def sweep_colored_library(self):
        """sweep_colored() with the slabs moved by the library (dpgo_team_run_group_ranks): no host-driven message"""
        assert getattr(self, "library_exchange", False), "sweep_colored_library needs enable_library_exchange()"
        if not hasattr(self, "groups"):
            self.groups = greedy_coloring(self.nbrs, self.N)
            self.be.set_groups(self.groups)
        with self._ctx():
            for g, members in enumerate(self.groups):
                self.be.run_group_ranks(g, members)
                self.k += len(members)
        self.version = [self.k] * self.N
        self.sent = {}

    def free_run(self, ticks):
        """The asynchronous (ASAPP) mode across ranks, src/PGOAgentROS.cpp:119-127: every rank steps its agents `ticks`
        times at its own pace, reading the other ranks' public poses in place (needs enable_peer_access).  No
        collective, no barrier: which iterate of a neighbour a step sees depends on timing, as in the reference's
        free-running optimization threads; the result is therefore not reproducible bit for bit -- the tests check
        what the asynchronous mode promises (the cost goes down to the synchronous answer's neighbourhood)."""
        assert self.peer_access, "free_run needs enable_peer_access()"
        with self._ctx():
            self.be.free_run(ticks)
        self.k += self.N * ticks
        self.version = [self.k] * self.N

    def update_weights(self):
        """UPDATE_WEIGHT round across ranks (src/PGOAgentROS.cpp:1211-1233): every agent re-weights the edges it owns
        (lower-ID endpoint), the weights of shared edges travel to the higher-ID endpoint's rank
        (publishMeasurementWeights :721-754 -> measurementWeightsCallback :1315-1353), which applies them and clears its
        data matrices; public poses are exchanged afterwards (:1224-1226).  Returns the number of weights applied here."""
        d = self.dist
        changed = 0
        with self._ctx():
            self.exchange_all_nolock()
            changed += self.be.local_update_weights()
            ops, todo, keep = [], [], []
            for a in range(self.N):
                for b in self.nbrs[a]:
                    if b < a or self.owner[a] == self.owner[b]:
                        continue  # a owns the weights of its edges with the higher-ID b; co-resident pairs are done
                    cnt = self.nshared[(a, b)]
                    if self.rank == self.owner[a]:
                        payload, _ = self.be.owned_weights(a, b)
                        t = self.be.to_transport(payload)
                        keep.append(t)
                        ops.append(d.P2POp(d.isend, t, self.owner[b]))
                    if self.rank == self.owner[b]:
                        t = self.be.transport_empty(2 * cnt)
                        ops.append(d.P2POp(d.irecv, t, self.owner[a]))
                        todo.append((b, a, t))
            if ops:
                for w in d.batch_isend_irecv(ops):
                    w.wait()
            for b, a, t in todo:
                changed += self.be.apply_weights(b, a, self.be.from_transport(t))
            self.exchange_all_nolock()
        return changed

    def exchange_all_nolock(self):
        for a in range(self.N):
            self._exchange_to(a, (0, 1), force=True)

    def sweep_colored(self):
        """one colour-parallel sweep of plain (non-accelerated) RBCD: for each colour class, every member
        receives its neighbours' public poses, then all members -- on whatever ranks they live -- take
        their block update concurrently.  Equals the sequential schedule [class 0 ..., class 1 ...]."""
        if not hasattr(self, "groups"):
            self.groups = greedy_coloring(self.nbrs, self.N)
            self.be.set_groups(self.groups)
        with self._ctx():
            for g, members in enumerate(self.groups):
                for a in members:
                    self._exchange_to(a, (0,))
                self.be.run_group(g, members)
                self.k += len(members)
                for a in members:
                    self.version[a] = self.k

    def global_cost(self, torch_module, device):
        """f of the concatenated iterate: owned-edge partial sums, one 1-double all-reduce."""
        self.exchange_all()
        with self._ctx():
            t = torch_module.tensor([self.be.partial_cost()], dtype=torch_module.float64, device=device)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
            return float(t.item())
