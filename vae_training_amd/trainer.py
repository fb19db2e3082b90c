"""Graph-captured training loop: the reference's hot loop (model.py:221-222 -- get_batch, sample_latent,
train_step, append loss) with NOTHING on the host per step.

The launches of a step -- the Philox dataset + latent draw (K7) and the two kernels of vaek_train_step --
are captured G steps at a time into a hipGraph; the Adam step counter, the RNG step and the loss ring
buffer all live on the device, so replays need no arguments.  At the reference's own batch size (100)
the per-step Python of the drop-in path (~50 us) is 3-4x the GPU time; this loop removes it without
changing what a step computes.

`pipeline=True` (default) draws batch n+1 WHILE step n trains (vaek_train_step_gen): the draw does not
depend on the weights, so its work items ride in the finalize launch of step n -- which by itself keeps
9 of 256 CUs busy -- and write the other of two batch buffers; at <= 256 rows the whole step, draw included,
is ONE launch (10 us per step at the reference's batch size 100).  The draw takes its step from a counter pair the
generator advances itself, not from the Adam step counter that same launch is incrementing.  Same
counters, same Philox streams: losses and parameters are bit-identical to `pipeline=False`
(vaek_make_batch, then vaek_train_step).  (A second stream / parallel graph branch for the draw was
measured first: cross-branch edges of a hipGraph cost far more than the 7 us they were meant to hide.)

`moments` (default: whenever the model qualifies -- a linear VAE on the linear_gaussian or sphere dataset): the steps go through
vaek_train_steps_gen, the headline kernel of bench.py (csrc/linear_moments.hip): up to 64 steps per persistent launch, every
step's batch drawn INSIDE the launch by the workgroups that multiply it -- the same Philox counters as above, so the same
batches bit for bit, but no batch buffer, no hipGraph and nothing per step on the host: one library call per run of steps
between two events of the reference's schedule (stats every 5 000, plot + save every 50 000: model.py:213-220).  The losses land
in the same device ring.  Losses and parameters agree with the per-sample loop to summation order (tests/test_gpu_loop.py)."""
from __future__ import annotations

import torch


class GraphLoop:
    def __init__(self, vae_model, steps_per_graph=200, seed=None, loss_capacity=1 << 20, pipeline=True, moments=None):
        m = vae_model
        self.m = m
        ds = m.dataset
        self.kind, self.A, self.dd, self.did, self.pad, self.var = ds.device_spec()
        from .datasets import DEVICE_DRAW_MAX_DIM
        if self.dd > DEVICE_DRAW_MAX_DIM or self.did > DEVICE_DRAW_MAX_DIM:
            raise RuntimeError(f"--fast_loop draws its batches with libvaek's Philox kernel, which supports -dd / -did <= "
                               f"{DEVICE_DRAW_MAX_DIM} (got {self.dd} / {self.did}); run without --fast_loop")
        self.B = m.batch_size
        self.eng = m.model.module.engine(self.B, m.optimizer.global_batch)
        ex = m.optimizer.exchange
        if self.eng.world > 1 and not (ex is not None and ex.in_library) and not self.eng.supports_train_steps_gen(self.kind):
            raise RuntimeError("GraphLoop under data parallelism needs the in-library P2P exchange (GradExchange mode 'p2p'): "
                               "an RCCL all-reduce between the two halves of the step is not captured")
        can = self.eng.supports_train_steps_gen(self.kind)
        if moments and not can:
            raise RuntimeError("GraphLoop(moments=True): vaek_train_steps_gen does not cover this model / dataset")
        self.moments = can if moments is None else bool(moments)
        self.row0 = self.eng.rank * self.B           # ranks draw disjoint rows of the global batch
        self.seed = (ds.key[0] ^ ds.key[1] ^ m.key[1]) if seed is None else seed
        self.pipeline = bool(pipeline) and not self.moments
        self.G = int(steps_per_graph)
        if self.pipeline and self.G % 2:
            self.G += 1                              # two batch buffers: a replay must start on the parity it was captured on
        dev = self.eng.device

        def bufs():
            return (torch.empty(self.B, self.eng.D, dtype=torch.float32, device=dev),
                    torch.empty(self.B, self.eng.L, dtype=torch.float32, device=dev),
                    torch.empty(self.B, self.eng.D, dtype=torch.float32, device=dev))
        self.bufs = [] if self.moments else [bufs() for _ in range(2 if self.pipeline else 1)]
        self.loss_ring = torch.zeros(loss_capacity, dtype=torch.float32, device=dev)
        self.eng.set_loss_history(self.loss_ring)
        self.graph = None
        self.graph_parity = 0
        self.steps_done_at_attach = m.optimizer.state.step
        if self.pipeline:
            n = m.optimizer.state.step
            # the generator's own step counter, a pair used alternately (vaek_make_batch_next): the draw of batch k
            # reads counter[k % 2] and stores k + 1 into the other slot.  Invariant between steps: the batch of the
            # next step n is in bufs[n % 2] and counter[(n + 1) % 2] == n + 1.
            self.counter = torch.tensor([n, n], dtype=torch.int32, device=dev)
            self._make(self.bufs[n % 2], counter=self.counter, which=n % 2)

    def _make(self, out, **kw):
        self.eng.make_batch(self.kind, self.A, self.dd, self.did, self.pad, self.var, self.B, self.seed,
                            tag=0, row0=self.row0, out=out, **kw)

    def _one(self):
        st = self.m.optimizer.state
        lr = self.m.optimizer.optimizer_def.learning_rate
        if self.pipeline:
            n = st.step
            self.eng.train_step_gen(self.m.model.flat, st.grads, st.m, st.v, st.step_dev, self.bufs[n % 2], lr,
                                    self.kind, self.A, self.dd, self.did, self.pad, self.var, self.bufs[(n + 1) % 2],
                                    self.seed, self.counter, (n + 1) % 2, tag=0, row0=self.row0)
        else:
            x, z1, z2 = self.bufs[0]
            self._make(self.bufs[0], step_dev=st.step_dev)
            self.eng.train_step(self.m.model.flat, st.grads, st.m, st.v, st.step_dev, x, z1, z2, lr)
        st.step += 1

    def _capture(self):
        for _ in range(2):                       # warm-up outside capture (lazy kernel attributes etc.)
            self._one()
        torch.cuda.synchronize()
        self.graph_parity = self.m.optimizer.state.step % 2
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                for _ in range(self.G):
                    self._one()
        torch.cuda.current_stream().wait_stream(side)
        # capture does not execute: take back the host-side step mirror it advanced
        self.m.optimizer.state.step -= self.G
        self.graph = g
        return 2

    def run(self, n_steps):
        """Exactly n_steps train steps."""
        if self.moments:
            if n_steps > 0:
                st = self.m.optimizer.state
                self.eng.train_steps_gen(self.m.model.flat, st.grads, st.m, st.v, st.step_dev, n_steps,
                                         self.m.optimizer.optimizer_def.learning_rate, self.kind, self.A, self.dd, self.did, self.pad,
                                         self.var, self.seed, tag=0, row0=self.row0)
                st.step += n_steps
            return
        done = 0
        if self.graph is None and n_steps >= self.G + 2:
            done += self._capture()
        st = self.m.optimizer.state
        if self.graph is not None and self.pipeline and n_steps - done > self.G and st.step % 2 != self.graph_parity:
            self._one()                          # back onto the buffer parity the graph was captured on
            done += 1
        if self.graph is not None and (not self.pipeline or st.step % 2 == self.graph_parity):
            while n_steps - done >= self.G:
                self.graph.replay()
                st.step += self.G
                done += self.G
        for _ in range(n_steps - done):
            self._one()

    def check(self):
        """Synchronous.  The persistent launches of the moments path wait for each other's hand-offs with bounded spins: one that
        expired has produced garbage -- stop loudly."""
        if self.moments:
            torch.cuda.synchronize()
            if self.eng.train_steps_gave_up():
                raise RuntimeError(f"a bounded in-launch wait of vaek_train_steps_gen expired (status {self.eng.train_steps_status_word:#x}): "
                                   "the steps since the last check are invalid")

    def losses(self):
        """Losses of all steps run so far in order (device -> host once)."""
        n = self.m.optimizer.state.step
        cap = self.loss_ring.numel()
        ring = self.loss_ring.cpu()
        if n <= cap:
            return ring[:n]
        k = n % cap
        return torch.cat([ring[k:], ring[:k]])
