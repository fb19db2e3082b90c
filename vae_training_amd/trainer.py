"""Graph-captured training loop: the reference's hot loop (model.py:221-222 -- get_batch, sample_latent,
train_step, append loss) with NOTHING on the host per step.

Per step three launches -- vaek_make_batch (Philox dataset + latent draw, K7), and the two kernels of
vaek_train_step -- are captured G at a time into a hipGraph; the Adam step counter, the RNG step and
the loss ring buffer all live on the device, so replays need no arguments.  At the reference's own
batch size (100) the per-step Python of the drop-in path (~50 us) is 3-4x the GPU time; this loop
removes it without changing what a step computes."""
from __future__ import annotations

import torch


class GraphLoop:
    def __init__(self, vae_model, steps_per_graph=50, seed=None, loss_capacity=1 << 20):
        m = vae_model
        self.m = m
        ds = m.dataset
        self.kind, self.A, self.dd, self.did, self.pad, self.var = ds.device_spec()
        self.B = m.batch_size
        self.eng = m.model.module.engine(self.B, m.optimizer.global_batch)
        self.seed = (ds.key[0] ^ ds.key[1] ^ m.key[1]) if seed is None else seed
        self.G = int(steps_per_graph)
        dev = self.eng.device
        self.x = torch.empty(self.B, self.eng.D, dtype=torch.float32, device=dev)
        self.z1 = torch.empty(self.B, self.eng.L, dtype=torch.float32, device=dev)
        self.z2 = torch.empty(self.B, self.eng.D, dtype=torch.float32, device=dev)
        self.loss_ring = torch.zeros(loss_capacity, dtype=torch.float32, device=dev)
        self.eng.set_loss_history(self.loss_ring)
        self.graph = None
        self.steps_done_at_attach = m.optimizer.state.step

    def _one(self):
        st = self.m.optimizer.state
        self.eng.make_batch(self.kind, self.A, self.dd, self.did, self.pad, self.var, self.B, self.seed,
                            step_dev=st.step_dev, tag=0, out=(self.x, self.z1, self.z2))
        self.eng.train_step(self.m.model.flat, st.grads, st.m, st.v, st.step_dev, self.x, self.z1, self.z2,
                            self.m.optimizer.optimizer_def.learning_rate)
        st.step += 1

    def _capture(self):
        for _ in range(2):                       # warm-up outside capture (lazy kernel attributes etc.)
            self._one()
        torch.cuda.synchronize()
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
        done = 0
        if self.graph is None and n_steps >= self.G + 2:
            done += self._capture()
        while self.graph is not None and n_steps - done >= self.G:
            self.graph.replay()
            self.m.optimizer.state.step += self.G
            done += self.G
        for _ in range(n_steps - done):
            self._one()

    def losses(self):
        """Losses of all steps run so far in order (device -> host once)."""
        n = self.m.optimizer.state.step
        cap = self.loss_ring.numel()
        ring = self.loss_ring.cpu()
        if n <= cap:
            return ring[:n]
        k = n % cap
        return torch.cat([ring[k:], ring[:k]])
