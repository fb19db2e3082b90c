"""Convolutional VAE (BASELINE config 5) -- host side.  NO reference counterpart: the reference has no convolutional model; the
architecture is this repository's own specification and oracle/conv_vae_oracle.py is the checker.  What IS the
reference's is everything around the two networks: epsilon_p as the latent log-variance (networks.py:68-72), the
reparameterisation (:73-74), decoder noise and the tunable decoder variance (:70-71, 81-83), the ELBO (:94-98), Adam (:100).

A train step is assembled here from the library's blocks (include/vaek.h), every product on the GPU through the C ABI:
  encoder   4 x vaek_conv2d_forward (relu)  ->  vaek_dense_fwd_reparam (mu, samples)
  decoder   vaek_dense_fwd (relu)  ->  4 x vaek_conv2d_transpose_forward (relu between, linear output)
  loss      vaek_elbo_fwd_bwd (decoder noise, KL, Gaussian likelihood, dL/dx_hat, dL/d eps)
  backward  vaek_conv2d_weight_grad / vaek_conv2d_bias_grad / vaek_conv2d_forward (the transposed layers' input gradient) /
            vaek_dense_bwd_dw / vaek_dense_bwd_dx / vaek_reparam_bwd / vaek_conv2d_transpose_forward (the convolutions' input
            gradient, the relu of the layer below applied as a mask)
  update    vaek_adam_step over the flat parameter vector
Parameters, gradients and Adam moments are ONE flat float32 buffer each, in the leaf order of ConvVAE.leaves() (the oracle's);
PyTorch holds the device memory and slices views -- it computes nothing.  This is a launch-per-layer assembly (about 95
launches per step, none of which touches the host: capture() turns the step into a hipGraph).  Tensors that a bf16-operand kernel reads again travel
with a bf16 copy written by the epilogue that produced them (conv.py: x16 / want16), so nothing is converted twice."""
import math

import torch

from .conv import conv2d_bias_grad, conv2d_forward, conv2d_transpose_forward, conv2d_weight_grad, to_bf16
from .engine import Engine

KS = 4


def _pair(r):
    """(tensor, bf16 copy or None) from a conv wrapper's result."""
    return r if isinstance(r, tuple) else (r, None)


class ConvVAE:
    def __init__(self, batch, size=64, widths=(32, 64, 128, 256), latent_dim=32, epsilon=-3.0, tunable_decoder_var=True, device=0, world=1,
                 lean=True):
        """batch: THIS rank's rows; world > 1: data parallel -- every mean is taken over batch * world rows, so the SUM of the ranks' flat
        gradients (train_step's `all_reduce`) is the global batch's gradient and every replica applies the same Adam update.
        lean (round 3): the hidden activations and their gradients live in HBM as bf16 ONLY -- no float32 twin is written, relu masks
        come from the bf16 copies, the transposed layers' bias gradients are column sums of the bf16 gradient.  Every product reads
        the same bf16 operands either way, so all leaves but those bias gradients are bitwise the same as with lean=False
        (tests/test_gpu_conv.py).  Needs the LDS-DMA kernels' shapes (power-of-two widths in 32 .. 256, power-of-two image): other
        models run as before.  lean=2 (the default): the two 32-channel images at the one-channel ends too -- the last transposed
        layer's input and the gradient reaching the first convolution's kernel gradient -- so the streaming one-channel kernels read
        bf16 copies like every other layer does (the model's arithmetic changes there: the bf16 envelope, not bitwise)."""
        assert size % 16 == 0 and len(widths) == 4
        self.lean = bool(lean) and all(32 <= w <= 256 and (w & (w - 1)) == 0 for w in widths) and (size & (size - 1)) == 0 and \
            batch * (size // 16) ** 2 >= 64
        self.lean2 = self.lean and (lean is True or int(lean) >= 2) and size % 8 == 0
        self.world = int(world)
        self.B, self.S, self.widths, self.L, self.eps_cli, self.tdv = batch, size, tuple(widths), latent_dim, float(epsilon), tunable_decoder_var
        self.bott = (size // 16) ** 2 * widths[3]
        # the block entry points want a context: a linear VAE of the same batch / widest Dense side sizes their workspaces
        self.eng = Engine(batch, max(size * size, self.bott), latent_dim, (), (), epsilon, tunable_decoder_var, False, device=device,
                          force_generic=True)
        self.device = self.eng.device
        # every leaf starts on a 16-byte boundary of the flat buffers (the kernels' 16-byte operand loads); the padding floats stay
        # zero in parameters, gradients and moments.  n_params counts the leaves only.
        off, self.leaves, self.n_params = 0, {}, 0
        for name, shape in self.leaf_shapes():
            n = math.prod(shape)
            self.leaves[name] = (off, shape)
            self.n_params += n
            off += (n + 3) // 4 * 4
        self.P = off

    def leaf_shapes(self):
        out, cin = [], 1
        for i, c in enumerate(self.widths):
            out += [(f"Encoder/Conv{i}/kernel", (KS, KS, cin, c)), (f"Encoder/Conv{i}/bias", (c,))]
            cin = c
        out += [("Encoder/FC/kernel", (self.bott, self.L)), ("Encoder/FC/bias", (self.L,)),
                ("Decoder/FC/kernel", (self.L, self.bott)), ("Decoder/FC/bias", (self.bott,))]
        chans = [self.widths[3], self.widths[2], self.widths[1], self.widths[0], 1]
        for i in range(4):
            out += [(f"Decoder/ConvT{i}/kernel", (KS, KS, chans[i + 1], chans[i])), (f"Decoder/ConvT{i}/bias", (chans[i + 1],))]
        out.append(("epsilon_p", (self.L,)))
        if self.tdv:
            out.append(("epsilon", (1,)))
        return out

    def view(self, flat, name):
        off, shape = self.leaves[name]
        return flat[off:off + math.prod(shape)].view(*shape)

    def new_flat(self):
        return torch.zeros(self.P, dtype=torch.float32, device=self.device)

    def loss_and_grad(self, params, grads, x, z1, z2):
        """x, z2 [B, S, S, 1], z1 [B, L] (contiguous float32 on the device); writes the flat gradient, returns the device
        tensor {loss, mean Dkl, mean mse, dL/d eps}."""
        B, S, L, e = x.shape[0], self.S, self.L, self.eng
        P = lambda n: self.view(params, n)
        G = lambda n: self.view(grads, n)
        # ---- forward (every tensor a bf16-operand kernel will read again gets its bf16 copy from the epilogue that produces it; lean:
        # the hidden ones get NOTHING else -- acts[1..3], dec[1..2] and the gradients below are None beside their bf16 copies)
        lean, lean2 = self.lean, self.lean2
        acts, acts16 = [x], [None]
        for i in range(4):
            y, y16 = _pair(conv2d_forward(acts[-1], P(f"Encoder/Conv{i}/kernel"), P(f"Encoder/Conv{i}/bias"), relu=True, x16=acts16[-1], want16=i < 3,
                                          want32=not (lean and i < 3)))
            acts.append(y); acts16.append(y16)
        flat = acts[-1].view(B, self.bott)
        lv = P("epsilon_p")
        mu, samples = e.dense_fwd_reparam(flat, P("Encoder/FC/kernel"), P("Encoder/FC/bias"), z1, lv)
        dec = [e.dense_fwd(samples, P("Decoder/FC/kernel"), P("Decoder/FC/bias"), relu=True).view(B, S // 16, S // 16, self.widths[3])]
        dec16 = [to_bf16(dec[0])]
        for i in range(4):
            y, y16 = _pair(conv2d_transpose_forward(dec[-1], P(f"Decoder/ConvT{i}/kernel"), P(f"Decoder/ConvT{i}/bias"), relu=i < 3,
                                                    y16=dec16[-1], want16=i < 2 or (lean2 and i == 2), want32=not (lean and i < 2) and not (lean2 and i == 2)))
            dec.append(y); dec16.append(y16)
        bt = B * self.world                                                                       # the means' denominator: the GLOBAL batch
        out4, d, _ = e.elbo_fwd_bwd(x.view(B, S * S), dec[4].view(B, S * S), None, z2.view(B, S * S), mu, lv, self.eps_cli, batch_total=bt,
                                    eps_param=P("epsilon") if self.tdv else None)        # the tunable eps is read on the device
        # ---- backward: decoder
        d, d16 = d.view(B, S, S, 1), None
        for i in reversed(range(4)):
            inp, inp16 = dec[i], dec16[i]
            conv2d_weight_grad(d, inp, want_bias=False, dw=G(f"Decoder/ConvT{i}/kernel"), x16=d16, dy16=inp16)
            conv2d_bias_grad(d if d is not None else d16, G(f"Decoder/ConvT{i}/bias"))
            m16 = inp16 if lean and inp16 is not None else None                                   # the relu mask from the bf16 copy where there is one
            d, d16 = _pair(conv2d_forward(d, P(f"Decoder/ConvT{i}/kernel"), None, relu=False, mask=None if m16 is not None else inp, mask16=m16, x16=d16,
                                          want16=i > 0, want32=not (lean and i > 0)))             # adjoint of the adjoint + relu below
        d = d.view(B, self.bott)
        dwb = e.dense_bwd_dw(samples, d)                                                          # [kernel | bias] rows
        G("Decoder/FC/kernel").copy_(dwb[:L]); G("Decoder/FC/bias").copy_(dwb[L])
        d_s = e.dense_bwd_dx(d, P("Decoder/FC/kernel"))
        e.reparam_bwd(d_s, mu, z1, lv, batch_total=bt, out=G("epsilon_p"))                        # d_s becomes d_mu in place
        dwb = e.dense_bwd_dw(flat, d_s)
        G("Encoder/FC/kernel").copy_(dwb[:self.bott]); G("Encoder/FC/bias").copy_(dwb[self.bott])
        d = e.dense_bwd_dx(d_s, P("Encoder/FC/kernel"), flat, relu=True).view(B, S // 16, S // 16, self.widths[3])
        d16 = to_bf16(d)
        # ---- backward: encoder
        for i in reversed(range(4)):
            inp, inp16 = acts[i], acts16[i]
            conv2d_weight_grad(inp, d, dw=G(f"Encoder/Conv{i}/kernel"), db=G(f"Encoder/Conv{i}/bias"), x16=inp16, dy16=d16)
            if i > 0:
                m16 = inp16 if lean else None
                d, d16 = _pair(conv2d_transpose_forward(d, P(f"Encoder/Conv{i}/kernel"), None, relu=False, mask=None if m16 is not None else inp, mask16=m16,
                                                        y16=d16, want16=i > 1 or lean2, want32=not (lean and i > 1) and not lean2))
        if self.tdv:
            G("epsilon").copy_(out4[3:4] * self.eps_cli)                                          # eps = param * eps_cli
        return out4

    def train_step(self, params, grads, m, v, step_dev, x, z1, z2, lr, all_reduce=None):
        """all_reduce (world > 1): an in-place SUM over the ranks of a device tensor (parallel.py: GradExchange.all_reduce -- RCCL)."""
        out4 = self.loss_and_grad(params, grads, x, z1, z2)
        if self.world > 1:
            all_reduce(grads)
            all_reduce(out4)                             # {loss, Dkl, mse} partial means -> the global batch's; the d eps slot likewise
        step_dev += 1
        self.eng.adam_step(params, grads, m, v, lr, step_dev=step_dev)
        return out4

    def capture(self, params, grads, m, v, step_dev, x, z1, z2, lr, warmup=2):
        """The train step as a hipGraph over these buffers (nothing in it touches the host): returns (replay, out4) -- replay() runs one
        step on the CURRENT contents of params / m / v / step_dev / x / z1 / z2, out4 is rewritten by every replay.  The `warmup` eager
        steps it runs first (per-kernel attributes are set on first use) DO update the parameters.  One GPU only (the data-parallel
        step runs eagerly around its all-reduce)."""
        assert self.world == 1
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self.train_step(params, grads, m, v, step_dev, x, z1, z2, lr)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out4 = self.train_step(params, grads, m, v, step_dev, x, z1, z2, lr)
        return graph.replay, out4
