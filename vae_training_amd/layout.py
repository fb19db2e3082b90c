"""Flat parameter layout shared with libvaek (include/vaek.h "conventions"): leaf order
Encoder/FC*/{kernel,bias}, Decoder/..., [SigDecoder/...], epsilon_p, [epsilon].  Pure Python so
host logic (param trees, checkpoints) works without a GPU; Engine asserts it matches the
library's own leaf table."""
from __future__ import annotations

from collections import OrderedDict


def leaves(data_dim, latent_dim, enc_hidden=(), dec_hidden=(), sigmoid_decoder=False, tunable_eps=False):
    out, off = OrderedDict(), 0
    nets = [("Encoder", data_dim, list(enc_hidden) + [latent_dim]), ("Decoder", latent_dim, list(dec_hidden) + [data_dim])]
    if sigmoid_decoder:
        nets.append(("SigDecoder", latent_dim, list(dec_hidden) + [data_dim]))
    for name, k, sizes in nets:
        for i, n in enumerate(sizes):
            out[f"{name}/FC{i}/kernel"] = (off, (k, n)); off += k * n
            out[f"{name}/FC{i}/bias"] = (off, (n,)); off += n
            k = n
    out["epsilon_p"] = (off, (latent_dim,)); off += latent_dim
    if tunable_eps:
        out["epsilon"] = (off, (1,)); off += 1
    return out, off


def views(flat, leaf_table):
    """Nested dict of views into `flat`, reference param-tree names (vae.py:73-80)."""
    tree = OrderedDict()
    for name, (off, shape) in leaf_table.items():
        numel = 1
        for s in shape:
            numel *= s
        node, parts = tree, name.split("/")
        for p in parts[:-1]:
            node = node.setdefault(p, OrderedDict())
        node[parts[-1]] = flat[off:off + numel].view(*shape)
    return tree
