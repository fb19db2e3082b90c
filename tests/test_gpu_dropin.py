"""-m gpu: the reference-shaped Python interface (VAEModel, VAE.train_step / VAE.loss, model(...),
run.py) end to end on the HIP path, checked against the oracle on the SAME explicit inputs
(params, x, z1, z2 pulled from the device each step: RNG streams are this build's own)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import elbo_oracle as O
from tests.gpu_util import host

pytestmark = pytest.mark.gpu


def _tree_from_model(cfg, model):
    flat = host(model.flat)
    return O.unflatten(cfg, flat)


@pytest.mark.parametrize("dataset,enc,dec,latent,pad,eps,lr", [
    ("linear_gaussian", "", "", 20, 9, -1.0, 1e-3),        # seed_linpadding_expts.sh:1  (fused path)
    ("sigmoid", "", "", 6, 3, -3.0, 1e-4),                 # sigmoid_vae_padding_expts.sh:1 (fused, two decoders)
    ("sphere", "40|40|40", "40|40|40", 6, 3, -3.0, 1e-4),   # sphere script shape at reduced width (layer-by-layer)
])
def test_vaemodel_steps_match_oracle(tmp_path, dataset, enc, dec, latent, pad, eps, lr):
    from vae_training_amd.run import get_dataset, parse_arguments
    from vae_training_amd.vae import VAEModel
    from vae_training_amd.networks import VAE
    args = parse_arguments(["t", "--dataset", dataset, "--padding_dim", str(pad), "-dd", "3", "-ds", "2"])
    ds = get_dataset(dataset, 2, pad, 128, args)
    m = VAEModel(dirname=str(tmp_path), num_batches=4, num_epochs=1, batch_size=128, learning_rate=lr, layer_sizes=dec,
                 encoder_layer_sizes=enc, state_dict=None, data_fn=None, epsilon=eps, tqdm=False, dataset=ds,
                 latent_dimension=latent, tunable_decoder_var=True, dataset_name=dataset)
    D = ds.dimension
    cfg = O.Config(D, latent, [int(s) for s in enc.split("|")] if enc else [], [int(s) for s in dec.split("|")] if dec else [],
                   eps, True, dataset)
    assert m.model.flat.numel() == cfg.n_params()
    p = _tree_from_model(cfg, m.model)
    st = O.adam_init(p)
    for step in range(3):
        x = ds.get_batch(128)
        z = m.sample_latent(m.get_key(), 128)
        z1, z2 = z[:, :latent].contiguous(), z[:, latent:].contiguous()
        assert z.shape == (128, latent + D)                       # model.py:227
        p, st, loss_ref = O.train_step(cfg, p, st, host(x), host(z1), host(z2), lr)
        m.optimizer, m.model, loss = VAE.train_step(m.optimizer, x, z1, z2, m.epsilon)
        assert abs(float(loss) - loss_ref) <= 1e-5 * abs(loss_ref)
    assert m.optimizer.state.step == 3 and int(m.optimizer.state.step_dev.item()) == 3
    assert np.max(np.abs(host(m.model.flat) - O.flatten(cfg, p))) <= 0.02 * lr
    # eval twin and sampling forward on the trained model
    x = ds.get_batch(1000); z = m.sample_latent(m.get_key(), 1000)
    z1, z2 = z[:, :latent].contiguous(), z[:, latent:].contiguous()
    l, dkl, mse, lv, e = VAE.loss(m.model, x, z1, z2, m.epsilon)
    pt = _tree_from_model(cfg, m.model)
    ref = O.loss_eval(cfg, pt, host(x), host(z1), host(z2))
    assert abs(float(l) - ref[0]) <= 1e-5 * abs(ref[0]) and abs(float(dkl) - ref[1]) <= 1e-5 * abs(ref[0])
    assert np.allclose(host(lv), ref[3], atol=1e-7) and abs(float(e.reshape(-1)[0]) - float(np.asarray(ref[4]).reshape(-1)[0])) < 1e-6
    stats = m.compute_model_stats(x, None, z)
    assert set(stats) == {"VAE Loss", "KL divergence", "mse"}
    xs, zs = m.sample_batch(m.get_key(), 1000)
    (xs_ref, _, _, _), _ = O.vae_forward(cfg, pt, None, host(zs[:, :latent]), host(zs[:, latent:]), sampling=True,
                                         epsilon=float(np.asarray(ref[4]).reshape(-1)[0]))
    assert np.max(np.abs(host(xs) - xs_ref)) <= 1e-4 * max(1.0, np.max(np.abs(xs_ref)))
    xh, mu, lv2, e2 = m.model(x, z1, z2)
    (xh_ref, mu_ref, _, _), _ = O.vae_forward(cfg, pt, host(x), host(z1), host(z2))
    assert np.max(np.abs(host(xh) - xh_ref)) <= 1e-4 * max(1.0, np.max(np.abs(xh_ref)))
    assert np.max(np.abs(host(mu) - mu_ref)) <= 1e-4 * max(1.0, np.max(np.abs(mu_ref)))


@pytest.mark.parametrize("dataset,latent,pad", [("linear_gaussian", 8, 5), ("sigmoid", 6, 3)])
def test_warm_start_starts_near_the_data_manifold(tmp_path, dataset, latent, pad):
    """-ws (vae.py:62-107): the warm-started model's first loss is far below the cold-started one's on the same batch, and it trains."""
    from vae_training_amd.run import get_dataset, parse_arguments
    from vae_training_amd.vae import VAEModel
    from vae_training_amd.networks import VAE
    args = parse_arguments(["t", "--dataset", dataset, "--padding_dim", str(pad), "-dd", "2" if dataset == "sigmoid" else "3", "-ds", "2"])
    first = {}
    for ws in (False, True):
        ds = get_dataset(dataset, 2, pad, 256, args)
        m = VAEModel(dirname=str(tmp_path), num_batches=4, num_epochs=1, batch_size=256, learning_rate=1e-3, layer_sizes="",
                     encoder_layer_sizes="", state_dict=None, data_fn=None, epsilon=-3.0, tqdm=False, dataset=ds, latent_dimension=latent,
                     tunable_decoder_var=True, dataset_name=dataset, warm_start=ws, latent_off_dimension=1)
        x = ds.get_batch(256)
        z = m.sample_latent(m.get_key(), 256)
        z1, z2 = z[:, :latent].contiguous(), z[:, latent:].contiguous()
        losses = []
        for _ in range(5):
            m.optimizer, m.model, loss = VAE.train_step(m.optimizer, x, z1, z2, m.epsilon)
            losses.append(float(loss))
        assert np.isfinite(losses).all() and losses[-1] < losses[0]
        first[ws] = losses[0]
    assert first[True] < 0.5 * first[False], first


def test_run_py_end_to_end_and_resume(tmp_path, monkeypatch, capsys):
    """`python run.py NAME ...` side effects (utils.py:46-60, model.py:246-255) and a --state_dict resume."""
    from vae_training_amd import run, utils
    monkeypatch.setattr(utils, "DATA_DIR", str(tmp_path) + "/")
    argv = ["e2e", "--dataset", "linear_gaussian", "--encoder_layer_sizes", "", "--layer_sizes", "", "-ow", "--latent_dim", "20",
            "--padding_dim", "9", "-dd", "3", "--num_batches", "60", "--batch_size", "256", "--epsilon", "-1", "-tdv", "-ds", "2",
            "-lr", "1e-3"]
    assert run.main(run.parse_arguments(argv)) == 0
    out = capsys.readouterr().out
    assert "Score for real data" in out and "Batch | 0 | VAE Loss" in out and "Squared Norm of padding dimensions" in out
    d = os.path.join(str(tmp_path), "e2e")
    assert {"args.json", "losses.npz", "model.pkl"} <= set(os.listdir(d))
    assert json.load(open(os.path.join(d, "args.json")))["latent_dimension"] == 20
    z = np.load(os.path.join(d, "losses.npz"), allow_pickle=True)
    assert len(z["VAE Loss"]) == 61 and np.isfinite(np.asarray(z["VAE Loss"], dtype=np.float64)).all()
    first, last = float(z["VAE Loss"][1]), float(z["VAE Loss"][-1])
    assert last < first                                        # it trains
    argv2 = ["e2e_resume"] + argv[1:] + ["--state_dict", os.path.join(d, "model.pkl")]
    argv2[argv2.index("--num_batches") + 1] = "5"
    assert run.main(run.parse_arguments(argv2)) == 0
    z2 = np.load(os.path.join(str(tmp_path), "e2e_resume", "losses.npz"), allow_pickle=True)
    assert float(z2["VAE Loss"][0]) < first                    # resumed from the trained parameters
