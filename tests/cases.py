"""Shared parity cases (small shapes of BASELINE.json's configs)."""
from oracle.elbo_oracle import Config

# name -> (Config kwargs, dataset kwargs, B, lr)
CASES = {
    # C1: seed_linpadding_expts.sh:1  (dd3 pad9 ld20 eps-1 tdv lr1e-3, linear enc/dec)
    "c1_linear_L20": (dict(data_dim=12, latent_dim=20, epsilon=-1.0, tunable_decoder_var=True,
                           dataset_name="linear_gaussian"),
                      dict(name="linear_gaussian", seed=2, dd=3, did=3, pad=9), 16, 1e-3),
    # C1 as BASELINE.json words it: latent 2
    "c1_linear_L2": (dict(data_dim=12, latent_dim=2, epsilon=-1.0, tunable_decoder_var=True,
                          dataset_name="linear_gaussian"),
                     dict(name="linear_gaussian", seed=2, dd=3, did=3, pad=9), 16, 1e-3),
    # fixed decoder variance (no -tdv), epsilon = CLI default 0
    "linear_notdv": (dict(data_dim=20, latent_dim=20, epsilon=0.0, tunable_decoder_var=False,
                          dataset_name="linear_gaussian"),
                     dict(name="linear_gaussian", seed=2, dd=3, did=3, pad=17), 8, 1e-4),
    # sigmoid script row 1: two decoders, linear  (sigmoid_vae_padding_expts.sh:1)
    "sigmoid_linear": (dict(data_dim=7, latent_dim=6, epsilon=-3.0, tunable_decoder_var=True,
                            dataset_name="sigmoid"),
                       dict(name="sigmoid", seed=69, dd=3, pad=3), 16, 1e-4),
    # C2 shape at reduced width: sigmoid with one hidden layer, two decoders
    "c2_sigmoid_mlp": (dict(data_dim=7, latent_dim=6, enc_hidden=(32,), dec_hidden=(32,), epsilon=-3.0,
                            tunable_decoder_var=True, dataset_name="sigmoid"),
                       dict(name="sigmoid", seed=69, dd=3, pad=3), 16, 1e-4),
    # C3 shape at reduced width: sphere, 3 hidden layers (sphere_vae_padding_expts.sh:1 uses 200|200|200)
    "c3_sphere_mlp": (dict(data_dim=6, latent_dim=6, enc_hidden=(24, 16, 24), dec_hidden=(24, 16, 24),
                           epsilon=-3.0, tunable_decoder_var=True, dataset_name="sphere"),
                      dict(name="sphere", seed=69, dd=3, pad=3), 16, 1e-4),
    # C4 shape at reduced ambient dim: wide linear
    "c4_linear_wide": (dict(data_dim=96, latent_dim=20, epsilon=-1.0, tunable_decoder_var=True,
                            dataset_name="linear_gaussian"),
                       dict(name="linear_gaussian", seed=2, dd=3, did=3, pad=93), 8, 1e-3),
}


def build(name):
    ck, dk, B, lr = CASES[name]
    return Config(**ck), dk, B, lr
