"""Configuration of the hot path: the reference's Hydra/OmegaConf keys as a plain attribute dict.

`load("default")` returns the values of the reference's `config/default.yaml` for the keys that
`sampler.py`, `network.py`, `renderer.py`, `specular_brdf.py` and `loss.py` read; the named
variants are the reference's own few-line diffs of it (config/triplaneline.yaml:19-21,
config/no_voxel.yaml:19, config/custom.yaml).  Overrides use the reference's dotted
`key=value` form: `load("default", ["renderer.n_upsamples=0"])`.
"""
import copy
import os

import yaml

_HERE = os.path.dirname(os.path.abspath(__file__))


class Conf(dict):
    """dict with attribute access (OmegaConf-style `conf.renderer.n_samples0`)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return Conf({k: copy.deepcopy(v, memo) for k, v in self.items()})


def _wrap(d):
    if isinstance(d, dict):
        return Conf({k: _wrap(v) for k, v in d.items()})
    return d


# the reference's shipped variants, as diffs of default.yaml
_VARIANTS = {
    "default": [],
    "triplaneline": ["geometric_network.voxel.type=triplaneline",
                     "geometric_network.voxel.grid_size=2048",
                     "geometric_network.voxel.feature_size=8"],
    "no_voxel": ["geometric_network.voxel.type=none"],
    "ste": ["geometric_network.voxel.use_ste=true"],
    "custom": ["geometric_network.initial_sphere_radius=0.5",
               "geometric_network.voxel.type=lanczos_voxel",
               "renderer.eps_normal=1.0e-08", "train.tv_weight=1.0",
               "train.roughness_prior_weight=1.0e-04"],
}


def apply_overrides(conf, overrides):
    for ov in overrides or []:
        key, val = ov.split("=", 1)
        node = conf
        parts = key.split(".")
        for p in parts[:-1]:
            node = node[p]
        node[parts[-1]] = yaml.safe_load(val)
    return conf


def load(name="default", overrides=None):
    from .defaults import DEFAULTS
    conf = Conf()
    for key, val in DEFAULTS.items():
        node = conf
        parts = key.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, Conf())
        node[parts[-1]] = copy.deepcopy(val)
    apply_overrides(conf, _VARIANTS[name])
    apply_overrides(conf, overrides)
    return conf
